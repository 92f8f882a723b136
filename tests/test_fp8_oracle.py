"""The FP8 oracle's number format against PyTorch's independent CPU implementation of OCP e4m3fn."""
import numpy as np
import torch

from oracle import fp8_oracle as F


def test_decode_all_codes_matches_torch():
    codes = np.arange(256, dtype=np.uint8)
    ours = F.e4m3_decode(codes)
    theirs = torch.from_numpy(codes).view(torch.float8_e4m3fn).float().numpy()
    assert np.array_equal(np.isnan(ours), np.isnan(theirs))
    ok = ~np.isnan(ours)
    assert np.array_equal(ours[ok], theirs[ok])
    assert ours[0x7E] == 448.0 and ours[0x01] == 2.0 ** -9


def test_encode_matches_torch_rne():
    rng = np.random.RandomState(0)
    x = np.concatenate([
        rng.uniform(-448, 448, 200000), rng.standard_normal(200000) * 0.01, rng.standard_normal(100000) * 2.0 ** -8,
        F.e4m3_decode(np.arange(256, dtype=np.uint8))[~np.isnan(F.e4m3_decode(np.arange(256, dtype=np.uint8)))],
        [0.0, -0.0, 448.0, -448.0, 2.0 ** -10, 3 * 2.0 ** -10, 2.0 ** -9 * 0.5, 17.0, 18.0, 19.0, 447.9],
    ]).astype(np.float32)
    # every midpoint between neighbouring codes (ties-to-even cases)
    vals = np.sort(F.e4m3_decode(np.arange(0, 0x7F, dtype=np.uint8)))
    mid = ((vals[1:].astype(np.float64) + vals[:-1]) / 2).astype(np.float32)
    x = np.concatenate([x, mid, -mid])
    ours = F.e4m3_encode(x)
    theirs = torch.from_numpy(x).to(torch.float8_e4m3fn).view(torch.uint8).numpy()
    # -0.0 / +0.0: same value, sign bit kept by both
    assert np.array_equal(ours, theirs), np.flatnonzero(ours != theirs)[:10]


def test_quantize_rows_roundtrip_error_bound():
    rng = np.random.RandomState(1)
    x = (rng.standard_normal((64, 256)) * rng.uniform(0.01, 30, (64, 1))).astype(np.float32)
    x[5] = 0
    q, s = F.quantize_rows(x)
    assert s[5] == 1.0 and not q[5].any()
    back = F.e4m3_decode(q) * s[:, None]
    amax = np.abs(x).max(axis=1, keepdims=True)
    # half a quantum: relative 2^-4 in the normal range, absolute 2^-10 (in units of the row scale) below it
    assert np.all(np.abs(back - x) <= np.maximum(np.abs(x) * 2.0 ** -4, s[:, None] * 2.0 ** -10) * (1 + 1e-6))
    # the row maximum itself is exact
    assert np.allclose(np.abs(back).max(axis=1)[np.arange(64) != 5], amax[np.arange(64) != 5, 0], rtol=1e-6)


def test_linear_fp8_close_to_fp32():
    rng = np.random.RandomState(2)
    x = rng.standard_normal((32, 768)).astype(np.float32)
    w = (rng.standard_normal((64, 768)) * 0.02).astype(np.float32)
    y8 = F.linear_fp8(x, w)
    y = x.astype(np.float64) @ w.astype(np.float64).T
    rel = np.linalg.norm(y8 - y) / np.linalg.norm(y)
    assert rel < 0.05, rel
