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


# ---- MX (block-scaled) statement, round 4 -----------------------------------------------------------------------------
def test_mx_scale_rule_is_the_smallest_power_of_two_that_fits():
    rng = np.random.default_rng(0)
    amax = np.concatenate([np.exp2(rng.uniform(-40, 40, 20000)).astype(np.float32),
                           np.array([448.0, 449.0, 447.99, 224.0, 224.01, 1.0, 1.75, 1.7500001, 0.875, 3.5], np.float32)])
    byte = F.mx_scale_bytes(amax).astype(np.int32)
    scale = np.ldexp(1.0, byte - 127)
    assert (amax.astype(np.float64) / scale <= 448.0).all()                 # fits ...
    assert (amax.astype(np.float64) / (scale / 2) > 448.0).all()            # ... and the next smaller power of two does not
    assert F.mx_scale_bytes(np.float32(0.0)) == 0 and F.mx_scale_bytes(np.float32(448.0)) == 127
    assert F.mx_scale_bytes(np.float32(449.0)) == 128 and F.mx_scale_bytes(np.float32(1e-45)) == 0


def test_mx_codes_are_torch_float8_of_the_scaled_block_and_round_trip():
    rng = np.random.default_rng(1)
    x = (rng.standard_normal((64, 256)) * np.exp2(rng.integers(-12, 12, (64, 8)).repeat(32, axis=1))).astype(np.float32)
    x[3] = 0
    x[5, 32:64] = 0
    q, b = F.mx_quantize(x)
    inv = np.ldexp(np.float32(1), 127 - b.astype(np.int32)).repeat(32, axis=1).astype(np.float32)
    want = torch.from_numpy(x * inv).to(torch.float8_e4m3fn).view(torch.uint8).numpy()
    assert np.array_equal(q, want)
    assert (b[3] == 0).all() and b[5, 1] == 0 and (q[3] == 0).all()
    back = F.mx_dequantize(q, b)
    amax = np.abs(x.reshape(64, 8, 32)).max(axis=2).repeat(32, axis=1)
    # one e4m3 rounding relative to the block: <= 2^-4 relative for normals, <= 2^-10 of the block scale for subnormals
    scale = np.ldexp(1.0, b.astype(np.int32) - 127).repeat(32, axis=1)
    assert (np.abs(back - x) <= np.abs(x) / 16 + scale / 1024 + 1e-300).all()
    assert (np.abs(back) <= amax * (1 + 1 / 16) + 1e-300).all() and (np.abs(back) <= 448 * scale).all()
    words = F.mx_scale_words(b, 256)
    assert words.shape == (2, 256) and words[0, 5] == (int(b[5, 0]) | int(b[5, 1]) << 8 | int(b[5, 2]) << 16 | int(b[5, 3]) << 24)
    assert np.array_equal(F.mx_words_to_bytes(words, 64), b)


def test_mx_linear_is_closer_to_fp32_than_its_bound():
    rng = np.random.default_rng(2)
    x = rng.standard_normal((32, 512)).astype(np.float32)
    w = (rng.standard_normal((48, 512)) * 0.05).astype(np.float32)
    y = F.linear_mx(x, w)
    exact = x.astype(np.float64) @ w.astype(np.float64).T
    rel = np.linalg.norm(y - exact) / np.linalg.norm(exact)
    assert 1e-4 < rel < 0.06, rel


def test_mx_gelu_polynomial_is_within_its_stated_bound_of_the_erf_form():
    x = np.linspace(-12, 12, 200001)
    want = torch.nn.functional.gelu(torch.from_numpy(x)).numpy()
    err = np.abs(F.mx_gelu(x) - want)
    assert (err <= 4e-4 * np.abs(x) + 6e-4).all(), float((err - 4e-4 * np.abs(x)).max())
