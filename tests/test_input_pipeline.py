"""Row f3: device-side finishing of a pre-training batch.
CPU: the oracle (oracle/batch_oracle.py) against the reference's own code and the golden fixture.
GPU: vb_concap_finish_batch / DeviceBatchPipeline against the oracle - bit-exact (integer, copy and
correctly-rounded fp32 work)."""
import os

import numpy as np
import pytest
import torch

import ref_loader_source as rls
from oracle import batch_oracle as bo

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "concap_batch.npz")


def _golden():
    z = np.load(GOLDEN)
    raw = {k[4:]: z[k] for k in z.files if k.startswith("raw_")}
    ref = {k[4:]: z[k] for k in z.files if k.startswith("ref_") and not k.startswith("ref_obj1_")}
    return raw, ref, z["ref_obj1_image_label"], z["ref_obj1_lm_label_ids"]


def _same(a, b, name):
    a, b = np.asarray(a), np.asarray(b)
    assert a.shape == b.shape and a.dtype == b.dtype, (name, a.shape, b.shape, a.dtype, b.dtype)
    assert np.array_equal(a, b), name


def test_oracle_matches_golden_reference_outputs():
    raw, ref, il1, lm1 = _golden()
    out = bo.finish_batch(raw, objective=0)
    for n in bo.OUT_FIELDS:
        _same(out[n], ref[n], n)
    out1 = bo.finish_batch(raw, objective=1)
    _same(out1["image_label"], il1, "objective-1 image_label")
    _same(out1["lm_label_ids"], lm1, "objective-1 lm_label_ids")
    assert (lm1 == 0).sum() == 0 and (raw["lm_label_ids"] == 0).sum() >= 1      # the label-0 quirk is exercised
    assert (raw["masked_label"] == 0).sum(1).min() == 0                        # and the divisor clamp


@pytest.mark.skipif(not rls.available(), reason="needs /root/reference (build container only)")
@pytest.mark.parametrize("seed,shape", [(3, dict(batch=5, tokens=12, regions=9, feat_dim=32, n_classes=7)),
                                        (4, dict(batch=3, tokens=36, regions=36, feat_dim=2048, n_classes=1601))])
def test_oracle_matches_the_reference_code_run_live(seed, shape):
    raw = bo.make_raw_batch(seed=seed, **shape)
    tup = tuple(raw[n].copy() for n in bo.RAW_FIELDS) + (np.arange(shape["batch"]),)
    got = rls.reference_loader_iter([tup])[0]
    out = bo.finish_batch(raw, objective=0)
    for i, n in enumerate(bo.OUT_FIELDS):
        _same(out[n], got[i].numpy(), n)
    il1, lm1 = rls.reference_objective1_edit(torch.tensor(raw["image_label"]), torch.tensor(raw["lm_label_ids"]),
                                             torch.tensor(raw["is_next"]))
    out1 = bo.finish_batch(raw, objective=1)
    _same(out1["image_label"], il1.numpy(), "image_label")
    _same(out1["lm_label_ids"], lm1.numpy(), "lm_label_ids")


def _to_dev(raw):
    return {k: torch.from_numpy(np.ascontiguousarray(v)).cuda() for k, v in raw.items()}


@pytest.mark.gpu
@pytest.mark.parametrize("objective", [0, 1])
def test_kernel_matches_golden_and_oracle(objective):
    from vilbert import input_pipeline as ip
    raw, ref, il1, lm1 = _golden()
    out = ip.finish_batch(_to_dev(raw), objective)
    want = bo.finish_batch(raw, objective)
    for i, n in enumerate(bo.OUT_FIELDS):
        _same(out[i].cpu().numpy(), want[n], n)
    if objective == 0:
        for i, n in enumerate(bo.OUT_FIELDS):
            _same(out[i].cpu().numpy(), ref[n], "golden " + n)
    else:
        _same(out[8].cpu().numpy(), il1, "golden objective-1 image_label")
        _same(out[3].cpu().numpy(), lm1, "golden objective-1 lm_label_ids")


@pytest.mark.gpu
def test_kernel_full_size_batch_is_bit_exact():
    """BASELINE shapes (36 regions x 2048, 36 tokens) at batch 64: every output equals the oracle exactly,
    including the fp32 mean row (same summation order, fp64 division, one rounding)."""
    from vilbert import input_pipeline as ip
    raw = bo.make_raw_batch(64, seed=21)
    out = ip.finish_batch(_to_dev(raw), 1)
    want = bo.finish_batch(raw, 1)
    for i, n in enumerate(bo.OUT_FIELDS):
        _same(out[i].cpu().numpy(), want[n], n)
    # size-independent properties: row 0 * count == column sums (to rounding), rows 1.. are the input
    feat = out[5]
    assert torch.equal(feat[:, 1:], torch.from_numpy(raw["image_feat"]).cuda())
    cnt = np.maximum((raw["masked_label"] == 0).sum(1), 1)
    np.testing.assert_allclose(feat[:, 0].cpu().numpy() * cnt[:, None], raw["image_feat"].sum(1), rtol=1e-5, atol=1e-5)


@pytest.mark.gpu
def test_bad_arguments_raise():
    from vilbert import input_pipeline as ip
    raw = bo.make_raw_batch(2, tokens=8, regions=5, feat_dim=16, n_classes=3, vocab=20)
    dev = _to_dev(raw)
    dev["image_mask"] = dev["image_mask"][:, :-1]
    with pytest.raises(RuntimeError, match="image_mask"):
        ip.finish_batch(dev, 0)
    with pytest.raises(RuntimeError, match="HIP device"):
        ip.finish_batch({k: torch.from_numpy(v) for k, v in raw.items()}, 0)


@pytest.mark.gpu
def test_double_buffered_pipeline_over_several_batches():
    from vilbert import input_pipeline as ip
    raws = [bo.make_raw_batch(4 + (i % 2), tokens=10, regions=6, feat_dim=64, n_classes=9, vocab=99, seed=30 + i)
            for i in range(5)]
    source = [tuple(r[n] if n != "input_ids" else r[n].astype(np.int32) for n in bo.RAW_FIELDS) + (["id%d" % i],)
              for i, r in enumerate(raws)]
    seen = 0
    for i, batch in enumerate(ip.DeviceBatchPipeline(source, "cuda:0", objective=1)):
        assert len(batch) == 11 and batch[10] == ["id%d" % i]
        want = bo.finish_batch(raws[i], 1)
        got = [t.cpu().numpy() for t in batch[:10]]      # read before the slot is reused
        for j, n in enumerate(bo.OUT_FIELDS):
            _same(got[j], want[n], "batch %d %s" % (i, n))
        torch.cuda.current_stream().synchronize()
        seen += 1
    assert seen == 5
