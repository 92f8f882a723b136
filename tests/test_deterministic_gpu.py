"""Deterministic split-K weight gradients (vb_set_deterministic, round-2 verdict missing item 6): partial products of the
K splits go to a workspace and are added in split order instead of with fp32 atomics. Two runs must be BIT-identical
(they are not required to be with atomics), the values must agree with the atomic path and with fp64; a launch that finds
no workspace slice (too small, ninth stream, another device) falls back to atomics and is counted."""
import pytest
import torch

from oracle import synth

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture
def det():
    from vilbert import _native
    wanted = _native._DET["wanted"]           # (on by default; registered lazily at the first split launch)
    _native.set_deterministic(True)
    yield _native
    _native.set_deterministic(wanted)


def _rand(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


# token rows, out-features per segment, in-features, segments: 4-wave kernel with several splits (768 x 768 at 9216 rows,
# the q | k | v stack, a ragged-row image-stream shape), the persistent kernel (FFN weights: 64 tiles x 4 splits)
@pytest.mark.parametrize("M,N,K,nseg", [(9216, 768, 768, 1), (9216, 768, 768, 3), (9472, 1024, 1024, 1), (9216, 3072, 768, 1),
                                        (9216, 768, 3072, 1), (1000, 256, 128, 1)])
def test_weight_gradient_is_bit_identical_and_correct(det, M, N, K, nseg):
    from vilbert import ops
    x, dy = _rand(M, K, seed=1).to(DEV), _rand(M, nseg * N, seed=2).to(DEV)
    runs = []
    for _ in range(3):
        dws, dbs = ops.linear_bwd_weight(dy, x, nseg, N, [True] * nseg)
        runs.append([t.clone() for t in dws + dbs])
    for other in runs[1:]:
        for a, b in zip(runs[0], other):
            assert torch.equal(a, b), "deterministic mode gave different bits in two runs"
    for s in range(nseg):
        seg = dy[:, s * N:(s + 1) * N].double()
        want = (seg.t() @ x.double()).cpu()
        err = (runs[0][s].cpu().double() - want).abs().max().item()
        assert err <= 3e-5 * max(1.0, want.abs().max().item()), err
        wb = seg.sum(0).cpu()
        assert (runs[0][nseg + s].cpu().double() - wb).abs().max().item() <= 3e-5 * max(1.0, M / 64)
    # accumulation into an existing gradient
    g0 = _rand(N, K, seed=3).to(DEV)
    g1 = g0.clone()
    ops.linear_bwd_weight(dy[:, :N].contiguous(), x, 1, N, [False], dw_out=[g1])
    want = g0.cpu().double() + (dy[:, :N].double().t() @ x.double()).cpu()
    assert (g1.cpu().double() - want).abs().max().item() <= 3e-5 * max(1.0, want.abs().max().item())
    # same values as the atomic path up to summation order
    det.set_deterministic(False)
    dws0, _ = ops.linear_bwd_weight(dy, x, nseg, N, [True] * nseg)
    det.set_deterministic(True)
    for a, b in zip(runs[0][:nseg], dws0):
        assert (a - b).abs().max().item() <= 3e-5 * max(1.0, b.abs().max().item())


@pytest.mark.parametrize("mode,tol", [("bf16x6", 3e-5), ("bf16", 2.0 ** -6)])
@pytest.mark.parametrize("M,N,K", [(9216, 768, 768), (9472, 1024, 1024), (1604, 1601, 1024)])
def test_reduced_precision_modes_and_the_fallback_kernel_split_through_the_workspace(det, mode, tol, M, N, K):
    """The bf16-plane kernels and the round-1 fallback kernel (ragged shapes: 1601 out-features over 1604 rows) split
    the contraction of a weight gradient too. With atomics only, the deterministic default forced them to ONE split
    (bf16 train step 4,871 -> 4,250 samples/s when the default changed); they store to the same workspace now."""
    from vilbert import ops
    x, dy = _rand(M, K, seed=1).to(DEV), _rand(M, N, seed=2).to(DEV)
    want = (dy.double().t() @ x.double()).cpu()
    prev = det.set_gemm_mode(mode)
    try:
        (a,), (ab,) = ops.linear_bwd_weight(dy, x, 1, N, [True])
        (b,), (bb,) = ops.linear_bwd_weight(dy, x, 1, N, [True])
    finally:
        det.set_gemm_mode(prev)
    assert torch.equal(a, b) and torch.equal(ab, bb)
    assert (a.cpu().double() - want).abs().max().item() <= tol * want.abs().max().item()
    assert (ab.cpu().double() - dy.double().sum(0).cpu()).abs().max().item() <= 1e-3
    # fp32 mode, ragged shape: the fallback kernel
    (c,), _ = ops.linear_bwd_weight(dy, x, 1, N, [True])
    (d,), _ = ops.linear_bwd_weight(dy, x, 1, N, [True])
    assert torch.equal(c, d)
    assert (c.cpu().double() - want).abs().max().item() <= 3e-5 * want.abs().max().item()


def test_split_k_dgrad_of_a_small_output_is_bit_identical(det):
    """vb_linear_bwd_input splits a long contraction over a small output (the MLM decoder: dX[rows, 768] over 30522
    out-features) - same mechanism."""
    from vilbert import ops
    dy, w = _rand(1628, 30522, seed=4, scale=0.1).to(DEV), _rand(30522, 768, seed=5, scale=0.05).to(DEV)
    a = ops.linear_bwd_input(dy, [w], 768)
    b = ops.linear_bwd_input(dy, [w], 768)
    assert torch.equal(a, b)
    want = (dy.double() @ w.double()).cpu()
    assert (a.cpu().double() - want).abs().max().item() <= 1e-4 * max(1.0, want.abs().max().item())


def test_workspace_too_small_falls_back_to_atomics_and_is_counted():
    """Round-3 advisor: a launch whose partials do not fit its slice used to fail with VB_E_WORKSPACE; it now runs with
    the fp32 atomics (correct, not bit-reproducible) and `deterministic_fallbacks()` says so."""
    from vilbert import _native, ops
    prev = _native._DET["wanted"]
    _native.set_deterministic(False)
    ws = torch.empty(8 * 1024, dtype=torch.float32, device=DEV)
    assert _native.lib().vb_set_deterministic(1, ws.data_ptr(), ws.numel() * 4) == 0
    try:
        assert _native.deterministic_fallbacks() == 0
        x, dy = _rand(9216, 768, seed=1).to(DEV), _rand(9216, 768, seed=2).to(DEV)
        (dw,), (db,) = ops.linear_bwd_weight(dy, x, 1, 768, [True])
        torch.cuda.synchronize()
        assert _native.deterministic_fallbacks() >= 1
        want = (dy.double().t() @ x.double()).cpu()
        assert (dw.cpu().double() - want).abs().max().item() <= 1e-4 * want.abs().max().item()
        assert (db.cpu().double() - dy.double().sum(0).cpu()).abs().max().item() <= 1e-3
    finally:
        _native.lib().vb_set_deterministic(0, None, 0)
        _native.set_deterministic(prev)


def test_ninth_stream_falls_back_instead_of_raising_and_slots_survive_reregistration(det):
    """A workspace has 8 per-stream slices. Round-3 advisor: a ninth stream made every split launch on it fail for the
    rest of the process (GraphedTrainStep warm-ups draw fresh streams). Now: streams 1-8 are ordered (bit-identical),
    stream 9 runs with atomics and is counted; registering the SAME buffer again keeps the assignment (captured graphs
    have it baked in), a NEW buffer starts with free slices."""
    from vilbert import _native, ops
    _native.set_deterministic(False)
    _native.set_deterministic(True, device=DEV)
    x, dy = _rand(9216, 768, seed=1).to(DEV), _rand(9216, 768, seed=2).to(DEV)
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream() for _ in range(9)]
    first = None
    for i, st in enumerate(streams):
        with torch.cuda.stream(st):
            (dw,), _ = ops.linear_bwd_weight(dy, x, 1, 768, [True])
        st.synchronize()
        if i < 8:
            assert _native.deterministic_fallbacks() == 0, i
            first = dw if first is None else first
            assert torch.equal(dw, first)                        # ordered reduce: same bits on every stream
        else:
            assert _native.deterministic_fallbacks() >= 1
            assert (dw - first).abs().max().item() <= 1e-3 * first.abs().max().item()
    ws = _native.deterministic_workspace(DEV)
    _native.set_deterministic(True, device=DEV)                  # same buffer: assignment kept -> stream 9 still has no slice
    assert _native.deterministic_workspace(DEV) is ws
    with torch.cuda.stream(streams[8]):
        ops.linear_bwd_weight(dy, x, 1, 768, [True])
    streams[8].synchronize()
    assert _native.deterministic_fallbacks() >= 1
    _native.set_deterministic(True, workspace_mb=2304, device=DEV)   # larger -> new buffer (the old one lives on only in graphs that hold it)
    assert _native.deterministic_workspace(DEV) is not ws
    with torch.cuda.stream(streams[8]):
        (dw,), _ = ops.linear_bwd_weight(dy, x, 1, 768, [True])
    streams[8].synchronize()
    assert _native.deterministic_fallbacks() == 0 and torch.equal(dw, first)
    _native.set_deterministic(True, workspace_mb=2048)
    # an off / on toggle re-registers the SAME buffer: no 2 GiB leak per toggle (round-4 advisor)
    cur = _native.deterministic_workspace(DEV)
    before = torch.cuda.memory_allocated()
    for _ in range(3):
        _native.set_deterministic(False)
        assert _native.deterministic_workspace(DEV) is None
        _native.set_deterministic(True, device=DEV)
        assert _native.deterministic_workspace(DEV) is cur
    assert torch.cuda.memory_allocated() == before


def test_workspace_is_per_device_and_lazily_registered(det):
    """Round-3 advisor (high): the workspace used to be one process-global buffer on whichever device ran backward first.
    It is now keyed by device; `ensure_deterministic` registers the launch device's own on first use."""
    from vilbert import _native
    _native.set_deterministic(False)
    assert _native._DET["ws"] == {}
    _native._DET["wanted"] = True
    _native.ensure_deterministic(torch.device(DEV))
    ws = _native.deterministic_workspace(DEV)
    assert ws is not None and ws.device == torch.device(DEV) and list(_native._DET["ws"]) == [0]
    _native.ensure_deterministic(torch.device(DEV))
    assert _native.deterministic_workspace(DEV) is ws


def test_default_is_deterministic_and_streams_stay_on():
    """The setting is on by default (VB_DETERMINISTIC unset) and keeps the two-stream overlap and the weight-gradient side
    streams (every stream has its own workspace slice)."""
    import vilbert.vilbert as V
    from vilbert import _native, autograd_ops as AO
    assert _native._DET["wanted"] is True
    assert V._TWO_STREAMS and AO._WGRAD["on"]


def test_model_gradients_bit_identical_except_embedding_tables(det):
    """Two backward passes of the 2L/2C pre-training model on the same batch: every gradient produced by a GEMM (weights,
    biases) and by the LayerNorm / attention kernels is bit-identical; the embedding tables (scatter with atomics) are
    excluded, as documented in include/vilbert_hip.h."""
    import vilbert.vilbert as V
    from vilbert.vilbert import BertConfig, BertForMultiModalPreTraining
    cfg = synth.load_config("bert_base_2layer_2conect.json")
    sd = synth.make_state_dict(cfg, "pretraining")
    x = synth.make_inputs(cfg, 32, 36, 37, with_labels=True)
    names = ["input_ids", "image_feat", "image_loc", "token_type_ids", "attention_mask", "image_attention_mask",
             "masked_lm_labels", "image_label", "image_target", "next_sentence_label"]
    args = [x[n].to(DEV) for n in names]
    orig, V._drop_p = V._drop_p, (lambda m: 0.0)
    try:
        m = BertForMultiModalPreTraining(BertConfig.from_dict(cfg))
        m.load_state_dict(sd)
        m = m.to(DEV).train()
        grads = []
        for _ in range(2):
            m.zero_grad(set_to_none=True)
            sum(l.mean() for l in m(*args)).backward()
            torch.cuda.synchronize()
            grads.append({n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None})
        checked = 0
        for n, g in grads[0].items():
            if "embeddings.word_embeddings" in n or "position_embeddings" in n or "token_type_embeddings" in n:
                continue
            assert torch.equal(g, grads[1][n]), n
            checked += 1
        assert checked > 150
    finally:
        V._drop_p = orig
