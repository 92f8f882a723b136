"""The bf16 TRAINING kernels at the shapes bench.py TIMES them on (round-5 review, weak 1).

tests/test_bf16_stream_gpu.py stops at M = 2,368: at most 240 output tiles of 256 x 128 - fewer than the 256 persistent blocks of
`gemm_bf16_kernel`, so no case there sends a SECOND output tile through a block (the LDS ring carried across tile boundaries, the
peeled last K tile, the loaders' prefetch under the epilogue, csrc/gemm_bf16.hip `gemm_bf16_kernel`), nor the weight-gradient
kernel's XCD-aware unit order / contraction-split model at 144 - 148 contraction tiles. The timed step (`alt_gemm_modes.bf16`,
B = 256, T = 36, R = 37) launches M = 9,216 (text) and 9,472 (image, ragged last row tile) with 216 - 888 tiles. Here:

  * `vb_linear_bf16` at M = 9,216 / 9,472 for every (N, K) the encoder launches - (2304, 768) q | k | v, (3072, 768) FFN up and the
    text-side co-attention projections, (768, 3072) FFN down, (768, 768) attention output, (768, 1024) co-attention output,
    (3072, 1024) image q | k | v, (1024, 1024), (1024, 2048) region features - in every epilogue the model uses (plain, GELU +
    stored derivative, residual, dropout + residual, fp32 out) AND as the input gradient through the transposed shadow (plain,
    + residual gradient, x saved derivative). Rows checked: both sides of EVERY 256-row tile seam + the first / last rows + a
    stride of interior rows, against float64 arithmetic on the same bf16 values; every element must be finite (an unwritten
    tile shows up as the NaN the output buffer was filled with);
  * `vb_wgrad_bf16` at the same M for every weight shape, the stacked q | k | v segments and the fused bias gradient included:
    the WHOLE dW / db against float64, targets pre-filled (the kernel adds);
  * one B = 256 training step of the model (6L/6C, T = 36, R = 37, dropout off) against autograd through the CPU oracle in 4
    chunks of 64: the three losses, every parameter gradient as relative L2 per tensor (median / 90th percentile / worst
    printed and bounded).
Reference lines these launches replace: /root/reference/vilbert/vilbert.py:425-427, 501, 514, 749-751 under
`model.half()` (/root/reference/train_concap.py:443-461).
"""
import pytest
import torch

import helpers
from oracle import synth

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
BF16 = torch.bfloat16


def _rand(*shape, seed=0, scale=1.0):
    return torch.randn(*shape, generator=torch.Generator(device=DEV).manual_seed(seed), device=DEV) * scale


def _rows_to_check(M):
    rows = {0, 1, M - 2, M - 1}
    for s in range(256, M, 256):
        rows.update((s - 1, s))
    for s in range(128, M, 128):              # (the half-size blocks' seams, should a launch ever take them)
        rows.add(s)
    rows.update(range(7, M, 499))
    return torch.tensor(sorted(r for r in rows if 0 <= r < M), device=DEV)


def _check(got, rows, want64, mag64, what):
    """got [M, N] (bf16 or fp32) on the device: finite EVERYWHERE, and on `rows` within fp32 accumulation + one bf16 rounding of
    the float64 statement."""
    assert torch.isfinite(got).all(), "%s: non-finite output (an unwritten tile?)" % what
    g = got[rows].double()
    tol = 3e-6 * mag64 + 1e-5 + (want64.abs() / 256 if got.dtype == BF16 else 0.0)
    err = (g - want64).abs()
    bad = err > tol
    assert not bad.any(), "%s: %d of %d checked values off, worst err / tol %.2f at row %d" % (
        what, int(bad.sum()), bad.numel(), float((err / tol).max()), int(rows[int((err / tol).max(dim=1).values.argmax())]))
    return float((err / tol).max())


# (M, nseg, seg_n, K, GELU launch too?, fp32-out launch too?)
NT_SHAPES = [(9216, 3, 768, 768, False, False),      # text q | k | v
             (9216, 1, 3072, 768, True, False),      # text FFN up
             (9216, 1, 768, 3072, False, False),     # text FFN down
             (9216, 1, 768, 768, False, False),      # text attention output
             (9216, 3, 1024, 768, False, False),     # co-attention: text-side q | k | v
             (9216, 1, 768, 1024, False, False),     # co-attention output, text side
             (9472, 3, 1024, 1024, False, False),    # image q | k | v (37 regions: ragged last row tile)
             (9472, 1, 1024, 1024, True, False),     # image attention output / FFN
             (9472, 1, 1024, 2048, False, True)]     # region-feature projection (fp32 out)


@pytest.mark.parametrize("M,nseg,seg_n,K,gelu,f32out", NT_SHAPES)
def test_linear16_at_the_benchmarked_shapes_every_epilogue_and_dgrad(M, nseg, seg_n, K, gelu, f32out):
    from vilbert import ops, ops16
    N_ = nseg * seg_n
    tiles_f, tiles_d = ((M + 255) // 256) * (N_ // 128), ((M + 255) // 256) * (K // 128)
    assert max(tiles_f, tiles_d) > 128, "full-size persistent blocks are the point of this test"
    x = _rand(M, K, seed=1).to(BF16)
    ws = [_rand(seg_n, K, seed=10 + i, scale=0.05) for i in range(nseg)]
    bs = [_rand(seg_n, seed=20 + i) for i in range(nseg)]
    r = _rand(M, N_, seed=3).to(BF16)
    rows = _rows_to_check(M)
    w16 = torch.cat(ws).to(BF16).double()
    xr = x[rows].double()
    pre = xr @ w16.t() + torch.cat(bs).double()
    mag = xr.abs() @ w16.abs().t() + 1.0
    worst = {}
    y, _ = ops16.linear_fwd(x, ws, bs)
    worst["plain"] = _check(y, rows, pre, mag, "forward")
    y, _ = ops16.linear_fwd(x, ws, bs, None, r)
    worst["res"] = _check(y, rows, pre + r[rows].double(), mag, "forward + residual")
    seed, p = 0x5EED5EED5EED, 0.1
    y, _ = ops16.linear_fwd(x, ws, bs, None, r, drop_p=p, seed=seed)
    keep = ops.dropout(torch.ones(M, N_, device=DEV), p, seed)[rows] != 0
    worst["dropres"] = _check(y, rows, torch.where(keep, pre / (1 - p), torch.zeros_like(pre)) + r[rows].double(), mag,
                              "forward + dropout + residual")
    if gelu:
        y, d = ops16.linear_fwd(x, ws, bs, "gelu", want_act_grad=True)
        worst["gelu"] = _check(y, rows, torch.nn.functional.gelu(pre), mag, "forward + GELU")
        phi = 0.5 * (1 + torch.erf(pre / 2 ** 0.5))
        _check(d, rows, phi + pre * torch.exp(-0.5 * pre * pre) / (2 * torch.pi) ** 0.5, mag, "GELU derivative")
    if f32out:
        y, _ = ops16.linear_fwd(x, ws, bs, out_f32=True)
        assert y.dtype == torch.float32
        worst["f32"] = _check(y, rows, pre, mag, "forward, fp32 out")
    # input gradient through the transposed shadow: dX [M, K] = dY [M, N] W
    dy = _rand(M, N_, seed=4).to(BF16)
    rk, mk = _rand(M, K, seed=5).to(BF16), _rand(M, K, seed=6).to(BF16)
    dyr = dy[rows].double()
    want = dyr @ w16
    magd = dyr.abs() @ w16.abs() + 1.0
    worst["dgrad"] = _check(ops16.linear_bwd_input(dy, ws, bs, K), rows, want, magd, "dgrad")
    worst["dgrad+res"] = _check(ops16.linear_bwd_input(dy, ws, bs, K, residual=rk), rows, want + rk[rows].double(), magd,
                                "dgrad + residual")
    worst["dgrad*mul"] = _check(ops16.linear_bwd_input(dy, ws, bs, K, mul=mk), rows, want * mk[rows].double(),
                                magd * mk[rows].double().abs() + 1.0, "dgrad x multiplier")
    print("bf16 NT %dx%dx%d (%d segments): %d forward tiles (%.2f per block), %d dgrad tiles (%.2f); worst err / tolerance %s"
          % (M, N_, K, nseg, tiles_f, tiles_f / 256.0, tiles_d, tiles_d / 256.0,
             ", ".join("%s %.2f" % kv for kv in worst.items())))


WG_SHAPES = [(9216, 3, 768, 768), (9216, 1, 3072, 768), (9216, 1, 768, 3072), (9216, 1, 768, 768), (9216, 3, 1024, 768),
             (9216, 1, 768, 1024), (9472, 3, 1024, 1024), (9472, 1, 1024, 1024), (9472, 1, 1024, 2048)]


@pytest.mark.parametrize("M,nseg,seg_n,K", WG_SHAPES)
def test_wgrad16_at_the_benchmarked_shapes_whole_gradient(M, nseg, seg_n, K):
    from vilbert import ops16
    N_ = nseg * seg_n
    x = _rand(M, K, seed=1).to(BF16)
    dy = _rand(M, N_, seed=4).to(BF16)
    init_w = [_rand(seg_n, K, seed=30 + i) for i in range(nseg)]
    init_b = [_rand(seg_n, seed=40 + i) for i in range(nseg)]
    tw, tb = [t.clone() for t in init_w], [t.clone() for t in init_b]
    dws, dbs = ops16.linear_bwd_weight(dy, x, nseg, seg_n, [True] * nseg, dw_out=tw, db_out=tb)
    xd, yd = x.double(), dy.double()
    want = yd.t() @ xd                       # [N, K]: the whole gradient
    mag = yd.abs().t() @ xd.abs() + 1.0
    worst = 0.0
    for s in range(nseg):
        assert dws[s] is tw[s] and dbs[s] is tb[s]
        sl = slice(s * seg_n, (s + 1) * seg_n)
        g = dws[s].double() - init_w[s].double()
        assert torch.isfinite(g).all()
        tol = 3e-6 * mag[sl] + 2e-5 + init_w[s].double().abs() * 1e-6
        err = (g - want[sl]).abs()
        assert (err <= tol).all(), "wgrad segment %d of %dx%dx%d: worst err / tol %.2f" % (s, M, N_, K, float((err / tol).max()))
        worst = max(worst, float((err / tol).max()))
        gb = dbs[s].double() - init_b[s].double()
        wb, mb = yd[:, sl].sum(0), yd[:, sl].abs().sum(0) + 1.0
        assert ((gb - wb).abs() <= 3e-6 * mb + 2e-5).all(), "bias gradient of segment %d" % s
    # only some segments want a bias gradient, targets allocated by the launcher (zero-filled)
    dws2, dbs2 = ops16.linear_bwd_weight(dy, x, nseg, seg_n, [s == 0 for s in range(nseg)])
    assert dbs2[0] is not None and all(b is None for b in dbs2[1:])
    err = (dws2[-1].double() - want[-seg_n:]).abs()
    assert (err <= 3e-6 * mag[-seg_n:] + 2e-5).all()
    print("bf16 wgrad %dx%dx%d (%d segments, %d tiles x %d contraction tiles): worst err / tolerance %.2f"
          % (M, N_, K, nseg, (N_ // 256) * (K // 128), (M + 63) // 64, worst))


def test_wgrad16_is_bit_reproducible_in_the_deterministic_setting():
    """VB_DETERMINISTIC (default on) covers the bf16 weight gradient too (round 6): the contraction splits store partial tiles
    to the per-stream workspace and an ordered reduce adds them - two runs of the same launch agree bit for bit; with the
    setting off (fp32 atomics from every split) they are only close."""
    from vilbert import _native, ops16
    if not _native.deterministic_enabled():
        pytest.skip("deterministic split-K switched off in this process")
    M, nseg, seg_n, K = 9216, 3, 768, 768
    x = _rand(M, K, seed=1).to(BF16)
    dy = _rand(M, nseg * seg_n, seed=4).to(BF16)
    runs = []
    for _ in range(3):
        dws, dbs = ops16.linear_bwd_weight(dy, x, nseg, seg_n, [True] * nseg)
        torch.cuda.synchronize()
        runs.append([t.clone() for t in dws + dbs])
    for other in runs[1:]:
        for a, b in zip(runs[0], other):
            assert torch.equal(a, b), "bf16 weight gradient differs between two runs of the same launch"


NAMES = ["input_ids", "image_feat", "image_loc", "token_type_ids", "attention_mask", "image_attention_mask",
         "masked_lm_labels", "image_label", "image_target", "next_sentence_label"]


@pytest.mark.slow
def test_bf16_step_at_the_timed_batch_losses_and_every_gradient_vs_the_oracle():
    """`alt_gemm_modes.bf16` of bench.py: 6L/6C, B = 256, T = 36, R = 37 - here with dropout off, against autograd through the
    fp32 CPU oracle in 4 chunks of 64 (losses re-weighted to the whole-batch means, gradients summed)."""
    import vilbert.vilbert as V
    from oracle import vilbert_oracle as vo
    from vilbert import _native
    from vilbert.optim import AdamW
    from vilbert.vilbert import BertConfig, BertForMultiModalPreTraining
    B, CH = 256, 64
    cfg = synth.load_config("bert_base_6layer_6conect.json")
    sd = synth.make_state_dict(cfg, "pretraining", seed=17)
    x = synth.make_inputs(cfg, B, 36, 37, seed=17, with_labels=True)
    args = [x[n] for n in NAMES]
    leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items() if k != "cls.predictions.decoder.weight"}
    leaves["cls.predictions.decoder.weight"] = leaves["bert.embeddings.word_embeddings.weight"]
    n_lm = float((x["masked_lm_labels"] != -1).sum())
    n_img = float((x["image_label"] == 1).sum())
    n_nsp = float((x["next_sentence_label"] != -1).sum())
    want = [0.0, 0.0, 0.0]
    for lo in range(0, B, CH):
        part = [a[lo:lo + CH] for a in args]
        lm, img, nsp = vo.pretraining_forward(leaves, cfg, *part)
        w = (float((part[6] != -1).sum()) / n_lm, float((part[7] == 1).sum()) / n_img, float((part[9] != -1).sum()) / n_nsp)
        (lm.mean() * w[0] + img.mean() * w[1] + nsp.mean() * w[2]).backward()
        for i, l in enumerate((lm, img, nsp)):
            want[i] += l.mean().item() * w[i]
    prev = _native.set_gemm_mode("bf16")
    orig, V._drop_p = V._drop_p, (lambda mod: 0.0)
    try:
        net = BertForMultiModalPreTraining(BertConfig.from_dict(cfg))
        net.load_state_dict(sd)
        net = net.to(DEV).train()
        opt = AdamW(net.parameters(), lr=1e-4)          # (owns the gradient arena, as in bench.py)
        lm, img, nsp = net(*helpers.to_device(args, DEV))
        (lm.mean() + img.mean() + nsp.mean()).backward()
        torch.cuda.synchronize()
        got = [lm.mean().item(), img.mean().item(), nsp.mean().item()]
    finally:
        V._drop_p = orig
        _native.set_gemm_mode(prev)
    for name, g, w in zip(("masked_lm_loss", "masked_img_loss", "next_sentence_loss"), got, want):
        rel = abs(g - w) / abs(w)
        print("bf16 stream, 6L/6C B=256: %s %.5f vs fp32 oracle %.5f (relative %.2e)" % (name, g, w, rel))
        assert rel <= 2e-2, (name, g, w)
    ref = {n: leaves[n].grad.double() for n, p in net.named_parameters() if leaves[n].grad is not None}
    grads = {n: p.grad.detach().cpu().double() for n, p in net.named_parameters() if p.grad is not None}
    assert set(ref) == set(grads) and len(ref) > 400
    typical = sorted(g.norm().item() for g in ref.values())[len(ref) // 2]
    rel = sorted(((grads[n] - g).norm().item() / max(g.norm().item(), 1e-3 * typical), n) for n, g in ref.items())
    median, p90, worst = rel[len(rel) // 2][0], rel[len(rel) * 9 // 10][0], rel[-1][0]
    print("bf16 stream, 6L/6C B=256: gradient relative L2 error vs fp32 oracle autograd over %d tensors - median %.3e, 90th "
          "percentile %.3e, worst %.3e (%s); five worst: %s"
          % (len(rel), median, p90, worst, rel[-1][1], ", ".join("%s %.3f" % (n, e) for e, n in rel[-5:])))
    assert median <= 0.05 and p90 <= 0.15 and worst <= 0.25, (median, p90, worst)
    assert opt is not None


@pytest.mark.parametrize("M,n,K,bias", [(1628, 30522, 768, True), (389, 30522, 768, True), (400, 4100, 1024, False),
                                        (70, 30522, 768, True)])
def test_ragged_weight_gradient_of_the_wide_heads_on_the_bf16_kernel(M, n, K, bias):
    """Round 6 (review: "bf16 heads", reference vilbert.py:1178-1196): the weight gradient of an fp32-tensor linear whose width
    is no tile multiple - the tied 30,522 x 768 MLM decoder at the labelled rows of B = 256 (1,628) and B = 64 (389) - runs on
    wgrad_bf16_kernel over a bf16 copy of dY padded with zero columns to 30,720 (vb_cast_rows_f32_bf16); n_valid keeps every
    store inside the 30,522 rows. Whole dW / db against float64 on the bf16-rounded operands, targets pre-filled (the kernel
    adds), the gradient read through a PADDED row stride like the logits gradient is, guard rows behind dW untouched, two
    runs bit-identical."""
    from vilbert import _native, ops16
    prev = _native.set_gemm_mode("bf16")
    try:
        ld = (n + 3) // 4 * 4 + 8
        dy_store = _rand(M, ld, seed=M + n, scale=0.05)
        dy = dy_store[:, :n]
        dy_store[:, n:] = float("nan")                # what lies in the padding of the logits gradient must never be read
        x = _rand(M, K, seed=K + M)
        guard = 64
        store = torch.full(((n + guard) * K,), 0.25, device=DEV)
        dw = store[:n * K].view(n, K)
        db0 = torch.full((n,), -0.5, device=DEV)
        outs = []
        for run in range(2):
            store.fill_(0.25)
            db = db0.clone()
            ops16.linear_bwd_weight_ragged(dy, x, [bias], [dw], [db] if bias else None)
            torch.cuda.synchronize()
            outs.append((dw.clone(), db.clone()))
        assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1]), "two runs differ"
        assert float((store[n * K:] - 0.25).abs().max()) == 0.0, "rows past n_valid were written"
        dy64, x64 = dy.to(BF16).double(), x.to(BF16).double()
        want = dy64.t() @ x64 + 0.25
        mag = dy64.abs().t() @ x64.abs()
        err = (outs[0][0].double() - want).abs()
        tol = 3e-6 * mag + 1e-5
        assert torch.isfinite(outs[0][0]).all() and not (err > tol).any(), "dW: worst err / tol %.2f" % float((err / tol).max())
        if bias:
            wantb = dy64.sum(0) - 0.5
            errb = (outs[0][1].double() - wantb).abs()
            assert not (errb > 3e-6 * dy64.abs().sum(0) + 1e-5).any(), "db: %.3e" % float(errb.max())
        else:
            assert torch.equal(outs[0][1], db0)
    finally:
        _native.set_gemm_mode(prev)


def test_bf16_mode_sends_the_decoder_weight_gradient_to_the_bf16_kernel(monkeypatch):
    """The model in the bf16 mode: the 30,522-wide decoder's weight gradient goes through linear_bwd_weight_ragged (counted),
    and the tied word-embedding gradient stays within bf16 rounding of the fp32-tensor kernel's (VB_RAGGED off by patching)."""
    from vilbert import _native, ops16
    from vilbert.vilbert import BertConfig, BertForMultiModalPreTraining
    cfg = synth.load_config("bert_base_2layer_2conect.json")
    sd = synth.make_state_dict(cfg, "pretraining")
    names = ["input_ids", "image_feat", "image_loc", "token_type_ids", "attention_mask", "image_attention_mask",
             "masked_lm_labels", "image_label", "image_target", "next_sentence_label"]
    x = synth.make_inputs(cfg, 32, 36, 37, seed=11, with_labels=True)
    args = [x[k].to(DEV) for k in names]
    prev = _native.set_gemm_mode("bf16")
    try:
        calls = []
        real = ops16.linear_bwd_weight_ragged

        def counted(*a, **k):
            calls.append(a[0].shape)
            return real(*a, **k)

        def grads(ragged):
            monkeypatch.setattr(ops16, "linear_bwd_weight_ragged", counted)
            monkeypatch.setattr(ops16, "ragged_wgrad_ok", (lambda *a: ragged and _native.bf16_stream() and a[1] % 128 == 0
                                                           and a[0] >= 4096 and a[2] >= 64))
            m = BertForMultiModalPreTraining(BertConfig.from_dict(cfg))
            m.load_state_dict(sd)
            m = m.to(DEV).eval()                     # (dropout off: the two runs see the same function)
            sum(l.mean() for l in m(*args)).backward()
            torch.cuda.synchronize()
            return m.bert.embeddings.word_embeddings.weight.grad.clone(), m.cls.predictions.bias.grad.clone()
        gw0, gb0 = grads(False)
        assert not calls
        gw1, gb1 = grads(True)
        assert len(calls) == 1 and calls[0][-1] == cfg["vocab_size"], calls
        relw = float((gw1.double() - gw0.double()).norm() / gw0.double().norm())
        relb = float((gb1.double() - gb0.double()).norm() / gb0.double().norm())
        print("decoder weight gradient, bf16 kernel vs fp32-tensor kernel (both on bf16-rounded operands): relative L2 %.2e, "
              "bias %.2e" % (relw, relb))
        assert relw <= 2e-3 and relb <= 2e-3
    finally:
        _native.set_gemm_mode(prev)
