"""Backward kernels on the GPU: each *_bwd entry point against fp64 torch autograd of the reference
arithmetic, then whole-model gradient parity against autograd through the CPU oracle."""
import math

import pytest
import torch

import helpers
from oracle import synth, vilbert_oracle as vo

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def ops():
    from vilbert import ops as _ops
    return _ops


def _rand(*shape, seed=0, scale=1.0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale


def _close(got, want64, rtol=3e-5, atol=3e-5):
    got = got.detach().cpu().double()
    assert got.shape == want64.shape, (got.shape, want64.shape)
    assert torch.isfinite(got).all()
    scale = max(1.0, want64.abs().max().item())
    err = (got - want64).abs().max().item()
    assert err <= atol * scale + rtol * scale, "max err %.3e (scale %.3e)" % (err, scale)


@pytest.mark.parametrize("M,N,K,nseg", [(200, 256, 96, 1), (36 * 8, 768, 768, 1), (180, 128, 192, 3),
                                        (90, 64, 96, 3), (50, 20, 36, 3), (300, 1024, 5, 1), (77, 1, 64, 1)])
def test_linear_backward(ops, M, N, K, nseg):
    x = _rand(M, K, seed=1)
    ws = [_rand(N, K, seed=10 + i, scale=0.1) for i in range(nseg)]
    dy = _rand(M, nseg * N, seed=3)
    dx = ops.linear_bwd_input(dy.cuda(), [w.cuda() for w in ws], K)
    _close(dx, dy.double() @ torch.cat(ws, 0).double())
    dws, dbs = ops.linear_bwd_weight(dy.cuda(), x.cuda(), nseg, N, [True] * nseg)
    for s in range(nseg):
        seg = dy[:, s * N:(s + 1) * N].double()
        _close(dws[s], seg.t() @ x.double())
        _close(dbs[s], seg.sum(0))


def test_linear_backward_strided_input_rows(ops):
    # wgrad reading the first-token rows in place (pooler backward)
    B, S, H, N = 6, 9, 64, 48
    h, dy = _rand(B, S, H, seed=1).cuda(), _rand(B, N, seed=2)
    (dw,), (db,) = ops.linear_bwd_weight(dy.cuda(), h[:, 0], 1, N, [True])
    _close(dw, dy.double().t() @ h[:, 0].cpu().double())
    _close(db, dy.double().sum(0))


@pytest.mark.parametrize("act", ["gelu", "relu"])
def test_act_backward(ops, act):
    pre, dy = _rand(64, 96, seed=1, scale=2.0), _rand(64, 96, seed=2)
    p64 = pre.double().requires_grad_(True)
    out = p64 * 0.5 * (1.0 + torch.erf(p64 / math.sqrt(2.0))) if act == "gelu" else torch.relu(p64)
    out.backward(dy.double())
    _close(ops.act_bwd(dy.cuda(), pre.cuda(), act), p64.grad)


def test_dropout_mask_is_a_function_of_seed_and_index(ops):
    x = torch.ones(1000, 999).cuda()
    y1, y2, y3 = ops.dropout(x, 0.1, 1234), ops.dropout(x, 0.1, 1234), ops.dropout(x, 0.1, 1235)
    assert torch.equal(y1, y2) and not torch.equal(y1, y3)
    keep = (y1 != 0).float().mean().item()
    assert abs(keep - 0.9) < 2e-3
    assert torch.allclose(y1[y1 != 0], torch.tensor(1.0 / 0.9, device=DEV))
    # backward = the same launch on the gradient; residual is added un-masked
    g = _rand(1000, 999, seed=5).cuda()
    r = _rand(1000, 999, seed=6).cuda()
    assert torch.equal(ops.dropout(g, 0.1, 1234), g * y1)
    assert torch.allclose(ops.dropout(g, 0.1, 1234, r), g * y1 + r, rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("rows,cols", [(37, 64), (300, 768), (129, 1024), (5, 2048), (9, 4096), (7, 8192), (6, 5000)])
def test_layernorm_backward(ops, rows, cols):
    x, dy = _rand(rows, cols, seed=1, scale=2.0) + 0.3, _rand(rows, cols, seed=2)
    g, b = 1 + 0.1 * _rand(cols, seed=3), 0.1 * _rand(cols, seed=4)
    y, mean, rstd = ops.layernorm_fwd(x.cuda(), g.cuda(), b.cuda(), 1e-12, None, want_stats=True)
    dx, dg, db = ops.layernorm_bwd(dy.cuda(), x.cuda(), mean, rstd, g.cuda())
    x64, g64, b64 = (t.double().requires_grad_(True) for t in (x, g, b))
    vo.layer_norm(x64, g64, b64).backward(dy.double())
    _close(dx, x64.grad)
    _close(dg, g64.grad)
    _close(db, b64.grad)


def _attn_ref(q, k, v, madd, heads, keep=None, p=0.0):
    B, Sq, H = q.shape
    d = H // heads
    sp = lambda t: t.view(t.shape[0], t.shape[1], heads, d).permute(0, 2, 1, 3)
    s = sp(q) @ sp(k).transpose(-1, -2) / math.sqrt(d) + madd.view(B, 1, 1, -1)
    pr = torch.softmax(s, -1)
    if keep is not None:
        pr = pr * keep / (1.0 - p)
    return (pr @ sp(v)).permute(0, 2, 1, 3).reshape(B, Sq, H)


@pytest.mark.parametrize("heads,d,Sq,Sk,p", [(12, 64, 36, 36, 0.0), (8, 128, 36, 37, 0.0), (2, 32, 9, 7, 0.0),
                                             (4, 64, 50, 101, 0.0), (2, 128, 20, 300, 0.0),
                                             (8, 128, 36, 36, 0.1), (3, 32, 17, 40, 0.3)])
def test_attention_backward(ops, heads, d, Sq, Sk, p):
    B, H = 3, heads * d
    g = torch.Generator().manual_seed(9)
    q, k, v = torch.randn(B, Sq, H, generator=g), torch.randn(B, Sk, H, generator=g), torch.randn(B, Sk, H, generator=g)
    lens = torch.randint(1, Sk + 1, (B,), generator=g)
    madd = (1.0 - (torch.arange(Sk)[None] < lens[:, None]).float()) * -10000.0
    d_out = torch.randn(B, Sq, H, generator=g)
    qd, kd, vd = q.cuda(), k.cuda(), v.cuda()
    seed = 4242
    out, probs, lse = ops.attention_fwd(qd, kd, vd, madd.cuda(), heads, True, True, p, seed)
    keep = (probs != 0).double().cpu() if p > 0 else None
    if p > 0:
        rate = keep[..., :1].numel() and keep.mean().item()
        assert 0.0 < rate < 1.0
    q64, k64, v64 = (t.double().requires_grad_(True) for t in (q, k, v))
    ref = _attn_ref(q64, k64, v64, madd.double(), heads, keep, p)
    _close(out, ref.detach())
    ref.backward(d_out.double())
    dq, dk, dv = torch.empty_like(qd), torch.empty_like(kd), torch.empty_like(vd)
    ops.attention_bwd(d_out.cuda(), qd, kd, vd, madd.cuda(), heads, lse, dq, dk, dv, p, seed)
    _close(dq, q64.grad)
    _close(dk, k64.grad)
    _close(dv, v64.grad)


def test_attention_backward_into_fused_buffers(ops):
    # q/k/v are column slices of one projection and the gradients land in one buffer, each slice once
    heads, d, S, B = 4, 64, 20, 2
    H = heads * d
    qkv = _rand(B, S, 3 * H, seed=3)
    d_out = _rand(B, S, H, seed=4)
    qd = qkv.cuda()
    sl = lambda t: (t[..., :H], t[..., H:2 * H], t[..., 2 * H:])
    out, _, lse = ops.attention_fwd(*sl(qd), None, heads, False, True)
    dqkv = torch.full_like(qd, float("nan"))
    ops.attention_bwd(d_out.cuda(), *sl(qd), None, heads, lse, *sl(dqkv))
    q64 = qkv.double().requires_grad_(True)
    _attn_ref(*sl(q64), torch.zeros(B, S).double(), heads).backward(d_out.double())
    _close(dqkv, q64.grad)


def _grad_parity(cfg, kind, batch, n_tok, n_reg, seed=5):
    from vilbert.vilbert import BertConfig, BertForMultiModalPreTraining, VILBertForVLTasks
    sd = synth.make_state_dict(cfg, kind, seed=seed)
    x = synth.make_inputs(cfg, batch, n_tok, n_reg, seed=seed, with_labels=(kind == "pretraining"),
                          task_id=3 if cfg["task_specific_tokens"] else None)
    c = BertConfig.from_dict(cfg)
    model = VILBertForVLTasks(c, num_labels=1) if kind == "vltasks" else BertForMultiModalPreTraining(c)
    model.load_state_dict(sd)
    model = model.to(DEV).train()
    leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items() if k != "cls.predictions.decoder.weight"}
    leaves["cls.predictions.decoder.weight"] = leaves["bert.embeddings.word_embeddings.weight"]
    if kind == "pretraining":
        args = (x["input_ids"], x["image_feat"], x["image_loc"], x["token_type_ids"], x["attention_mask"],
                x["image_attention_mask"], x["masked_lm_labels"], x["image_label"], x["image_target"],
                x["next_sentence_label"])
        got = model(*helpers.to_device(args, DEV))
        want = vo.pretraining_forward(leaves, cfg, *args)
        loss_g, loss_w = sum(l.sum() for l in got), sum(l.sum() for l in want)
    else:
        args = (x["input_ids"], x["image_feat"], x["image_loc"], x["token_type_ids"], x["attention_mask"],
                x["image_attention_mask"], x["co_attention_mask"])
        if "task_ids" in x:
            args = args + (x["task_ids"],)
        got = model(*helpers.to_device(args, DEV))[:9]
        want = vo.vltasks_forward(leaves, cfg, *args)
        # a loss touching every head; vision_logit's -10000 entries are excluded through the mask
        m = x["image_attention_mask"].float().unsqueeze(2)
        w8 = [1.0, 0.7, 1.3, 0.9, 1.1, 0.01, None, 0.002, 1.0]
        loss_g = sum((w * o).sum() for w, o in zip(w8, got) if w is not None) + (got[6] * m.to(DEV)).sum()
        loss_w = sum((w * o).sum() for w, o in zip(w8, want) if w is not None) + (want[6] * m).sum()
    helpers.assert_close(loss_g, loss_w, "loss", atol=1e-4 * max(1.0, abs(loss_w.item())))
    loss_g.backward()
    loss_w.backward()
    # Criterion per parameter tensor: |got - ref| <= 2e-4 * max|ref| + 2e-7 * (largest gradient entry in
    # the model). The second term only matters for gradients that are pure rounding noise in BOTH
    # implementations (e.g. key.bias: softmax is invariant to a constant shift of the keys, so its true
    # gradient is 0). Returns the worst error relative to that bound.
    gmax = max(v.grad.abs().max().item() for v in leaves.values() if v.grad is not None)
    worst = 0.0
    for name, p in model.named_parameters():
        ref = leaves[name].grad
        if ref is None:
            assert p.grad is None or p.grad.abs().max().item() == 0.0, name
            continue
        assert p.grad is not None, name
        scale = ref.abs().max().item()
        err = (p.grad.cpu().double() - ref.double()).abs().max().item()
        bound = 2e-4 * scale + 2e-7 * gmax
        assert err <= bound, "%s: grad err %.3e > bound %.3e (max|ref| %.3e, model max %.3e)" % (
            name, err, bound, scale, gmax)
        worst = max(worst, err / bound)
    return worst


NO_DROPOUT = dict(hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0, v_hidden_dropout_prob=0.0,
                  v_attention_probs_dropout_prob=0.0)


@pytest.mark.parametrize("kind", ["pretraining", "vltasks"])
@pytest.mark.parametrize("over", [{}, {"task_specific_tokens": True}, {"dynamic_attention": True, "fusion_method": "sum"},
                                  {"model": "roberta", "type_vocab_size": 1},    # roberta_base_6layer_6connect.json
                                  {"hidden_act": "swish", "v_hidden_act": "swish"}])   # ACT2FN["swish"], vilbert.py:120-121
def test_model_gradients_match_oracle_autograd(kind, over):
    if kind == "pretraining" and over.get("task_specific_tokens"):
        pytest.skip("the pre-training wrapper does not pass task ids (reference vilbert.py:1486-1495)")
    cfg = synth.tiny_config(**NO_DROPOUT, **over)
    # the wrappers hard-code nn.Dropout(0.1) on the pooled output (vilbert.py:1226,1606), so train mode
    # is made deterministic by forcing every effective dropout probability to 0
    assert _grad_parity_no_dropout(cfg, kind) <= 1.0


def _grad_parity_no_dropout(cfg, kind):
    import vilbert.vilbert as V
    orig = V._drop_p
    V._drop_p = lambda m: 0.0
    try:
        return _grad_parity(cfg, kind, 4, 9, 8)
    finally:
        V._drop_p = orig


def test_model_gradients_base_2l2c():
    import vilbert.vilbert as V
    cfg = dict(synth.load_config("bert_base_2layer_2conect.json"), **NO_DROPOUT)
    orig = V._drop_p
    V._drop_p = lambda m: 0.0
    try:
        assert _grad_parity(cfg, "pretraining", 4, 20, 37) <= 1.0
    finally:
        V._drop_p = orig


@pytest.mark.parametrize("cfgname,batch", [("bert_base_6layer_6conect.json", 4), ("bert_large_6layer_6conect.json", 2)])
def test_model_gradients_north_star_configs(cfgname, batch):
    """Every parameter gradient of the train_concap objective (T = 36, R = 36 + 1 global row, loader label
    conventions) against autograd through the CPU oracle, on the north-star model (BASELINE.json configs[1-2])
    and on bert_large_6layer_6conect (configs[3]: H = 1024, 16 x 64 heads, I = 4096, 24 text layers)."""
    import vilbert.vilbert as V
    cfg = dict(synth.load_config(cfgname), **NO_DROPOUT)
    orig = V._drop_p
    V._drop_p = lambda m: 0.0
    try:
        assert _grad_parity(cfg, "pretraining", batch, 36, 37) <= 1.0
    finally:
        V._drop_p = orig


def test_model_gradients_large_vltasks():
    """bert_large multi-task wrapper (BASELINE.json configs[3]): gradients of a loss over every head."""
    import vilbert.vilbert as V
    cfg = dict(synth.load_config("bert_large_6layer_6conect.json"), **NO_DROPOUT)
    orig = V._drop_p
    V._drop_p = lambda m: 0.0
    try:
        assert _grad_parity(cfg, "vltasks", 2, 24, 101) <= 1.0      # real task shape: T = 23 + task token, R = 101
    finally:
        V._drop_p = orig


def test_training_mode_with_dropout_runs_and_is_seeded():
    from vilbert.vilbert import BertConfig, BertForMultiModalPreTraining
    cfg = synth.tiny_config()
    sd = synth.make_state_dict(cfg, "pretraining")
    x = synth.make_inputs(cfg, 6, 9, 8, with_labels=True)
    args = helpers.to_device((x["input_ids"], x["image_feat"], x["image_loc"], x["token_type_ids"],
                              x["attention_mask"], x["image_attention_mask"], x["masked_lm_labels"],
                              x["image_label"], x["image_target"], x["next_sentence_label"]), DEV)
    model = BertForMultiModalPreTraining(BertConfig.from_dict(cfg))
    model.load_state_dict(sd)
    model = model.to(DEV).train()
    losses = []
    for _ in range(3):
        model.zero_grad()
        out = model(*args)
        loss = sum(l.sum() for l in out)
        loss.backward()
        losses.append(loss.item())
        for n, p in model.named_parameters():
            if "q_dense" in n:
                assert p.grad is None, n
            else:
                assert p.grad is not None and torch.isfinite(p.grad).all(), n
    assert all(math.isfinite(l) for l in losses)
    assert len(set(losses)) == 3  # a fresh dropout mask every step
    model.eval()
    with torch.no_grad():
        a, b = model(*args), model(*args)
    assert all(torch.equal(u, v) for u, v in zip(a, b))


def test_dropout_fused_into_gemm_epilogue_uses_the_same_mask(ops):
    # dropout(dense(x)) + residual as ONE launch == GEMM followed by the standalone dropout(+residual) kernel
    M, N, K = 300, 256, 96   # interior tiles (fast epilogue) and ragged edge tiles (generic epilogue)
    x, w, b, r = _rand(M, K, seed=1).cuda(), _rand(N, K, seed=2, scale=0.1).cuda(), _rand(N, seed=3).cuda(), \
        _rand(M, N, seed=4).cuda()
    fused, _ = ops.linear_fwd(x, [w], [b], residual=r, drop_p=0.25, seed=777)
    plain, _ = ops.linear_fwd(x, [w], [b])
    two_step = ops.dropout(plain, 0.25, 777, r)
    assert torch.allclose(fused, two_step, rtol=1e-6, atol=1e-6)
    kept = ((fused - r).abs() > 0).float().mean().item()
    assert abs(kept - 0.75) < 0.02


def test_hip_graph_capture_and_replay_of_the_forward():
    """Every launcher enqueues on torch's current stream and never allocates or synchronises behind torch's
    back, so a whole forward can be captured into a HIP graph (launch-bound small-batch inference)."""
    from vilbert.vilbert import BertConfig, VILBertForVLTasks
    cfg = synth.tiny_config()
    sd = synth.make_state_dict(cfg, "vltasks")
    model = VILBertForVLTasks(BertConfig.from_dict(cfg), num_labels=1)
    model.load_state_dict(sd)
    model = model.eval().to(DEV)
    x = synth.make_inputs(cfg, 4, 9, 7)
    names = ["input_ids", "image_feat", "image_loc", "token_type_ids", "attention_mask", "image_attention_mask",
             "co_attention_mask"]
    static = [x[n].to(DEV) for n in names]
    with torch.no_grad():
        eager = [o.clone() for o in model(*static)[:9]]
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            model(*static)                      # warm-up on the capture stream
        torch.cuda.current_stream().wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            captured = model(*static)[:9]
        # new inputs through the static buffers, replayed without any Python-side launch
        y = synth.make_inputs(cfg, 4, 9, 7, seed=99)
        for buf, n in zip(static, names):
            buf.copy_(y[n].to(DEV))
        graph.replay()
        torch.cuda.synchronize()
        replayed = [o.clone() for o in captured]
        fresh = model(*static)[:9]
    for a, b2 in zip(replayed, fresh):
        assert torch.equal(a, b2)
    assert not all(torch.equal(a, e) for a, e in zip(replayed, eager))   # the inputs did change


def test_fixed_layers_run_without_grad_and_train_the_rest():
    # fixed_t_layer prefix under no_grad (reference vilbert.py:968-995)
    from vilbert.vilbert import BertConfig, VILBertForVLTasks
    cfg = synth.tiny_config(fixed_t_layer=1, **NO_DROPOUT)   # text layer 0 frozen; t_biattention_id = [1, 2]
    sd = synth.make_state_dict(cfg, "vltasks")
    x = synth.make_inputs(cfg, 3, 6, 5)
    model = VILBertForVLTasks(BertConfig.from_dict(cfg), num_labels=1)
    model.load_state_dict(sd)
    model = model.to(DEV).train()
    args = helpers.to_device((x["input_ids"], x["image_feat"], x["image_loc"], x["token_type_ids"],
                              x["attention_mask"], x["image_attention_mask"], x["co_attention_mask"]), DEV)
    out = model(*args)
    (out[0].sum() + out[7].sum() * 1e-3).backward()
    params = dict(model.named_parameters())
    frozen = [n for n in params if n.startswith("bert.encoder.layer.0.")]
    assert frozen and all(params[n].grad is None for n in frozen)
    assert model.bert.encoder.layer[1].attention.self.query.weight.grad is not None
    # the text embeddings feed only the frozen prefix, so they receive no gradient through the encoder
    assert model.bert.embeddings.position_embeddings.weight.grad is None
    assert model.bert.v_embeddings.image_embeddings.weight.grad is not None


def test_gradient_side_dropout_rides_on_the_layernorm_backward(per_op_path):
    """Round 3: the dropout mask of the dense layer in front of a LayerNorm is applied by the LayerNorm backward kernel
    itself (vb_layernorm_bwd_drop) and handed to the dense node through a tensor tag. With dropout ON: (a) the
    stand-alone vb_dropout launches of backward all but disappear, (b) every gradient is BIT-identical to the path
    without the hand-over (same mask function, same arithmetic)."""
    import vilbert.autograd_ops as AO
    from vilbert import _native
    from vilbert.vilbert import BertConfig, BertForMultiModalPreTraining
    was_det = _native._DET["wanted"]
    _native.set_deterministic(True)          # bit-identity needs the ordered split-K reduce
    cfg = synth.load_config("bert_base_2layer_2conect.json")
    sd = synth.make_state_dict(cfg, "pretraining")
    x = synth.make_inputs(cfg, 8, 20, 37, with_labels=True)
    names = ["input_ids", "image_feat", "image_loc", "token_type_ids", "attention_mask", "image_attention_mask",
             "masked_lm_labels", "image_label", "image_target", "next_sentence_label"]
    args = [x[n].to(DEV) for n in names]
    calls = {"n": 0}
    real_dropout = AO.ops.dropout

    def counting(*a, **k):
        calls["n"] += 1
        return real_dropout(*a, **k)

    def run(handover):
        tag = AO._tag_drop
        if not handover:
            AO._tag_drop = lambda y, p, s: y
        AO.ops.dropout = counting
        try:
            m = BertForMultiModalPreTraining(BertConfig.from_dict(cfg))
            m.load_state_dict(sd)
            m = m.to(DEV).train()
            AO._seed_counter = __import__("itertools").count(1)      # the same dropout seeds in both runs
            loss = sum(l.mean() for l in m(*args))
            calls["n"] = 0
            loss.backward()
            torch.cuda.synchronize()
            return calls["n"], {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}
        finally:
            AO._tag_drop = tag
            AO.ops.dropout = real_dropout
    try:
        n_off, g_off = run(False)
        n_on, g_on = run(True)
    finally:
        _native.set_deterministic(was_det)
    assert n_off >= 12 and n_on <= n_off - 12, (n_off, n_on)
    skip = ("word_embeddings", "position_embeddings", "token_type_embeddings")     # scatter with atomics
    for n in g_off:
        if not any(k in n for k in skip):
            assert torch.equal(g_off[n], g_on[n]), n


def test_dropout_twin_is_dropped_after_an_in_place_write():
    """Round-3 advisor: the hand-over tag (`_vb_dropped`) rides on a Python attribute of the gradient tensor; if autograd
    ever accumulated a second consumer's gradient IN PLACE into that tensor the stale twin would still be used. The tag
    now carries the tensor's address and version counter; after an in-place write `_dropped` recomputes."""
    import vilbert.autograd_ops as AO
    from vilbert import ops
    g = torch.Generator().manual_seed(5)
    dy = torch.randn(64, 768, generator=g).to(DEV)
    p, seed = 0.1, 1234
    twin = ops.dropout(dy, p, seed)
    dy._vb_dropped = (p, seed, twin, dy.data_ptr(), dy._version)
    assert AO._dropped(dy, (p, seed)) is twin                       # untouched: the twin is handed over
    assert AO._dropped(dy, (p, seed + 1)) is not twin               # another mask: recomputed
    dy.add_(1.0)                                                    # what an in-place accumulation would do
    got = AO._dropped(dy, (p, seed))
    assert got is not twin
    assert torch.equal(got, ops.dropout(dy, p, seed))
