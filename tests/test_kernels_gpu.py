"""Per-kernel parity on the GPU: each C-ABI entry point against an fp64 torch restatement of the
reference arithmetic it replaces (tolerances are fp32-roundoff class, far inside the 1e-4 bar)."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    from vilbert import ops as _ops
    return _ops


def _rand(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale)


def _gelu64(x):
    return x * 0.5 * (1.0 + torch.erf(x / math.sqrt(2.0)))


def _close(got, want64, rtol=2e-5, atol=2e-5):
    got = got.detach().cpu().double()
    err = (got - want64).abs()
    bound = atol + rtol * want64.abs()
    assert got.shape == want64.shape
    assert torch.isfinite(got).all()
    assert (err <= bound).all(), "max err %.3e (max |ref| %.3e)" % (err.max().item(), want64.abs().max().item())


@pytest.mark.parametrize("M,N,K", [
    (128, 128, 32),      # exactly one tile, one k step
    (256, 384, 768),     # several tiles
    (36 * 8, 768, 768),  # config-1 text shape, ragged M
    (100, 200, 52),      # ragged everywhere (K % 32 != 0, N % 128 != 0)
    (1, 1, 4),           # degenerate
    (77, 3129, 64),      # wide ragged N (VQA head width)
    (130, 1024, 2048),   # image embedding K
])
def test_linear_plain(ops, M, N, K):
    x, w, b = _rand(M, K, seed=1), _rand(N, K, seed=2, scale=0.05), _rand(N, seed=3)
    y, _ = ops.linear_fwd(x.cuda(), [w.cuda()], [b.cuda()])
    _close(y, x.double() @ w.double().t() + b.double())


@pytest.fixture
def gemm_tile():
    """Forces a tile of the second-generation GEMM kernel (or the round-1 kernel) for one test."""
    from vilbert import _native
    prev = _native.set_gemm_tile(0)

    def use(code):
        _native.set_gemm_tile(code)
    yield use
    _native.set_gemm_tile(prev)


@pytest.mark.parametrize("code", [-1, 0, 22, 33, 34, 43, 44, 434, 433, 324, 323])
@pytest.mark.parametrize("M,N,K,nseg", [(384, 384, 64, 1), (36 * 11, 768, 16, 1), (300, 260, 48, 1), (96, 96, 16 * 7, 3),
                                        (1000, 128, 256, 3), (50, 20, 32, 1), (37 * 32, 256, 64, 1)])
def test_linear_forward_dgrad_wgrad_every_tile(ops, gemm_tile, code, M, N, K, nseg):
    """Every tile shape of the second-generation kernel (forced) and the round-1 kernel on aligned shapes with full,
    ragged-M, ragged-N tiles, 1..7 K steps (prologue / steady state / tail of the K loop) and stacked segments:
    forward, dgrad and wgrad (+ fused bias gradient) against fp64."""
    gemm_tile(code)
    x = _rand(M, K, seed=1)
    ws = [_rand(N, K, seed=10 + i, scale=0.1) for i in range(nseg)]
    bs = [_rand(N, seed=20 + i) for i in range(nseg)]
    dy = _rand(M, nseg * N, seed=3)
    xd, wd = x.cuda(), [w.cuda() for w in ws]
    y, _ = ops.linear_fwd(xd, wd, [b.cuda() for b in bs])
    _close(y, torch.cat([x.double() @ w.double().t() + b.double() for w, b in zip(ws, bs)], 1), 3e-5, 3e-5)
    dx = ops.linear_bwd_input(dy.cuda(), wd, K)
    _close(dx, dy.double() @ torch.cat(ws, 0).double(), 3e-5, 3e-5 * max(1.0, N * nseg / 256))
    dws, dbs = ops.linear_bwd_weight(dy.cuda(), xd, nseg, N, [True] * nseg)
    for s in range(nseg):
        seg = dy[:, s * N:(s + 1) * N].double()
        want = seg.t() @ x.double()
        _close(dws[s], want, 3e-5, 3e-5 * max(1.0, want.abs().max().item()))
        _close(dbs[s], seg.sum(0), 3e-5, 3e-5 * max(1.0, M / 64))


@pytest.mark.parametrize("code", [-1, 0, 33, 44])
@pytest.mark.parametrize("M,N,K", [(192, 384, 64), (100, 200, 52)])   # aligned (fused epilogue) and ragged (post-pass)
def test_linear_activation_derivative_and_multiplier_epilogues(ops, gemm_tile, code, M, N, K):
    """act_grad (forward stores gelu'(pre-activation)) and mul (dgrad multiplies its result) - the fused GELU
    backward of the feed-forward blocks - in the second-generation kernel and through the round-1 kernel's post-passes."""
    gemm_tile(code)
    x, w, b = _rand(M, K, seed=7), _rand(N, K, seed=8, scale=0.1), _rand(N, seed=9)
    y, d = ops.linear_fwd(x.cuda(), [w.cuda()], [b.cuda()], act="gelu", want_act_grad=True)
    pre = (x.double() @ w.double().t() + b.double()).requires_grad_(True)
    act = _gelu64(pre)
    act.sum().backward()
    _close(y, act.detach())
    _close(d, pre.grad)
    dy, m = _rand(M, N, seed=11), _rand(M, K, seed=12)
    dx = ops.linear_bwd_input(dy.cuda(), [w.cuda()], K, mul=m.cuda())
    _close(dx, (dy.double() @ w.double()) * m.double(), 3e-5, 3e-5)


@pytest.mark.parametrize("M,N,K", [(424, 1002, 256), (1628, 3054, 768), (300, 8190, 256), (1628, 30522, 768)])
def test_linear_padded_logits_path_ragged_contraction(ops, M, N, K):
    """The MLM-decoder shape class: out-features N with N % 4 == 2 (30522), a row count that is no multiple of 16. The
    logits live in a buffer whose row stride is rounded up to 4 floats (pad_cols); the backward GEMMs read the strided
    gradient in place and split their contraction into an aligned bulk (second-generation kernel) + a <= 15-element
    tail launch. With >= 4096 out-features and a small dX (the last two cases, incl. the real decoder shape) the dgrad
    bulk is additionally cut along K into atomically combined splits."""
    x, w, b = _rand(M, K, seed=1), _rand(N, K, seed=2, scale=0.05), _rand(N, seed=3)
    y, _ = ops.linear_fwd(x.cuda(), [w.cuda()], [b.cuda()], pad_cols=True)
    assert y.shape == (M, N) and y.stride() == ((N + 3) // 4 * 4, 1)
    _close(y, x.double() @ w.double().t() + b.double())
    dy_buf = torch.full((M, (N + 3) // 4 * 4), float("nan"))      # NaN in the pad columns: they must never be read
    dy_buf[:, :N] = _rand(M, N, seed=4)
    dy = dy_buf.cuda()[:, :N]
    dx = ops.linear_bwd_input(dy, [w.cuda()], K)
    _close(dx, dy_buf[:, :N].double() @ w.double(), 3e-5, 3e-5 * N / 256)
    (dw,), (db,) = ops.linear_bwd_weight(dy, x.cuda(), 1, N, [True])
    want = dy_buf[:, :N].double().t() @ x.double()
    _close(dw, want, 3e-5, 3e-5 * max(1.0, want.abs().max().item()))
    _close(db, dy_buf[:, :N].double().sum(0), 3e-5, 3e-5 * M / 64)


def test_cross_entropy_on_padded_rows_keeps_the_row_stride():
    from vilbert import ops as O
    rows, n = 37, 1002
    buf = torch.full((rows, 1004), float("nan"))
    buf[:, :n] = _rand(rows, n, seed=5)
    logits = buf.cuda()[:, :n]
    labels = torch.randint(0, n, (rows,), generator=torch.Generator().manual_seed(1))
    labels[::5] = -1
    loss, lse, count = O.xent_fwd(logits, labels.cuda(), -1)
    ref = torch.nn.functional.cross_entropy(buf[:, :n].double(), labels, ignore_index=-1)
    _close(loss, ref, 1e-5, 1e-5)
    d = O.xent_bwd(torch.ones(1).cuda(), logits, labels.cuda(), -1, lse, count)
    assert d.stride() == (1004, 1)
    l64 = buf[:, :n].double().requires_grad_(True)
    torch.nn.functional.cross_entropy(l64, labels, ignore_index=-1).backward()
    _close(d, l64.grad, 1e-5, 1e-6)


def test_linear_is_transpose_detecting(ops):
    # asymmetric A = I-like check: y = x @ w.T with x = one-hot rows picks rows of w.T
    K, N = 64, 160
    x = torch.zeros(K, K)
    x[torch.arange(K), torch.arange(K)] = 1.0
    w = torch.arange(N * K, dtype=torch.float32).reshape(N, K) / 1000.0
    y, _ = ops.linear_fwd(x.cuda(), [w.cuda()], [None])
    _close(y, w.double().t(), 1e-6, 1e-6)


def test_linear_unaligned_k_scalar_path(ops):
    # K = 5 (image location embeddings) takes the scalar loader
    x, w, b = _rand(73, 5, seed=4), _rand(96, 5, seed=5), _rand(96, seed=6)
    y, _ = ops.linear_fwd(x.cuda(), [w.cuda()], [b.cuda()])
    _close(y, x.double() @ w.double().t() + b.double())


@pytest.mark.parametrize("act", ["gelu", "relu", "swish"])
def test_linear_act_residual_preact(ops, act):
    M, N, K = 200, 256, 96
    x, w, b, r = _rand(M, K, seed=7), _rand(N, K, seed=8, scale=0.1), _rand(N, seed=9), _rand(M, N, seed=10)
    y, pre = ops.linear_fwd(x.cuda(), [w.cuda()], [b.cuda()], act=act, residual=r.cuda(), want_preact=True)
    pre64 = x.double() @ w.double().t() + b.double()
    act64 = _gelu64(pre64) if act == "gelu" else (torch.relu(pre64) if act == "relu" else pre64 * torch.sigmoid(pre64))
    _close(pre, pre64)
    _close(y, act64 + r.double())
    # activation derivative (saved for the fused backward) and the stand-alone activation backward
    _, d = ops.linear_fwd(x.cuda(), [w.cuda()], [b.cuda()], act=act, want_act_grad=True)
    p64 = pre64.clone().requires_grad_(True)
    (_gelu64(p64) if act == "gelu" else (torch.relu(p64) if act == "relu" else p64 * torch.sigmoid(p64))).sum().backward()
    _close(d, p64.grad)
    dy = _rand(M, N, seed=12)
    _close(ops.act_bwd(dy.cuda(), pre, act), dy.double() * p64.grad)


@pytest.mark.parametrize("H", [128, 64, 96, 20])
def test_linear_three_segments_qkv(ops, H):
    M, K = 36 * 5, 192
    x = _rand(M, K, seed=11)
    ws = [_rand(H, K, seed=20 + i, scale=0.1) for i in range(3)]
    bs = [_rand(H, seed=30 + i) for i in range(3)]
    y, _ = ops.linear_fwd(x.cuda(), [w.cuda() for w in ws], [b.cuda() for b in bs])
    want = torch.cat([x.double() @ w.double().t() + b.double() for w, b in zip(ws, bs)], dim=1)
    _close(y, want)


def test_linear_first_token_row_stride(ops):
    B, S, H, N = 6, 9, 64, 48
    h = _rand(B, S, H, seed=12).cuda()
    w, b = _rand(N, H, seed=13), _rand(N, seed=14)
    y, _ = ops.linear_fwd(h[:, 0], [w.cuda()], [b.cuda()], act="relu")
    _close(y, torch.relu(h[:, 0].cpu().double() @ w.double().t() + b.double()))


def test_linear_errors(ops):
    x = _rand(4, 8).cuda()
    with pytest.raises(RuntimeError):
        ops.linear_fwd(x, [_rand(4, 7).cuda()], [None])
    with pytest.raises(RuntimeError):
        ops.linear_fwd(x.cpu(), [_rand(4, 8).cuda()], [None])
    with pytest.raises(RuntimeError):  # at most 4 weight segments
        ops.linear_fwd(x, [_rand(4, 8).cuda()] * 5, [None] * 5)


def _ln64(x, g, b, eps=1e-12):
    u = x.mean(-1, keepdim=True)
    s = (x - u).pow(2).mean(-1, keepdim=True)
    return g * ((x - u) / torch.sqrt(s + eps)) + b


@pytest.mark.parametrize("cols", [64, 96, 768, 1024, 2048, 4096])
def test_layernorm(ops, cols):
    x, x2 = _rand(37, cols, seed=1, scale=3.0) + 0.7, _rand(37, cols, seed=2)
    g, b = 1 + 0.1 * _rand(cols, seed=3), 0.1 * _rand(cols, seed=4)
    y, mean, rstd = ops.layernorm_fwd(x.cuda(), g.cuda(), b.cuda(), 1e-12, x2.cuda(), want_stats=True)
    s = x.double() + x2.double()
    _close(y, _ln64(s, g.double(), b.double()))
    _close(mean, s.mean(-1))
    _close(rstd, 1.0 / torch.sqrt(s.var(-1, unbiased=False) + 1e-12), 1e-5, 1e-6)
    y1, _, _ = ops.layernorm_fwd(x.cuda(), g.cuda(), b.cuda(), 1e-12)
    _close(y1, _ln64(x.double(), g.double(), b.double()))


def test_layernorm_constant_row_is_finite(ops):
    x = torch.full((3, 64), 2.5)
    g, b = torch.ones(64), torch.zeros(64)
    y, _, _ = ops.layernorm_fwd(x.cuda(), g.cuda(), b.cuda(), 1e-12)
    assert torch.isfinite(y).all() and y.abs().max().item() == 0.0


@pytest.mark.parametrize("task", [False, True])
def test_text_embedding(ops, task):
    B, T, H, V = 5, 11, 96, 50
    g0 = torch.Generator().manual_seed(3)
    ids = torch.randint(0, V, (B, T), generator=g0)
    seg = torch.randint(0, 2, (B, T), generator=g0)
    word, pos, typ = _rand(V, H, seed=1), _rand(40, H, seed=2), _rand(2, H, seed=3)
    temb, tid = _rand(20, H, seed=4), torch.randint(0, 20, (B, 1), generator=g0)
    g, b = 1 + 0.1 * _rand(H, seed=5), 0.1 * _rand(H, seed=6)
    out = ops.text_embed_ln_fwd(ids.cuda(), seg.cuda(), word.cuda(), pos.cuda(), typ.cuda(), g.cuda(),
                                      b.cuda(), 1e-12, tid.cuda() if task else None, temb.cuda() if task else None)[0]
    e = word.double()[ids] + pos.double()[torch.arange(T)][None] + typ.double()[seg]
    if task:
        e = torch.cat([e[:, :1], temb.double()[tid], e[:, 1:]], dim=1)
    _close(out, _ln64(e, g.double(), b.double()))


def test_text_embedding_single_type_row_and_out_of_range_ids(ops):
    """type_vocab_size = 1 (roberta_base_6layer_6connect.json): the backward must not touch a second type row;
    ids outside their table read / write nothing (no access past the allocations)."""
    B, T, H, V = 4, 7, 64, 30
    g0 = torch.Generator().manual_seed(8)
    ids = torch.randint(1, V, (B, T), generator=g0)
    seg = torch.zeros(B, T, dtype=torch.long)
    word, pos, typ = _rand(V, H, seed=1), _rand(16, H, seed=2), _rand(1, H, seed=3)
    g, b = 1 + 0.1 * _rand(H, seed=5), 0.1 * _rand(H, seed=6)
    dx = _rand(B, T, H, seed=9)
    # the gradient tables sit inside one allocation with NaN canaries right behind each of them
    pool = torch.full((V * H + 16 * H + H + 3 * H,), float("nan")).cuda()
    dword, dpos, dtype = pool[:V * H].view(V, H), pool[V * H + H:V * H + 17 * H].view(16, H), \
        pool[V * H + 18 * H:V * H + 19 * H].view(1, H)
    for t in (dword, dpos, dtype):
        t.zero_()
    from vilbert import _native as N
    ids_d, seg_d, dx_d = ids.cuda(), seg.cuda(), dx.cuda()       # (kept alive across the raw C-ABI call)
    N.check(N.lib().vb_text_embed_bwd(N.stream_ptr(), B, T, H, V, 1, 0, ids_d.data_ptr(), seg_d.data_ptr(),
                                      None, dx_d.data_ptr(), dword.data_ptr(), dpos.data_ptr(), dtype.data_ptr(),
                                      None), "vb_text_embed_bwd")
    torch.cuda.synchronize()
    want_w = torch.zeros(V, H, dtype=torch.float64).index_add_(0, ids.reshape(-1), dx.double().reshape(-1, H))
    _close(dword, want_w)
    _close(dpos[:T], dx.double().sum(0))
    _close(dtype, dx.double().sum((0, 1))[None])
    canaries = torch.cat([pool[V * H:V * H + H], pool[V * H + 17 * H:V * H + 18 * H], pool[V * H + 19 * H:]])
    assert torch.isnan(canaries).all(), "the backward wrote past a gradient table"
    # forward + backward with ids / types / labels outside their tables: zero rows, nothing out of bounds
    bad_ids, bad_seg = ids.clone(), seg.clone()
    bad_ids[0, 1], bad_ids[1, 2], bad_seg[2, 3] = V + 5, -3, 1
    out = ops.text_embed_ln_fwd(bad_ids.cuda(), bad_seg.cuda(), word.cuda(), pos.cuda(), typ.cuda(), g.cuda(), b.cuda(),
                                1e-12)[0]
    ok_w = ((bad_ids >= 0) & (bad_ids < V)).double()[..., None]
    e = word.double()[bad_ids.clamp(0, V - 1)] * ok_w + pos.double()[torch.arange(T)][None] \
        + typ.double()[bad_seg.clamp(0, 0)] * (bad_seg == 0).double()[..., None]
    _close(out, _ln64(e, g.double(), b.double()))
    dw, dp, dt, _ = ops.text_embed_bwd(dx.cuda(), bad_ids.cuda(), bad_seg.cuda(), None, (V, H), (16, H), (1, H), None)
    assert torch.isfinite(dw).all() and torch.isfinite(dt).all()


def test_cross_entropy_out_of_range_label_poisons_the_loss():
    from vilbert import ops as O
    logits = _rand(6, 11, seed=2).cuda()
    labels = torch.tensor([1, -1, 11, 3, -1, 2]).cuda()         # 11 is outside [0, 11)
    loss, lse, count = O.xent_fwd(logits, labels, -1)
    assert torch.isnan(loss)                                      # loud, and no read past the row
    assert torch.isfinite(O.xent_bwd(torch.ones(1).cuda(), logits, labels, -1, lse, count)).all()


def test_image_embedding(ops):
    B, R, H = 4, 9, 1024
    proj, loc = _rand(B, R, H, seed=1), torch.rand(B, R, 5, generator=torch.Generator().manual_seed(2))
    wl, bl = _rand(H, 5, seed=3), _rand(H, seed=4)
    g, b = 1 + 0.1 * _rand(H, seed=5), 0.1 * _rand(H, seed=6)
    out = ops.image_embed_ln_fwd(proj.cuda(), loc.cuda(), wl.cuda(), bl.cuda(), g.cuda(), b.cuda(), 1e-12)[0]
    s = proj.double() + loc.double() @ wl.double().t() + bl.double()
    _close(out, _ln64(s, g.double(), b.double()))


def test_additive_mask(ops):
    m = torch.tensor([[1, 1, 0, 0], [1, 0, 0, 1]])
    want = (1.0 - m.double()) * -10000.0
    _close(ops.additive_mask(m.cuda()), want, 0, 0)
    _close(ops.additive_mask(m.float().cuda()), want, 0, 0)
    _close(ops.additive_mask(m.int().cuda()), want, 0, 0)


def _attn64(q, k, v, mask_add, heads):
    B, Sq, H = q.shape
    d = H // heads
    sp = lambda t: t.double().view(t.shape[0], t.shape[1], heads, d).permute(0, 2, 1, 3)
    s = sp(q) @ sp(k).transpose(-1, -2) / math.sqrt(d)
    if mask_add is not None:
        s = s + mask_add.double().view(mask_add.shape[0], 1, 1, -1)
    p = torch.softmax(s, -1)
    ctx = (p @ sp(v)).permute(0, 2, 1, 3).reshape(max(q.shape[0], k.shape[0]), Sq, H)
    return ctx, p


@pytest.mark.parametrize("heads,d,Sq,Sk", [
    (12, 64, 36, 36),    # text self-attention
    (8, 128, 36, 36),    # image self-attention / co-attention
    (8, 128, 21, 37),    # co-attention, text queries over 37 regions
    (2, 32, 9, 7),       # tiny test config
    (3, 32, 1, 1),       # single token
    (4, 64, 50, 101),    # > 3 key tiles (NT = 8)
    (2, 128, 33, 306),   # longest region count in the task table (NT = 20)
    (2, 64, 256, 256),
])
def test_attention_from_fused_qkv(ops, heads, d, Sq, Sk):
    B, H = 3, heads * d
    g = torch.Generator().manual_seed(5)
    qsrc = torch.randn(B, Sq, 3 * H, generator=g)
    ksrc = qsrc if Sq == Sk else torch.randn(B, Sk, 3 * H, generator=g)
    lens = torch.randint(1, Sk + 1, (B,), generator=g)
    mask = (torch.arange(Sk)[None] < lens[:, None]).float()
    madd = (1.0 - mask) * -10000.0
    qd, kd = qsrc.cuda(), ksrc.cuda()
    ctx, probs, _ = ops.attention_fwd(qd[..., :H], kd[..., H:2 * H], kd[..., 2 * H:], madd.cuda(), heads,
                                   want_probs=True)
    want_ctx, want_p = _attn64(qsrc[..., :H], ksrc[..., H:2 * H], ksrc[..., 2 * H:], madd, heads)
    _close(ctx, want_ctx)
    _close(probs, want_p, 2e-5, 1e-7)
    ctx2, none, _ = ops.attention_fwd(qd[..., :H], kd[..., H:2 * H], kd[..., 2 * H:], None, heads)
    assert none is None
    _close(ctx2, _attn64(qsrc[..., :H], ksrc[..., H:2 * H], ksrc[..., 2 * H:], None, heads)[0])


def test_attention_query_broadcast(ops):
    # one caption against many images (eval_retrieval): q batch 1, k/v batch 5
    heads, d, T, R, Bn = 8, 128, 12, 36, 5
    H = heads * d
    g = torch.Generator().manual_seed(6)
    q, k, v = torch.randn(1, T, H, generator=g), torch.randn(Bn, R, H, generator=g), torch.randn(Bn, R, H, generator=g)
    madd = torch.zeros(Bn, R)
    madd[:, 30:] = -10000.0
    ctx, _, _ = ops.attention_fwd(q.cuda(), k.cuda(), v.cuda(), madd.cuda(), heads)
    _close(ctx, _attn64(q.expand(Bn, T, H), k, v, madd, heads)[0])


def test_attention_fully_masked_row_matches_reference_semantics(ops):
    # additive -10000 (not -inf): an all-masked row is a uniform softmax, as in the reference
    heads, d, S = 2, 64, 5
    g = torch.Generator().manual_seed(7)
    q, k, v = (torch.randn(1, S, heads * d, generator=g) for _ in range(3))
    madd = torch.full((1, S), -10000.0)
    ctx, _, _ = ops.attention_fwd(q.cuda(), k.cuda(), v.cuda(), madd.cuda(), heads)
    # next to -10000 one fp32 ulp is 1e-3: compare with the fp32 evaluation order of the reference
    sp = lambda t: t.view(1, S, heads, d).permute(0, 2, 1, 3)
    s32 = sp(q) @ sp(k).transpose(-1, -2) / math.sqrt(d) + madd.view(1, 1, 1, S)
    want = (torch.softmax(s32, -1) @ sp(v)).permute(0, 2, 1, 3).reshape(1, S, heads * d)
    _close(ctx, want.double(), 2e-3, 2e-3)


def test_attention_range_error(ops):
    """One LAUNCH handles at most VB_MAX_KEYS = 320 keys: the C entry point refuses more; the Python launcher serves longer
    key sequences chunk by chunk (round 4; with dropout and under autograd since round 6: tests/test_attention_chunks.py) and
    raises only when the probabilities tensor is wanted."""
    import ctypes
    from vilbert import _native as N
    q = torch.zeros(1, 4, 64).cuda()
    k = torch.zeros(1, 400, 64).cuda()
    a, keep, _ = ops._attn_args(q, k, k, None, 1, 0.0, 0)
    out = torch.empty(1, 4, 64, device="cuda")
    a.O, a.ldo = out.data_ptr(), 64
    assert N.lib().vb_attention_fwd(N.stream_ptr(), ctypes.byref(a)) == -3          # VB_E_RANGE
    ctx, _, _ = ops.attention_fwd(q, k, k, None, 1)
    assert ctx.shape == (1, 4, 64) and float(ctx.abs().max()) == 0.0
    ctx, _, _ = ops.attention_fwd(q, k, k, None, 1, drop_p=0.1, seed=3)
    assert ctx.shape == (1, 4, 64) and float(ctx.abs().max()) == 0.0
    with pytest.raises(RuntimeError):
        ops.attention_fwd(q, k, k, None, 1, want_probs=True)


@pytest.mark.parametrize("rows,n", [(1, 2), (37, 2), (50, 1601), (123, 30522), (5, 7)])
def test_cross_entropy_forward_backward(rows, n):
    """vb_xent_fwd / vb_xent_bwd vs nn.CrossEntropyLoss(ignore_index=-1) in fp64 (reference vilbert.py:1453,
    1578-1585), through the autograd Function the model uses."""
    from vilbert import functional as F
    g = torch.Generator().manual_seed(rows * 31 + n)
    logits = torch.randn(rows, n, generator=g) * 3.0
    labels = torch.randint(0, n, (rows,), generator=g)
    if rows > 2:
        labels[torch.rand(rows, generator=g) < 0.3] = -1
        labels[0] = n - 1
    x = logits.cuda().requires_grad_(True)
    loss = F.cross_entropy(x, labels.cuda(), ignore_index=-1)
    (loss * 1.7).backward()
    x64 = logits.double().requires_grad_(True)
    want = torch.nn.CrossEntropyLoss(ignore_index=-1)(x64, labels)
    (want * 1.7).backward()
    assert loss.dim() == 0
    _close(loss, want.detach(), rtol=2e-6, atol=1e-6)
    # p - onehot cancels when p -> 1: absolute error is fp32 roundoff of p (6e-8) times the loss scale
    _close(x.grad, x64.grad, rtol=2e-5, atol=2e-7)
    with torch.no_grad():
        _close(F.cross_entropy(x, labels.cuda()), want.detach(), rtol=2e-6, atol=1e-6)


def test_cross_entropy_all_ignored_is_nan_like_torch():
    from vilbert import functional as F
    x = torch.randn(4, 9).cuda().requires_grad_(True)
    loss = F.cross_entropy(x, torch.full((4,), -1, dtype=torch.long).cuda())
    assert torch.isnan(loss)


@pytest.mark.parametrize("rows,n", [(1, 1601), (64, 1601), (9, 33)])
def test_kl_div_of_log_softmax_forward_backward(rows, n):
    """vb_kl_fwd / vb_kl_bwd vs sum(KLDivLoss(reduction='none')(log_softmax(s), t)) / count in fp64
    (reference vilbert.py:1454,1516-1522); targets with exact zeros included."""
    from vilbert import functional as F
    g = torch.Generator().manual_seed(rows + n)
    scores = torch.randn(rows, n, generator=g) * 2.0
    target = torch.softmax(torch.randn(rows, n, generator=g) * 2.0, -1)
    target[:, ::5] = 0.0
    s = scores.cuda().requires_grad_(True)
    loss = F.kl_div_log_softmax(s, target.cuda(), float(rows))
    (loss * 0.6).backward()
    s64 = scores.double().requires_grad_(True)
    want = torch.nn.KLDivLoss(reduction="none")(torch.log_softmax(s64, 1), target.double()).sum() / rows
    (want * 0.6).backward()
    _close(loss, want.detach(), rtol=5e-6, atol=1e-6)
    _close(s.grad, s64.grad, rtol=2e-5, atol=2e-7)


@pytest.mark.parametrize("M,N,K", [(1628, 30522, 768), (2048, 30522, 768), (389, 30522, 768), (300, 8190, 256)])
def test_decoder_input_gradient_is_split_along_the_contraction_in_the_bf16_operand_modes(M, N, K):
    """Round 6: the small-output / long-contraction input gradient (the MLM decoder: dX [rows x 768] over 30,522 out-features)
    is cut along K in the bf16-operand modes as it is in fp32 (it ran unsplit on 96 blocks there: 0.90 ms of the 27 ms bf16
    step at B = 256, 0.36 ms split). Against float64 on the bf16-rounded operands, through a padded row stride with NaN in
    the padding; two runs bit-identical (ordered reduce through the deterministic workspace)."""
    from vilbert import _native, ops
    prev = _native.set_gemm_mode("bf16")
    try:
        w = _rand(N, K, seed=2, scale=0.05)
        dy_buf = torch.full((M, (N + 3) // 4 * 4), float("nan"))
        dy_buf[:, :N] = _rand(M, N, seed=4)
        dy = dy_buf.cuda()[:, :N]
        wd = w.cuda()
        dx1 = ops.linear_bwd_input(dy, [wd], K).clone()
        dx2 = ops.linear_bwd_input(dy, [wd], K)
        assert torch.equal(dx1, dx2), "two runs differ"
        want = dy_buf[:, :N].bfloat16().double() @ w.bfloat16().double()
        mag = dy_buf[:, :N].bfloat16().double().abs() @ w.bfloat16().double().abs()
        err = (dx1.cpu().double() - want).abs()
        tol = 3e-6 * mag + 1e-5
        assert not (err > tol).any(), "worst err / tol %.2f" % float((err / tol).max())
    finally:
        _native.set_gemm_mode(prev)
