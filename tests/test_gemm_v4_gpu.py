"""Persistent one-block-per-CU GEMM kernel (csrc/gemm_v4.h: 12 MFMA waves + 1 LDS-DMA loader wave, 288 x 96 / 288 x 128
tiles, K tiles of all output tiles of a block streamed through one LDS ring) against fp64, forced onto EVERY eligible
launch (mode 2) so that the cases the planner would not pick are covered too: ragged last row tile, fewer tiles than
compute units, several rounds of tiles per block (loader running across output-tile boundaries, the extra boundary
barrier), stacked weight segments (forward: along N, dgrad: along K), both tile widths, every fused epilogue of the
forward and dgrad layouts, the shortest legal contraction (two K steps), and bit-equality of repeated launches."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    from vilbert import ops as _ops
    return _ops


@pytest.fixture
def v4():
    from vilbert import _native
    prev = _native.set_gemm_v4(2)
    yield _native
    _native.set_gemm_v4(prev)


def _rand(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale)


def _gelu64(x):
    return x * 0.5 * (1.0 + torch.erf(x / math.sqrt(2.0)))


def _close(got, want64, rtol=3e-5, atol=3e-5):
    got = got.detach().cpu().double()
    err = (got - want64).abs()
    assert got.shape == want64.shape and torch.isfinite(got).all()
    assert (err <= atol + rtol * want64.abs()).all(), "max err %.3e (max |ref| %.3e)" % (err.max().item(), want64.abs().max().item())


# (M, N per segment, K, nseg): one tile; ragged M (rows past the matrix are computed, never stored); 5 x 8 = 40 tiles
# < 256 blocks; 3 segments x 96 (tile width 96 only); 1024-wide (128 only); two K steps; > 256 tiles (two rounds, the
# second one partial: blocks with and without a second tile); three full rounds
SHAPES = [(288, 96, 64, 1), (300, 384, 96, 1), (1440, 768, 160, 1), (576, 96, 128, 3), (900, 1024, 64, 1),
          (288, 128, 32, 1), (288 * 9 + 17, 96 * 30, 64, 1), (288 * 32, 768, 64, 3)]


@pytest.mark.parametrize("M,N,K,nseg", SHAPES)
def test_forward_and_dgrad_match_fp64(ops, v4, M, N, K, nseg):
    x = _rand(M, K, seed=1)
    ws = [_rand(N, K, seed=10 + i, scale=0.1) for i in range(nseg)]
    bs = [_rand(N, seed=20 + i) for i in range(nseg)]
    dy = _rand(M, nseg * N, seed=3)
    xd, wd = x.cuda(), [w.cuda() for w in ws]
    y, _ = ops.linear_fwd(xd, wd, [b.cuda() for b in bs])
    _close(y, torch.cat([x.double() @ w.double().t() + b.double() for w, b in zip(ws, bs)], 1))
    dx = ops.linear_bwd_input(dy.cuda(), wd, K)
    _close(dx, dy.double() @ torch.cat(ws, 0).double(), 3e-5, 3e-5 * max(1.0, N * nseg / 256))
    # the planner's own choice (mode 1) and the 4-wave kernels (mode 0) give the same numbers up to summation order
    for mode in (1, 0):
        v4.set_gemm_v4(mode)
        y2, _ = ops.linear_fwd(xd, wd, [b.cuda() for b in bs])
        assert (y2 - y).abs().max().item() <= 2e-5 * max(1.0, y.abs().max().item())
    v4.set_gemm_v4(2)


def test_repeated_launches_are_bit_identical(ops, v4):
    x, w = _rand(288 * 3, 256, seed=5).cuda(), _rand(96 * 4, 256, seed=6, scale=0.1).cuda()
    a, _ = ops.linear_fwd(x, [w], [None])
    for _ in range(3):
        b, _ = ops.linear_fwd(x, [w], [None])
        assert torch.equal(a, b)


@pytest.mark.parametrize("M,N,K", [(288 * 2, 96 * 4, 64), (320, 256, 96)])
def test_fused_epilogues(ops, v4, M, N, K):
    """gelu, gelu + stored derivative, residual, dropout + residual (mask = the vb_dropout function of seed and
    element index), dgrad with the multiplier / residual-gradient epilogues."""
    x, w, b = _rand(M, K, seed=7), _rand(N, K, seed=8, scale=0.1), _rand(N, seed=9)
    r = _rand(M, N, seed=10)
    xd, wd, bd = x.cuda(), [w.cuda()], [b.cuda()]
    pre = x.double() @ w.double().t() + b.double()
    y, _ = ops.linear_fwd(xd, wd, bd, act="gelu")
    _close(y, _gelu64(pre))
    y, d = ops.linear_fwd(xd, wd, bd, act="gelu", want_act_grad=True)
    p2 = pre.clone().requires_grad_(True)
    _gelu64(p2).sum().backward()
    _close(y, _gelu64(pre))
    _close(d, p2.grad)
    y, _ = ops.linear_fwd(xd, wd, bd, residual=r.cuda())
    _close(y, pre + r.double())
    # dropout before the residual: compare with the stand-alone dropout kernel applied to the plain product
    plain, _ = ops.linear_fwd(xd, wd, bd)
    y, _ = ops.linear_fwd(xd, wd, bd, residual=r.cuda(), drop_p=0.25, seed=1234)
    want = ops.dropout(plain, 0.25, 1234) + r.cuda()
    assert (y - want).abs().max().item() <= 1e-5 * max(1.0, want.abs().max().item())
    kept = (y - r.cuda()).abs() > 0
    assert 0.6 < kept.float().mean().item() < 0.9
    dy, m = _rand(M, N, seed=11), _rand(M, K, seed=12)
    dx = ops.linear_bwd_input(dy.cuda(), wd, K, mul=m.cuda())
    _close(dx, (dy.double() @ w.double()) * m.double())
    dx = ops.linear_bwd_input(dy.cuda(), wd, K, residual=m.cuda())
    _close(dx, dy.double() @ w.double() + m.double())


def test_planner_picks_the_persistent_kernel_only_where_it_fills_the_chip(ops):
    """Mode 1 (default): same results whichever kernel runs; this test pins that the default mode is 1 and that the
    text-stream shape of the batch-256 step (9216 rows) goes through it without error, next to a shape that stays on
    the 4-wave kernels (batch-64 rows)."""
    from vilbert import _native
    assert _native.set_gemm_v4(1) == 1
    for M in (9216, 2304):
        x, w = _rand(M, 768, seed=1), _rand(768, 768, seed=2, scale=0.05)
        y, _ = ops.linear_fwd(x.cuda(), [w.cuda()], [None])
        rows = torch.arange(0, M, 97)
        _close(y[rows], x[rows].double() @ w.double().t())


# (token rows = contraction, out-features per segment, in-features, segments): one tile with 1 / 2 splits; the three
# stacked q | k | v weights (one dW tensor per segment, bias gradients fused); the text-stream shapes of the batch-256
# step (W[768, 768]: 16 tiles x 16 splits = 256 units; W[3072, 768] / W[768, 3072]: 64 tiles x 4 splits); several
# rounds of units per block (2 tiles x 8 col tiles x 18 splits = 288 units)
WGRAD_SHAPES = [(256, 384, 96, 1), (512, 384, 192, 1), (1152, 384, 96, 3), (9216, 768, 768, 1), (9216, 3072, 768, 1),
                (2304, 768, 3072, 1), (9216, 768, 768, 3), (4608, 768, 384, 1),
                # 256-row tiles on 8 MFMA waves (1024 / 4096-row weights: image stream, connection layers, bert_large)
                (512, 256, 128, 1), (9216, 1024, 1024, 1), (9472, 1024, 1024, 1), (2304, 1024, 768, 3), (4608, 4096, 1024, 1),
                (4608, 1024, 4096, 1)]


@pytest.mark.parametrize("M,N,K,nseg", WGRAD_SHAPES)
def test_weight_gradient_matches_fp64(ops, v4, M, N, K, nseg):
    x = _rand(M, K, seed=1)
    dy = _rand(M, nseg * N, seed=3)
    dws, dbs = ops.linear_bwd_weight(dy.cuda(), x.cuda(), nseg, N, [True] * nseg)
    for s in range(nseg):
        seg = dy[:, s * N:(s + 1) * N].double()
        want = seg.t() @ x.double()
        _close(dws[s], want, 3e-5, 3e-5 * max(1.0, want.abs().max().item()))
        _close(dbs[s], seg.sum(0), 3e-5, 3e-5 * max(1.0, M / 64))
    # accumulation into existing gradients (micro-batches / tied weights): the kernel ADDS
    if nseg == 1:
        dw0, db0 = _rand(N, K, seed=5).cuda(), _rand(N, seed=6).cuda()
        dw1, db1 = dw0.clone(), db0.clone()
        ops.linear_bwd_weight(dy.cuda(), x.cuda(), 1, N, [True], dw_out=[dw1], db_out=[db1])
        want = dy.double().t() @ x.double()
        _close(dw1, dw0.cpu().double() + want, 3e-5, 3e-5 * max(1.0, want.abs().max().item()))
        _close(db1, db0.cpu().double() + dy.double().sum(0), 3e-5, 3e-5 * max(1.0, M / 64))
    v4.set_gemm_v4(0)
    dws0, _ = ops.linear_bwd_weight(dy.cuda(), x.cuda(), nseg, N, [True] * nseg)
    v4.set_gemm_v4(2)
    for a, b in zip(dws, dws0):
        assert (a - b).abs().max().item() <= 3e-5 * max(1.0, b.abs().max().item())


# ---- round 4: the tile menu of the persistent kernel (wave grid x wave tile as template parameters) ------------------------
# configuration code WM * 1000 + TM * 100 + TM2 * 10 + TN (plan_v4 in csrc/gemm.hip), forced through the laboratory hook
# vblab_set_gemm_v4_cfg so that every instantiation is exercised whatever the planner would choose.
#   4544 / 4543: MIXED 320 | 256-row tiles on 8 MFMA waves - M = 320 a + 256 (32 - a): the image stream at batch 256
#   (9472 = 20 x 320 + 12 x 256) and the other row counts of that form; N a multiple of 8 column tiles.
MENU = [(6303, 9216, 768, 64, 1), (6304, 9216, 1024, 64, 1),
        (4544, 9472, 1024, 64, 1), (4544, 9472, 1024, 96, 3), (4544, 8256, 1024, 64, 1), (4544, 10176, 2048, 32, 1),
        (4543, 9472, 768, 64, 1), (4543, 9472, 768, 64, 2),
        (6204, 2304, 1024, 64, 1), (6204, 2368, 1024, 96, 3), (6204, 1000, 128, 32, 1),
        (6104, 2304, 1024, 64, 1), (6103, 2368, 768, 64, 1), (6103, 2304, 96 * 30, 32, 1),
        (4202, 2304, 768, 64, 3), (4202, 2368, 1024, 64, 1),
        (4104, 2304, 3072, 64, 1), (4104, 2368, 1024, 128, 1), (4104, 70, 128, 32, 1)]


@pytest.fixture
def force_cfg():
    import ctypes
    from vilbert import _native
    hook = _native.lib().vblab_set_gemm_v4_cfg
    hook.restype, hook.argtypes = ctypes.c_int, [ctypes.c_int]
    hook.last = _native.lib().vblab_last_gemm_v4_cfg
    hook.last.restype, hook.last.argtypes = ctypes.c_int, []
    prev_mode = _native.set_gemm_v4(2)
    yield hook
    hook(0)
    _native.set_gemm_v4(prev_mode)


@pytest.mark.parametrize("cfg,M,N,K,nseg", MENU)
def test_menu_configuration_matches_fp64(ops, force_cfg, cfg, M, N, K, nseg):
    """Forward (bias; GELU + stored derivative; dropout + residual) and dgrad (plain; multiplier; residual gradient) of
    one configuration against fp64, all rows."""
    from vilbert import _native
    force_cfg(cfg)
    x = _rand(M, K, seed=1)
    ws = [_rand(N, K, seed=10 + i, scale=0.1) for i in range(nseg)]
    bs = [_rand(N, seed=20 + i) for i in range(nseg)]
    dy = _rand(M, nseg * N, seed=3)
    xd, wd, bd = x.cuda(), [w.cuda() for w in ws], [b.cuda() for b in bs]
    pre = torch.cat([x.double() @ w.double().t() + b.double() for w, b in zip(ws, bs)], 1)
    y, _ = ops.linear_fwd(xd, wd, bd)
    assert force_cfg.last() == cfg, "the forced configuration did not run"
    _close(y, pre)
    if nseg == 1:
        y, d = ops.linear_fwd(xd, wd, bd, act="gelu", want_act_grad=True)
        p2 = pre.clone().requires_grad_(True)
        _gelu64(p2).sum().backward()
        _close(y, _gelu64(pre))
        _close(d, p2.grad)
        r = _rand(M, N, seed=10).cuda()
        plain, _ = ops.linear_fwd(xd, wd, bd)
        y, _ = ops.linear_fwd(xd, wd, bd, residual=r, drop_p=0.25, seed=77)
        want = ops.dropout(plain, 0.25, 77) + r
        assert (y - want).abs().max().item() <= 1e-5 * max(1.0, want.abs().max().item())
    dyd = dy.cuda()
    want = dy.double() @ torch.cat(ws, 0).double()
    tol = 3e-5 * max(1.0, N * nseg / 256)
    _close(ops.linear_bwd_input(dyd, wd, K), want, 3e-5, tol)
    # (the dgrad's output width is K: it runs on `cfg` only where K divides into that configuration's columns)
    assert force_cfg.last() in (cfg, 0)
    m = _rand(M, K, seed=12)
    _close(ops.linear_bwd_input(dyd, wd, K, mul=m.cuda()), want * m.double(), 3e-5, tol)
    _close(ops.linear_bwd_input(dyd, wd, K, residual=m.cuda()), want + m.double(), 3e-5, tol)


def test_image_stream_shape_of_the_headline_runs_on_the_mixed_tiles_by_default(ops):
    """Mode 1 (the default planner): M = 9472 = 37 regions x 256 samples, N = 1024 -> configuration 4544; results equal
    fp64 on sampled rows. The batch-64 shapes (M = 2304 / 2368) stay on the 4-wave blocks by default (the small-M menu
    wins isolated A/B runs and loses inside the multi-stream training step: DESIGN.md) - VB_GEMM_V4_SMALLM=1 opts in."""
    import ctypes
    from vilbert import _native
    last = _native.lib().vblab_last_gemm_v4_cfg
    last.restype, last.argtypes = ctypes.c_int, []
    assert _native.set_gemm_v4(1) == 1
    for M, N, K in ((2304, 768, 768), (2368, 1024, 1024)):
        xs, ws_ = _rand(M, K, seed=1), _rand(N, K, seed=2, scale=0.05)
        ys, _ = ops.linear_fwd(xs.cuda(), [ws_.cuda()], [None])
        assert last() == 0, (M, N, K)
        _close(ys, xs.double() @ ws_.double().t())
    x, w = _rand(9472, 1024, seed=1), _rand(1024, 1024, seed=2, scale=0.05)
    xd, wd = x.cuda(), [w.cuda()]
    y, _ = ops.linear_fwd(xd, wd, [None])
    assert last() == 4544
    rows = torch.cat([torch.arange(0, 9472, 97), torch.tensor([6399, 6400, 9471])])      # incl. the 320 | 256 boundary
    _close(y[rows], x[rows].double() @ w.double().t())
    _native.set_gemm_v4(0)
    y0, _ = ops.linear_fwd(xd, wd, [None])
    _native.set_gemm_v4(1)
    assert (y - y0).abs().max().item() <= 2e-5 * y.abs().max().item()
