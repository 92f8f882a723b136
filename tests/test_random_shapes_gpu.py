"""Seeded random-shape sweep of the kernels against fp64 torch references: ragged sizes, tile edges, segment
counts, strides, masks. Complements the hand-picked cases of test_kernels_gpu.py / test_backward_gpu.py."""
import math
import random

import pytest
import torch

pytestmark = pytest.mark.gpu


def _close(got, want64, tol=3e-5):
    got = got.detach().cpu().double()
    assert got.shape == want64.shape and torch.isfinite(got).all()
    scale = max(1.0, want64.abs().max().item())
    err = (got - want64).abs().max().item()
    assert err <= tol * scale, "max err %.3e (scale %.3e)" % (err, scale)


def test_linear_random_shapes():
    from vilbert import ops
    rng = random.Random(1234)
    g = torch.Generator().manual_seed(99)
    for trial in range(40):
        M = rng.choice([1, 7, 63, 64, 65, 127, 128, 129, 200, 256, 300, 513])
        K = rng.choice([1, 3, 4, 5, 12, 16, 17, 31, 32, 33, 64, 100, 256])
        nseg = rng.choice([1, 1, 2, 3, 4])
        n = rng.choice([1, 2, 5, 16, 31, 64, 96, 128, 130, 192, 256])
        act = rng.choice([None, "gelu", "relu"])
        use_res, use_bias = rng.random() < 0.5, rng.random() < 0.8
        x = torch.randn(M, K, generator=g)
        ws = [torch.randn(n, K, generator=g) * 0.2 for _ in range(nseg)]
        bs = [torch.randn(n, generator=g) if use_bias else None for _ in range(nseg)]
        r = torch.randn(M, nseg * n, generator=g) if use_res else None
        y, pre = ops.linear_fwd(x.cuda(), [w.cuda() for w in ws], [b.cuda() if b is not None else None for b in bs],
                                act=act, residual=r.cuda() if use_res else None, want_preact=True)
        pre64 = torch.cat([x.double() @ w.double().t() + (b.double() if b is not None else 0) for w, b in zip(ws, bs)], 1)
        a64 = pre64 if act is None else (torch.relu(pre64) if act == "relu"
                                         else pre64 * 0.5 * (1 + torch.erf(pre64 / math.sqrt(2.0))))
        _close(pre, pre64)
        _close(y, a64 + (r.double() if use_res else 0))
        dy = torch.randn(M, nseg * n, generator=g)
        _close(ops.linear_bwd_input(dy.cuda(), [w.cuda() for w in ws], K), dy.double() @ torch.cat(ws, 0).double())
        dws, dbs = ops.linear_bwd_weight(dy.cuda(), x.cuda(), nseg, n, [True] * nseg)
        for s in range(nseg):
            seg = dy[:, s * n:(s + 1) * n].double()
            _close(dws[s], seg.t() @ x.double())
            _close(dbs[s], seg.sum(0))


def test_attention_random_shapes_forward_and_backward():
    from vilbert import ops
    rng = random.Random(4321)
    g = torch.Generator().manual_seed(77)
    for trial in range(24):
        heads = rng.choice([1, 2, 3, 8, 12])
        d = rng.choice([32, 64, 128])
        Sq, Sk = rng.choice([1, 2, 15, 16, 17, 36, 37, 48, 49, 100]), rng.choice([1, 3, 16, 33, 36, 37, 47, 48, 65, 129, 200])
        B, H = rng.choice([1, 2, 3]), heads * d
        q, k, v = (torch.randn(B, s, H, generator=g) for s in (Sq, Sk, Sk))
        lens = torch.randint(1, Sk + 1, (B,), generator=g)
        madd = (1.0 - (torch.arange(Sk)[None] < lens[:, None]).float()) * -10000.0
        use_mask = rng.random() < 0.8
        dout = torch.randn(B, Sq, H, generator=g)
        q64, k64, v64 = (t.double().requires_grad_(True) for t in (q, k, v))
        sp = lambda t: t.view(t.shape[0], t.shape[1], heads, d).permute(0, 2, 1, 3)
        s = sp(q64) @ sp(k64).transpose(-1, -2) / math.sqrt(d)
        if use_mask:
            s = s + madd.double().view(B, 1, 1, Sk)
        ref = (torch.softmax(s, -1) @ sp(v64)).permute(0, 2, 1, 3).reshape(B, Sq, H)
        ref.backward(dout.double())
        qd, kd, vd = q.cuda(), k.cuda(), v.cuda()
        m = madd.cuda() if use_mask else None
        out, _, lse = ops.attention_fwd(qd, kd, vd, m, heads, False, True)
        _close(out, ref.detach())
        dq, dk, dv = torch.empty_like(qd), torch.empty_like(kd), torch.empty_like(vd)
        ops.attention_bwd(dout.cuda(), qd, kd, vd, m, heads, lse, dq, dk, dv)
        _close(dq, q64.grad)
        _close(dk, k64.grad)
        _close(dv, v64.grad)


def test_layernorm_random_shapes():
    from vilbert import ops
    from oracle import vilbert_oracle as vo
    rng = random.Random(5)
    g = torch.Generator().manual_seed(6)
    for trial in range(16):
        rows = rng.choice([1, 3, 4, 5, 15, 16, 17, 63, 64, 65, 200, 1000])
        cols = rng.choice([4, 8, 64, 96, 252, 256, 260, 768, 1024, 1028, 2048, 3072])
        x = torch.randn(rows, cols, generator=g) * 2 + 0.5
        dy = torch.randn(rows, cols, generator=g)
        gam, bet = 1 + 0.1 * torch.randn(cols, generator=g), 0.1 * torch.randn(cols, generator=g)
        y, mean, rstd = ops.layernorm_fwd(x.cuda(), gam.cuda(), bet.cuda(), 1e-12, None, want_stats=True)
        x64, g64, b64 = (t.double().requires_grad_(True) for t in (x, gam, bet))
        ref = vo.layer_norm(x64, g64, b64)
        ref.backward(dy.double())
        _close(y, ref.detach())
        dx, dg, db = ops.layernorm_bwd(dy.cuda(), x.cuda(), mean, rstd, gam.cuda())
        _close(dx, x64.grad)
        _close(dg, g64.grad)
        _close(db, b64.grad)
