"""Attention over more than VB_MAX_KEYS (320) keys (round-3 review: "attention beyond 320 keys raises"): the launcher serves
the keys chunk by chunk and merges the chunk contexts with their log-sum-exps (`ops.merge_attention_chunks` - the exact
identity flash attention tiles by). The merge is pure tensor arithmetic and is checked here on the CPU against a softmax over
all keys; the GPU test runs the whole path (several launches + merge) against a float64 statement of
vilbert.py:429-449 / 768-809 in eval mode."""
import math
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vilbert-multi-task_amd"))


def _reference(q, k, v, mask, heads):
    B, Sq, H = q.shape[0] if q.shape[0] > k.shape[0] else k.shape[0], q.shape[1], q.shape[2]
    d = H // heads
    qh = q.double().expand(B, -1, -1).reshape(B, Sq, heads, d).permute(0, 2, 1, 3)
    kh = k.double().expand(B, -1, -1).reshape(B, k.shape[1], heads, d).permute(0, 2, 1, 3)
    vh = v.double().expand(B, -1, -1).reshape(B, k.shape[1], heads, d).permute(0, 2, 1, 3)
    s = qh @ kh.transpose(2, 3) / math.sqrt(d)
    if mask is not None:
        s = s + mask.double().expand(B, -1)[:, None, None, :]
    return (torch.softmax(s, -1) @ vh).permute(0, 2, 1, 3).reshape(B, Sq, H), torch.logsumexp(s, -1)


@pytest.mark.parametrize("chunks", [[5, 7], [320, 320, 60], [1, 1, 1]])
def test_merge_of_chunk_contexts_is_the_softmax_over_all_keys(chunks):
    from vilbert import ops
    g = torch.Generator().manual_seed(sum(chunks))
    B, Sq, heads, d = 2, 6, 3, 8
    H, Sk = heads * d, sum(chunks)
    q, k, v = (torch.randn(B, n, H, generator=g, dtype=torch.float64) for n in (Sq, Sk, Sk))
    mask = (torch.rand(B, Sk, generator=g) > 0.7).double() * -10000.0
    want, want_lse = _reference(q, k, v, mask, heads)
    outs, lses, c0 = [], [], 0
    for n in chunks:
        o, l = _reference(q, k[:, c0:c0 + n], v[:, c0:c0 + n], mask[:, c0:c0 + n], heads)
        outs.append(o)
        lses.append(l)
        c0 += n
    got, got_lse = ops.merge_attention_chunks(outs, lses, heads)
    assert torch.allclose(got, want, rtol=1e-12, atol=1e-12) and torch.allclose(got_lse, want_lse, rtol=1e-12, atol=1e-12)


@pytest.mark.gpu
@pytest.mark.parametrize("B,heads,d,Sq,Sk,kb", [(2, 8, 128, 30, 404, 2), (3, 12, 64, 24, 1000, 3), (4, 8, 128, 23, 505, 1)])
def test_attention_over_more_than_320_keys(B, heads, d, Sq, Sk, kb):
    from vilbert import ops
    dev = "cuda:0"
    H = heads * d
    g = torch.Generator().manual_seed(Sk)
    q = torch.randn(B, Sq, H, generator=g) * 0.5
    kv = torch.randn(kb, Sk, 2 * H, generator=g) * 0.5
    keep = (torch.rand(kb, Sk, generator=g) > 0.2).float()
    keep[:, 0] = 1
    mask = (1.0 - keep) * -10000.0
    dkv = kv.to(dev)
    with torch.no_grad():
        out, probs, lse = ops.attention_fwd(q.to(dev), dkv[..., :H], dkv[..., H:], mask.to(dev), heads, want_lse=True)
    want, want_lse = _reference(q, kv[..., :H], kv[..., H:], mask, heads)
    assert probs is None
    assert (out.cpu().double() - want).abs().max().item() <= 1e-5 and (lse.cpu().double() - want_lse).abs().max().item() <= 1e-4
    with pytest.raises(RuntimeError, match="probabilities"):
        ops.attention_fwd(q.to(dev), dkv[..., :H], dkv[..., H:], mask.to(dev), heads, want_probs=True)


@pytest.mark.gpu
@pytest.mark.parametrize("B,heads,d,S1,S2", [(2, 8, 128, 30, 404), (2, 12, 64, 24, 700), (3, 4, 32, 330, 17)])
def test_more_than_320_keys_under_autograd_match_float64_autograd(B, heads, d, S1, S2):
    """Round 6 (review: "attention > 320 keys under autograd still raises"; reference vilbert.py:1008-1040 trains
    in_batch_pairs with stacked options): the autograd nodes serve long key sequences chunk by chunk in BOTH directions -
    the backward hands every chunk launch the log-sum-exp over all keys and runs two passes over the chunks (D = rowsum(P dP)
    accumulated, then given). Self attention and both co-attention directions against float64 autograd, masks on."""
    from vilbert import functional as VF
    dev = "cuda:0"
    H = heads * d
    g = torch.Generator().manual_seed(S1 + S2)
    qkv1 = (torch.randn(B, S1, 3 * H, generator=g) * 0.5)
    qkv2 = (torch.randn(B, S2, 3 * H, generator=g) * 0.5)
    keep1, keep2 = (torch.rand(B, S1, generator=g) > 0.2).float(), (torch.rand(B, S2, generator=g) > 0.2).float()
    keep1[:, 0], keep2[:, 0] = 1, 1
    m1, m2 = (1.0 - keep1) * -10000.0, (1.0 - keep2) * -10000.0
    w1, w2 = torch.randn(B, S2, H, generator=g), torch.randn(B, S1, H, generator=g)
    ws = torch.randn(B, S2, H, generator=g)

    def ref(a, b_, mask):        # attention of a's queries over b_'s keys / values
        return _reference(a[..., :H], b_[..., H:2 * H], b_[..., 2 * H:], mask, heads)[0]
    r1, r2 = qkv1.double().requires_grad_(True), qkv2.double().requires_grad_(True)
    (ref(r2, r1, m1) * w1.double()).sum().add((ref(r1, r2, m2) * w2.double()).sum()).add(
        (ref(r2, r2, m2) * ws.double()).sum()).backward()

    x1, x2 = qkv1.to(dev).requires_grad_(True), qkv2.to(dev).requires_grad_(True)
    c1, c2, _, _ = VF.bi_attention(x1, x2, m1.to(dev), m2.to(dev), heads)       # c1 = attn(q2; k1, v1), c2 = attn(q1; k2, v2)
    cs, _ = VF.self_attention(x2, m2.to(dev), heads)
    assert (c1.detach().cpu().double() - ref(r2, r1, m1).detach()).abs().max().item() <= 1e-5
    ((c1 * w1.to(dev)).sum() + (c2 * w2.to(dev)).sum() + (cs * ws.to(dev)).sum()).backward()
    for got, want, nm in ((x1.grad, r1.grad, "dqkv1"), (x2.grad, r2.grad, "dqkv2")):
        err = (got.cpu().double() - want).abs().max().item()
        assert err <= 2e-5 * max(1.0, want.abs().max().item()), "%s: %.3e (range %.3e)" % (nm, err, want.abs().max().item())


@pytest.mark.gpu
def test_dropout_over_more_than_320_keys_uses_the_same_mask_in_both_directions():
    """Dropout ON over 505 keys: the context is LINEAR in V for a fixed keep mask, so <dO, ctx> = <dV, V> holds exactly when
    backward regenerates the forward's mask chunk by chunk (and fails by ~p otherwise); the kept fraction matches p; two
    different seeds give different masks; dQ / dK against a central difference of the same (seeded, hence repeatable) function."""
    from vilbert import ops
    dev = "cuda:0"
    B, heads, d, Sq, Sk, p = 2, 4, 64, 20, 505, 0.25
    H = heads * d
    g = torch.Generator().manual_seed(5)
    q, k, v = (torch.randn(B, n, H, generator=g).to(dev) * 0.5 for n in (Sq, Sk, Sk))
    dO = torch.randn(B, Sq, H, generator=g).to(dev)
    ones = torch.ones_like(v)
    with torch.no_grad():
        out, _, lse = ops.attention_fwd(q, k, v, None, heads, want_lse=True, drop_p=p, seed=77)
        kept = ops.attention_fwd(q, k, ones, None, heads, drop_p=p, seed=77)[0]         # = sum of the kept, rescaled probabilities
        other = ops.attention_fwd(q, k, v, None, heads, drop_p=p, seed=78)[0]
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        ops.attention_bwd(dO, q, k, v, None, heads, lse, dq, dk, dv, drop_p=p, seed=77)
    assert abs(float(kept.mean()) - 1.0) <= 0.05 and float((kept - 1.0).abs().max()) > 1e-3
    assert float((out - other).abs().max()) > 1e-3
    lhs, rhs = float((dO.double() * out.double()).sum()), float((dv.double() * v.double()).sum())
    assert abs(lhs - rhs) <= 1e-4 * max(1.0, abs(lhs)), (lhs, rhs)
    for x, gx, nm in ((q, dq, "dq"), (k, dk, "dk")):
        u = torch.randn(x.shape, generator=g).to(dev)
        eps = 1e-2
        def f(t):
            args = (t, k, v) if nm == "dq" else (q, t, v)
            with torch.no_grad():
                return float((ops.attention_fwd(*args, None, heads, drop_p=p, seed=77)[0].double() * dO.double()).sum())
        fd = (f(x + eps * u) - f(x - eps * u)) / (2 * eps)
        an = float((gx.double() * u.double()).sum())
        assert abs(fd - an) <= 2e-2 * max(1.0, abs(an)), "%s: finite difference %.5f vs %.5f" % (nm, fd, an)


@pytest.mark.gpu
def test_attention_maps_of_more_than_320_keys_say_so_at_the_call():
    """What the chunked path cannot return is the probabilities tensor (`visualization`): raised where the attention is called."""
    from vilbert import functional as VF
    H, heads, S = 128, 2, 400
    qkv = torch.randn(1, S, 3 * H, device="cuda:0", requires_grad=True)
    with pytest.raises(RuntimeError, match="attention maps"):
        VF.self_attention(qkv, None, heads, want_probs=True)
    ctx, _ = VF.self_attention(qkv, None, heads)
    ctx.sum().backward()
    assert ctx.shape == (1, S, H) and torch.isfinite(qkv.grad).all()
