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
    with pytest.raises(RuntimeError, match="keys"):
        ops.attention_fwd(q.to(dev), dkv[..., :H], dkv[..., H:], mask.to(dev), heads, drop_p=0.1, seed=1)


@pytest.mark.gpu
def test_more_than_320_keys_under_autograd_says_so_at_the_call():
    """Round-4 advisor: the chunked path is forward-only; under autograd the error used to appear inside backward()
    (VB_E_RANGE, no hint). It is raised where the attention is called, with the reason."""
    from vilbert import functional as VF
    H, heads, S = 128, 2, 400
    qkv = torch.randn(1, S, 3 * H, device="cuda:0", requires_grad=True)
    with pytest.raises(RuntimeError, match="under autograd"):
        VF.self_attention(qkv, None, heads)
    with torch.no_grad():
        ctx, _ = VF.self_attention(qkv, None, heads)        # inference: served chunk by chunk
    assert ctx.shape == (1, S, H) and torch.isfinite(ctx).all()
