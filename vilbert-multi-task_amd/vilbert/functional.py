"""Fused ops of the model, autograd-visible.

When no input needs a gradient (or grad mode is off) the forward launcher is called directly with
nothing saved; otherwise the op goes through its ``torch.autograd.Function`` (autograd_ops.py), whose
backward also runs native kernels only.
"""
import torch

from . import _native as N
from . import autograd_ops as A
from . import ops
from . import ops16

BF16 = torch.bfloat16


def _needs_grad(*tensors):
    return torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in tensors)


# ---------------------------------------------------------------------------------------------------------------
# bf16 stream (round 5, _native.bf16_stream()): between the embeddings and the poolers / heads the hidden states are
# bfloat16 tensors; every op below dispatches on the dtype of its input. A shape the bf16 kernels do not serve (the tiny
# test configurations, an activation other than GELU inside an FFN, attention probabilities wanted) goes through the fp32
# kernels between two casts - same stream dtype on both sides, so the layers around it never notice.
# ---------------------------------------------------------------------------------------------------------------
def _is16(x):
    return torch.is_tensor(x) and x.dtype == BF16 and N.bf16_stream()


def to_bf16(x):
    if x.dtype == BF16:
        return x
    return A.CastFn.apply(x, True) if _needs_grad(x) else ops16.cast_bf16(x)


def to_f32(x):
    if x is None or x.dtype != BF16:
        return x
    return A.CastFn.apply(x, False) if _needs_grad(x) else ops16.cast_f32(x)


def _uniform_bias(biases):
    return all(b is None for b in biases) or all(b is not None for b in biases)


def _linear16(x, weights, biases, act, residual, drop_p, out_f32=False):
    seg_n, K = weights[0].shape
    ok = (ops16.eligible(K, len(weights) * seg_n, seg_n, act) and x.is_cuda and _uniform_bias(biases)
          and (residual is None or residual.dtype == BF16))
    grad = _needs_grad(x, residual, *weights, *biases)
    if not ok or (grad and act is not None):
        y = linear(to_f32(x), weights, biases, act, to_f32(residual), drop_p)
        return y if out_f32 else to_bf16(y)
    if grad:
        return A.Linear16Fn.apply(x, residual, len(weights), drop_p, out_f32, *weights, *biases)
    seed = A.next_seed() if drop_p > 0.0 else 0
    return ops16.linear_fwd(x, weights, biases, act, residual, drop_p=drop_p, seed=seed, out_f32=out_f32)[0]


def linear_f32_out(x, weight, bias):
    """x bf16 -> fp32 result (the 2048 -> 1024 region-feature projection in front of the fp32 embedding kernel)."""
    return _linear16(x, [weight], [bias], None, None, 0.0, out_f32=True)


def linear(x, weights, biases=None, act=None, residual=None, drop_p=0.0, pad_cols=False, out="f32"):
    """dropout(act(x @ cat(weights).T + cat(biases)), drop_p) (+ residual); weights / biases may be single
    tensors; drop_p is the EFFECTIVE probability (0 in eval mode). pad_cols, out: see ops.linear_fwd (out is a request
    the MX inference mode honours where the shape allows it; everywhere else the result is an fp32 tensor)."""
    if not isinstance(weights, (list, tuple)):
        weights, biases = [weights], [biases]
    if biases is None:
        biases = [None] * len(weights)
    if _is16(x):
        return _linear16(x, list(weights), list(biases), act, residual, drop_p)
    if _needs_grad(x, residual, *weights, *biases):
        return A.LinearFn.apply(x, residual, act, -len(weights) if pad_cols else len(weights), drop_p, *weights, *biases)
    seed = A.next_seed() if drop_p > 0.0 else 0
    return ops.linear_fwd(x, weights, biases, act, residual, drop_p=drop_p, seed=seed, pad_cols=pad_cols, out=out)[0]


def ffn(x, w1, b1, act, w2, b2, drop_p=0.0):
    """dropout(act(x @ w1.T + b1) @ w2.T + b2, drop_p) + x  (the block in front of the output LayerNorm)."""
    if _is16(x):
        ok = (act == "gelu" and x.is_cuda and ops16.eligible(w1.shape[1], w1.shape[0]) and ops16.eligible(w2.shape[1], w2.shape[0])
              and _uniform_bias([b1]) and _uniform_bias([b2]))
        if not ok:
            return to_bf16(ffn(to_f32(x), w1, b1, act, w2, b2, drop_p))
        if _needs_grad(x, w1, b1, w2, b2):
            return A.FFN16Fn.apply(x, w1, b1, w2, b2, drop_p)
        h = ops16.linear_fwd(x, [w1], [b1], "gelu")[0]
        seed = A.next_seed() if drop_p > 0.0 else 0
        return ops16.linear_fwd(h, [w2], [b2], None, x, drop_p=drop_p, seed=seed)[0]
    if _needs_grad(x, w1, b1, w2, b2):
        return A.FFNFn.apply(x, w1, b1, w2, b2, act, drop_p)
    # MX mode (inference): the activation of the up-projection leaves its GEMM as MX codes for the down-projection - it is
    # never written in fp32 and never re-read by a quantiser
    mx = (drop_p == 0.0 and ops.mx_eligible(w1.shape[1], w1.shape[0], act, biases=[b1])
          and ops.mx_eligible(w2.shape[1], w2.shape[0], None, biases=[b2]) and x.is_cuda)
    h = ops.linear_fwd(x, [w1], [b1], act, out="mx" if mx else "f32")[0]
    seed = A.next_seed() if drop_p > 0.0 else 0
    return ops.linear_fwd(h, [w2], [b2], None, x, drop_p=drop_p, seed=seed, out="bf16" if mx and ops.mx_stream_bf16() else "f32")[0]


def layer_norm(x, gamma, beta, eps=1e-12):
    if _is16(x):
        if x.shape[-1] > 1024 or x.shape[-1] % 4 != 0 or not x.is_cuda:
            return to_bf16(layer_norm(to_f32(x), gamma, beta, eps))
        if _needs_grad(x, gamma, beta):
            return A.LayerNorm16Fn.apply(x, gamma, beta, eps)
        return ops16.layernorm_fwd(x, gamma, beta, eps)[0]
    if _needs_grad(x, gamma, beta):
        return A.LayerNormFn.apply(x, gamma, beta, eps)
    return ops.layernorm_fwd(x, gamma, beta, eps)[0]


def dropout(x, p, residual=None):
    """dropout(x, p) (+ residual); p is the EFFECTIVE probability (0 in eval mode)."""
    if p <= 0.0:
        # identity (+ residual): the callers fuse the residual into the producing GEMM instead
        assert residual is None
        return x
    if _needs_grad(x, residual):
        return A.dropout(x, p, residual)
    return ops.dropout(x, p, A.next_seed(), residual)


def self_attention(qkv, mask_add, heads, drop_p=0.0, want_probs=False):
    """Attention over a fused [B, S, 3H] = [q | k | v] projection. Returns (context [B,S,H], probs|None)."""
    if _is16(qkv):
        H = qkv.shape[-1] // 3
        if want_probs or qkv.shape[1] > ops.MAX_KEYS or (H // heads) not in (32, 64, 128) or not qkv.is_cuda:
            ctx, probs = self_attention(to_f32(qkv), mask_add, heads, drop_p, want_probs)
            return to_bf16(ctx), probs
        if _needs_grad(qkv):
            return A.SelfAttn16Fn.apply(qkv, mask_add, heads, drop_p), None
        seed = A.next_seed() if drop_p > 0.0 else 0
        return ops16.attention_fwd(qkv[..., :H], qkv[..., H:2 * H], qkv[..., 2 * H:], mask_add, heads, False, drop_p, seed)[0], None
    if _needs_grad(qkv):
        out, probs = A.SelfAttnFn.apply(qkv, mask_add, heads, drop_p, want_probs)
        return out, (probs if want_probs else None)
    H = qkv.shape[-1] // 3
    if qkv.dtype == torch.bfloat16:     # MX inference mode: bf16 projection in, MX context out (ops.mx_attention_ok held)
        return ops.attention_fwd_mx_any(qkv[..., :H], qkv[..., H:2 * H], qkv[..., 2 * H:], mask_add, heads), None
    seed = A.next_seed() if drop_p > 0.0 else 0
    out, probs, _ = ops.attention_fwd(qkv[..., :H], qkv[..., H:2 * H], qkv[..., 2 * H:], mask_add, heads,
                                      want_probs, False, drop_p, seed)
    return out, probs


def bi_attention(qkv1, qkv2, mask1, mask2, heads, p1=0.0, p2=0.0, want_probs=False):
    """Co-attention between stream 1 (image, qkv1 / mask1) and stream 2 (text, qkv2 / mask2):
    returns (ctx1 = attn(q2; k1, v1), ctx2 = attn(q1; k2, v2), probs1|None, probs2|None)."""
    if _is16(qkv1) or _is16(qkv2):
        H = qkv1.shape[-1] // 3
        if (want_probs or max(qkv1.shape[1], qkv2.shape[1]) > ops.MAX_KEYS or (H // heads) not in (32, 64, 128)
                or not (qkv1.is_cuda and _is16(qkv1) and _is16(qkv2))):
            c1, c2, pr1, pr2 = bi_attention(to_f32(qkv1), to_f32(qkv2), mask1, mask2, heads, p1, p2, want_probs)
            return to_bf16(c1), to_bf16(c2), pr1, pr2
        if _needs_grad(qkv1, qkv2):
            c1, c2 = A.BiAttn16Fn.apply(qkv1, qkv2, mask1, mask2, heads, p1, p2)
            return c1, c2, None, None
        s1 = A.next_seed() if p1 > 0.0 else 0
        s2 = A.next_seed() if p2 > 0.0 else 0
        c1 = ops16.attention_fwd(qkv2[..., :H], qkv1[..., H:2 * H], qkv1[..., 2 * H:], mask1, heads, False, p1, s1)[0]
        c2 = ops16.attention_fwd(qkv1[..., :H], qkv2[..., H:2 * H], qkv2[..., 2 * H:], mask2, heads, False, p2, s2)[0]
        return c1, c2, None, None
    if _needs_grad(qkv1, qkv2):
        c1, c2, pr1, pr2 = A.BiAttnFn.apply(qkv1, qkv2, mask1, mask2, heads, p1, p2, want_probs)
        return c1, c2, (pr1 if want_probs else None), (pr2 if want_probs else None)
    H = qkv1.shape[-1] // 3
    if qkv1.dtype == torch.bfloat16 and qkv2.dtype == torch.bfloat16:
        c1 = ops.attention_fwd_mx_any(qkv2[..., :H], qkv1[..., H:2 * H], qkv1[..., 2 * H:], mask1, heads)
        c2 = ops.attention_fwd_mx_any(qkv1[..., :H], qkv2[..., H:2 * H], qkv2[..., 2 * H:], mask2, heads)
        return c1, c2, None, None
    s1 = A.next_seed() if p1 > 0.0 else 0
    s2 = A.next_seed() if p2 > 0.0 else 0
    c1, pr1, _ = ops.attention_fwd(qkv2[..., :H], qkv1[..., H:2 * H], qkv1[..., 2 * H:], mask1, heads,
                                   want_probs, False, p1, s1)
    c2, pr2, _ = ops.attention_fwd(qkv1[..., :H], qkv2[..., H:2 * H], qkv2[..., 2 * H:], mask2, heads,
                                   want_probs, False, p2, s2)
    return c1, c2, pr1, pr2


def text_embed_ln(ids, seg, word, pos, typ, gamma, beta, eps, task_ids=None, task_emb=None):
    if _needs_grad(word, pos, typ, gamma, beta, task_emb):
        return A.TextEmbedFn.apply(ids, seg, word, pos, typ, gamma, beta, eps, task_ids, task_emb)
    return ops.text_embed_ln_fwd(ids, seg, word, pos, typ, gamma, beta, eps, task_ids, task_emb)[0]


def image_embed_ln(feat_proj, loc, w_loc, b_loc, gamma, beta, eps):
    if _needs_grad(feat_proj, w_loc, b_loc, gamma, beta):
        return A.ImageEmbedFn.apply(feat_proj, loc, w_loc, b_loc, gamma, beta, eps)
    return ops.image_embed_ln_fwd(feat_proj, loc, w_loc, b_loc, gamma, beta, eps)[0]


def cross_entropy(logits, labels, ignore_index=-1):
    """nn.CrossEntropyLoss(ignore_index=...)(logits [rows, n], labels [rows]) -> 0-dim loss."""
    if _needs_grad(logits):
        return A.CrossEntropyFn.apply(logits, labels, ignore_index)
    return ops.xent_fwd(logits, labels, ignore_index)[0]


def kl_div_log_softmax(scores, target, divisor):
    """sum(nn.KLDivLoss(reduction="none")(log_softmax(scores, 1), target)) / divisor -> 0-dim loss."""
    if _needs_grad(scores):
        return A.KLDivFn.apply(scores, target, divisor)
    return ops.kl_fwd(scores, target, divisor)[0]
