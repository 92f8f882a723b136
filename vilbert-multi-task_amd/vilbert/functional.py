"""Autograd-visible fused ops built on the native kernels.

Each op is one ``torch.autograd.Function`` whose forward AND backward enqueue only kernels of
libvilbert_hip.so. When no input needs a gradient (or grad mode is off) the forward launcher is
called directly, with nothing saved.
"""
import torch

from . import ops


def _needs_grad(*tensors):
    return torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in tensors)


def linear(x, weights, biases=None, act=None, residual=None):
    """act(x @ cat(weights).T + cat(biases)) (+ residual); weights / biases may be single tensors."""
    if not isinstance(weights, (list, tuple)):
        weights, biases = [weights], [biases]
    if biases is None:
        biases = [None] * len(weights)
    if _needs_grad(x, residual, *weights, *[b for b in biases if b is not None]):
        from . import autograd_ops
        return autograd_ops.LinearFn.apply(x, residual, act, len(weights), *weights, *biases)
    return ops.linear_fwd(x, weights, biases, act, residual)[0]


def layer_norm(x, gamma, beta, eps=1e-12, x2=None):
    """TF-style LayerNorm of (x + x2)."""
    if _needs_grad(x, x2, gamma, beta):
        from . import autograd_ops
        return autograd_ops.LayerNormFn.apply(x, x2, gamma, beta, eps)
    return ops.layernorm_fwd(x, gamma, beta, eps, x2)[0]


def attention(q, k, v, mask_add, heads, want_probs=False):
    """softmax(q k^T / sqrt(d) + mask) v with merged heads; returns (context, probs or None)."""
    if _needs_grad(q, k, v):
        from . import autograd_ops
        return autograd_ops.attention(q, k, v, mask_add, heads, want_probs)
    return ops.attention_fwd(q, k, v, mask_add, heads, want_probs)


def text_embed_ln(ids, seg, word, pos, typ, gamma, beta, eps, task_ids=None, task_emb=None):
    if _needs_grad(word, pos, typ, gamma, beta, task_emb):
        from . import autograd_ops
        return autograd_ops.TextEmbedFn.apply(ids, seg, word, pos, typ, gamma, beta, eps, task_ids, task_emb)
    return ops.text_embed_ln_fwd(ids, seg, word, pos, typ, gamma, beta, eps, task_ids, task_emb)[0]


def image_embed_ln(feat_proj, loc, w_loc, b_loc, gamma, beta, eps):
    if _needs_grad(feat_proj, w_loc, b_loc, gamma, beta):
        from . import autograd_ops
        return autograd_ops.ImageEmbedFn.apply(feat_proj, loc, w_loc, b_loc, gamma, beta, eps)
    return ops.image_embed_ln_fwd(feat_proj, loc, w_loc, b_loc, gamma, beta, eps)[0]
