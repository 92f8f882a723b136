"""``torch.autograd.Function`` wrappers: forward AND backward run only kernels of libvilbert_hip.so.

What torch autograd contributes is the graph bookkeeping (and the few gradient additions where a
tensor feeds two consumers); every gradient *computation* below is a native kernel:
dgrad / wgrad MFMA GEMMs (bias gradient fused into wgrad), GELU / ReLU backward, LayerNorm backward
(deterministic two-stage column reduction), the two-pass attention backward, embedding scatter-add
and the counter-based dropout mask that forward and backward regenerate from (seed, index).
"""
import itertools

import torch
from torch.autograd import Function

from . import _native as N
from . import arena as _arena
from . import ops

_seed_counter = itertools.count(1)


def next_seed():
    """64-bit dropout seed: torch's initial seed (so torch.manual_seed makes runs repeatable) mixed
    with a per-process call counter. No device sync."""
    return ((torch.initial_seed() & 0xFFFFFFFF) * 0x9E3779B1 + next(_seed_counter) * 0x632BE59BD9B4E019) \
        & 0xFFFFFFFFFFFFFFFF


# ---------------------------------------------------------------------------------------------------------------
# Weight-gradient side streams. dW / db are not on the critical path of backward (only the optimizer / the gradient
# all-reduce read them), so every wgrad GEMM whose targets are FRESH gradient-arena slices is enqueued on a side stream
# of the stream its backward node runs on: the GPU then always holds a second, independent GEMM whose blocks fill the
# tail / prologue / epilogue bubbles of the dgrad chain (tools/gemm_lab LAB_STREAMS=1: a dgrad + wgrad pair runs
# 6-12 % faster on two streams than back to back on one). Rules that keep it race-free:
#   * only "fresh" claims go to the side stream; a second writer of a slice ("accum": tied weights, gradient
#     accumulation) first joins the side streams and then runs in order;
#   * the side streams are joined into the calling stream at the end of the backward pass (arena end-pass hook) and
#     before a data-parallel bucket is all-reduced (distributed.py);
#   * dy / x are record_stream()ed on the side stream (autograd may free them as soon as the node returns).
# VB_WGRAD_STREAM=0 disables it.
# ---------------------------------------------------------------------------------------------------------------
import collections as _collections
import os as _os

_WGRAD = {"on": _os.environ.get("VB_WGRAD_STREAM", "1") != "0", "streams": {}, "used": {}, "held": {}}


def set_wgrad_stream(on):
    prev, _WGRAD["on"] = _WGRAD["on"], bool(on)
    return prev


def _wgrad_stream(device, raw):
    """(side stream, torch.cuda.Stream object of the raw stream `raw`) for weight gradients of work launched on `raw`.
    Keyed on the raw handle so that the hot path never builds Stream objects (tools/host_profile.py); `raw` must be
    the stream that is current on `device` when this is called."""
    key = (device.index, raw)
    ent = _WGRAD["streams"].get(key)
    if ent is None:
        ent = _WGRAD["streams"][key] = (torch.cuda.Stream(device=device), torch.cuda.current_stream(device))
    return ent


def join_wgrad_streams(into=None, clear=False):
    """Make `into` (default: the current stream) wait for every weight-gradient side stream of ITS device with work in
    flight (the bookkeeping is per device: backward threads of different devices never touch each other's entry)."""
    used = _WGRAD["used"]
    if not used:
        return
    mine = used.get(into.device.index if into is not None else N.current_device())
    if not mine:
        return
    cur = into if into is not None else torch.cuda.current_stream()
    for st in list(mine.values()):
        cur.wait_stream(st)
    if clear:
        mine.clear()
        for q in _WGRAD["held"].values():
            q.clear()


def hold_for_side_stream(ws, *tensors):
    """A gradient tensor that a weight-gradient launch on side stream `ws` still READS and that the same backward node hands
    back to autograd (the skip-connection gradient passing through a dense + residual node) must not be updated in place
    while that launch is pending - and autograd's input buffers do exactly that (`old += new`) when they hold the only
    reference to the first arrival. The stream order of the in-place add covers the node's own stream, not its side
    stream (seen as a rare wrong dW with two processes time-slicing one GPU: tests/test_ddp_two_ranks_one_gpu.py, round
    6). A second reference, kept until the launch has finished (event on `ws`; under stream capture: until the end of the
    pass), sends the engine down its out-of-place path."""
    q = _WGRAD["held"].get(ws.cuda_stream)
    if q is None:
        q = _WGRAD["held"][ws.cuda_stream] = _collections.deque()
    if torch.cuda.is_current_stream_capturing():
        q.append((None, tensors))
        return
    ev = torch.cuda.Event()
    ev.record(ws)
    q.append((ev, tensors))
    while q and q[0][0] is not None and q[0][0].query():
        q.popleft()


_arena.END_PASS_HOOKS.append(lambda: join_wgrad_streams(clear=True))


class _Claims(object):
    """Gradient targets of a group of parameters for one backward node: gradient-arena slices where the arena
    manages the parameter (the kernels add in place, autograd gets an alias or None - see arena.py), else None
    (the launcher allocates)."""

    def __init__(self, params, needed):
        self.c = [(_arena.claim(p) if (n and p is not None) else (None, None, None, None)) for p, n in zip(params, needed)]
        if any(c[1] == "accum" for c in self.c):
            join_wgrad_streams()          # the earlier writer of that slice may still be running on a side stream

    def views(self):
        return [c[0] for c in self.c]

    def out(self, i, fallback):
        _view, mode, ar, idx = self.c[i]
        return _arena.result(mode, ar, idx, fallback)

    def fresh_or_none(self, i):
        """target for a kernel that OVERWRITES (LayerNorm column sums): only a fresh (zeroed, first-writer) slice."""
        return self.c[i][0] if self.c[i][1] == "fresh" else None

    def finish_overwrite(self, i, value):
        """value was computed by an overwriting kernel; returns what the backward node hands to autograd."""
        view, mode, ar, idx = self.c[i]
        if mode == "fresh":
            return ar.alias(idx)          # written in place
        if mode == "accum":
            view.add_(value)
            return None
        return value


def _wgrad(dy, x, weights, biases_present, need_w, need_b, launcher=None):
    """dW / db of the stacked segments into their arena slices (or fresh buffers); returns what autograd gets.
    launcher: ops.linear_bwd_weight (fp32 tensors, default) or ops16.linear_bwd_weight (bf16 tensors) - same contract."""
    launcher = launcher or ops.linear_bwd_weight
    nseg, seg_n = len(weights), weights[0].shape[0]
    cw = _Claims(weights, need_w)
    cb = _Claims(biases_present, need_b)
    wanted = [c for c, n in zip(cw.c, need_w) if n] + [c for c, n in zip(cb.c, need_b) if n]
    side = _WGRAD["on"] and dy.is_cuda and len(wanted) > 0 and all(c[1] == "fresh" for c in wanted)
    if side:
        ws, cur = _wgrad_stream(dy.device, N.raw_stream(dy.device.index))
        ws.wait_stream(cur)
        N.set_stream(ws)
        try:
            dws, dbs = launcher(dy, x, nseg, seg_n, need_b, dw_out=cw.views(), db_out=cb.views())
        finally:
            N.set_stream(cur)
        dy.record_stream(ws)
        x.record_stream(ws)
        _WGRAD["used"].setdefault(dy.device.index, {})[ws.cuda_stream] = ws
        hold_for_side_stream(ws, dy)     # (dy may also be on its way back to autograd as the residual's gradient)
    else:
        dws, dbs = launcher(dy, x, nseg, seg_n, need_b, dw_out=cw.views(), db_out=cb.views())
    return ([cw.out(s, dws[s]) if need_w[s] else None for s in range(nseg)],
            [cb.out(s, dbs[s]) if need_b[s] else None for s in range(nseg)])


class LinearFn(Function):
    """y = dropout(act(x @ cat(W).T + cat(b)), p) (+ residual). Inputs: x, residual, act, nseg, drop_p, W..., b...
    The dropout mask is regenerated in backward from the saved seed (vb_dropout on the incoming gradient)."""

    @staticmethod
    def forward(ctx, x, residual, act, nseg, drop_p, *wb):
        pad_cols = nseg < 0          # (flag folded into the sign of nseg: the output keeps a 16-byte row stride)
        nseg = abs(nseg)
        weights, biases = list(wb[:nseg]), list(wb[nseg:])
        seed = next_seed() if drop_p > 0.0 else 0
        y, pre = ops.linear_fwd(x, weights, biases, act, residual, want_preact=act is not None, drop_p=drop_p,
                                seed=seed, pad_cols=pad_cols)
        ctx.save_for_backward(x, pre, *weights, *[b for b in biases if b is not None])
        ctx.act, ctx.nseg = act, nseg
        ctx.drop = (drop_p, seed)
        ctx.has_bias = [b is not None for b in biases]
        return _tag_drop(y, drop_p, seed) if residual is not None and act is None else y

    @staticmethod
    def backward(ctx, dy):
        x, pre = ctx.saved_tensors[:2]
        nseg = ctx.nseg
        weights = list(ctx.saved_tensors[2:2 + nseg])
        rest = list(ctx.saved_tensors[2 + nseg:])
        biases = [rest.pop(0) if h else None for h in ctx.has_bias]
        seg_n, K = weights[0].shape[0], weights[0].shape[1]
        if not (ops._row_strided(dy) and ctx.drop[0] == 0.0 and ctx.act is None and not ctx.needs_input_grad[1]):
            dy = dy.contiguous()     # (a row-strided gradient of padded logits feeds the GEMMs in place)
        dres = dy if ctx.needs_input_grad[1] else None
        if ctx.drop[0] > 0.0:
            dy = _dropped(dy, ctx.drop)
        dpre = ops.act_bwd(dy, pre, ctx.act) if ctx.act is not None else dy
        dx = None
        if ctx.needs_input_grad[0]:
            dx = ops.linear_bwd_input(dpre, weights, K).view(x.shape)
        need_w = [bool(ctx.needs_input_grad[5 + s]) for s in range(nseg)]
        need_b = [ctx.has_bias[s] and ctx.needs_input_grad[5 + nseg + s] for s in range(nseg)]
        dws, dbs = [None] * nseg, [None] * nseg
        if any(need_w) or any(need_b):
            # (the fused launch computes every segment; segments nobody asked for land in scratch buffers)
            launcher = None
            if nseg == 1 and dpre.is_cuda and ops16.ragged_wgrad_ok(seg_n, K, dpre.numel() // max(seg_n, 1)):
                # bf16 mode: the wide ragged heads (30,522-wide MLM decoder) on the bf16 weight-gradient kernel
                launcher = lambda dy_, x_, nseg_, seg_n_, want_b, dw_out=None, db_out=None: \
                    ops16.linear_bwd_weight_ragged(dy_, x_, want_b, dw_out, db_out)
            dws, dbs = _wgrad(dpre, x, weights, biases, need_w, need_b, launcher)
        return (dx, dres, None, None, None) + tuple(dws) + tuple(dbs)


class FFNFn(Function):
    """y = dropout(act(x @ W1.T + b1) @ W2.T + b2, p) + x - the feed-forward block in front of its LayerNorm
    (reference vilbert.py:500-503 + :513-517 and the image / connection twins), as ONE autograd node.
    Backward: the dgrad through W1 adds the skip-connection gradient in its GEMM epilogue (no torch add). The
    activation backward is fused too: the forward epilogue of the up-projection stores act'(pre-activation) next to
    the activation (one extra exp per element where erf is computed anyway) and the dgrad through W2 multiplies by
    it in its epilogue - no erf / exp in any backward kernel, no separate elementwise pass over [M, intermediate]."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, act, drop_p):
        seed = next_seed() if drop_p > 0.0 else 0
        h, dact = ops.linear_fwd(x, [w1], [b1], act, None, want_act_grad=True)
        y, _ = ops.linear_fwd(h, [w2], [b2], None, x, drop_p=drop_p, seed=seed)
        ctx.save_for_backward(x, dact, h, w1, w2, *[b for b in (b1, b2) if b is not None])
        ctx.act, ctx.drop = act, (drop_p, seed)
        ctx.has_bias = (b1 is not None, b2 is not None)
        return _tag_drop(y, drop_p, seed)

    @staticmethod
    def backward(ctx, dy):
        x, dact, h, w1, w2 = ctx.saved_tensors[:5]
        rest = list(ctx.saved_tensors[5:])
        b1 = rest.pop(0) if ctx.has_bias[0] else None
        b2 = rest.pop(0) if ctx.has_bias[1] else None
        dy = dy.contiguous()
        dyd = _dropped(dy, ctx.drop) if ctx.drop[0] > 0.0 else dy
        inter, hidden = w1.shape[0], w1.shape[1]
        dpre = ops.linear_bwd_input(dyd, [w2], inter, mul=dact)
        dw2 = db2 = dw1 = db1 = dx = None
        nb2 = bool(ctx.has_bias[1] and ctx.needs_input_grad[4])
        if ctx.needs_input_grad[3] or nb2:
            (dw2,), (db2,) = _wgrad(dyd, h, [w2], [b2], [bool(ctx.needs_input_grad[3])], [nb2])
        if ctx.needs_input_grad[0]:
            dx = ops.linear_bwd_input(dpre, [w1], hidden, residual=dy).view(x.shape)
        nb1 = bool(ctx.has_bias[0] and ctx.needs_input_grad[2])
        if ctx.needs_input_grad[1] or nb1:
            (dw1,), (db1,) = _wgrad(dpre, x, [w1], [b1], [bool(ctx.needs_input_grad[1])], [nb1])
        return dx, dw1, db1, dw2, db2, None, None


def _ln_bwd(dy, x, mean, rstd, gamma, beta, need_g, need_b, drop=None):
    """LayerNorm backward with dgamma / dbeta written straight into their arena slices when those are fresh.
    drop = (p, seed) of the dense layer in front: dx comes back tagged with its dropout-masked twin (see _DROP_HINT)."""
    c = _Claims([gamma, beta], [need_g, need_b])
    res = ops.layernorm_bwd(dy, x, mean, rstd, gamma, c.fresh_or_none(0), c.fresh_or_none(1), drop=drop)
    dx, dgamma, dbeta = res[:3]
    if len(res) == 4:
        dx._vb_dropped = (drop[0], drop[1], res[3], dx.data_ptr(), dx._version)
    return dx, (c.finish_overwrite(0, dgamma) if need_g else None), (c.finish_overwrite(1, dbeta) if need_b else None)


# Dropout-mask hand-over between two autograd nodes (round 3, -66 launches per step). Every BertSelfOutput / BertOutput /
# BertBiOutput is y = LayerNorm(dropout(dense(h), p) + x): LinearFn / FFNFn produce the pre-LayerNorm sum, LayerNormFn
# consumes it. In backward the LayerNorm node computes dx and the dense node needs dropout(dx) - the same elements, the
# mask is a function of (seed, index). The forward of the dense node tags its output tensor with (p, seed)
# (`_vb_drop`), LayerNormFn.forward copies the tag into its context, its backward lets the LayerNorm kernel write the
# masked gradient next to dx and tags dx with it (`_vb_dropped`), the dense node's backward uses the twin when the tag
# matches its own (p, seed). Pure optimisation: without a matching tag every node falls back to its own vb_dropout.
def _tag_drop(y, drop_p, seed):
    if drop_p > 0.0:
        y._vb_drop = (drop_p, seed)
    return y


def _dropped(dy, drop):
    """dropout(dy) for the dense node's backward: the twin the LayerNorm kernel already wrote, or a vb_dropout launch."""
    tag = getattr(dy, "_vb_dropped", None)
    # the twin equals dropout(dy) only while dy still IS the tensor the LayerNorm kernel wrote: same storage and no
    # in-place write since (autograd's InputBuffer may accumulate a second consumer's gradient in place into a tensor it
    # holds the only reference to - the Python attribute would survive that, the version counter does not)
    if (tag is not None and tag[0] == drop[0] and tag[1] == drop[1] and tag[2].shape == dy.shape
            and tag[3] == dy.data_ptr() and tag[4] == dy._version):
        return tag[2]
    return ops.dropout(dy, drop[0], drop[1])


class LayerNormFn(Function):
    """TF-style LayerNorm of one tensor."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps):
        y, mean, rstd = ops.layernorm_fwd(x, gamma, beta, eps, None, want_stats=True)
        ctx.save_for_backward(x, mean, rstd, gamma, beta)
        ctx.drop_hint = getattr(x, "_vb_drop", None)      # x = dropout(dense(h)) + residual of the node in front
        return y

    @staticmethod
    def backward(ctx, dy):
        x, mean, rstd, gamma, beta = ctx.saved_tensors
        dx, dgamma, dbeta = _ln_bwd(dy, x, mean, rstd, gamma, beta, ctx.needs_input_grad[1], ctx.needs_input_grad[2],
                                    drop=ctx.drop_hint)
        out = dx.view(x.shape)
        tag = getattr(dx, "_vb_dropped", None)
        if tag is not None:
            out._vb_dropped = (tag[0], tag[1], tag[2].view(x.shape), out.data_ptr(), out._version)
        return out, dgamma, dbeta, None


class DropoutFn(Function):
    """y = dropout(x, p) (+ residual)."""

    @staticmethod
    def forward(ctx, x, residual, p):
        ctx.p, ctx.seed = p, next_seed()
        return ops.dropout(x, p, ctx.seed, residual)

    @staticmethod
    def backward(ctx, dy):
        dy = dy.contiguous()
        dx = ops.dropout(dy, ctx.p, ctx.seed) if ctx.needs_input_grad[0] else None
        return dx, (dy if ctx.needs_input_grad[1] else None), None


def dropout(x, p, residual=None):
    return DropoutFn.apply(x, residual, p)


def _check_keys_for_backward(n_keys, want_probs):
    """More than ops.MAX_KEYS keys run chunk by chunk, forward and backward (ops._attention_fwd_long / _attention_bwd_long);
    only the probabilities tensor (`visualization`) does not exist there - say so where the call is made."""
    if n_keys > ops.MAX_KEYS and want_probs:
        raise RuntimeError("attention: %d keys with attention maps - one launch serves at most %d keys, and the chunked path "
                           "of longer key sequences (stacked retrieval options, in_batch_pairs) returns no probabilities"
                           % (n_keys, ops.MAX_KEYS))


class SelfAttnFn(Function):
    """ctx = attention over one fused [q | k | v] projection; dqkv is written in one buffer."""

    @staticmethod
    def forward(ctx, qkv, mask_add, heads, drop_p, want_probs):
        H = qkv.shape[-1] // 3
        _check_keys_for_backward(qkv.shape[1], want_probs)
        seed = next_seed() if drop_p > 0.0 else 0
        out, probs, lse = ops.attention_fwd(qkv[..., :H], qkv[..., H:2 * H], qkv[..., 2 * H:], mask_add, heads,
                                            want_probs, True, drop_p, seed)
        ctx.save_for_backward(qkv, mask_add, lse)
        ctx.meta = (heads, drop_p, seed)
        if probs is None:
            probs = qkv.new_empty(0)
        ctx.mark_non_differentiable(probs)
        ctx.set_materialize_grads(False)      # no zeros() for the gradient slot of the non-differentiable probs
        return out, probs

    @staticmethod
    def backward(ctx, d_out, _d_probs):
        qkv, mask_add, lse = ctx.saved_tensors
        if d_out is None:
            return None, None, None, None, None
        heads, drop_p, seed = ctx.meta
        H = qkv.shape[-1] // 3
        dqkv = torch.empty_like(qkv)
        ops.attention_bwd(d_out, qkv[..., :H], qkv[..., H:2 * H], qkv[..., 2 * H:], mask_add, heads, lse,
                          dqkv[..., :H], dqkv[..., H:2 * H], dqkv[..., 2 * H:], drop_p, seed)
        return dqkv, None, None, None, None


class BiAttnFn(Function):
    """Both directions of the co-attention (reference vilbert.py:768-809) over the two fused projections:
    ctx1 = attn(q2; k1, v1 | mask1) for the text stream, ctx2 = attn(q1; k2, v2 | mask2) for the image
    stream. The backward fills dqkv1 / dqkv2 with each slice written exactly once."""

    @staticmethod
    def forward(ctx, qkv1, qkv2, mask1, mask2, heads, p1, p2, want_probs):
        H = qkv1.shape[-1] // 3
        _check_keys_for_backward(max(qkv1.shape[1], qkv2.shape[1]), want_probs)
        s1 = next_seed() if p1 > 0.0 else 0
        s2 = next_seed() if p2 > 0.0 else 0
        q1, k1, v1 = qkv1[..., :H], qkv1[..., H:2 * H], qkv1[..., 2 * H:]
        q2, k2, v2 = qkv2[..., :H], qkv2[..., H:2 * H], qkv2[..., 2 * H:]
        ctx1, probs1, lse1 = ops.attention_fwd(q2, k1, v1, mask1, heads, want_probs, True, p1, s1)
        ctx2, probs2, lse2 = ops.attention_fwd(q1, k2, v2, mask2, heads, want_probs, True, p2, s2)
        ctx.save_for_backward(qkv1, qkv2, mask1, mask2, lse1, lse2)
        ctx.meta = (heads, p1, p2, s1, s2)
        if probs1 is None:
            probs1, probs2 = qkv1.new_empty(0), qkv1.new_empty(0)
        ctx.mark_non_differentiable(probs1, probs2)
        ctx.set_materialize_grads(False)
        return ctx1, ctx2, probs1, probs2

    @staticmethod
    def backward(ctx, d1, d2, _dp1, _dp2):
        qkv1, qkv2, mask1, mask2, lse1, lse2 = ctx.saved_tensors
        heads, p1, p2, s1, s2 = ctx.meta
        if d1 is None and d2 is None:
            return (None,) * 8
        if d1 is None:                         # only one direction reached the loss: the other one's gradient is zero
            d1 = torch.zeros((qkv2.shape[0], qkv2.shape[1], qkv1.shape[-1] // 3), dtype=qkv1.dtype, device=qkv1.device)
        if d2 is None:
            d2 = torch.zeros((qkv1.shape[0], qkv1.shape[1], qkv1.shape[-1] // 3), dtype=qkv1.dtype, device=qkv1.device)
        H = qkv1.shape[-1] // 3
        sl = lambda t: (t[..., :H], t[..., H:2 * H], t[..., 2 * H:])
        q1, k1, v1 = sl(qkv1)
        q2, k2, v2 = sl(qkv2)
        dqkv1, dqkv2 = torch.empty_like(qkv1), torch.empty_like(qkv2)
        dq1, dk1, dv1 = sl(dqkv1)
        dq2, dk2, dv2 = sl(dqkv2)
        ops.attention_bwd(d1, q2, k1, v1, mask1, heads, lse1, dq2, dk1, dv1, p1, s1)
        ops.attention_bwd(d2, q1, k2, v2, mask2, heads, lse2, dq1, dk2, dv2, p2, s2)
        return dqkv1, dqkv2, None, None, None, None, None, None


class TextEmbedFn(Function):
    @staticmethod
    def forward(ctx, ids, seg, word, pos, typ, gamma, beta, eps, task_ids, task_emb):
        out, mean, rstd, presum = ops.text_embed_ln_fwd(ids, seg, word, pos, typ, gamma, beta, eps, task_ids,
                                                        task_emb, want_stats=True)
        ctx.save_for_backward(ids, seg, task_ids, presum, mean, rstd, gamma, beta, word, pos, typ, task_emb)
        return out

    @staticmethod
    def backward(ctx, dy):
        ids, seg, task_ids, presum, mean, rstd, gamma, beta, word, pos, typ, task_emb = ctx.saved_tensors
        need = ctx.needs_input_grad
        dx, dgamma, dbeta = _ln_bwd(dy, presum, mean, rstd, gamma, beta, need[5], need[6])
        tables = [word, pos, typ, task_emb]
        want = [bool(need[2]), bool(need[3]), bool(need[4]), bool(task_emb is not None and need[9])]
        c = _Claims(tables, want)
        got = ops.text_embed_bwd(dx, ids, seg, task_ids, tuple(word.shape), tuple(pos.shape), tuple(typ.shape),
                                 tuple(task_emb.shape) if task_emb is not None else None, out=c.views())
        dword, dpos, dtype, dtask = [(c.out(i, got[i]) if want[i] else None) for i in range(4)]
        return None, None, dword, dpos, dtype, dgamma, dbeta, None, None, dtask


class ImageEmbedFn(Function):
    @staticmethod
    def forward(ctx, feat_proj, loc, w_loc, b_loc, gamma, beta, eps):
        out, mean, rstd, presum = ops.image_embed_ln_fwd(feat_proj, loc, w_loc, b_loc, gamma, beta, eps,
                                                         want_stats=True)
        ctx.save_for_backward(loc, presum, mean, rstd, gamma, beta, w_loc, b_loc)
        return out

    @staticmethod
    def backward(ctx, dy):
        loc, presum, mean, rstd, gamma, beta, w_loc, b_loc = ctx.saved_tensors
        need = ctx.needs_input_grad
        dsum, dgamma, dbeta = _ln_bwd(dy, presum, mean, rstd, gamma, beta, need[4], need[5])
        (dw_loc,), (db_loc,) = _wgrad(dsum, loc.reshape(-1, 5), [w_loc], [b_loc], [bool(need[2])], [bool(need[3])])
        return dsum, None, dw_loc, db_loc, dgamma, dbeta, None


class CrossEntropyFn(torch.autograd.Function):
    """nn.CrossEntropyLoss(ignore_index) (mean over counted rows) - reference vilbert.py:1453,1578-1585."""

    @staticmethod
    def forward(ctx, logits, labels, ignore_index):
        loss, lse, count = ops.xent_fwd(logits, labels, ignore_index)
        ctx.save_for_backward(logits, labels, lse, count)
        ctx.ignore_index = ignore_index
        return loss

    @staticmethod
    def backward(ctx, grad_loss):
        logits, labels, lse, count = ctx.saved_tensors
        return ops.xent_bwd(grad_loss, logits, labels, ctx.ignore_index, lse, count), None, None


class KLDivFn(torch.autograd.Function):
    """sum(KLDivLoss(reduction="none")(log_softmax(scores), target)) / divisor - reference :1454,1516-1522."""

    @staticmethod
    def forward(ctx, scores, target, divisor):
        loss, lse, tsum = ops.kl_fwd(scores, target, divisor)
        ctx.save_for_backward(scores, target, lse, tsum)
        ctx.divisor = divisor
        return loss

    @staticmethod
    def backward(ctx, grad_loss):
        scores, target, lse, tsum = ctx.saved_tensors
        return ops.kl_bwd(grad_loss, scores, target, lse, tsum, ctx.divisor), None, None


# ---------------------------------------------------------------------------------------------------------------
# bf16 TRAINING path (round 5; ops16.py, csrc/gemm_bf16.hip): the same autograd nodes on bfloat16 activations. Parameters,
# their gradients (arena slices, fp32), LayerNorm statistics and the softmax statistics stay fp32. What the reference does
# with `model.half()` + apex FP16_Optimizer (train_concap.py:443-461,504-505), without loss scaling (bf16 keeps fp32's
# exponent range).
# ---------------------------------------------------------------------------------------------------------------
from . import ops16  # noqa: E402


class CastFn(Function):
    """fp32 <-> bfloat16 at the edges of the bf16 stream (embeddings in, sequence outputs out); backward = the other cast."""

    @staticmethod
    def forward(ctx, x, to_bf16):
        ctx.to_bf16 = to_bf16
        return ops16.cast_bf16(x) if to_bf16 else ops16.cast_f32(x)

    @staticmethod
    def backward(ctx, dy):
        return (ops16.cast_f32(dy) if ctx.to_bf16 else ops16.cast_bf16(dy)), None


def _dropped16(dy, drop):
    tag = getattr(dy, "_vb_dropped", None)
    if (tag is not None and tag[0] == drop[0] and tag[1] == drop[1] and tag[2].shape == dy.shape
            and tag[3] == dy.data_ptr() and tag[4] == dy._version):
        return tag[2]
    return ops16.dropout(dy, drop[0], drop[1])


class Linear16Fn(Function):
    """y = dropout(x @ cat(W).T + cat(b), p) (+ residual) on bf16 tensors (y fp32 when out_f32: the image-feature projection
    in front of the fp32 embedding kernel). Inputs: x, residual, nseg, drop_p, out_f32, W..., b... (a linear with an
    activation of its own is not an autograd node of this path: the FFN is FFN16Fn, functional.linear sends the rest
    through the fp32 node)."""

    @staticmethod
    def forward(ctx, x, residual, nseg, drop_p, out_f32, *wb):
        weights, biases = list(wb[:nseg]), list(wb[nseg:])
        seed = next_seed() if drop_p > 0.0 else 0
        y, _ = ops16.linear_fwd(x, weights, biases, None, residual, drop_p=drop_p, seed=seed, out_f32=out_f32)
        ctx.save_for_backward(x, *weights, *[b for b in biases if b is not None])
        ctx.nseg = nseg
        ctx.drop = (drop_p, seed)
        ctx.has_bias = [b is not None for b in biases]
        return _tag_drop(y, drop_p, seed) if residual is not None else y

    @staticmethod
    def backward(ctx, dy):
        x = ctx.saved_tensors[0]
        nseg = ctx.nseg
        weights = list(ctx.saved_tensors[1:1 + nseg])
        rest = list(ctx.saved_tensors[1 + nseg:])
        biases = [rest.pop(0) if h else None for h in ctx.has_bias]
        K = weights[0].shape[1]
        dy = dy.contiguous()
        if dy.dtype != ops16.BF16:          # fp32 output: the gradient arrives in fp32
            dy = ops16.cast_bf16(dy)
        dres = dy if ctx.needs_input_grad[1] else None
        if ctx.drop[0] > 0.0:
            dy = _dropped16(dy, ctx.drop)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = ops16.linear_bwd_input(dy, weights, biases, K).view(x.shape)
        need_w = [bool(ctx.needs_input_grad[5 + s]) for s in range(nseg)]
        need_b = [ctx.has_bias[s] and ctx.needs_input_grad[5 + nseg + s] for s in range(nseg)]
        dws, dbs = [None] * nseg, [None] * nseg
        if any(need_w) or any(need_b):
            dws, dbs = _wgrad(dy, x, weights, biases, need_w, need_b, launcher=ops16.linear_bwd_weight)
        return (dx, dres, None, None, None) + tuple(dws) + tuple(dbs)


class FFN16Fn(Function):
    """y = dropout(gelu(x @ W1.T + b1) @ W2.T + b2, p) + x on bf16 tensors - FFNFn's structure: the up-projection's epilogue
    stores gelu'(pre) next to the activation, the dgrad through W2 multiplies by it, the dgrad through W1 adds the skip
    gradient; no elementwise pass over [M, intermediate] in either direction."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, drop_p):
        seed = next_seed() if drop_p > 0.0 else 0
        h, dact = ops16.linear_fwd(x, [w1], [b1], "gelu", want_act_grad=True)
        y, _ = ops16.linear_fwd(h, [w2], [b2], None, x, drop_p=drop_p, seed=seed)
        ctx.save_for_backward(x, dact, h, w1, w2, *[b for b in (b1, b2) if b is not None])
        ctx.drop = (drop_p, seed)
        ctx.has_bias = (b1 is not None, b2 is not None)
        return _tag_drop(y, drop_p, seed)

    @staticmethod
    def backward(ctx, dy):
        x, dact, h, w1, w2 = ctx.saved_tensors[:5]
        rest = list(ctx.saved_tensors[5:])
        b1 = rest.pop(0) if ctx.has_bias[0] else None
        b2 = rest.pop(0) if ctx.has_bias[1] else None
        dy = dy.contiguous()
        dyd = _dropped16(dy, ctx.drop) if ctx.drop[0] > 0.0 else dy
        inter, hidden = w1.shape[0], w1.shape[1]
        dpre = ops16.linear_bwd_input(dyd, [w2], [b2], inter, mul=dact)
        dw2 = db2 = dw1 = db1 = dx = None
        nb2 = bool(ctx.has_bias[1] and ctx.needs_input_grad[4])
        if ctx.needs_input_grad[3] or nb2:
            (dw2,), (db2,) = _wgrad(dyd, h, [w2], [b2], [bool(ctx.needs_input_grad[3])], [nb2], launcher=ops16.linear_bwd_weight)
        if ctx.needs_input_grad[0]:
            dx = ops16.linear_bwd_input(dpre, [w1], [b1], hidden, residual=dy).view(x.shape)
        nb1 = bool(ctx.has_bias[0] and ctx.needs_input_grad[2])
        if ctx.needs_input_grad[1] or nb1:
            (dw1,), (db1,) = _wgrad(dpre, x, [w1], [b1], [bool(ctx.needs_input_grad[1])], [nb1], launcher=ops16.linear_bwd_weight)
        return dx, dw1, db1, dw2, db2, None


class LayerNorm16Fn(Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, eps):
        y, mean, rstd = ops16.layernorm_fwd(x, gamma, beta, eps, want_stats=True)
        ctx.save_for_backward(x, mean, rstd, gamma, beta)
        ctx.drop_hint = getattr(x, "_vb_drop", None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, mean, rstd, gamma, beta = ctx.saved_tensors
        need_g, need_b = ctx.needs_input_grad[1], ctx.needs_input_grad[2]
        c = _Claims([gamma, beta], [need_g, need_b])
        res = ops16.layernorm_bwd(dy, x, mean, rstd, gamma, c.fresh_or_none(0), c.fresh_or_none(1), drop=ctx.drop_hint)
        dx, dgamma, dbeta = res[:3]
        out = dx.view(x.shape)
        if len(res) == 4:
            out._vb_dropped = (ctx.drop_hint[0], ctx.drop_hint[1], res[3].view(x.shape), out.data_ptr(), out._version)
        return (out, (c.finish_overwrite(0, dgamma) if need_g else None), (c.finish_overwrite(1, dbeta) if need_b else None),
                None)


class SelfAttn16Fn(Function):
    """Attention over one fused bf16 [q | k | v] projection; context and dqkv bf16, softmax statistics fp32."""

    @staticmethod
    def forward(ctx, qkv, mask_add, heads, drop_p):
        H = qkv.shape[-1] // 3
        seed = next_seed() if drop_p > 0.0 else 0
        out, lse = ops16.attention_fwd(qkv[..., :H], qkv[..., H:2 * H], qkv[..., 2 * H:], mask_add, heads, True, drop_p, seed)
        ctx.save_for_backward(qkv, mask_add, lse)
        ctx.meta = (heads, drop_p, seed)
        return out

    @staticmethod
    def backward(ctx, d_out):
        qkv, mask_add, lse = ctx.saved_tensors
        heads, drop_p, seed = ctx.meta
        H = qkv.shape[-1] // 3
        dqkv = torch.empty_like(qkv)
        ops16.attention_bwd(d_out, qkv[..., :H], qkv[..., H:2 * H], qkv[..., 2 * H:], mask_add, heads, lse,
                            dqkv[..., :H], dqkv[..., H:2 * H], dqkv[..., 2 * H:], drop_p, seed)
        return dqkv, None, None, None


class BiAttn16Fn(Function):
    """Both directions of the co-attention on the two fused bf16 projections (BiAttnFn's structure)."""

    @staticmethod
    def forward(ctx, qkv1, qkv2, mask1, mask2, heads, p1, p2):
        H = qkv1.shape[-1] // 3
        s1 = next_seed() if p1 > 0.0 else 0
        s2 = next_seed() if p2 > 0.0 else 0
        sl = lambda t: (t[..., :H], t[..., H:2 * H], t[..., 2 * H:])
        q1, k1, v1 = sl(qkv1)
        q2, k2, v2 = sl(qkv2)
        ctx1, lse1 = ops16.attention_fwd(q2, k1, v1, mask1, heads, True, p1, s1)
        ctx2, lse2 = ops16.attention_fwd(q1, k2, v2, mask2, heads, True, p2, s2)
        ctx.save_for_backward(qkv1, qkv2, mask1, mask2, lse1, lse2)
        ctx.meta = (heads, p1, p2, s1, s2)
        ctx.set_materialize_grads(False)
        return ctx1, ctx2

    @staticmethod
    def backward(ctx, d1, d2):
        qkv1, qkv2, mask1, mask2, lse1, lse2 = ctx.saved_tensors
        heads, p1, p2, s1, s2 = ctx.meta
        if d1 is None and d2 is None:
            return (None,) * 7
        H = qkv1.shape[-1] // 3
        if d1 is None:
            d1 = torch.zeros((max(qkv1.shape[0], qkv2.shape[0]), qkv2.shape[1], H), dtype=qkv1.dtype, device=qkv1.device)
        if d2 is None:
            d2 = torch.zeros((max(qkv1.shape[0], qkv2.shape[0]), qkv1.shape[1], H), dtype=qkv1.dtype, device=qkv1.device)
        sl = lambda t: (t[..., :H], t[..., H:2 * H], t[..., 2 * H:])
        q1, k1, v1 = sl(qkv1)
        q2, k2, v2 = sl(qkv2)
        dqkv1, dqkv2 = torch.empty_like(qkv1), torch.empty_like(qkv2)
        dq1, dk1, dv1 = sl(dqkv1)
        dq2, dk2, dv2 = sl(dqkv2)
        ops16.attention_bwd(d1, q2, k1, v1, mask1, heads, lse1, dq2, dk1, dv1, p1, s1)
        ops16.attention_bwd(d2, q1, k2, v2, mask2, heads, lse2, dq1, dk2, dv2, p2, s2)
        return dqkv1, dqkv2, None, None, None, None, None
