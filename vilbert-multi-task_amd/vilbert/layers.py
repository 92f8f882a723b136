"""Whole-layer autograd nodes on the native layer launcher (csrc/layers.hip: vb_layer_fwd / vb_layer_bwd; round 6).

A BertLayer / BertImageLayer (reference vilbert.py:527-533, 688-694) is ONE autograd node and ONE call across the C ABI per
direction; a BertConnectionLayer (:871-900) is three (the co-attention block, then the image-side and the text-side output +
feed-forward blocks, which still run on two streams). The kernels, their order and their arithmetic are those of the per-op
nodes in autograd_ops.py (the launcher calls the same entry points; the dropout seeds are drawn in the same order, so the
two paths give bit-identical results: tests/test_layers_native_gpu.py) - what is gone is the host work between the
kernels: ~30 `torch.autograd.Function.apply`, ~90 `torch.empty` and ~25 ctypes argument structs per layer and step. At the
reference's per-GPU batch 64 that host work, not the GPU, bounded the eager step.

Buffers: one allocation per call holds everything the forward saves for backward, one more the temporaries of a backward
call; the launcher gets plain pointers into them. Weight gradients go straight into their gradient-arena slices (arena.py
protocol: fresh slices are returned to autograd as aliases) and, when every target is a fresh slice, onto the
weight-gradient side stream (autograd_ops.py rules).

`VB_LAYER_NATIVE=0` (or `set_native(False)`) keeps the per-op path; the modes that path serves alone: fp8 / MX inference,
attention maps (`visualization`), dynamic attention, activations other than GELU, shapes the bf16 kernels do not take,
partially frozen layers.
"""
import ctypes
import os

import torch
from torch.autograd import Function

from . import _native as N
from . import arena as _arena
from . import autograd_ops as A
from . import ops
from . import ops16

BF16 = torch.bfloat16
_STATE = {"on": os.environ.get("VB_LAYER_NATIVE", "1") != "0", "calls": 0}


def set_native(on):
    """Switch the whole-layer launcher on / off at run time; returns the previous setting."""
    prev, _STATE["on"] = _STATE["on"], bool(on)
    return prev


def native_calls():
    """Number of vb_layer_fwd / vb_layer_bwd calls made so far (tests: which path served a model)."""
    return _STATE["calls"]


def _align(n):
    return (n + 255) // 256 * 256


class _Carver(object):
    """Sub-buffers of one flat allocation: add() collects (name, bytes), alloc() makes the tensor, ptr() hands out addresses."""

    def __init__(self):
        self.off, self.total = {}, 0

    def add(self, name, nbytes):
        self.off[name] = self.total
        self.total += _align(int(nbytes))

    def alloc(self, device):
        self.buf = torch.empty(max(self.total, 256), dtype=torch.uint8, device=device)
        self.base = self.buf.data_ptr()
        return self.buf

    def ptr(self, name):
        return self.base + self.off[name]


class _P(object):
    """A sub-buffer of a flat allocation, by address (what the fill helpers need of a tensor)."""
    __slots__ = ("_p",)

    def __init__(self, ptr):
        self._p = ptr

    def data_ptr(self):
        return self._p


def _fill_linear(L, weights, biases, b16):
    seg_n, K = weights[0].shape
    L.nseg, L.seg_n, L.K = len(weights), seg_n, K
    for s, (w, b) in enumerate(zip(weights, biases)):
        L.w[s] = w.data_ptr()
        L.bias[s] = b.data_ptr()
    if b16:
        w16, wt16 = ops16.shadows(weights)
        L.w16, L.wt16 = w16.data_ptr(), wt16.data_ptr()


def _linear_ok(weights, biases, b16):
    seg_n, K = weights[0].shape
    if any(b is None for b in biases) or any(w.shape != (seg_n, K) or not w.is_contiguous() for w in weights):
        return False
    if b16:
        return ops16.eligible(K, len(weights) * seg_n, seg_n, "gelu")
    return True


class _Targets(object):
    """Gradient targets of a node's parameters (autograd_ops._Claims protocol), resolved to raw pointers for the launcher."""

    def __init__(self, params, device):
        self.claims = [_arena.claim(p) for p in params]            # (view, mode, arena, index)
        self.params = params
        if any(c[1] == "accum" for c in self.claims):
            A.join_wgrad_streams()          # an earlier writer of such a slice may still run on a side stream
        need = sum((p.numel() + 3) // 4 * 4 for p, c in zip(params, self.claims) if c[0] is None)
        self.scratch = torch.zeros(need, dtype=torch.float32, device=device) if need else None
        self.views, off = [], 0
        for p, c in zip(params, self.claims):
            if c[0] is not None:
                self.views.append(c[0])
            else:
                n = p.numel()
                self.views.append(self.scratch[off:off + n].view(p.shape))
                off += (n + 3) // 4 * 4
        self.all_fresh = all(c[1] == "fresh" for c in self.claims)

    def ptr(self, i):
        return self.views[i].data_ptr()

    def overwrite_target(self, i, device):
        """LayerNorm column sums are OVERWRITTEN by their kernel: a fresh (zeroed, first-writer) slice or private scratch takes
        them in place, an accumulating slice gets them through a temporary (finish())."""
        if self.claims[i][1] == "accum":
            t = torch.empty_like(self.views[i])
            self._tmp = getattr(self, "_tmp", [])
            self._tmp.append((i, t))
            return t.data_ptr()
        return self.views[i].data_ptr()

    def finish(self):
        for i, t in getattr(self, "_tmp", []):
            self.views[i].add_(t)
        out = []
        for c, v in zip(self.claims, self.views):
            out.append(_arena.result(c[1], c[2], c[3], v) if c[0] is not None else v)
        return out


def _side_stream_for(device, targets):
    """(raw handle or None, stream object) of the weight-gradient side stream this backward call may use."""
    if not (A._WGRAD["on"] and targets.all_fresh) or torch.cuda.is_current_stream_capturing():
        return None, None
    ws, _cur = A._wgrad_stream(device, N.raw_stream(device.index))
    A._WGRAD["used"].setdefault(device.index, {})[ws.cuda_stream] = ws
    return ws.cuda_stream, ws


def _set_linear_targets(L, tg, idx_w, idx_b):
    for s, (iw, ib) in enumerate(zip(idx_w, idx_b)):
        L.dw[s] = tg.ptr(iw)
        L.dbias[s] = tg.ptr(ib)


# ---------------------------------------------------------------------------------------------------------------------
# output projection + feed-forward block:  y = LN2(dropout(W2 gelu(W1 a + b1) + b2) + a),  a = LN1(dropout(Wo ctx + bo) + x)
# params: o.w, o.b, ln1.g, ln1.b, f1.w, f1.b, f2.w, f2.b, ln2.g, ln2.b
# ---------------------------------------------------------------------------------------------------------------------
def _ffn_sizes(M, H, I, es, training):
    c = _Carver()
    c.add("sum1", M * H * es)
    c.add("a1", M * H * es)
    c.add("h", M * I * es)
    c.add("sum2", M * H * es)
    if training:
        c.add("dact", M * I * es)
        for n in ("mean1", "rstd1", "mean2", "rstd2"):
            c.add(n, M * 4)
    return c


def _fill_ffn_fwd(f, c, ctx_t, x, y, p, meta, b16, training):
    M, Hc, H, I = meta["M"], meta["Hc"], meta["H"], meta["I"]
    f.M, f.Hc, f.H, f.I = M, Hc, H, I
    f.ctx, f.x = ctx_t.data_ptr(), x.data_ptr()
    _fill_linear(f.o, [p[0]], [p[1]], b16)
    f.ln1.gamma, f.ln1.beta = p[2].data_ptr(), p[3].data_ptr()
    _fill_linear(f.f1, [p[4]], [p[5]], b16)
    _fill_linear(f.f2, [p[6]], [p[7]], b16)
    f.ln2.gamma, f.ln2.beta = p[8].data_ptr(), p[9].data_ptr()
    f.eps, f.p_o, f.p_f = meta["eps"], meta["p_o"], meta["p_f"]
    f.seed_o, f.seed_f = meta["seed_o"], meta["seed_f"]
    f.sum1, f.a1, f.h, f.sum2 = c.ptr("sum1"), c.ptr("a1"), c.ptr("h"), c.ptr("sum2")
    f.y = y.data_ptr()
    if training:
        f.dact = c.ptr("dact")
        f.mean1, f.rstd1, f.mean2, f.rstd2 = c.ptr("mean1"), c.ptr("rstd1"), c.ptr("mean2"), c.ptr("rstd2")


def _ffn_bwd_sizes(M, Hc, H, I, es, p_o, p_f):
    c = _Carver()
    c.add("d_sum2", M * H * es)
    if p_f > 0.0:
        c.add("d_sum2_drop", M * H * es)
    c.add("d_pre", M * I * es)
    c.add("d_a1", M * H * es)
    if p_o > 0.0:
        c.add("d_sum1_drop", M * H * es)
    return c


def _ln_ws(device, rows, cols, b16):
    floats = (rows + 15) // 16 * 2 * cols if b16 else N.lib().vb_layernorm_bwd_workspace(rows, cols)
    return ops16._ln_workspace(device, floats)


def _fill_ffn_bwd(f, t, dy, d_sum1, d_ctx, tg, base, ws, meta):
    """t: carver of the backward temporaries; tg: targets of the 10 parameters starting at index `base`."""
    f.dy = dy.data_ptr()
    f.d_sum2, f.d_pre, f.d_a1 = t.ptr("d_sum2"), t.ptr("d_pre"), t.ptr("d_a1")
    f.d_sum2_drop = t.ptr("d_sum2_drop") if meta["p_f"] > 0.0 else None
    f.d_sum1_drop = t.ptr("d_sum1_drop") if meta["p_o"] > 0.0 else None
    f.d_sum1, f.d_ctx = d_sum1.data_ptr(), d_ctx.data_ptr()
    f.ln_ws = ws.data_ptr()
    dev = dy.device
    _set_linear_targets(f.o, tg, [base + 0], [base + 1])
    f.ln1.dgamma, f.ln1.dbeta = tg.overwrite_target(base + 2, dev), tg.overwrite_target(base + 3, dev)
    _set_linear_targets(f.f1, tg, [base + 4], [base + 5])
    _set_linear_targets(f.f2, tg, [base + 6], [base + 7])
    f.ln2.dgamma, f.ln2.dbeta = tg.overwrite_target(base + 8, dev), tg.overwrite_target(base + 9, dev)


def _call(fn, a, what):
    _STATE["calls"] += 1
    N.check(fn(N.stream_ptr(), ctypes.byref(a)), what)


class FfnBlockFn(Function):
    """y = output + feed-forward block of one stream (connection layers: one per stream, on two HIP streams)."""

    @staticmethod
    def forward(ctx, ctx_t, x, meta, *p):
        b16 = x.dtype == BF16
        es = 2 if b16 else 4
        training = meta["training"]
        ctx_t, x = ops._contig(ctx_t), ops._contig(x)
        c = _ffn_sizes(meta["M"], meta["H"], meta["I"], es, training)
        buf = c.alloc(x.device)
        y = torch.empty_like(x)
        a = N.LayerArgs()
        a.dtype, a.training = (1 if b16 else 0), int(training)
        _fill_ffn_fwd(a.s1, c, ctx_t, x, y, p, meta, b16, training)
        _call(N.lib().vb_layer_fwd, a, "vb_layer_fwd (output + FFN block)")
        if training:
            ctx.save_for_backward(ctx_t, x, buf, *p)
            ctx.c, ctx.meta = c, meta
        return y

    @staticmethod
    def backward(ctx, dy):
        ctx_t, x, buf = ctx.saved_tensors[:3]
        p = ctx.saved_tensors[3:]
        meta, c = ctx.meta, ctx.c
        b16 = x.dtype == BF16
        es = 2 if b16 else 4
        dy = ops._contig(dy)
        dev = x.device
        N.ensure_deterministic(dev)
        tg = _Targets(list(p), dev)
        side, ws_stream = _side_stream_for(dev, tg)
        t = _ffn_bwd_sizes(meta["M"], meta["Hc"], meta["H"], meta["I"], es, meta["p_o"], meta["p_f"])
        tbuf = t.alloc(dev)
        d_x = torch.empty_like(x)                      # = d_sum1: the gradient arriving over the skip connection
        d_ctx = torch.empty_like(ctx_t)
        a = N.LayerArgs()
        a.dtype, a.training, a.wgrad_stream = (1 if b16 else 0), 1, side
        c.base = buf.data_ptr()
        y_unused = buf                                 # (the forward's y is not needed by backward; any valid pointer)
        _fill_ffn_fwd(a.s1, c, ctx_t, x, y_unused, p, meta, b16, True)
        _fill_ffn_bwd(a.s1, t, dy, d_x, d_ctx, tg, 0, _ln_ws(dev, meta["M"], meta["H"], b16), meta)
        _call(N.lib().vb_layer_bwd, a, "vb_layer_bwd (output + FFN block)")
        if ws_stream is not None:
            for tt in (buf, tbuf, ctx_t, d_x):
                tt.record_stream(ws_stream)
            if meta["p_o"] == 0.0:
                A.hold_for_side_stream(ws_stream, d_x)      # (d_sum1 is the o-projection's weight-gradient operand)
        grads = tg.finish()
        return (d_ctx, d_x, None) + tuple(grads)


# ---------------------------------------------------------------------------------------------------------------------
# attention block. self: params q.w q.b k.w k.b v.w v.b; co-attention: the same six for stream 1, then for stream 2
# ---------------------------------------------------------------------------------------------------------------------
def _fill_attn_fwd(b, meta, x1, x2, mask1, mask2, qkv1, qkv2, ctx1, ctx2, lse1, lse2, p, b16):
    b.batch, b.heads, b.head_dim, b.n1, b.n2 = meta["B"], meta["heads"], meta["d"], meta["n1"], meta["n2"]
    b.x1 = x1.data_ptr()
    b.mask1 = mask1.data_ptr() if mask1 is not None else None
    _fill_linear(b.qkv1, [p[0], p[2], p[4]], [p[1], p[3], p[5]], b16)
    b.p1, b.seed1 = meta["p1"], meta["seed1"]
    b.qkv1_out, b.ctx1 = qkv1.data_ptr(), ctx1.data_ptr()
    b.lse1 = lse1.data_ptr() if lse1 is not None else None
    if meta["n2"]:
        b.x2 = x2.data_ptr()
        b.mask2 = mask2.data_ptr() if mask2 is not None else None
        _fill_linear(b.qkv2, [p[6], p[8], p[10]], [p[7], p[9], p[11]], b16)
        b.p2, b.seed2 = meta["p2"], meta["seed2"]
        b.qkv2_out, b.ctx2 = qkv2.data_ptr(), ctx2.data_ptr()
        b.lse2 = lse2.data_ptr() if lse2 is not None else None


def _mask2d(mask, B, S):
    """additive mask [B,1,1,S] (or any shape with B*S values) -> contiguous fp32 [B, S]; None stays None."""
    if mask is None:
        return None
    m = ops._contig(mask)
    if m.numel() != B * S:
        raise RuntimeError("attention: mask must hold %d x %d values" % (B, S))
    N.dev_f32(m, "attention mask")          # (fp32 on a HIP device, or raise)
    return m


class SelfLayerFn(Function):
    """A whole BertLayer / BertImageLayer: q|k|v projection, attention, output projection + LayerNorm, feed-forward +
    LayerNorm. params: q.w q.b k.w k.b v.w v.b, then the ten of the output + FFN block."""

    @staticmethod
    def forward(ctx, x, mask, meta, *p):
        b16 = x.dtype == BF16
        es = 2 if b16 else 4
        training = meta["training"]
        x = ops._contig(x)
        B, S, H, I, heads = meta["B"], meta["n1"], meta["H"], meta["I"], meta["heads"]
        M = B * S
        mask = _mask2d(mask, B, S)
        c = _ffn_sizes(M, H, I, es, training)
        c.add("qkv", M * 3 * H * es)
        c.add("ctx", M * H * es)
        if training:
            c.add("lse", B * heads * S * 4)
        buf = c.alloc(x.device)
        y = torch.empty_like(x)
        a = N.LayerArgs()
        a.dtype, a.training = (1 if b16 else 0), int(training)

        qkv, cx = _P(c.ptr("qkv")), _P(c.ptr("ctx"))
        lse = _P(c.ptr("lse")) if training else None
        _fill_attn_fwd(a.attn, meta, x, None, mask, None, qkv, None, cx, None, lse, None, p, b16)
        _fill_ffn_fwd(a.s1, c, cx, x, y, p[6:], meta, b16, training)
        _call(N.lib().vb_layer_fwd, a, "vb_layer_fwd")
        if training:
            ctx.save_for_backward(x, mask, buf, *p)
            ctx.c, ctx.meta = c, meta
        return y

    @staticmethod
    def backward(ctx, dy):
        x, mask, buf = ctx.saved_tensors[:3]
        p = ctx.saved_tensors[3:]
        meta, c = ctx.meta, ctx.c
        b16 = x.dtype == BF16
        es = 2 if b16 else 4
        dy = ops._contig(dy)
        dev = x.device
        B, S, H, I, heads = meta["B"], meta["n1"], meta["H"], meta["I"], meta["heads"]
        M = B * S
        N.ensure_deterministic(dev)
        tg = _Targets(list(p), dev)
        side, ws_stream = _side_stream_for(dev, tg)
        t = _ffn_bwd_sizes(M, H, H, I, es, meta["p_o"], meta["p_f"])
        t.add("d_sum1", M * H * es)
        t.add("d_ctx", M * H * es)
        t.add("dqkv", M * 3 * H * es)
        t.add("dvec", B * heads * S * 4)
        tbuf = t.alloc(dev)
        dx = torch.empty_like(x)
        a = N.LayerArgs()
        a.dtype, a.training, a.wgrad_stream = (1 if b16 else 0), 1, side
        c.base = buf.data_ptr()

        qkv, cx, lse = _P(c.ptr("qkv")), _P(c.ptr("ctx")), _P(c.ptr("lse"))
        d_sum1, d_ctx = _P(t.ptr("d_sum1")), _P(t.ptr("d_ctx"))
        _fill_attn_fwd(a.attn, meta, x, None, mask, None, qkv, None, cx, None, lse, None, p, b16)
        _fill_ffn_fwd(a.s1, c, cx, x, _P(c.ptr("sum1")), p[6:], meta, b16, True)
        _fill_ffn_bwd(a.s1, t, dy, d_sum1, d_ctx, tg, 6, _ln_ws(dev, M, H, b16), meta)
        b = a.attn
        b.d_ctx1, b.dqkv1, b.dvec = d_ctx.data_ptr(), t.ptr("dqkv"), t.ptr("dvec")
        b.dres1, b.dx1 = d_sum1.data_ptr(), dx.data_ptr()
        _set_linear_targets(b.qkv1, tg, [0, 2, 4], [1, 3, 5])
        _call(N.lib().vb_layer_bwd, a, "vb_layer_bwd")
        if ws_stream is not None:
            for tt in (buf, tbuf, x):
                tt.record_stream(ws_stream)
        grads = tg.finish()
        return (dx, None, None) + tuple(grads)


class BiAttnBlockFn(Function):
    """Co-attention block of a connection layer: both fused projections and both attention directions.
    Returns (ctx1 [B, n2, Hb] for the TEXT stream, ctx2 [B, n1, Hb] for the IMAGE stream)."""

    @staticmethod
    def forward(ctx, x1, x2, mask1, mask2, meta, *p):
        b16 = x1.dtype == BF16
        es = 2 if b16 else 4
        training = meta["training"]
        x1, x2 = ops._contig(x1), ops._contig(x2)
        B, n1, n2, heads, d = meta["B"], meta["n1"], meta["n2"], meta["heads"], meta["d"]
        Hb = heads * d
        mask1, mask2 = _mask2d(mask1, B, n1), _mask2d(mask2, B, n2)
        c = _Carver()
        c.add("qkv1", B * n1 * 3 * Hb * es)
        c.add("qkv2", B * n2 * 3 * Hb * es)
        if training:
            c.add("lse1", B * heads * n2 * 4)
            c.add("lse2", B * heads * n1 * 4)
        buf = c.alloc(x1.device)
        ctx1 = torch.empty((B, n2, Hb), dtype=x1.dtype, device=x1.device)
        ctx2 = torch.empty((B, n1, Hb), dtype=x1.dtype, device=x1.device)
        a = N.LayerArgs()
        a.dtype, a.training = (1 if b16 else 0), int(training)

        _fill_attn_fwd(a.attn, meta, x1, x2, mask1, mask2, _P(c.ptr("qkv1")), _P(c.ptr("qkv2")), ctx1, ctx2,
                       _P(c.ptr("lse1")) if training else None, _P(c.ptr("lse2")) if training else None, p, b16)
        _call(N.lib().vb_layer_fwd, a, "vb_layer_fwd (co-attention block)")
        if training:
            ctx.save_for_backward(x1, x2, mask1, mask2, buf, *p)
            ctx.c, ctx.meta = c, meta
            ctx.set_materialize_grads(False)
        return ctx1, ctx2

    @staticmethod
    def backward(ctx, d1, d2):
        x1, x2, mask1, mask2, buf = ctx.saved_tensors[:5]
        p = ctx.saved_tensors[5:]
        meta, c = ctx.meta, ctx.c
        if d1 is None and d2 is None:
            return (None,) * (5 + len(p))
        b16 = x1.dtype == BF16
        es = 2 if b16 else 4
        dev = x1.device
        B, n1, n2, heads, d = meta["B"], meta["n1"], meta["n2"], meta["heads"], meta["d"]
        Hb = heads * d
        d1 = ops._contig(d1) if d1 is not None else torch.zeros((B, n2, Hb), dtype=x1.dtype, device=dev)
        d2 = ops._contig(d2) if d2 is not None else torch.zeros((B, n1, Hb), dtype=x1.dtype, device=dev)
        N.ensure_deterministic(dev)
        tg = _Targets(list(p), dev)
        side, ws_stream = _side_stream_for(dev, tg)
        t = _Carver()
        t.add("dqkv1", B * n1 * 3 * Hb * es)
        t.add("dqkv2", B * n2 * 3 * Hb * es)
        t.add("dvec", B * heads * max(n1, n2) * 4)
        tbuf = t.alloc(dev)
        dx1, dx2 = torch.empty_like(x1), torch.empty_like(x2)
        a = N.LayerArgs()
        a.dtype, a.training, a.wgrad_stream = (1 if b16 else 0), 1, side
        c.base = buf.data_ptr()

        _fill_attn_fwd(a.attn, meta, x1, x2, mask1, mask2, _P(c.ptr("qkv1")), _P(c.ptr("qkv2")), d1, d2,
                       _P(c.ptr("lse1")), _P(c.ptr("lse2")), p, b16)      # (ctx1 / ctx2 are not read by backward)
        b = a.attn
        b.d_ctx1, b.d_ctx2 = d1.data_ptr(), d2.data_ptr()
        b.dqkv1, b.dqkv2, b.dvec = t.ptr("dqkv1"), t.ptr("dqkv2"), t.ptr("dvec")
        b.dx1, b.dx2 = dx1.data_ptr(), dx2.data_ptr()
        _set_linear_targets(b.qkv1, tg, [0, 2, 4], [1, 3, 5])
        _set_linear_targets(b.qkv2, tg, [6, 8, 10], [7, 9, 11])
        _call(N.lib().vb_layer_bwd, a, "vb_layer_bwd (co-attention block)")
        if ws_stream is not None:
            for tt in (buf, tbuf, x1, x2):
                tt.record_stream(ws_stream)
        grads = tg.finish()
        return (dx1, dx2, None, None, None) + tuple(grads)


# ---------------------------------------------------------------------------------------------------------------------
# dispatch from the modules (vilbert.py)
# ---------------------------------------------------------------------------------------------------------------------
def _grad_state(x_list, params):
    """-> "train" (grad mode, every parameter trainable), "infer" (nothing to record), or None (mixed: per-op path)."""
    if not torch.is_grad_enabled():
        return "infer"
    req = [p.requires_grad for p in params]
    if all(req):
        return "train"
    if not any(req) and not any(t.requires_grad for t in x_list):
        return "infer"
    return None


def _dtype_ok(x):
    if not x.is_cuda:
        return False
    if x.dtype == BF16:
        return N.bf16_stream()
    return x.dtype == torch.float32 and not N.fp8_enabled()


def _block_params(dense, ln1, inter, out):
    return [dense.weight, dense.bias, ln1.weight, ln1.bias, inter.dense.weight, inter.dense.bias, out.dense.weight,
            out.dense.bias, out.LayerNorm.weight, out.LayerNorm.bias]


def _block_ok(params, x, inter_act):
    b16 = x.dtype == BF16
    if inter_act != "gelu" or any(p is None for p in params):
        return False
    if not (_linear_ok([params[0]], [params[1]], b16) and _linear_ok([params[4]], [params[5]], b16)
            and _linear_ok([params[6]], [params[7]], b16)):
        return False
    H = params[2].shape[0]
    return not b16 or (H <= 1024 and H % 4 == 0)


def self_layer(layer, x, mask, drop_attn, drop_o, drop_f):
    """BertLayer / BertImageLayer forward on the native launcher, or None when the per-op path has to serve this call.
    drop_*: the effective dropout probabilities of the three nn.Dropout children."""
    if not _STATE["on"] or x.dim() != 3 or not _dtype_ok(x):
        return None
    att = layer.attention.self
    if att.visualization or getattr(att, "dynamic_attention", False):
        return None
    qkv = [att.query.weight, att.query.bias, att.key.weight, att.key.bias, att.value.weight, att.value.bias]
    blk = _block_params(layer.attention.output.dense, layer.attention.output.LayerNorm, layer.intermediate, layer.output)
    params = qkv + blk
    if any(p is None for p in params):
        return None
    b16 = x.dtype == BF16
    B, S, H = x.shape
    if S > ops.MAX_KEYS or att.attention_head_size not in (32, 64, 128) or att.all_head_size != H:
        return None
    if not _linear_ok(qkv[0::2], qkv[1::2], b16) or not _block_ok(blk, x, layer.intermediate.intermediate_act_fn):
        return None
    state = _grad_state([x], params)
    if state is None:
        return None
    training = state == "train"
    meta = dict(B=B, n1=S, n2=0, H=H, Hc=H, I=layer.intermediate.dense.weight.shape[0], M=B * S, heads=att.num_attention_heads,
                d=att.attention_head_size, eps=layer.output.LayerNorm.variance_epsilon, training=training,
                p1=drop_attn, p_o=drop_o, p_f=drop_f)
    # seeds in the order the per-op path draws them: attention, output projection, feed-forward
    meta["seed1"] = A.next_seed() if drop_attn > 0.0 else 0
    meta["seed_o"] = A.next_seed() if drop_o > 0.0 else 0
    meta["seed_f"] = A.next_seed() if drop_f > 0.0 else 0
    if training:
        return SelfLayerFn.apply(x, mask, meta, *params)
    with torch.no_grad():
        return SelfLayerFn.forward(_NoCtx, x, mask, meta, *params)


class _NoCtx(object):
    """ctx stand-in for a direct (inference) call of a Function's forward: nothing is saved."""
    @staticmethod
    def save_for_backward(*a):
        pass


def connection_layer(layer, x1, mask1, x2, mask2, drops, concurrent):
    """BertConnectionLayer forward: co-attention block + the two output / feed-forward blocks (`concurrent(side_fn, main_fn,
    side_inputs)` = vilbert._concurrent: image block on the side stream). drops = (p1, p2, p_o1, p_f1, p_o2, p_f2).
    -> (y1, y2) or None (per-op path)."""
    if not _STATE["on"] or x1.dim() != 3 or x2.dim() != 3 or not (_dtype_ok(x1) and x2.dtype == x1.dtype and x2.is_cuda):
        return None
    bi, bo = layer.biattention, layer.biOutput
    if bi.visualization or x1.shape[0] != x2.shape[0]:
        return None
    qkv = [bi.query1.weight, bi.query1.bias, bi.key1.weight, bi.key1.bias, bi.value1.weight, bi.value1.bias,
           bi.query2.weight, bi.query2.bias, bi.key2.weight, bi.key2.bias, bi.value2.weight, bi.value2.bias]
    blk1 = _block_params(bo.dense1, bo.LayerNorm1, layer.v_intermediate, layer.v_output)
    blk2 = _block_params(bo.dense2, bo.LayerNorm2, layer.t_intermediate, layer.t_output)
    if any(p is None for p in qkv + blk1 + blk2):
        return None
    b16 = x1.dtype == BF16
    B, n1, H1 = x1.shape
    _, n2, H2 = x2.shape
    Hb = bi.all_head_size
    if max(n1, n2) > ops.MAX_KEYS or bi.attention_head_size not in (32, 64, 128):
        return None
    if not (_linear_ok(qkv[0:6:2], qkv[1:6:2], b16) and _linear_ok(qkv[6::2], qkv[7::2], b16)
            and _block_ok(blk1, x1, layer.v_intermediate.intermediate_act_fn)
            and _block_ok(blk2, x2, layer.t_intermediate.intermediate_act_fn)):
        return None
    state = _grad_state([x1, x2], qkv + blk1 + blk2)
    if state is None:
        return None
    training = state == "train"
    p1, p2, p_o1, p_f1, p_o2, p_f2 = drops
    am = dict(B=B, n1=n1, n2=n2, heads=bi.num_attention_heads, d=bi.attention_head_size, training=training, p1=p1, p2=p2)
    # seeds in the per-op order: attention 1, attention 2, image block (output, FFN), text block (output, FFN)
    am["seed1"] = A.next_seed() if p1 > 0.0 else 0
    am["seed2"] = A.next_seed() if p2 > 0.0 else 0
    m1 = dict(M=B * n1, Hc=Hb, H=H1, I=layer.v_intermediate.dense.weight.shape[0], eps=layer.v_output.LayerNorm.variance_epsilon,
              training=training, p_o=p_o1, p_f=p_f1)
    m1["seed_o"] = A.next_seed() if p_o1 > 0.0 else 0
    m1["seed_f"] = A.next_seed() if p_f1 > 0.0 else 0
    m2 = dict(M=B * n2, Hc=Hb, H=H2, I=layer.t_intermediate.dense.weight.shape[0], eps=layer.t_output.LayerNorm.variance_epsilon,
              training=training, p_o=p_o2, p_f=p_f2)
    m2["seed_o"] = A.next_seed() if p_o2 > 0.0 else 0
    m2["seed_f"] = A.next_seed() if p_f2 > 0.0 else 0
    if training:
        ctx1, ctx2 = BiAttnBlockFn.apply(x1, x2, mask1, mask2, am, *qkv)
        image = lambda: FfnBlockFn.apply(ctx2, x1, m1, *blk1)
        text = lambda: FfnBlockFn.apply(ctx1, x2, m2, *blk2)
    else:
        with torch.no_grad():
            ctx1, ctx2 = BiAttnBlockFn.forward(_NoCtx, x1, x2, mask1, mask2, am, *qkv)

        def image():
            with torch.no_grad():
                return FfnBlockFn.forward(_NoCtx, ctx2, x1, m1, *blk1)

        def text():
            with torch.no_grad():
                return FfnBlockFn.forward(_NoCtx, ctx1, x2, m2, *blk2)
    return concurrent(image, text, [ctx2, x1])
