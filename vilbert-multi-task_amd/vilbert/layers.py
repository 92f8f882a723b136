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

Plans: what does not change from call to call - the module's parameters in launcher order, its eligibility, the layout of the
two buffers, and the argument struct with every STATIC field filled (dimensions, fp32 weight / bias / LayerNorm pointers) - is
kept per (module, dtype, shape, mode) and revalidated per call: the module tree below the layer and every parameter OBJECT by
identity (plain dict lookups), every parameter address per plan; a call copies the
struct and fills the per-call pointers, probabilities and seeds. (The bf16 weight shadows are still asked for on every call:
that is where a stale shadow gets refreshed.)

`VB_LAYER_NATIVE=0` (or `set_native(False)`) keeps the per-op path; the modes that path serves alone: fp8 / MX inference,
attention maps (`visualization`), dynamic attention, activations other than GELU, shapes the bf16 kernels do not take,
partially frozen layers.
"""
import ctypes
import os
import weakref

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
    """Layout of the sub-buffers of one flat allocation: add() collects (name, bytes); at(base, name) = the address inside an
    allocation that starts at `base`. Immutable once built (shared by every call of a plan)."""

    def __init__(self):
        self.off, self.total = {}, 0

    def add(self, name, nbytes):
        self.off[name] = self.total
        self.total += _align(int(nbytes))

    def alloc(self, device):
        return torch.empty(max(self.total, 256), dtype=torch.uint8, device=device)

    def at(self, base, name):
        return base + self.off[name]


def _static_linear(L, weights, biases):
    seg_n, K = weights[0].shape
    L.nseg, L.seg_n, L.K = len(weights), seg_n, K
    for s, (w, b) in enumerate(zip(weights, biases)):
        L.w[s] = w.data_ptr()
        L.bias[s] = b.data_ptr()


def _shadow_linear(L, weights):
    """bf16 path: the (possibly just refreshed) shadows of the stacked weights."""
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
        self.claims = _arena.claim_many(params)                    # (view, mode, arena, index)
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
        self._tmp = None

    def ptr(self, i):
        return self.views[i].data_ptr()

    def overwrite_target(self, i):
        """LayerNorm column sums are OVERWRITTEN by their kernel: a fresh (zeroed, first-writer) slice or private scratch takes
        them in place, an accumulating slice gets them through a temporary (finish())."""
        if self.claims[i][1] == "accum":
            t = torch.empty_like(self.views[i])
            if self._tmp is None:
                self._tmp = []
            self._tmp.append((i, t))
            return t.data_ptr()
        return self.views[i].data_ptr()

    def finish(self):
        if self._tmp is not None:
            for i, t in self._tmp:
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


def _ln_ws(device, rows, cols, b16):
    floats = (rows + 15) // 16 * 2 * cols if b16 else N.lib().vb_layernorm_bwd_workspace(rows, cols)
    return ops16._ln_workspace(device, floats)


def _call(fn, a, what):
    _STATE["calls"] += 1
    N.check(fn(N.stream_ptr(), ctypes.byref(a)), what)


# ---------------------------------------------------------------------------------------------------------------------
# Plans: the static half of a node
# ---------------------------------------------------------------------------------------------------------------------
class _Plan(object):
    __slots__ = ("params", "ptrs", "b16", "training", "tmpl", "c_fwd", "c_bwd", "dims")

    def new_args(self):
        return N.LayerArgs.from_buffer_copy(self.tmpl)

    def valid(self):
        ptrs = self.ptrs
        for i, p in enumerate(self.params):
            if p.data_ptr() != ptrs[i]:
                return False
        return True


def _seal(plan, args):
    plan.ptrs = tuple(p.data_ptr() for p in plan.params)
    plan.tmpl = bytes(args)


# ---------------------------------------------------------------------------------------------------------------------
# output projection + feed-forward block:  y = LN2(dropout(W2 gelu(W1 a + b1) + b2) + a),  a = LN1(dropout(Wo ctx + bo) + x)
# params: o.w, o.b, ln1.g, ln1.b, f1.w, f1.b, f2.w, f2.b, ln2.g, ln2.b
# dims: (M, Hc, H, I, eps)
# ---------------------------------------------------------------------------------------------------------------------
def _ffn_sizes(c, M, H, I, es, training):
    c.add("sum1", M * H * es)
    c.add("a1", M * H * es)
    c.add("h", M * I * es)
    c.add("sum2", M * H * es)
    if training:
        c.add("dact", M * I * es)
        for n in ("mean1", "rstd1", "mean2", "rstd2"):
            c.add(n, M * 4)


def _ffn_bwd_sizes(c, M, H, I, es):
    # (both dropout twins are laid out: which of them a call uses depends on its probabilities)
    c.add("d_sum2", M * H * es)
    c.add("d_sum2_drop", M * H * es)
    c.add("d_pre", M * I * es)
    c.add("d_a1", M * H * es)
    c.add("d_sum1_drop", M * H * es)


def _static_ffn(f, p, dims):
    M, Hc, H, I, eps = dims
    f.M, f.Hc, f.H, f.I, f.eps = M, Hc, H, I, eps
    _static_linear(f.o, [p[0]], [p[1]])
    f.ln1.gamma, f.ln1.beta = p[2].data_ptr(), p[3].data_ptr()
    _static_linear(f.f1, [p[4]], [p[5]])
    _static_linear(f.f2, [p[6]], [p[7]])
    f.ln2.gamma, f.ln2.beta = p[8].data_ptr(), p[9].data_ptr()


def _dyn_ffn_fwd(f, c, base, ctx_ptr, x_ptr, y_ptr, p, dyn, b16, training):
    """dyn = (p_o, p_f, seed_o, seed_f)"""
    f.ctx, f.x, f.y = ctx_ptr, x_ptr, y_ptr
    f.p_o, f.p_f, f.seed_o, f.seed_f = dyn
    if b16:
        _shadow_linear(f.o, [p[0]])
        _shadow_linear(f.f1, [p[4]])
        _shadow_linear(f.f2, [p[6]])
    at = c.at
    f.sum1, f.a1, f.h, f.sum2 = at(base, "sum1"), at(base, "a1"), at(base, "h"), at(base, "sum2")
    if training:
        f.dact = at(base, "dact")
        f.mean1, f.rstd1, f.mean2, f.rstd2 = at(base, "mean1"), at(base, "rstd1"), at(base, "mean2"), at(base, "rstd2")


def _dyn_ffn_bwd(f, t, tbase, dy_ptr, d_sum1_ptr, d_ctx_ptr, tg, pbase, ws, dyn):
    """t: layout of the backward temporaries; tg: targets, the block's 10 parameters start at index `pbase`."""
    p_o, p_f = dyn[0], dyn[1]
    at = t.at
    f.dy = dy_ptr
    f.d_sum2, f.d_pre, f.d_a1 = at(tbase, "d_sum2"), at(tbase, "d_pre"), at(tbase, "d_a1")
    f.d_sum2_drop = at(tbase, "d_sum2_drop") if p_f > 0.0 else None
    f.d_sum1_drop = at(tbase, "d_sum1_drop") if p_o > 0.0 else None
    f.d_sum1, f.d_ctx = d_sum1_ptr, d_ctx_ptr
    f.ln_ws = ws.data_ptr()
    _set_linear_targets(f.o, tg, [pbase + 0], [pbase + 1])
    f.ln1.dgamma, f.ln1.dbeta = tg.overwrite_target(pbase + 2), tg.overwrite_target(pbase + 3)
    _set_linear_targets(f.f1, tg, [pbase + 4], [pbase + 5])
    _set_linear_targets(f.f2, tg, [pbase + 6], [pbase + 7])
    f.ln2.dgamma, f.ln2.dbeta = tg.overwrite_target(pbase + 8), tg.overwrite_target(pbase + 9)


def _ffn_plan(p, dims, b16, training):
    plan = _Plan()
    plan.params, plan.b16, plan.training, plan.dims = list(p), b16, training, dims
    M, _Hc, H, I, _eps = dims
    es = 2 if b16 else 4
    plan.c_fwd = _Carver()
    _ffn_sizes(plan.c_fwd, M, H, I, es, training)
    plan.c_bwd = _Carver()
    _ffn_bwd_sizes(plan.c_bwd, M, H, I, es)
    a = N.LayerArgs()
    a.dtype, a.training = (1 if b16 else 0), int(training)
    _static_ffn(a.s1, p, dims)
    _seal(plan, a)
    return plan


class FfnBlockFn(Function):
    """y = output + feed-forward block of one stream (connection layers: one per stream, on two HIP streams)."""

    @staticmethod
    def forward(ctx, ctx_t, x, plan, dyn, *p):
        b16, training, c = plan.b16, plan.training, plan.c_fwd
        ctx_t, x = ops._contig(ctx_t), ops._contig(x)
        buf = c.alloc(x.device)
        y = torch.empty_like(x)
        a = plan.new_args()
        _dyn_ffn_fwd(a.s1, c, buf.data_ptr(), ctx_t.data_ptr(), x.data_ptr(), y.data_ptr(), p, dyn, b16, training)
        _call(N.lib().vb_layer_fwd, a, "vb_layer_fwd (output + FFN block)")
        if training:
            ctx.save_for_backward(ctx_t, x, buf, *p)
            ctx.plan, ctx.dyn = plan, dyn
        return y

    @staticmethod
    def backward(ctx, dy):
        ctx_t, x, buf = ctx.saved_tensors[:3]
        p = ctx.saved_tensors[3:]
        plan, dyn = ctx.plan, ctx.dyn
        b16, c, t = plan.b16, plan.c_fwd, plan.c_bwd
        M, _Hc, H, _I, _eps = plan.dims
        dy = ops._contig(dy)
        dev = x.device
        N.ensure_deterministic(dev)
        tg = _Targets(list(p), dev)
        side, ws_stream = _side_stream_for(dev, tg)
        tbuf = t.alloc(dev)
        d_x = torch.empty_like(x)                      # = d_sum1: the gradient arriving over the skip connection
        d_ctx = torch.empty_like(ctx_t)
        a = plan.new_args()
        a.training, a.wgrad_stream = 1, side
        # (the forward's y is not needed by backward; any valid pointer)
        _dyn_ffn_fwd(a.s1, c, buf.data_ptr(), ctx_t.data_ptr(), x.data_ptr(), buf.data_ptr(), p, dyn, b16, True)
        _dyn_ffn_bwd(a.s1, t, tbuf.data_ptr(), dy.data_ptr(), d_x.data_ptr(), d_ctx.data_ptr(), tg, 0, _ln_ws(dev, M, H, b16), dyn)
        _call(N.lib().vb_layer_bwd, a, "vb_layer_bwd (output + FFN block)")
        if ws_stream is not None:
            for tt in (buf, tbuf, ctx_t, d_x):
                tt.record_stream(ws_stream)
            if dyn[0] == 0.0:
                A.hold_for_side_stream(ws_stream, d_x)      # (d_sum1 is the o-projection's weight-gradient operand)
        grads = tg.finish()
        return (d_ctx, d_x, None, None) + tuple(grads)


# ---------------------------------------------------------------------------------------------------------------------
# attention block. self: params q.w q.b k.w k.b v.w v.b; co-attention: the same six for stream 1, then for stream 2
# adims: (B, heads, d, n1, n2)
# ---------------------------------------------------------------------------------------------------------------------
def _static_attn(b, adims, p):
    b.batch, b.heads, b.head_dim, b.n1, b.n2 = adims
    _static_linear(b.qkv1, [p[0], p[2], p[4]], [p[1], p[3], p[5]])
    if adims[4]:
        _static_linear(b.qkv2, [p[6], p[8], p[10]], [p[7], p[9], p[11]])


def _dyn_attn_fwd(b, n2, x1, x2, mask1, mask2, qkv1, qkv2, ctx1, ctx2, lse1, lse2, p, adyn, b16):
    """pointers (ints / None); adyn = (p1, p2, seed1, seed2)"""
    b.x1, b.mask1, b.qkv1_out, b.ctx1, b.lse1 = x1, mask1, qkv1, ctx1, lse1
    b.p1, b.p2, b.seed1, b.seed2 = adyn
    if b16:
        _shadow_linear(b.qkv1, [p[0], p[2], p[4]])
    if n2:
        b.x2, b.mask2, b.qkv2_out, b.ctx2, b.lse2 = x2, mask2, qkv2, ctx2, lse2
        if b16:
            _shadow_linear(b.qkv2, [p[6], p[8], p[10]])


def _mask2d(mask, B, S):
    """additive mask [B,1,1,S] (or any shape with B*S values) -> contiguous fp32 [B, S]; None stays None."""
    if mask is None:
        return None
    m = ops._contig(mask)
    if m.numel() != B * S:
        raise RuntimeError("attention: mask must hold %d x %d values" % (B, S))
    N.dev_f32(m, "attention mask")          # (fp32 on a HIP device, or raise)
    return m


def _ptr(t):
    return t.data_ptr() if t is not None else None


def _self_plan(p, adims, dims, b16, training):
    plan = _Plan()
    plan.params, plan.b16, plan.training = list(p), b16, training
    plan.dims = (adims, dims)
    B, heads, _d, S, _n2 = adims
    M, _Hc, H, I, _eps = dims
    es = 2 if b16 else 4
    c = plan.c_fwd = _Carver()
    _ffn_sizes(c, M, H, I, es, training)
    c.add("qkv", M * 3 * H * es)
    c.add("ctx", M * H * es)
    if training:
        c.add("lse", B * heads * S * 4)
    t = plan.c_bwd = _Carver()
    _ffn_bwd_sizes(t, M, H, I, es)
    t.add("d_sum1", M * H * es)
    t.add("d_ctx", M * H * es)
    t.add("dqkv", M * 3 * H * es)
    t.add("dvec", B * heads * S * 4)
    a = N.LayerArgs()
    a.dtype, a.training = (1 if b16 else 0), int(training)
    _static_attn(a.attn, adims, p)
    _static_ffn(a.s1, p[6:], dims)
    _seal(plan, a)
    return plan


class SelfLayerFn(Function):
    """A whole BertLayer / BertImageLayer: q|k|v projection, attention, output projection + LayerNorm, feed-forward +
    LayerNorm. params: q.w q.b k.w k.b v.w v.b, then the ten of the output + FFN block.
    dyn = (p_o, p_f, seed_o, seed_f, p1, seed1)"""

    @staticmethod
    def forward(ctx, x, mask, plan, dyn, *p):
        b16, training, c = plan.b16, plan.training, plan.c_fwd
        (B, _heads, _d, S, _n2), _dims = plan.dims
        x = ops._contig(x)
        mask = _mask2d(mask, B, S)
        buf = c.alloc(x.device)
        base = buf.data_ptr()
        y = torch.empty_like(x)
        a = plan.new_args()
        cx = c.at(base, "ctx")
        _dyn_attn_fwd(a.attn, 0, x.data_ptr(), None, _ptr(mask), None, c.at(base, "qkv"), None, cx, None,
                      c.at(base, "lse") if training else None, None, p, (dyn[4], 0.0, dyn[5], 0), b16)
        _dyn_ffn_fwd(a.s1, c, base, cx, x.data_ptr(), y.data_ptr(), p[6:], dyn[:4], b16, training)
        _call(N.lib().vb_layer_fwd, a, "vb_layer_fwd")
        if training:
            ctx.save_for_backward(x, mask, buf, *p)
            ctx.plan, ctx.dyn = plan, dyn
        return y

    @staticmethod
    def backward(ctx, dy):
        x, mask, buf = ctx.saved_tensors[:3]
        p = ctx.saved_tensors[3:]
        plan, dyn = ctx.plan, ctx.dyn
        b16, c, t = plan.b16, plan.c_fwd, plan.c_bwd
        _adims, (M, _Hc, H, _I, _eps) = plan.dims
        dy = ops._contig(dy)
        dev = x.device
        N.ensure_deterministic(dev)
        tg = _Targets(list(p), dev)
        side, ws_stream = _side_stream_for(dev, tg)
        tbuf = t.alloc(dev)
        base, tbase = buf.data_ptr(), tbuf.data_ptr()
        dx = torch.empty_like(x)
        a = plan.new_args()
        a.training, a.wgrad_stream = 1, side
        cx, d_sum1, d_ctx = c.at(base, "ctx"), t.at(tbase, "d_sum1"), t.at(tbase, "d_ctx")
        _dyn_attn_fwd(a.attn, 0, x.data_ptr(), None, _ptr(mask), None, c.at(base, "qkv"), None, cx, None, c.at(base, "lse"), None,
                      p, (dyn[4], 0.0, dyn[5], 0), b16)
        _dyn_ffn_fwd(a.s1, c, base, cx, x.data_ptr(), c.at(base, "sum1"), p[6:], dyn[:4], b16, True)
        _dyn_ffn_bwd(a.s1, t, tbase, dy.data_ptr(), d_sum1, d_ctx, tg, 6, _ln_ws(dev, M, H, b16), dyn)
        b = a.attn
        b.d_ctx1, b.dqkv1, b.dvec = d_ctx, t.at(tbase, "dqkv"), t.at(tbase, "dvec")
        b.dres1, b.dx1 = d_sum1, dx.data_ptr()
        _set_linear_targets(b.qkv1, tg, [0, 2, 4], [1, 3, 5])
        _call(N.lib().vb_layer_bwd, a, "vb_layer_bwd")
        if ws_stream is not None:
            for tt in (buf, tbuf, x):
                tt.record_stream(ws_stream)
        grads = tg.finish()
        return (dx, None, None, None) + tuple(grads)


def _bi_plan(p, adims, b16, training):
    plan = _Plan()
    plan.params, plan.b16, plan.training, plan.dims = list(p), b16, training, adims
    B, heads, d, n1, n2 = adims
    Hb = heads * d
    es = 2 if b16 else 4
    c = plan.c_fwd = _Carver()
    c.add("qkv1", B * n1 * 3 * Hb * es)
    c.add("qkv2", B * n2 * 3 * Hb * es)
    if training:
        c.add("lse1", B * heads * n2 * 4)
        c.add("lse2", B * heads * n1 * 4)
    t = plan.c_bwd = _Carver()
    t.add("dqkv1", B * n1 * 3 * Hb * es)
    t.add("dqkv2", B * n2 * 3 * Hb * es)
    t.add("dvec", B * heads * max(n1, n2) * 4)
    a = N.LayerArgs()
    a.dtype, a.training = (1 if b16 else 0), int(training)
    _static_attn(a.attn, adims, p)
    _seal(plan, a)
    return plan


class BiAttnBlockFn(Function):
    """Co-attention block of a connection layer: both fused projections and both attention directions.
    Returns (ctx1 [B, n2, Hb] for the TEXT stream, ctx2 [B, n1, Hb] for the IMAGE stream). adyn = (p1, p2, seed1, seed2)"""

    @staticmethod
    def forward(ctx, x1, x2, mask1, mask2, plan, adyn, *p):
        b16, training, c = plan.b16, plan.training, plan.c_fwd
        B, heads, d, n1, n2 = plan.dims
        Hb = heads * d
        x1, x2 = ops._contig(x1), ops._contig(x2)
        mask1, mask2 = _mask2d(mask1, B, n1), _mask2d(mask2, B, n2)
        buf = c.alloc(x1.device)
        base = buf.data_ptr()
        ctx1 = torch.empty((B, n2, Hb), dtype=x1.dtype, device=x1.device)
        ctx2 = torch.empty((B, n1, Hb), dtype=x1.dtype, device=x1.device)
        a = plan.new_args()
        _dyn_attn_fwd(a.attn, n2, x1.data_ptr(), x2.data_ptr(), _ptr(mask1), _ptr(mask2), c.at(base, "qkv1"), c.at(base, "qkv2"),
                      ctx1.data_ptr(), ctx2.data_ptr(), c.at(base, "lse1") if training else None,
                      c.at(base, "lse2") if training else None, p, adyn, b16)
        _call(N.lib().vb_layer_fwd, a, "vb_layer_fwd (co-attention block)")
        if training:
            ctx.save_for_backward(x1, x2, mask1, mask2, buf, *p)
            ctx.plan, ctx.adyn = plan, adyn
            ctx.set_materialize_grads(False)
        return ctx1, ctx2

    @staticmethod
    def backward(ctx, d1, d2):
        x1, x2, mask1, mask2, buf = ctx.saved_tensors[:5]
        p = ctx.saved_tensors[5:]
        plan, adyn = ctx.plan, ctx.adyn
        if d1 is None and d2 is None:
            return (None,) * (6 + len(p))
        b16, c, t = plan.b16, plan.c_fwd, plan.c_bwd
        B, heads, d, n1, n2 = plan.dims
        Hb = heads * d
        dev = x1.device
        d1 = ops._contig(d1) if d1 is not None else torch.zeros((B, n2, Hb), dtype=x1.dtype, device=dev)
        d2 = ops._contig(d2) if d2 is not None else torch.zeros((B, n1, Hb), dtype=x1.dtype, device=dev)
        N.ensure_deterministic(dev)
        tg = _Targets(list(p), dev)
        side, ws_stream = _side_stream_for(dev, tg)
        tbuf = t.alloc(dev)
        base, tbase = buf.data_ptr(), tbuf.data_ptr()
        dx1, dx2 = torch.empty_like(x1), torch.empty_like(x2)
        a = plan.new_args()
        a.training, a.wgrad_stream = 1, side
        # (ctx1 / ctx2 are not read by backward: any valid pointers)
        _dyn_attn_fwd(a.attn, n2, x1.data_ptr(), x2.data_ptr(), _ptr(mask1), _ptr(mask2), c.at(base, "qkv1"), c.at(base, "qkv2"),
                      d1.data_ptr(), d2.data_ptr(), c.at(base, "lse1"), c.at(base, "lse2"), p, adyn, b16)
        b = a.attn
        b.d_ctx1, b.d_ctx2 = d1.data_ptr(), d2.data_ptr()
        b.dqkv1, b.dqkv2, b.dvec = t.at(tbase, "dqkv1"), t.at(tbase, "dqkv2"), t.at(tbase, "dvec")
        b.dx1, b.dx2 = dx1.data_ptr(), dx2.data_ptr()
        _set_linear_targets(b.qkv1, tg, [0, 2, 4], [1, 3, 5])
        _set_linear_targets(b.qkv2, tg, [6, 8, 10], [7, 9, 11])
        _call(N.lib().vb_layer_bwd, a, "vb_layer_bwd (co-attention block)")
        if ws_stream is not None:
            for tt in (buf, tbuf, x1, x2):
                tt.record_stream(ws_stream)
        grads = tg.finish()
        return (dx1, dx2, None, None, None, None) + tuple(grads)


# ---------------------------------------------------------------------------------------------------------------------
# dispatch from the modules (vilbert.py)
# ---------------------------------------------------------------------------------------------------------------------
def _grad_state(x_list, params):
    """-> "train" (grad mode, every parameter trainable), "infer" (nothing to record), or None (mixed: per-op path)."""
    if not torch.is_grad_enabled():
        return "infer"
    n_req = 0
    for p in params:
        if p.requires_grad:
            n_req += 1
    if n_req == len(params):
        return "train"
    if n_req == 0 and not any(t.requires_grad for t in x_list):
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


def _block_ok(params, b16, inter_act):
    if inter_act != "gelu" or any(p is None for p in params):
        return False
    if not (_linear_ok([params[0]], [params[1]], b16) and _linear_ok([params[4]], [params[5]], b16)
            and _linear_ok([params[6]], [params[7]], b16)):
        return False
    H = params[2].shape[0]
    return not b16 or (H <= 1024 and H % 4 == 0)


class _NoCtx(object):
    """ctx stand-in for a direct (inference) call of a Function's forward: nothing is saved."""
    @staticmethod
    def save_for_backward(*a):
        pass


_REFUSED = object()      # cached verdict: this (module, key) goes op by op

_SELF_PATHS = ["attention.self.query", "attention.self.key", "attention.self.value", "attention.output.dense",
               "attention.output.LayerNorm", "intermediate.dense", "output.dense", "output.LayerNorm"]
_CONN_PATHS = ["biattention.query1", "biattention.key1", "biattention.value1", "biattention.query2", "biattention.key2",
               "biattention.value2", "biOutput.dense1", "biOutput.LayerNorm1", "v_intermediate.dense", "v_output.dense",
               "v_output.LayerNorm", "biOutput.dense2", "biOutput.LayerNorm2", "t_intermediate.dense", "t_output.dense",
               "t_output.LayerNorm"]


def _wiring(layer, paths):
    """The (parent, child name, child) edges from `layer` down to the leaf modules of `paths` and their (leaf, "weight" / "bias",
    parameter) slots, as found NOW: _wired() compares them with plain dict lookups on every call - a replaced sub-module
    (adapter wrappers), a replaced or re-parametrised Parameter object sends the module back through the full judgement."""
    edges, slots, seen = [], [], set()
    for path in paths:
        m = layer
        for name in path.split("."):
            child = m._modules[name]
            if (id(m), name) not in seen:
                seen.add((id(m), name))
                edges.append((m, name, child))
            m = child
        for pname in ("weight", "bias"):
            slots.append((m, pname, m._parameters.get(pname)))
    return edges, slots


def _wired(wiring):
    for m, name, child in wiring[0]:
        if m._modules.get(name) is not child:
            return False
    for m, pname, p in wiring[1]:
        if m._parameters.get(pname) is not p:
            return False
    return True


_PLANS = weakref.WeakKeyDictionary()      # module -> its plans (outside the module: deepcopy / pickle of a model never see them)


def _cache_of(layer):
    c = _PLANS.get(layer)
    if c is None:
        c = _PLANS[layer] = {}
    return c


def _self_params(layer):
    att = layer.attention.self
    qkv = [att.query.weight, att.query.bias, att.key.weight, att.key.bias, att.value.weight, att.value.bias]
    return qkv + _block_params(layer.attention.output.dense, layer.attention.output.LayerNorm, layer.intermediate, layer.output)


def self_layer(layer, x, mask, drop_attn, drop_o, drop_f):
    """BertLayer / BertImageLayer forward on the native launcher, or None when the per-op path has to serve this call.
    drop_*: the effective dropout probabilities of the three nn.Dropout children."""
    if not _STATE["on"] or x.dim() != 3 or not _dtype_ok(x):
        return None
    att = layer.attention.self
    if att.visualization or getattr(att, "dynamic_attention", False):
        return None
    b16 = x.dtype == BF16
    B, S, H = x.shape
    cache = _cache_of(layer)
    # one entry per (dtype, shape): (parameters in launcher order, {training: plan}) or the refusal
    key = (b16, B, S)
    ent = cache.get(key)
    if ent is not None and not _wired(ent[2]):
        ent = None       # (the module tree / its parameter objects are not the ones the verdict was made for; addresses: per plan)
    if ent is None:
        try:
            wiring = _wiring(layer, _SELF_PATHS)
            params = _self_params(layer)
        except (KeyError, AttributeError):
            return None                                   # (not the reference's module tree: op by op)
        ok = not any(p is None for p in params) and S <= ops.MAX_KEYS and att.attention_head_size in (32, 64, 128) \
            and att.all_head_size == H and _linear_ok(params[0:6:2], params[1:6:2], b16) \
            and _block_ok(params[6:], b16, layer.intermediate.intermediate_act_fn)
        ent = cache[key] = (params, {} if ok else _REFUSED, wiring)
    params, plans, _w = ent
    if plans is _REFUSED:
        return None
    state = _grad_state((x,), params)
    if state is None:
        return None
    training = state == "train"
    plan = plans.get(training)
    if plan is None or not plan.valid():
        if plan is not None:                 # moved parameters (model.to(), new storage): judge them again next call
            del cache[key]
            return self_layer(layer, x, mask, drop_attn, drop_o, drop_f)
        adims = (B, att.num_attention_heads, att.attention_head_size, S, 0)
        dims = (B * S, H, H, layer.intermediate.dense.weight.shape[0], layer.output.LayerNorm.variance_epsilon)
        plan = plans[training] = _self_plan(params, adims, dims, b16, training)
    # seeds in the order the per-op path draws them: attention, output projection, feed-forward
    seed1 = A.next_seed() if drop_attn > 0.0 else 0
    seed_o = A.next_seed() if drop_o > 0.0 else 0
    seed_f = A.next_seed() if drop_f > 0.0 else 0
    dyn = (drop_o, drop_f, seed_o, seed_f, drop_attn, seed1)
    if training:
        return SelfLayerFn.apply(x, mask, plan, dyn, *params)
    with torch.no_grad():
        return SelfLayerFn.forward(_NoCtx, x, mask, plan, dyn, *params)


def _conn_params(layer):
    bi, bo = layer.biattention, layer.biOutput
    qkv = [bi.query1.weight, bi.query1.bias, bi.key1.weight, bi.key1.bias, bi.value1.weight, bi.value1.bias,
           bi.query2.weight, bi.query2.bias, bi.key2.weight, bi.key2.bias, bi.value2.weight, bi.value2.bias]
    blk1 = _block_params(bo.dense1, bo.LayerNorm1, layer.v_intermediate, layer.v_output)
    blk2 = _block_params(bo.dense2, bo.LayerNorm2, layer.t_intermediate, layer.t_output)
    return qkv, blk1, blk2


def connection_layer(layer, x1, mask1, x2, mask2, drops, concurrent):
    """BertConnectionLayer forward: co-attention block + the two output / feed-forward blocks (`concurrent(side_fn, main_fn,
    side_inputs)` = vilbert._concurrent: image block on the side stream). drops = (p1, p2, p_o1, p_f1, p_o2, p_f2).
    -> (y1, y2) or None (per-op path)."""
    if not _STATE["on"] or x1.dim() != 3 or x2.dim() != 3 or not (_dtype_ok(x1) and x2.dtype == x1.dtype and x2.is_cuda):
        return None
    bi = layer.biattention
    if bi.visualization or x1.shape[0] != x2.shape[0]:
        return None
    b16 = x1.dtype == BF16
    B, n1, H1 = x1.shape
    _, n2, H2 = x2.shape
    cache = _cache_of(layer)
    key = (b16, B, n1, n2)
    ent = cache.get(key)
    if ent is not None and not _wired(ent[5]):
        ent = None
    if ent is None:
        try:
            wiring = _wiring(layer, _CONN_PATHS)
            qkv, blk1, blk2 = _conn_params(layer)
        except (KeyError, AttributeError):
            return None
        ok = not any(p is None for p in qkv + blk1 + blk2) and max(n1, n2) <= ops.MAX_KEYS \
            and bi.attention_head_size in (32, 64, 128) \
            and _linear_ok(qkv[0:6:2], qkv[1:6:2], b16) and _linear_ok(qkv[6::2], qkv[7::2], b16) \
            and _block_ok(blk1, b16, layer.v_intermediate.intermediate_act_fn) \
            and _block_ok(blk2, b16, layer.t_intermediate.intermediate_act_fn)
        ent = cache[key] = (qkv, blk1, blk2, qkv + blk1 + blk2, {} if ok else _REFUSED, wiring)
    qkv, blk1, blk2, allp, plans, _w = ent
    if plans is _REFUSED:
        return None
    state = _grad_state((x1, x2), allp)
    if state is None:
        return None
    training = state == "train"
    trio = plans.get(training)
    if trio is None or not (trio[0].valid() and trio[1].valid() and trio[2].valid()):
        if trio is not None:                 # moved parameters: judge them again
            del cache[key]
            return connection_layer(layer, x1, mask1, x2, mask2, drops, concurrent)
        Hb = bi.all_head_size
        adims = (B, bi.num_attention_heads, bi.attention_head_size, n1, n2)
        d1 = (B * n1, Hb, H1, layer.v_intermediate.dense.weight.shape[0], layer.v_output.LayerNorm.variance_epsilon)
        d2 = (B * n2, Hb, H2, layer.t_intermediate.dense.weight.shape[0], layer.t_output.LayerNorm.variance_epsilon)
        trio = plans[training] = (_bi_plan(qkv, adims, b16, training), _ffn_plan(blk1, d1, b16, training),
                                  _ffn_plan(blk2, d2, b16, training))
    pa, pf1, pf2 = trio
    p1, p2, p_o1, p_f1, p_o2, p_f2 = drops
    # seeds in the per-op order: attention 1, attention 2, image block (output, FFN), text block (output, FFN)
    seed1 = A.next_seed() if p1 > 0.0 else 0
    seed2 = A.next_seed() if p2 > 0.0 else 0
    adyn = (p1, p2, seed1, seed2)
    so1 = A.next_seed() if p_o1 > 0.0 else 0
    sf1 = A.next_seed() if p_f1 > 0.0 else 0
    m1 = (p_o1, p_f1, so1, sf1)
    so2 = A.next_seed() if p_o2 > 0.0 else 0
    sf2 = A.next_seed() if p_f2 > 0.0 else 0
    m2 = (p_o2, p_f2, so2, sf2)
    if training:
        ctx1, ctx2 = BiAttnBlockFn.apply(x1, x2, mask1, mask2, pa, adyn, *qkv)
        image = lambda: FfnBlockFn.apply(ctx2, x1, pf1, m1, *blk1)
        text = lambda: FfnBlockFn.apply(ctx1, x2, pf2, m2, *blk2)
    else:
        with torch.no_grad():
            ctx1, ctx2 = BiAttnBlockFn.forward(_NoCtx, x1, x2, mask1, mask2, pa, adyn, *qkv)

        def image():
            with torch.no_grad():
                return FfnBlockFn.forward(_NoCtx, ctx2, x1, pf1, m1, *blk1)

        def text():
            with torch.no_grad():
                return FfnBlockFn.forward(_NoCtx, ctx1, x2, pf2, m2, *blk2)
    return concurrent(image, text, [ctx2, x1])
