"""Tensor-level launchers of the bf16 TRAINING path (csrc/gemm_bf16.hip, rowops16.hip; include/vilbert_hip.h "bf16 TRAINING
path"): the reduced-precision mode that replaces the reference's `model.half()` + apex FP16_Optimizer
(/root/reference/train_concap.py:443-461,504-505). Activations, saved tensors and activation gradients are torch.bfloat16
tensors; parameters, their gradients (the arena), LayerNorm statistics and the optimizer stay fp32. No torch arithmetic.

Weights: every launch reads a bf16 SHADOW of the (stacked) fp32 weight - row-major [N, K] for the forward, transposed
[K, N] for the input gradient (dX = dY Wt^T runs on the forward kernel) - cached per weight and refreshed when the
parameter changes (torch's version counter, or the native optimizer's `weights_changed()` epoch; refreshed IN PLACE, so
captured graphs keep their addresses).
"""
import ctypes
import os
import math
import weakref

import torch

from . import _native as N
from . import ops

BF16 = torch.bfloat16
_WEIGHTS_EPOCH = N.WEIGHTS_EPOCH


def dev_bf16(t, what):
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("%s: expected a tensor on a HIP device, got %s - no CPU fallback" % (what, t.device))
    if t.dtype != BF16:
        raise RuntimeError("%s: expected bfloat16, got %s" % (what, t.dtype))
    return t.data_ptr()


def cast_bf16(x):
    """fp32 -> bfloat16 (round to nearest even), same shape."""
    x = ops._contig(x)
    y = torch.empty(x.shape, dtype=BF16, device=x.device)
    if x.numel():
        N.check(N.lib().vb_cast_f32_bf16(N.stream_ptr(), x.numel(), N.dev_f32(x, "cast input"), y.data_ptr()), "vb_cast_f32_bf16")
    return y


def cast_f32(x):
    """bfloat16 -> fp32 (exact), same shape."""
    x = ops._contig(x)
    y = torch.empty(x.shape, dtype=torch.float32, device=x.device)
    if x.numel():
        N.check(N.lib().vb_cast_bf16_f32(N.stream_ptr(), x.numel(), dev_bf16(x, "cast input"), y.data_ptr()), "vb_cast_bf16_f32")
    return y


def eligible(K, n_out, seg_n=None, act=None):
    """Shapes the bf16 kernels serve (every linear of the two-stream encoders: 768 / 1024 / 2048 / 3072 / 4096 wide):
    forward K % 64, N % 128; input gradient N % 64, K % 128; weight gradient seg_n % 256, K % 128."""
    seg_n = n_out if seg_n is None else seg_n
    return K % 128 == 0 and n_out % 128 == 0 and seg_n % 256 == 0 and act in (None, "none", "gelu", "relu")


class _Shadow(object):
    """bf16 copies of one (stacked) weight: w16 [N, K] row-major for the forward, wt16 [K, N] for the input gradient."""
    __slots__ = ("w16", "wt16", "wrefs", "keep", "vers", "epoch", "seg_n", "K", "nseg")


# (id(first weight tensor), number of stacked segments) -> _Shadow. "table": the device table of vb_weight_shadow_multi over every live entry of a device
# (rebuilt when an entry is added or dropped); "epoch": the _native.WEIGHTS_EPOCH the shadows of that device were last
# refreshed at by the one-launch refresh.
_SHADOWS = {}
_TABLES = {}


def shadow_cache_clear():
    _SHADOWS.clear()
    _TABLES.clear()


def _refresh_one(e, weights):
    n = e.seg_n * e.nseg
    for s, w in enumerate(weights):
        N.check(N.lib().vb_weight_shadow_bf16(
            N.stream_ptr(), e.seg_n, e.K, N.dev_f32(w.detach(), "linear weight"), e.K, e.w16.data_ptr() + 2 * s * e.seg_n * e.K,
            e.K, e.wt16.data_ptr() + 2 * s * e.seg_n, n), "vb_weight_shadow_bf16")


def _refresh_all(device):
    """ONE launch refreshes every registered shadow of `device` (the native optimizer rewrote all parameters). The table
    lives on the device and is rebuilt only when the set of registered weights changes."""
    import numpy as np
    dev_key = device.index
    t = _TABLES.get(dev_key)
    live = [(k, e) for k, e in _SHADOWS.items() if e.w16.device.index == dev_key and all(r() is not None for r in e.wrefs)]
    sig = tuple(k for k, _ in live)
    if t is None or t["sig"] != sig:
        for k in [k for k, e in _SHADOWS.items() if e.w16.device.index == dev_key and any(r() is None for r in e.wrefs)]:
            del _SHADOWS[k]                       # (weights whose model is gone)
        rec = np.dtype([("w", "<u8"), ("w16", "<u8"), ("wt16", "<u8"), ("rows", "<i4"), ("cols", "<i4"), ("ld16", "<i8"),
                        ("ldt", "<i8"), ("tile0", "<i8")])
        rows, tile0 = [], 0
        for _k, e in live:
            n = e.seg_n * e.nseg
            for s_, r in enumerate(e.wrefs):
                w = r()
                rows.append((w.data_ptr(), e.w16.data_ptr() + 2 * s_ * e.seg_n * e.K, e.wt16.data_ptr() + 2 * s_ * e.seg_n,
                             e.seg_n, e.K, e.K, n, tile0))
                tile0 += (e.seg_n // 64) * (e.K // 64)
        host = np.array(rows, dtype=rec)
        dev_tab = torch.from_numpy(host.view(np.uint8).reshape(-1).copy()).to(device)
        t = _TABLES[dev_key] = {"sig": sig, "tab": dev_tab, "n": len(rows), "tiles": tile0,
                                "ptrs": tuple(r[0] for r in rows)}
    elif t["ptrs"] != tuple(r().data_ptr() for _k, e in live for r in e.wrefs):
        _TABLES.pop(dev_key)                      # a parameter moved (model.to(), load into new storage): rebuild
        return _refresh_all(device)
    if t["n"]:
        N.check(N.lib().vb_weight_shadow_multi(N.stream_ptr(), t["n"], t["tab"].data_ptr(), t["tiles"]), "vb_weight_shadow_multi")
    ep = _WEIGHTS_EPOCH[0]
    t["epoch"] = ep
    for _k, e in live:
        e.epoch = ep
        e.vers = tuple(r()._version for r in e.wrefs)


def refresh_stale(device):
    """Start of a model forward (BertModel.forward, on the stream the text / image branches fork from): when the weights
    epoch moved since the shadows of `device` were refreshed - an optimizer stepped - refresh all of them now, with the one
    launch over the device table (built here, eagerly, so that it never is built inside a stream capture or a branch)."""
    ep = _WEIGHTS_EPOCH[0]
    t = _TABLES.get(device.index)
    if t is not None and t.get("epoch") == ep:
        return
    if not any(e.w16.device.index == device.index for e in _SHADOWS.values()):
        return
    with torch.no_grad():
        _refresh_all(device)


def shadows(weights, biases=None):
    """(W16 [N, K] bf16, Wt16 [K, N] bf16) of the stacked segments, cached until a segment is rewritten: torch's version
    counters are compared per call; a bump of the native optimizer's epoch (`_native.weights_changed()`: parameters
    rewritten through raw pointers) refreshes ALL registered shadows of the device with one launch at the next call."""
    w0 = weights[0]
    nseg = len(weights)
    key = (id(w0), nseg)
    e = _SHADOWS.get(key)
    if e is not None and any(r() is not w for r, w in zip(e.wrefs, weights)):
        e = None                                   # a recycled id
    if e is None:
        seg_n, K = w0.shape
        for w in weights:
            if w.shape != (seg_n, K) or not w.is_contiguous():
                raise RuntimeError("linear (bf16): weight segments must be contiguous and equally shaped")
        if seg_n % 64 != 0 or K % 64 != 0:
            raise RuntimeError("linear (bf16): weight dimensions must be multiples of 64")
        e = _Shadow()
        e.seg_n, e.K, e.nseg = seg_n, K, nseg
        e.w16 = torch.empty((nseg * seg_n, K), dtype=BF16, device=w0.device)
        e.wt16 = torch.empty((K, nseg * seg_n), dtype=BF16, device=w0.device)
        e.wrefs = [weakref.ref(w) for w in weights]
        e.keep = [w.detach() for w in weights]   # an alias of every segment: its address cannot be recycled under the entry
        with torch.no_grad():
            _refresh_one(e, weights)
        e.vers, e.epoch = tuple(w._version for w in weights), _WEIGHTS_EPOCH[0]
        _SHADOWS[key] = e
        return e.w16, e.wt16
    if e.epoch != _WEIGHTS_EPOCH[0]:
        with torch.no_grad():
            _refresh_all(w0.device)
        if e.epoch != _WEIGHTS_EPOCH[0]:          # (registered after the table was built and not live in it)
            with torch.no_grad():
                _refresh_one(e, weights)
            e.vers, e.epoch = tuple(w._version for w in weights), _WEIGHTS_EPOCH[0]
        return e.w16, e.wt16
    vers = tuple(w._version for w in weights)
    if vers != e.vers:
        with torch.no_grad():
            _refresh_one(e, weights)
        e.vers = vers
    return e.w16, e.wt16


def _rows2(x, K):
    """[..., K] bf16 -> (2-D contiguous view, leading shape)."""
    x = ops._contig(x)
    return x.view(-1, K), tuple(x.shape[:-1])


def linear_fwd(x, weights, biases, act=None, residual=None, drop_p=0.0, seed=0, want_act_grad=False, out_f32=False):
    """act(x @ cat(weights).T + cat(biases)), then dropout (+ residual). x / residual bf16; returns (y, act_grad or None):
    y bf16 [..., N] (fp32 with out_f32), act_grad = gelu'(pre-activation) bf16 when asked for."""
    nseg, seg_n, K = len(weights), weights[0].shape[0], weights[0].shape[1]
    n_out = nseg * seg_n
    if x.shape[-1] != K:
        raise RuntimeError("linear: input has %d features, weight expects %d" % (x.shape[-1], K))
    w16, _ = shadows(weights)
    x2, lead = _rows2(x, K)
    M = x2.shape[0]
    y = torch.empty(lead + (n_out,), dtype=torch.float32 if out_f32 else BF16, device=x.device)
    dact = torch.empty(lead + (n_out,), dtype=BF16, device=x.device) if want_act_grad else None
    a = N.LinearBf16Args()
    a.A, a.lda, a.W, a.ldw = dev_bf16(x2, "linear input"), K, w16.data_ptr(), K
    if biases is not None:
        a.bias_segments = nseg
        for s_ in range(nseg):
            a.bias[s_] = N.dev_f32(biases[s_], "linear bias") if biases[s_] is not None else None
    if out_f32:
        a.C32, a.ldc32 = y.data_ptr(), n_out
    else:
        a.C, a.ldc = y.data_ptr(), n_out
    if residual is not None:
        residual = ops._contig(residual)
        if residual.numel() != M * n_out:
            raise RuntimeError("linear: residual shape mismatch")
        a.residual, a.ldr = dev_bf16(residual, "linear residual"), n_out
    if dact is not None:
        a.act_grad, a.ldg = dact.data_ptr(), n_out
    a.M, a.N, a.K = M, n_out, K
    a.act = N.ACT_CODES[act]
    a.dropout_p, a.seed = float(drop_p), int(seed)
    ops._timed(lambda: N.check(N.lib().vb_linear_bf16(N.stream_ptr(), ctypes.byref(a)), "vb_linear_bf16"),
               2.0 * M * n_out * K, ("fwd16", M, seg_n, K, nseg))
    return y, dact


def linear_bwd_input(dy, weights, biases, in_features, residual=None, mul=None):
    """dX = (dY @ cat(weights)) (+ residual) (* mul): the forward kernel on the TRANSPOSED shadow. All tensors bf16."""
    nseg, seg_n = len(weights), weights[0].shape[0]
    n = nseg * seg_n
    _, wt16 = shadows(weights)
    dy2, lead = _rows2(dy, n)
    M = dy2.shape[0]
    dx = torch.empty(lead + (in_features,), dtype=BF16, device=dy.device)
    a = N.LinearBf16Args()
    a.A, a.lda, a.W, a.ldw = dev_bf16(dy2, "linear grad_output"), n, wt16.data_ptr(), n
    a.C, a.ldc = dx.data_ptr(), in_features
    if residual is not None:
        residual = ops._contig(residual)
        if residual.numel() != M * in_features:
            raise RuntimeError("linear_bwd_input: residual shape mismatch")
        a.residual, a.ldr = dev_bf16(residual, "linear residual grad"), in_features
    if mul is not None:
        mul = ops._contig(mul)
        if mul.numel() != M * in_features:
            raise RuntimeError("linear_bwd_input: multiplier shape mismatch")
        a.mul, a.ldm = dev_bf16(mul, "linear activation derivative"), in_features
    a.M, a.N, a.K = M, in_features, n
    ops._timed(lambda: N.check(N.lib().vb_linear_bf16(N.stream_ptr(), ctypes.byref(a)), "vb_linear_bf16 (dgrad)"),
               2.0 * M * n * in_features, ("dgrad16", M, seg_n, in_features, nseg))
    return dx


def linear_bwd_weight(dy, x, nseg, seg_n, want_bias, dw_out=None, db_out=None):
    """Per segment: dW_s += dY[:, s]^T @ X (fp32, atomics into the targets) and db_s += colsum(dY[:, s]) - one launch, the
    bias gradient comes out of the fragments the weight-gradient kernel holds anyway. dy / x bf16.
    Same contract as ops.linear_bwd_weight: targets not given are slices of one zero-filled buffer allocated here.
    Deterministic setting on (default): the contraction splits go through the per-stream workspace and an ordered reduce."""
    N.ensure_deterministic(dy.device)
    n = nseg * seg_n
    dy2, _ = _rows2(dy, n)
    K = x.shape[-1]
    x2, _ = _rows2(x, K)
    M = x2.shape[0]
    if dy2.shape[0] != M:
        raise RuntimeError("linear_bwd_weight: row count mismatch")
    dw_out = dw_out if dw_out is not None else [None] * nseg
    db_out = db_out if db_out is not None else [None] * nseg
    wsz, bsz = (seg_n * K + 3) // 4 * 4, (seg_n + 3) // 4 * 4
    need = sum(wsz for s in range(nseg) if dw_out[s] is None) + \
        sum(bsz for s in range(nseg) if want_bias[s] and db_out[s] is None)
    flat = torch.zeros(need, dtype=torch.float32, device=dy.device) if need else None
    a = N.WgradBf16Args()
    a.dY, a.ldy, a.X, a.ldx = dev_bf16(dy2, "linear grad_output"), n, dev_bf16(x2, "linear input"), K
    a.M, a.K, a.nseg, a.seg_n, a.ldw = M, K, nseg, seg_n, K
    dws, dbs, off = [], [], 0
    for s in range(nseg):
        dw = dw_out[s]
        if dw is None:
            dw = flat[off:off + seg_n * K].view(seg_n, K)
            off += wsz
        elif dw.shape != (seg_n, K) or not dw.is_contiguous():
            raise RuntimeError("linear_bwd_weight: gradient target must be a contiguous [seg_n, K] tensor")
        db = None
        if want_bias[s]:
            db = db_out[s]
            if db is None:
                db = flat[off:off + seg_n]
                off += bsz
        a.dW[s] = N.dev_f32(dw, "weight gradient")
        a.dbias[s] = N.dev_f32(db, "bias gradient") if db is not None else None
        dws.append(dw)
        dbs.append(db)
    ops._timed(lambda: N.check(N.lib().vb_wgrad_bf16(N.stream_ptr(), ctypes.byref(a)), "vb_wgrad_bf16"),
               2.0 * M * n * K, ("wgrad16", M, seg_n, K, nseg))
    return dws, dbs


# opt-in (VB_BF16_RAGGED_WGRAD=1): measured on one box, B = 256: the decoder's weight gradient 0.326 ms on the fp32-tensor kernel
# -> 0.191 ms here + ~0.1 ms of casts, the step unchanged within 0.4 % (profiles/r06_bf16_ragged_wgrad_ab.txt)
_RAGGED_WGRAD = os.environ.get("VB_BF16_RAGGED_WGRAD", "0") == "1"


def ragged_wgrad_ok(n_out, K, rows):
    """The weight gradient of an fp32-tensor linear whose width is NOT a tile multiple (the MLM decoder: 30,522 x 768) on the
    bf16 weight-gradient kernel (linear_bwd_weight_ragged): in the bf16 mode, for outputs wide enough to pay for the two casts."""
    return _RAGGED_WGRAD and N.bf16_stream() and K % 128 == 0 and n_out >= 4096 and rows >= 64


def linear_bwd_weight_ragged(dy, x, want_bias, dw_out=None, db_out=None):
    """dW += dY^T X, db += colsum(dY) for ONE fp32-tensor linear of any output width n (round 6: the heads of the bf16 mode,
    reference vilbert.py:1178-1196 - the tied 30,522 x 768 decoder was 3.4 % (B = 256) ... 7.7 % (B = 64) of the bf16 step on
    the fp32-tensor kernel). dy fp32 [rows, n] (row-strided view allowed), x fp32 [rows, K]: both are rounded to bf16 - what
    the fp32-tensor kernel of this mode does with its operands too - dy into a buffer padded with ZERO columns to a
    multiple of 256, and vb_wgrad_bf16 writes rows < n only (n_valid). Same contract as ops.linear_bwd_weight (nseg = 1)."""
    N.ensure_deterministic(dy.device)
    n, K = dy.shape[-1], x.shape[-1]
    dy2 = dy.reshape(-1, n) if dy.dim() != 2 else dy
    x2 = ops._contig(x).reshape(-1, K)
    M = x2.shape[0]
    if dy2.shape[0] != M:
        raise RuntimeError("linear_bwd_weight: row count mismatch")
    if dy2.stride(1) != 1 or dy2.stride(0) % 4 != 0:
        dy2 = dy2.contiguous() if n % 4 == 0 else torch.nn.functional.pad(dy2, (0, 4 - n % 4))[:, :n]
    n_pad = (n + 255) // 256 * 256
    dy16 = torch.empty(M, n_pad, dtype=BF16, device=dy.device)
    N.check(N.lib().vb_cast_rows_f32_bf16(N.stream_ptr(), M, n, N.dev_f32(dy2, "linear grad_output"), dy2.stride(0),
                                          dy16.data_ptr(), n_pad), "vb_cast_rows_f32_bf16")
    x16 = cast_bf16(x2)
    dw = dw_out[0] if dw_out is not None and dw_out[0] is not None else None
    db = db_out[0] if db_out is not None and db_out[0] is not None else None
    if dw is None:
        dw = torch.zeros(n, K, dtype=torch.float32, device=dy.device)
    elif dw.shape != (n, K) or not dw.is_contiguous():
        raise RuntimeError("linear_bwd_weight: gradient target must be a contiguous [n, K] tensor")
    if want_bias[0] and db is None:
        db = torch.zeros(n, dtype=torch.float32, device=dy.device)
    a = N.WgradBf16Args()
    a.dY, a.ldy, a.X, a.ldx = dy16.data_ptr(), n_pad, x16.data_ptr(), K
    a.M, a.K, a.nseg, a.seg_n, a.ldw, a.n_valid = M, K, 1, n_pad, K, n
    a.dW[0] = N.dev_f32(dw, "weight gradient")
    a.dbias[0] = N.dev_f32(db, "bias gradient") if (want_bias[0] and db is not None) else None
    ops._timed(lambda: N.check(N.lib().vb_wgrad_bf16(N.stream_ptr(), ctypes.byref(a)), "vb_wgrad_bf16"),
               2.0 * M * n * K, ("wgrad16", M, n, K, 1))
    # (on a weight-gradient side stream the caller has made that stream torch's current one: the two bf16 temporaries come
    # from its allocator pool)
    return [dw], [db if want_bias[0] else None]


def layernorm_fwd(x, gamma, beta, eps, want_stats=False):
    x = ops._contig(x)
    rows, cols = ops._rows(x)
    y = torch.empty_like(x)
    mean = rstd = None
    if want_stats:
        mean = torch.empty(rows, dtype=torch.float32, device=x.device)
        rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
    N.check(N.lib().vb_layernorm_fwd_bf16(
        N.stream_ptr(), rows, cols, dev_bf16(x, "layernorm input"), N.dev_f32(gamma, "layernorm weight"),
        N.dev_f32(beta, "layernorm bias"), eps, y.data_ptr(), mean.data_ptr() if want_stats else None,
        rstd.data_ptr() if want_stats else None), "vb_layernorm_fwd_bf16")
    return y, mean, rstd


_LN_WS = {}


def _ln_workspace(device, floats):
    """Partial dgamma / dbeta rows of one LayerNorm backward: one grow-only buffer per (device, stream) - the two kernels of a
    launch pair are ordered on their stream, and streams that run concurrently get their own."""
    key = (device.index, N.raw_stream(device.index))
    ws = _LN_WS.get(key)
    if ws is None or ws.numel() < floats:
        ws = _LN_WS[key] = torch.empty(max(floats, 1 << 20), dtype=torch.float32, device=device)
    return ws


def layernorm_bwd(dy, x, mean, rstd, gamma, dgamma=None, dbeta=None, drop=None):
    """(dx, dgamma, dbeta[, dx under the dropout mask `drop` = (p, seed) of the dense layer in front]); dy / x / dx bf16,
    dgamma / dbeta fp32 [cols] targets (OVERWRITTEN)."""
    dy, x = ops._contig(dy), ops._contig(x)
    rows, cols = ops._rows(x)
    dx = torch.empty_like(x)
    dgamma = dgamma if dgamma is not None else torch.empty(cols, dtype=torch.float32, device=x.device)
    dbeta = dbeta if dbeta is not None else torch.empty(cols, dtype=torch.float32, device=x.device)
    ws = _ln_workspace(x.device, (rows + 15) // 16 * 2 * cols)      # = vb_layernorm_bwd_bf16_workspace(rows, cols)
    twin = drop is not None and drop[0] > 0.0
    dxd = torch.empty_like(x) if twin else None
    N.check(N.lib().vb_layernorm_bwd_bf16(
        N.stream_ptr(), rows, cols, dev_bf16(dy, "layernorm grad_output"), dev_bf16(x, "layernorm input"),
        N.dev_f32(mean, "layernorm mean"), N.dev_f32(rstd, "layernorm rstd"), N.dev_f32(gamma, "layernorm weight"),
        dx.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(), ws.data_ptr(), dxd.data_ptr() if twin else None,
        float(drop[0]) if twin else 0.0, int(drop[1]) if twin else 0), "vb_layernorm_bwd_bf16")
    return (dx, dgamma, dbeta, dxd) if twin else (dx, dgamma, dbeta)


def dropout(x, p, seed):
    """dropout of a bf16 tensor (rare path: the LayerNorm backward normally hands over the masked gradient): through the fp32
    kernel and two casts."""
    return cast_bf16(ops.dropout(cast_f32(x), p, seed))


def _attn_args16(q, k, v, mask_add, heads, drop_p, seed):
    Bq, Sq, H = q.shape
    Bk, Sk, _ = k.shape
    B = max(Bq, Bk)
    d = H // heads
    for t, nm in ((q, "q"), (k, "k"), (v, "v")):
        if t.stride(2) != 1 or (t.shape[0] > 1 and t.stride(0) != t.shape[1] * t.stride(1)):
            raise RuntimeError("attention: %s must be a row-strided view" % nm)
    a = N.AttentionArgs()
    a.batch, a.heads, a.head_dim, a.n_q, a.n_k = B, heads, d, Sq, Sk
    a.q_batch, a.kv_batch = Bq, Bk
    a.Q, a.ldq = dev_bf16(q, "attention q"), q.stride(1)
    a.K, a.ldk = dev_bf16(k, "attention k"), k.stride(1)
    a.V, a.ldv = dev_bf16(v, "attention v"), v.stride(1)
    keep = []
    if mask_add is not None:
        mask_add = ops._contig(mask_add)
        if mask_add.numel() != Bk * Sk:
            raise RuntimeError("attention: mask must hold %d x %d values" % (Bk, Sk))
        a.mask_add = N.dev_f32(mask_add, "attention mask")
        keep.append(mask_add)
    a.scale = 1.0 / math.sqrt(d)
    a.dropout_p, a.seed = float(drop_p), int(seed)
    return a, keep, (B, Sq, Sk, H)


def attention_fwd(q, k, v, mask_add, heads, want_lse=False, drop_p=0.0, seed=0):
    """bf16 q [Bq, Sq, H*], k / v [Bk, Sk, H*] row-strided views (column slices of the fused projection); fp32 additive mask.
    Returns (ctx bf16 [B, Sq, H], lse fp32 [B, heads, Sq] or None). At most ops.MAX_KEYS keys."""
    if k.shape[1] > ops.MAX_KEYS:
        raise RuntimeError("attention (bf16): %d keys - one launch serves at most %d" % (k.shape[1], ops.MAX_KEYS))
    a, keep, (B, Sq, Sk, H) = _attn_args16(q, k, v, mask_add, heads, drop_p, seed)
    out = torch.empty(B, Sq, H, dtype=BF16, device=q.device)
    lse = torch.empty(B, heads, Sq, dtype=torch.float32, device=q.device) if want_lse else None
    a.O, a.ldo = out.data_ptr(), H
    a.lse = lse.data_ptr() if want_lse else None
    N.check(N.lib().vb_attention_fwd_bf16(N.stream_ptr(), ctypes.byref(a)), "vb_attention_fwd_bf16")
    return out, lse


def attention_bwd(d_out, q, k, v, mask_add, heads, lse, dq, dk, dv, drop_p=0.0, seed=0):
    """Writes dq / dk / dv (bf16 row-strided views, e.g. column slices of one fused gradient buffer) in place."""
    a, keep, (B, Sq, Sk, H) = _attn_args16(q, k, v, mask_add, heads, drop_p, seed)
    d_out = ops._contig(d_out)
    a.lse = N.dev_f32(lse, "attention lse")
    g = N.AttentionGrads()
    g.dO, g.lddo = dev_bf16(d_out, "attention grad_output"), H
    g.dQ, g.lddq = dev_bf16(dq, "attention dq"), dq.stride(1)
    g.dK, g.lddk = dev_bf16(dk, "attention dk"), dk.stride(1)
    g.dV, g.lddv = dev_bf16(dv, "attention dv"), dv.stride(1)
    dvec = torch.empty(B, heads, Sq, dtype=torch.float32, device=q.device)
    g.dvec = dvec.data_ptr()
    N.check(N.lib().vb_attention_bwd_bf16(N.stream_ptr(), ctypes.byref(a), ctypes.byref(g)), "vb_attention_bwd_bf16")
