"""Tensor-level launchers of the native layer (forward and backward kernels).

Every function here validates its tensors, allocates the outputs with ``torch.empty`` / ``torch.zeros``
and enqueues exactly the kernels of include/vilbert_hip.h on the current stream. No torch arithmetic.
"""
import ctypes
import os as _os
import weakref
import math

import torch

from . import _native as N


# Optional per-launch timing of the GEMM kernel (bench.py's roofline leg): when enabled every GEMM
# launch (forward, dgrad, wgrad) is bracketed by HIP events recorded on the launch stream.
_PROFILE = {"on": False, "events": [], "flops": 0.0, "tags": []}


def profile_linear(enable):
    """enable=True starts collecting; enable=False stops and returns (total_ms, total_flops, launches)."""
    if enable:
        _PROFILE.update(on=True, events=[], flops=0.0, tags=[])
        return None
    _PROFILE["on"] = False
    torch.cuda.synchronize()
    times = [a.elapsed_time(b) for a, b in _PROFILE["events"]]
    ms = sum(times)
    out = (ms, _PROFILE["flops"], len(_PROFILE["events"]))
    # per-shape breakdown of the last profiled region: {(kind, M, N, K, nseg): [launches, ms, flops]}
    by_shape = {}
    for (tag, fl), t in zip(_PROFILE["tags"], times):
        e = by_shape.setdefault(tag, [0, 0.0, 0.0])
        e[0] += 1
        e[1] += t
        e[2] += fl
    _PROFILE["by_shape"] = by_shape
    _PROFILE["events"] = []
    return out


def profile_breakdown():
    """Per-shape GEMM time of the last profile_linear region, slowest total first."""
    rows = sorted(_PROFILE.get("by_shape", {}).items(), key=lambda kv: -kv[1][1])
    return [(tag, n, ms, fl / (ms * 1e-3) / 1e12 if ms > 0 else 0.0) for tag, (n, ms, fl) in rows]


def _timed(fn, flops, tag=None):
    if not _PROFILE["on"]:
        return fn()
    _PROFILE["tags"].append((tag, flops))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    out = fn()
    e1.record()
    _PROFILE["events"].append((e0, e1))
    _PROFILE["flops"] += flops
    return out


def _rows(t):
    """[..., C] -> (rows, C) of a contiguous tensor."""
    return t.numel() // t.shape[-1], t.shape[-1]


def _contig(t):
    return t if t.is_contiguous() else t.contiguous()


def _row_view(x, K):
    """2-D row-strided view (x2, ld, leading shape) of [..., K]; copies only if it has to."""
    if x.dim() == 2 and x.stride(1) == 1 and x.stride(0) >= K:
        return x, x.stride(0), (x.shape[0],)
    x = _contig(x)
    return x.view(-1, K), K, tuple(x.shape[:-1])


def _row_strided(t):
    """2-D tensor whose rows are contiguous and uniformly strided (e.g. [rows, 30522] inside a [rows, 30524] buffer)."""
    return t.dim() == 2 and t.stride(1) == 1 and t.stride(0) >= t.shape[1]


# ---------------------------------------------------------------------------------------------------------------
# FP8 forward path (BASELINE config 5; numerics: oracle/fp8_oracle.py, kernels: csrc/fp8.hip)
# ---------------------------------------------------------------------------------------------------------------
FP8_K_MULTIPLE = 128     # the fp8 kernel's K step
_FP8_WEIGHTS = {}        # (data_ptr of every segment) -> (versions, weights epoch, codes [N, K] u8, scales [N], bias [N])
_WEIGHTS_EPOCH = N.WEIGHTS_EPOCH   # bumped by the native optimizer (parameters rewritten behind torch's version counters)


def quantize_rows_fp8(x2, out=None, scale_out=None):
    """x2 [rows, K] fp32 (row stride allowed) -> (codes [rows, K] uint8, scale [rows] fp32); one scale per row."""
    rows, K = x2.shape
    if x2.stride(1) != 1:
        x2 = x2.contiguous()
    q = out if out is not None else torch.empty((rows, K), dtype=torch.uint8, device=x2.device)
    sc = scale_out if scale_out is not None else torch.empty((rows,), dtype=torch.float32, device=x2.device)
    N.check(N.lib().vb_quantize_rows_fp8(N.stream_ptr(), rows, K, N.dev_f32(x2, "fp8 quantise input"), x2.stride(0),
                                         q.data_ptr(), q.stride(0), sc.data_ptr()), "vb_quantize_rows_fp8")
    return q, sc


def fp8_cache_clear():
    _FP8_WEIGHTS.clear()


def _fp8_sweep():
    """Drop the entries whose weights nobody else holds any more (their model was deleted)."""
    dead = [k for k, e in _FP8_WEIGHTS.items() if all(r() is None for r in e[6])]
    for k in dead:
        del _FP8_WEIGHTS[k]


def _fp8_weights(weights, biases):
    """Quantised copy of the (stacked) weight, cached until a segment is rewritten. The entry keeps an alias of every
    segment alive, so a key (the segments' addresses) can never be recycled by a different tensor while it is cached;
    entries whose weight tensors (the nn.Parameter objects the callers pass) are gone are dropped at the next miss."""
    key = tuple(w.data_ptr() for w in weights)
    vers = tuple(w._version for w in weights) + tuple(-1 if b is None else b._version for b in (biases or []))
    seg_n, K = weights[0].shape
    n = seg_n * len(weights)
    hit = _FP8_WEIGHTS.get(key)
    if hit is not None and hit[2].shape != (n, K):
        hit = None                                  # same address, different view of a buffer
    if hit is not None and hit[0] == vers and hit[1] == _WEIGHTS_EPOCH[0]:
        return hit[2], hit[3], hit[4]
    dev = weights[0].device
    if hit is not None:
        q, sc, bias = hit[2], hit[3], hit[4]        # refresh in place: same addresses (HIP graphs keep them)
    else:
        _fp8_sweep()
        q = torch.empty((n, K), dtype=torch.uint8, device=dev)
        sc = torch.empty((n,), dtype=torch.float32, device=dev)
        bias = None
    with torch.no_grad():
        for s, w in enumerate(weights):
            quantize_rows_fp8(w.detach(), q[s * seg_n:(s + 1) * seg_n], sc[s * seg_n:(s + 1) * seg_n])
        if biases is not None and all(b is not None for b in biases):
            if len(biases) == 1:
                bias = biases[0].detach()
            elif bias is None:
                bias = torch.cat([b.detach() for b in biases])
            else:
                # refresh in place: the concatenation kernel writes the cached buffer itself (no temporary + device-to-device
                # copy, which inside a captured step is a memcpy node between two kernel nodes)
                torch.cat([b.detach() for b in biases], out=bias)
        elif biases is not None and any(b is not None for b in biases):
            raise RuntimeError("linear (fp8): either every weight segment has a bias or none")
        else:
            bias = None
    _FP8_WEIGHTS[key] = (vers, _WEIGHTS_EPOCH[0], q, sc, bias, [w.detach() for w in weights],
                         [weakref.ref(w) for w in weights])
    return q, sc, bias


def _fp8_eligible(x2, K, n_out, biases):
    # K a multiple of the kernel's K step; tiny heads (N < 64), ragged K and stacked weights of which only some
    # segments have a bias stay exact fp32
    uniform_bias = biases is None or all(b is None for b in biases) or all(b is not None for b in biases)
    return K % FP8_K_MULTIPLE == 0 and n_out >= 64 and x2.is_cuda and uniform_bias


def _linear_fwd_fp8(x, x2, M, K, weights, biases, n_out, y, ldc, act, residual, pre, want_act_grad, drop_p, seed):
    wq, ws, bias = _fp8_weights(weights, biases)
    pre_q = getattr(x, "_vb_fp8", None)     # codes emitted by the producing LayerNorm (layernorm_fwd)
    if pre_q is not None and pre_q[2] == x._version and pre_q[0].shape == (M, K) and x.is_contiguous():
        xq, xs = pre_q[0], pre_q[1]
    else:
        xq, xs = quantize_rows_fp8(x2)
    a = N.LinearFp8Args()
    a.A, a.lda, a.a_scale = xq.data_ptr(), K, xs.data_ptr()
    a.W, a.ldw, a.w_scale = wq.data_ptr(), K, ws.data_ptr()
    a.bias = N.dev_f32(bias, "linear bias") if bias is not None else None
    a.C, a.ldc = y.data_ptr(), ldc
    if residual is not None:
        a.residual, a.ldr = N.dev_f32(residual, "linear residual"), n_out
    if pre is not None and want_act_grad:
        a.act_grad, a.ldg = pre.data_ptr(), n_out
    elif pre is not None:
        a.preact, a.ldp = pre.data_ptr(), n_out
    a.M, a.N, a.K = M, n_out, K
    a.act = N.ACT_CODES[act]
    a.dropout_p, a.seed = float(drop_p), int(seed)
    _timed(lambda: N.check(N.lib().vb_linear_fwd_fp8(N.stream_ptr(), ctypes.byref(a)), "vb_linear_fwd_fp8"),
           2.0 * M * n_out * K, ("fwd_fp8", M, n_out, K, 1))


# ---------------------------------------------------------------------------------------------------------------
# MX e4m3 forward path (round 4; kernels csrc/mx8.hip, numerics oracle/fp8_oracle.py `mx_*`): block-scaled operands, the
# producers (LayerNorm, a GEMM epilogue) emit the codes the next linear consumes
# ---------------------------------------------------------------------------------------------------------------
MX_BLOCK = 32            # elements per scale
MX_K_MULTIPLE = 128      # K of a linear (= the 4 scale bytes of one uint32 word), also the GEMM's column tile
_MX_WEIGHTS = {}


class MxRows(object):
    """A [rows, K] activation in the MX format: `q` codes [rows, K] uint8, `s` scale words [K / 128, srows] int32 (word
    (kt, r) = the four E8M0 bytes of row r's K range [128 kt, 128 kt + 128)); `lead` = the leading shape of the tensor it
    stands for. Not a torch tensor: only `linear_fwd` consumes it."""
    __slots__ = ("q", "s", "srows", "rows", "K", "lead")
    requires_grad = False      # (inference only; lets the autograd dispatch of functional.py treat it like a tensor)
    is_cuda = True

    def __init__(self, rows, K, device, lead):
        self.rows, self.K, self.lead = rows, K, tuple(lead)
        self.srows = (rows + 255) // 256 * 256          # the GEMM fetches the scale words of a 256-row tile as one piece
        self.q = torch.empty((rows, K), dtype=torch.uint8, device=device)
        self.s = torch.empty((K // MX_K_MULTIPLE, self.srows), dtype=torch.int32, device=device)

    @property
    def shape(self):
        return self.lead + (self.K,)

    @property
    def device(self):
        return self.q.device


def quantize_rows_mx(x2, lead=None, out=None):
    """x2 [rows, K] fp32 or bfloat16 (row stride allowed, K % 128 == 0) -> MxRows."""
    rows, K = x2.shape
    if x2.stride(1) != 1:
        x2 = x2.contiguous()
    m = out if out is not None else MxRows(rows, K, x2.device, lead if lead is not None else (rows,))
    if x2.dtype == torch.bfloat16:
        if not x2.is_cuda:
            raise RuntimeError("mx quantise input: expected a tensor on a HIP device - no CPU fallback")
        N.check(N.lib().vb_quantize_rows_mx_bf16(N.stream_ptr(), rows, K, x2.data_ptr(), x2.stride(0), m.q.data_ptr(),
                                                 m.q.stride(0), m.s.data_ptr(), m.srows), "vb_quantize_rows_mx_bf16")
        return m
    N.check(N.lib().vb_quantize_rows_mx(N.stream_ptr(), rows, K, N.dev_f32(x2, "mx quantise input"), x2.stride(0),
                                        m.q.data_ptr(), m.q.stride(0), m.s.data_ptr(), m.srows), "vb_quantize_rows_mx")
    return m


def mx_cache_clear():
    _MX_WEIGHTS.clear()


def _mx_weights(weights, biases, n_pad=None):
    """MX copy of the (stacked) weight, cached like `_fp8_weights` (same invalidation rules). n_pad > rows: the copy has
    n_pad rows, the extra ones all-zero codes with scale byte 0 (a head whose width is not a multiple of the GEMM's 128-column
    tile: its extra output columns are computed as 0 + 0 and never shown)."""
    key = tuple(w.data_ptr() for w in weights)
    vers = tuple(w._version for w in weights) + tuple(-1 if b is None else b._version for b in (biases or []))
    seg_n, K = weights[0].shape
    n_real = seg_n * len(weights)
    n = n_real if n_pad is None else n_pad
    hit = _MX_WEIGHTS.get(key)
    if hit is not None and hit[2].q.shape != (n, K):
        hit = None
    if hit is not None and hit[0] == vers and hit[1] == _WEIGHTS_EPOCH[0]:
        return hit[2], hit[3]
    dev = weights[0].device
    if hit is not None:
        m, bias = hit[2], hit[3]                    # refresh in place (captured graphs keep the addresses)
    else:
        dead = [k for k, e in _MX_WEIGHTS.items() if all(r() is None for r in e[5])]
        for k in dead:
            del _MX_WEIGHTS[k]
        m, bias = MxRows(n, K, dev, (n,)), None
        if n != n_real:
            m.q.zero_()
            m.s.zero_()
    with torch.no_grad():
        cat = weights[0].detach() if len(weights) == 1 else torch.cat([w.detach() for w in weights])
        quantize_rows_mx(cat, out=m)                # writes rows [0, n_real) of the codes and of every scale plane
        if biases is not None and all(b is not None for b in biases):
            bcat = biases[0].detach() if len(biases) == 1 else torch.cat([b.detach() for b in biases])
            if n != n_real:
                if bias is None:
                    bias = torch.zeros(n, dtype=torch.float32, device=dev)
                bias[:n_real].copy_(bcat)
            elif bias is None or len(biases) == 1:
                bias = bcat
            else:
                bias.copy_(bcat)
        elif biases is not None and any(b is not None for b in biases):
            raise RuntimeError("linear (mx): either every weight segment has a bias or none")
        else:
            bias = None
    _MX_WEIGHTS[key] = (vers, _WEIGHTS_EPOCH[0], m, bias, [w.detach() for w in weights], [weakref.ref(w) for w in weights])
    return m, bias


def mx_stream_bf16():
    """MX inference mode: the residual stream between the layers (GEMM + residual -> LayerNorm -> next residual) is kept in
    bf16; the fp32 tensors reappear at the encoder's exit (vilbert.py BertModel.forward)."""
    return N.mx_enabled() and not torch.is_grad_enabled() and _os.environ.get("VB_MX_STREAM", "bf16") == "bf16"


def mx_eligible(K, n_out, act=None, drop_p=0.0, want_pre=False, biases=None):
    uniform_bias = biases is None or all(b is None for b in biases) or all(b is not None for b in biases)
    return (N.mx_enabled() and not torch.is_grad_enabled() and K % MX_K_MULTIPLE == 0 and n_out % MX_K_MULTIPLE == 0
            and act in (None, "none", "gelu") and drop_p == 0.0 and not want_pre and uniform_bias)


MX_PAD_MIN_N = 256       # heads at least this wide whose width is not a multiple of 128 run on a zero-padded weight copy


def _mx_pad_eligible(K, n_out, act, drop_p, want_pre, biases, residual, out):
    """A wide head with a ragged width (the 30522-wide MLM decoder, 1601 region classes, 3129 / 1533 answers): served by the
    MX GEMM on a weight copy padded to the next multiple of 128 rows, result = a column slice of the padded output."""
    return (n_out % MX_K_MULTIPLE != 0 and n_out >= MX_PAD_MIN_N and residual is None and out == "f32"
            and mx_eligible(K, (n_out + MX_K_MULTIPLE - 1) // MX_K_MULTIPLE * MX_K_MULTIPLE, act, drop_p, want_pre, biases))


def _mx_of(x, x2, M, K, lead):
    """MX form of the input: the object itself, the codes its producer attached (LayerNorm), or a quantiser pass."""
    if isinstance(x, MxRows):
        return x
    tag = getattr(x, "_vb_mx", None)
    if tag is not None and tag[1] == x._version and tag[0].rows == M and tag[0].K == K and x.is_contiguous():
        return tag[0]
    # (a bf16 tensor without codes: the context of the key-tiled bf16 attention, an expand / index of a hidden state)
    return quantize_rows_mx(x2, lead)


def _linear_fwd_mx(x, x2, M, K, lead, weights, biases, n_out, act, residual, out):
    """out: "f32" -> fp32 tensor, "mx" -> MxRows of the result (no fp32 copy), "bf16" -> bfloat16 tensor."""
    n_real = n_out
    n_out = (n_out + MX_K_MULTIPLE - 1) // MX_K_MULTIPLE * MX_K_MULTIPLE      # (a padded head: see _mx_pad_eligible)
    wm, bias = _mx_weights(weights, biases, n_out if n_out != n_real else None)
    xm = _mx_of(x, x2, M, K, lead)
    a = N.LinearMxArgs()
    a.A, a.lda, a.a_scales, a.a_srows = xm.q.data_ptr(), K, xm.s.data_ptr(), xm.srows
    a.W, a.ldw, a.w_scales, a.w_srows = wm.q.data_ptr(), K, wm.s.data_ptr(), wm.srows
    a.bias = N.dev_f32(bias, "linear bias") if bias is not None else None
    if residual is not None:
        if residual.dtype == torch.bfloat16:            # the MX mode's bf16 residual stream
            a.residual_bf16, a.ldr16 = residual.data_ptr(), n_out
        else:
            a.residual, a.ldr = N.dev_f32(residual, "linear residual"), n_out
    dev = xm.q.device
    if out == "mx":
        y = MxRows(M, n_out, dev, lead)
        a.Cq, a.ldq, a.c_scales, a.c_srows = y.q.data_ptr(), n_out, y.s.data_ptr(), y.srows
    elif out == "bf16":
        y = torch.empty(tuple(lead) + (n_out,), dtype=torch.bfloat16, device=dev)
        a.Cb, a.ldb16 = y.data_ptr(), n_out
    else:
        y = torch.empty(tuple(lead) + (n_out,), dtype=torch.float32, device=dev)
        a.C, a.ldc = y.data_ptr(), n_out
    a.M, a.N, a.K = M, n_out, K
    a.act = N.ACT_CODES[act]
    _timed(lambda: N.check(N.lib().vb_linear_fwd_mx(N.stream_ptr(), ctypes.byref(a)), "vb_linear_fwd_mx"),
           2.0 * M * n_real * K, ("fwd_mx", M, n_real, K, 1))
    return y if n_out == n_real else y[..., :n_real]


def linear_fwd(x, weights, biases, act=None, residual=None, want_preact=False, drop_p=0.0, seed=0,
               want_act_grad=False, pad_cols=False, out="f32"):
    """act(x @ cat(weights).T + cat(biases)) (+ residual).
    out: "f32" (default); in the MX mode (set_gemm_mode("mxfp8"), no grad) an eligible linear may instead return its
    result as "mx" (MxRows: the codes the next linear consumes, no fp32 tensor) or "bf16"; x may be an MxRows.
    want_act_grad: the second return value is act'(pre-activation) instead of the pre-activation (the backward
    of the activation then is one multiply in the epilogue of linear_bwd_input(mul=...)).
    pad_cols: a 2-D result whose width is not a multiple of 4 (the 30522-wide MLM logits) is returned as a view of a
    buffer with the row stride rounded up to 4 floats, so that its rows (and its gradient's) stay 16-byte aligned
    for the backward GEMMs.

    x: [..., K] (the last dim must be contiguous; a uniform row stride is allowed, e.g. the first-token
    view ``h[:, 0]`` of the poolers). weights: list of [n, K] with equal n; biases: list of [n] or None.
    Returns (y [..., nseg*n], preact or None).
    """
    if not isinstance(weights, (list, tuple)):
        weights, biases = [weights], [biases]
    nseg, seg_n, K = len(weights), weights[0].shape[0], weights[0].shape[1]
    if nseg > N.VB_MAX_SEGMENTS:
        raise RuntimeError("linear: at most %d weight segments per launch" % N.VB_MAX_SEGMENTS)
    if x.shape[-1] != K:
        raise RuntimeError("linear: input has %d features, weight expects %d" % (x.shape[-1], K))
    n_out = nseg * seg_n
    want_pre_any = want_preact or want_act_grad
    mx_ok = N.mx_enabled() and (mx_eligible(K, n_out, act, drop_p, want_pre_any, biases)
                                or _mx_pad_eligible(K, n_out, act, drop_p, want_pre_any, biases, residual, out))
    if isinstance(x, MxRows) or (mx_ok and x.is_cuda):
        if not mx_ok:
            raise RuntimeError("linear: an MX input needs an MX-eligible linear (K, N multiples of 128, no dropout)")
        for w in weights:
            if w.shape != (seg_n, K) or not w.is_contiguous():
                raise RuntimeError("linear: weight segments must be contiguous and equally shaped")
        if isinstance(x, MxRows):
            x2, M, lead = None, x.rows, x.lead
        else:
            x2, _, lead = _row_view(x, K)
            M = x2.shape[0]
        if residual is not None:
            residual = _contig(residual)
            if residual.numel() != M * n_out:
                raise RuntimeError("linear: residual shape mismatch")
        return _linear_fwd_mx(x, x2, M, K, lead, weights, biases, n_out, act, residual, out), None
    x2, lda, lead = _row_view(x, K)
    M = x2.shape[0]
    ldc = n_out
    if pad_cols and n_out % 4 != 0 and len(lead) == 1 and residual is None and not (want_preact or want_act_grad) \
            and drop_p == 0.0:
        ldc = (n_out + 3) // 4 * 4
        y = torch.empty(lead + (ldc,), dtype=torch.float32, device=x.device)[:, :n_out]
    else:
        y = torch.empty(lead + (n_out,), dtype=torch.float32, device=x.device)
    pre = torch.empty_like(y) if (want_preact or want_act_grad) else None
    if N.fp8_enabled() and _fp8_eligible(x2, K, n_out, biases):
        for w in weights:
            if w.shape != (seg_n, K) or not w.is_contiguous():
                raise RuntimeError("linear: weight segments must be contiguous and equally shaped")
        if residual is not None:
            residual = _contig(residual)
            if residual.numel() != M * n_out:
                raise RuntimeError("linear: residual shape mismatch")
        _linear_fwd_fp8(x, x2, M, K, weights, biases, n_out, y, ldc, act, residual, pre, want_act_grad, drop_p, seed)
        return y, pre
    a = N.LinearArgs()
    a.M, a.K, a.nseg, a.seg_n = M, K, nseg, seg_n
    a.A, a.lda = N.dev_f32(x2, "linear input"), lda
    for s in range(nseg):
        w = weights[s]
        if w.shape != (seg_n, K) or not w.is_contiguous():
            raise RuntimeError("linear: weight segments must be contiguous and equally shaped")
        a.W[s] = N.dev_f32(w, "linear weight")
        b = biases[s] if biases is not None else None
        a.bias[s] = N.dev_f32(b, "linear bias") if b is not None else None
    a.ldw = K
    a.C, a.ldc = y.data_ptr(), ldc
    if residual is not None:
        residual = _contig(residual)
        if residual.numel() != M * n_out:
            raise RuntimeError("linear: residual shape mismatch")
        a.residual, a.ldr = N.dev_f32(residual, "linear residual"), n_out
    if pre is not None and want_act_grad:
        a.act_grad, a.ldg = pre.data_ptr(), n_out
    elif pre is not None:
        a.preact, a.ldp = pre.data_ptr(), n_out
    a.act = N.ACT_CODES[act]
    a.dropout_p, a.seed = float(drop_p), int(seed)
    _timed(lambda: N.check(N.lib().vb_linear_fwd(N.stream_ptr(), ctypes.byref(a)), "vb_linear_fwd"),
           2.0 * M * n_out * K, ("fwd", M, seg_n, K, nseg))
    return y, pre


def linear_bwd_input(dy, weights, in_features, residual=None, mul=None):
    """dX = (dY @ cat(weights) + residual) * mul for dY [..., nseg*n]; returns [..., in_features].
    residual: a gradient of the same shape arriving over a skip connection (added in the GEMM epilogue);
    mul: elementwise multiplier of the same shape (the saved activation derivative of the producing layer)."""
    N.ensure_deterministic(dy.device)
    nseg, seg_n = len(weights), weights[0].shape[0]
    ldy = nseg * seg_n
    if _row_strided(dy) and dy.shape[1] == ldy:
        ldy = dy.stride(0)            # rows of a padded buffer (pad_cols): used in place
    else:
        dy = _contig(dy)
    M = dy.numel() // (nseg * seg_n)
    dx = torch.empty(tuple(dy.shape[:-1]) + (in_features,), dtype=torch.float32, device=dy.device)
    a = N.LinearBwdInputArgs()
    a.M, a.K, a.nseg, a.seg_n = M, in_features, nseg, seg_n
    a.dY, a.ldy = N.dev_f32(dy, "linear grad_output"), ldy
    for s in range(nseg):
        a.W[s] = N.dev_f32(weights[s], "linear weight")
    a.ldw = in_features
    a.dX, a.ldx = dx.data_ptr(), in_features
    a.accumulate = 0
    if residual is not None:
        residual = _contig(residual)
        if residual.numel() != M * in_features:
            raise RuntimeError("linear_bwd_input: residual shape mismatch")
        a.residual, a.ldr = N.dev_f32(residual, "linear residual grad"), in_features
    if mul is not None:
        mul = _contig(mul)
        if mul.numel() != M * in_features:
            raise RuntimeError("linear_bwd_input: multiplier shape mismatch")
        a.mul, a.ldm = N.dev_f32(mul, "linear activation derivative"), in_features
    _timed(lambda: N.check(N.lib().vb_linear_bwd_input(N.stream_ptr(), ctypes.byref(a)), "vb_linear_bwd_input"),
           2.0 * M * nseg * seg_n * in_features, ("dgrad", M, seg_n, in_features, nseg))
    return dx


def linear_bwd_weight(dy, x, nseg, seg_n, want_bias, dw_out=None, db_out=None):
    """Per segment: dW_s = dY[:, s]^T @ X and db_s = colsum(dY[:, s]) in one call.
    Returns (list dW, list db-or-None). The split-K kernel ADDS into its targets with atomics. dw_out / db_out:
    per-segment target tensors (gradient-arena slices: zero-filled once per backward pass, or holding an earlier
    contribution) or None; the targets not given are slices of ONE zero-filled buffer allocated here (one fill
    launch instead of two per segment)."""
    N.ensure_deterministic(dy.device)
    ldy = nseg * seg_n
    if _row_strided(dy) and dy.shape[1] == ldy:
        ldy = dy.stride(0)
    else:
        dy = _contig(dy)
    K = x.shape[-1]
    x2, ldx, _ = _row_view(x, K)
    M = x2.shape[0]
    a = N.LinearBwdWeightArgs()
    a.M, a.K, a.nseg, a.seg_n = M, K, nseg, seg_n
    a.dY, a.ldy = N.dev_f32(dy, "linear grad_output"), ldy
    a.X, a.ldx = N.dev_f32(x2, "linear input"), ldx
    a.ldw, a.accumulate = K, 1
    wsz, bsz = (seg_n * K + 3) // 4 * 4, (seg_n + 3) // 4 * 4          # every slice stays 16-byte aligned
    dw_out = dw_out if dw_out is not None else [None] * nseg
    db_out = db_out if db_out is not None else [None] * nseg
    need = sum(wsz for s in range(nseg) if dw_out[s] is None) + \
        sum(bsz for s in range(nseg) if want_bias[s] and db_out[s] is None)
    flat = torch.zeros(need, dtype=torch.float32, device=dy.device) if need else None
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
        a.dW[s] = dw.data_ptr()
        a.dbias[s] = db.data_ptr() if db is not None else None
        dws.append(dw)
        dbs.append(db)
    _timed(lambda: N.check(N.lib().vb_linear_bwd_weight(N.stream_ptr(), ctypes.byref(a)), "vb_linear_bwd_weight"),
           2.0 * M * nseg * seg_n * K, ("wgrad", M, seg_n, K, nseg))
    return dws, dbs


def act_bwd(dy, preact, act):
    dy, preact = _contig(dy), _contig(preact)
    dx = torch.empty_like(dy)
    N.check(N.lib().vb_act_bwd(N.stream_ptr(), dy.numel(), N.ACT_CODES[act], N.dev_f32(dy, "grad_output"),
                               N.dev_f32(preact, "preactivation"), dx.data_ptr()), "vb_act_bwd")
    return dx


def dropout(x, p, seed, residual=None):
    """x * keep(seed, i) / (1 - p) (+ residual); on a gradient with the same seed it is the backward."""
    x = _contig(x)
    if residual is not None:
        residual = _contig(residual)
        if residual.shape != x.shape:
            raise RuntimeError("dropout: residual shape mismatch")
    y = torch.empty_like(x)
    N.check(N.lib().vb_dropout(N.stream_ptr(), x.numel(), N.dev_f32(x, "dropout input"),
                               N.dev_f32(residual, "dropout residual"), y.data_ptr(), p, seed), "vb_dropout")
    return y


def layernorm_fwd(x, gamma, beta, eps, x2=None, want_stats=False):
    x = _contig(x)
    rows, cols = _rows(x)
    y = torch.empty_like(x)
    mean = rstd = None
    if x.dtype == torch.bfloat16:
        # the MX mode's bf16 residual stream: bf16 pre-LayerNorm sum in, bf16 row + MX codes out
        if not (N.mx_enabled() and not want_stats and x2 is None and cols % MX_K_MULTIPLE == 0 and x.is_cuda):
            raise RuntimeError("layernorm: a bfloat16 input is only served in the MX inference mode")
        m = MxRows(rows, cols, x.device, tuple(x.shape[:-1]))
        N.check(N.lib().vb_layernorm_fwd_mx16(
            N.stream_ptr(), rows, cols, x.data_ptr(), N.dev_f32(gamma, "layernorm weight"), N.dev_f32(beta, "layernorm bias"),
            eps, y.data_ptr(), m.q.data_ptr(), cols, m.s.data_ptr(), m.srows), "vb_layernorm_fwd_mx16")
        y._vb_mx = (m, y._version)
        return y, None, None
    if N.mx_enabled() and not want_stats and not torch.is_grad_enabled() and cols % MX_K_MULTIPLE == 0 and x.is_cuda:
        # inference in the MX mode: the LayerNorm kernel also emits its output rows as MX codes + scale words
        if x2 is not None:
            x2 = _contig(x2)
            if x2.shape != x.shape:
                raise RuntimeError("layernorm: x2 shape mismatch")
        m = MxRows(rows, cols, x.device, tuple(x.shape[:-1]))
        N.check(N.lib().vb_layernorm_fwd_mx(
            N.stream_ptr(), rows, cols, N.dev_f32(x, "layernorm input"), N.dev_f32(x2, "layernorm x2"),
            N.dev_f32(gamma, "layernorm weight"), N.dev_f32(beta, "layernorm bias"), eps, y.data_ptr(), m.q.data_ptr(),
            cols, m.s.data_ptr(), m.srows), "vb_layernorm_fwd_mx")
        y._vb_mx = (m, y._version)
        return y, None, None
    if N.fp8_enabled() and not want_stats and not torch.is_grad_enabled() and cols % FP8_K_MULTIPLE == 0 and x.is_cuda:
        # inference in fp8 mode: the LayerNorm kernel also emits the e4m3 codes of its output rows; the linears that
        # consume this very tensor (same object, not modified since) skip their quantiser pass
        if x2 is not None:
            x2 = _contig(x2)
            if x2.shape != x.shape:
                raise RuntimeError("layernorm: x2 shape mismatch")
        q = torch.empty((rows, cols), dtype=torch.uint8, device=x.device)
        sc = torch.empty((rows,), dtype=torch.float32, device=x.device)
        N.check(N.lib().vb_layernorm_fwd_fp8(
            N.stream_ptr(), rows, cols, N.dev_f32(x, "layernorm input"), N.dev_f32(x2, "layernorm x2"),
            N.dev_f32(gamma, "layernorm weight"), N.dev_f32(beta, "layernorm bias"), eps, y.data_ptr(), q.data_ptr(),
            cols, sc.data_ptr()), "vb_layernorm_fwd_fp8")
        y._vb_fp8 = (q, sc, y._version)
        return y, None, None
    if want_stats:
        mean = torch.empty(rows, dtype=torch.float32, device=x.device)
        rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
    if x2 is not None:
        x2 = _contig(x2)
        if x2.shape != x.shape:
            raise RuntimeError("layernorm: x2 shape mismatch")
    N.check(N.lib().vb_layernorm_fwd(
        N.stream_ptr(), rows, cols, N.dev_f32(x, "layernorm input"), N.dev_f32(x2, "layernorm x2"),
        N.dev_f32(gamma, "layernorm weight"), N.dev_f32(beta, "layernorm bias"), eps, y.data_ptr(),
        mean.data_ptr() if want_stats else None, rstd.data_ptr() if want_stats else None), "vb_layernorm_fwd")
    return y, mean, rstd


def layernorm_bwd(dy, x, mean, rstd, gamma, dgamma=None, dbeta=None, drop=None):
    """Returns (dx, dgamma, dbeta); x is the normalised input (the sum when the forward had x2). dgamma / dbeta:
    optional [cols] targets (OVERWRITTEN - the column reduction is a deterministic two-stage sum).
    drop = (p, seed): also returns, as a 4th value, dx with that dropout mask applied (the mask of the dense layer in
    front of the LayerNorm), produced in the same pass."""
    dy, x = _contig(dy), _contig(x)
    rows, cols = _rows(x)
    dx = torch.empty_like(x)
    if drop is not None and drop[0] > 0.0 and cols <= 4096:
        dgamma = dgamma if dgamma is not None else torch.empty(cols, dtype=torch.float32, device=x.device)
        dbeta = dbeta if dbeta is not None else torch.empty(cols, dtype=torch.float32, device=x.device)
        ws = torch.empty(N.lib().vb_layernorm_bwd_workspace(rows, cols), dtype=torch.float32, device=x.device)
        dxd = torch.empty_like(x)
        N.check(N.lib().vb_layernorm_bwd_drop(
            N.stream_ptr(), rows, cols, N.dev_f32(dy, "layernorm grad_output"), N.dev_f32(x, "layernorm input"),
            N.dev_f32(mean, "layernorm mean"), N.dev_f32(rstd, "layernorm rstd"), N.dev_f32(gamma, "layernorm weight"),
            dx.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(), ws.data_ptr(), dxd.data_ptr(), drop[0], drop[1]),
            "vb_layernorm_bwd_drop")
        return dx, dgamma, dbeta, dxd
    dgamma = dgamma if dgamma is not None else torch.empty(cols, dtype=torch.float32, device=x.device)
    dbeta = dbeta if dbeta is not None else torch.empty(cols, dtype=torch.float32, device=x.device)
    ws = torch.empty(N.lib().vb_layernorm_bwd_workspace(rows, cols), dtype=torch.float32, device=x.device)
    N.check(N.lib().vb_layernorm_bwd(
        N.stream_ptr(), rows, cols, N.dev_f32(dy, "layernorm grad_output"), N.dev_f32(x, "layernorm input"),
        N.dev_f32(mean, "layernorm mean"), N.dev_f32(rstd, "layernorm rstd"), N.dev_f32(gamma, "layernorm weight"),
        dx.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(), ws.data_ptr()), "vb_layernorm_bwd")
    return dx, dgamma, dbeta


def text_embed_ln_fwd(ids, seg, word, pos, typ, gamma, beta, eps, task_ids=None, task_emb=None,
                      want_stats=False):
    """Returns (out, mean, rstd, presum); the last three are None unless want_stats."""
    ids, seg = _contig(ids), _contig(seg)
    B, T = ids.shape
    H = word.shape[1]
    if T > pos.shape[0]:
        raise RuntimeError("sequence length %d exceeds max_position_embeddings %d" % (T, pos.shape[0]))
    n_out = T + (1 if task_ids is not None else 0)
    out = torch.empty(B, n_out, H, dtype=torch.float32, device=word.device)
    mean = rstd = presum = None
    if want_stats:
        mean = torch.empty(B * n_out, dtype=torch.float32, device=word.device)
        rstd = torch.empty(B * n_out, dtype=torch.float32, device=word.device)
        presum = torch.empty_like(out)
    if task_ids is not None:
        task_ids = _contig(task_ids.view(-1))
        if task_ids.numel() != B:
            raise RuntimeError("task_ids must hold one id per sample")
    N.check(N.lib().vb_text_embed_ln_fwd(
        N.stream_ptr(), B, T, H, word.shape[0], typ.shape[0], task_emb.shape[0] if task_emb is not None else 0,
        N.dev_i64(ids, "input_ids"), N.dev_i64(seg, "token_type_ids"), 0, N.dev_f32(word, "word_embeddings"), N.dev_f32(pos, "position_embeddings"),
        N.dev_f32(typ, "token_type_embeddings"), N.dev_i64(task_ids, "task_ids"),
        N.dev_f32(task_emb, "task_embeddings"), N.dev_f32(gamma, "LayerNorm.weight"),
        N.dev_f32(beta, "LayerNorm.bias"), eps, out.data_ptr(),
        mean.data_ptr() if want_stats else None, rstd.data_ptr() if want_stats else None,
        presum.data_ptr() if want_stats else None), "vb_text_embed_ln_fwd")
    return out, mean, rstd, presum


def text_embed_bwd(dx, ids, seg, task_ids, word_shape, pos_shape, type_shape, task_shape, out=None):
    """Scatter-add dx (gradient of the pre-LayerNorm sum) into zero tables (fresh ones, or the accumulating targets
    `out` = [dword, dpos, dtype, dtask] with None for the ones to allocate)."""
    dx = _contig(dx)
    ids, seg = _contig(ids), _contig(seg)
    B, T = ids.shape
    dev = dx.device
    out = out if out is not None else [None] * 4
    dword = out[0] if out[0] is not None else torch.zeros(word_shape, dtype=torch.float32, device=dev)
    dpos = out[1] if out[1] is not None else torch.zeros(pos_shape, dtype=torch.float32, device=dev)
    dtype = out[2] if out[2] is not None else torch.zeros(type_shape, dtype=torch.float32, device=dev)
    dtask = None
    if task_ids is not None:
        dtask = out[3] if out[3] is not None else torch.zeros(task_shape, dtype=torch.float32, device=dev)
    if task_ids is not None:
        task_ids = _contig(task_ids.view(-1))
    N.check(N.lib().vb_text_embed_bwd(
        N.stream_ptr(), B, T, word_shape[1], word_shape[0], type_shape[0], task_shape[0] if task_shape else 0,
        N.dev_i64(ids, "input_ids"), N.dev_i64(seg, "token_type_ids"),
        N.dev_i64(task_ids, "task_ids"), N.dev_f32(dx, "embedding grad"), dword.data_ptr(), dpos.data_ptr(),
        dtype.data_ptr(), dtask.data_ptr() if dtask is not None else None), "vb_text_embed_bwd")
    return dword, dpos, dtype, dtask


def image_embed_ln_fwd(feat_proj, loc, w_loc, b_loc, gamma, beta, eps, want_stats=False):
    """Returns (out, mean, rstd, presum)."""
    feat_proj, loc = _contig(feat_proj), _contig(loc)
    rows, H = _rows(feat_proj)
    if loc.shape[-1] != 5 or loc.numel() != rows * 5:
        raise RuntimeError("image_loc must be [..., 5] matching the features")
    out = torch.empty_like(feat_proj)
    mean = rstd = presum = None
    if want_stats:
        mean = torch.empty(rows, dtype=torch.float32, device=out.device)
        rstd = torch.empty(rows, dtype=torch.float32, device=out.device)
        presum = torch.empty_like(out)
    N.check(N.lib().vb_image_embed_ln_fwd(
        N.stream_ptr(), rows, H, N.dev_f32(feat_proj, "image projection"), N.dev_f32(loc, "image_loc"),
        N.dev_f32(w_loc, "image_location_embeddings.weight"), N.dev_f32(b_loc, "image_location_embeddings.bias"),
        N.dev_f32(gamma, "LayerNorm.weight"), N.dev_f32(beta, "LayerNorm.bias"), eps, out.data_ptr(),
        mean.data_ptr() if want_stats else None, rstd.data_ptr() if want_stats else None,
        presum.data_ptr() if want_stats else None), "vb_image_embed_ln_fwd")
    return out, mean, rstd, presum


def additive_mask(mask):
    """(1 - mask) * -10000 as fp32, same shape (reference vilbert.py:1341-1362)."""
    mask = _contig(mask)
    if mask.dtype == torch.float32:
        is_f32 = 1
    elif mask.dtype == torch.int64:
        is_f32 = 0
    else:
        mask, is_f32 = mask.to(torch.int64), 0
    if not mask.is_cuda:
        raise RuntimeError("attention mask must live on a HIP device - no CPU fallback")
    out = torch.empty(mask.shape, dtype=torch.float32, device=mask.device)
    N.check(N.lib().vb_additive_mask(N.stream_ptr(), mask.numel(), mask.data_ptr(), is_f32, out.data_ptr()),
            "vb_additive_mask")
    return out


def _attn_args(q, k, v, mask_add, heads, drop_p, seed):
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
    a.Q, a.ldq = N.dev_f32(q, "attention q"), q.stride(1)
    a.K, a.ldk = N.dev_f32(k, "attention k"), k.stride(1)
    a.V, a.ldv = N.dev_f32(v, "attention v"), v.stride(1)
    keep = []
    if mask_add is not None:
        mask_add = _contig(mask_add)
        if mask_add.numel() != Bk * Sk:
            raise RuntimeError("attention: mask must hold %d x %d values" % (Bk, Sk))
        a.mask_add = N.dev_f32(mask_add, "attention mask")
        keep.append(mask_add)
    a.scale = 1.0 / math.sqrt(d)
    a.dropout_p, a.seed = float(drop_p), int(seed)
    return a, keep, (B, Sq, Sk, H)


MAX_KEYS = 320          # include/vilbert_hip.h VB_MAX_KEYS: longest key sequence ONE attention launch handles


def merge_attention_chunks(outs, lses, heads):
    """Contexts of the SAME queries over disjoint key chunks -> the context over the union of the keys.
    outs[c]: [B, Sq, H], softmax-normalised over chunk c's keys only; lses[c]: [B, heads, Sq] = log sum_k exp(score) of
    that chunk. softmax over the union = sum_c w_c softmax_c with w_c = exp(lse_c - logsumexp_c lse_c) - exact, the
    identity flash attention tiles by. Returns (ctx [B, Sq, H], lse over all keys [B, heads, Sq])."""
    lse = torch.stack(lses)                                      # [C, B, heads, Sq]
    total = torch.logsumexp(lse, dim=0)
    w = torch.exp(lse - total)
    B, Sq, H = outs[0].shape
    d = H // heads
    out = None
    for c, o in enumerate(outs):
        term = o.reshape(B, Sq, heads, d) * w[c].permute(0, 2, 1).unsqueeze(-1)
        out = term if out is None else out + term
    return out.reshape(B, Sq, H), total


def _key_chunks(Sk):
    n_chunks = (Sk + MAX_KEYS - 1) // MAX_KEYS
    step = (Sk + n_chunks - 1) // n_chunks
    return [(c0, min(Sk, c0 + step)) for c0 in range(0, Sk, step)]


def _chunk_seed(seed, c):
    """Dropout seed of key chunk c (the keep mask is a function of (seed, element index inside the launch): every chunk needs
    its own seed, and backward must find it again from the node's one saved seed)."""
    return (int(seed) + c * 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF if seed else 0


def _attention_fwd_long(q, k, v, mask_add, heads, want_lse, drop_p=0.0, seed=0):
    """More than MAX_KEYS keys (stacked retrieval options, in_batch_pairs): one launch per chunk of <= MAX_KEYS keys, merged
    with merge_attention_chunks (a few small torch ops - this is the rare path; the shapes of every task in
    vilbert_tasks.yml fit one launch). Dropout acts on the probabilities element by element, so the mask of a chunk
    commutes with the chunk's weight in the merge: dropout(w_c P_c) = w_c dropout(P_c)."""
    Bk, Sk, _ = k.shape
    if mask_add is not None:
        mask_add = _contig(mask_add).reshape(Bk, Sk)
    outs, lses = [], []
    for c, (c0, c1) in enumerate(_key_chunks(Sk)):
        m = mask_add[:, c0:c1].contiguous() if mask_add is not None else None
        o, _, l = attention_fwd(q, k[:, c0:c1].contiguous(), v[:, c0:c1].contiguous(), m, heads, False, True, drop_p,
                                _chunk_seed(seed, c))
        outs.append(o)
        lses.append(l)
    out, total = merge_attention_chunks(outs, lses, heads)
    return out, None, (total if want_lse else None)


def _attention_bwd_long(d_out, q, k, v, mask_add, heads, lse, dq, dk, dv, drop_p, seed):
    """Backward of _attention_fwd_long (round 6; reference vilbert.py:1008-1040 under autograd). `lse` is the log-sum-exp over
    ALL keys, so P = exp(S - lse) inside a chunk launch is the chunk's part of the full softmax; what couples the chunks is
    D = rowsum(P dP) over all keys: pass A adds every chunk's share into one buffer (VB_DVEC_ACCUMULATE), pass B hands the
    complete D to every chunk (VB_DVEC_GIVEN) for its dK / dV and its share of dQ."""
    B, Sq, H = d_out.shape
    Bk, Sk, _ = k.shape
    if mask_add is not None:
        mask_add = _contig(mask_add).reshape(Bk, Sk)
    chunks = _key_chunks(Sk)
    parts = [(k[:, c0:c1].contiguous(), v[:, c0:c1].contiguous(),
              mask_add[:, c0:c1].contiguous() if mask_add is not None else None) for c0, c1 in chunks]
    dvec = torch.zeros(B, heads, Sq, dtype=torch.float32, device=q.device)
    dq_c = torch.empty(B, Sq, H, dtype=torch.float32, device=q.device)
    for c, (kc, vc, mc) in enumerate(parts):
        dkv = torch.empty(2, kc.shape[0], kc.shape[1], H, dtype=torch.float32, device=q.device)   # (not written in this mode)
        attention_bwd(d_out, q, kc, vc, mc, heads, lse, dq_c, dkv[0], dkv[1], drop_p, _chunk_seed(seed, c), dvec=dvec,
                      dvec_mode=N.DVEC_ACCUMULATE)
    total = None
    for c, ((c0, c1), (kc, vc, mc)) in enumerate(zip(chunks, parts)):
        dkv = torch.empty(2, kc.shape[0], kc.shape[1], H, dtype=torch.float32, device=q.device)
        dq_c = torch.empty(B, Sq, H, dtype=torch.float32, device=q.device)
        attention_bwd(d_out, q, kc, vc, mc, heads, lse, dq_c, dkv[0], dkv[1], drop_p, _chunk_seed(seed, c), dvec=dvec,
                      dvec_mode=N.DVEC_GIVEN)
        dk[:, c0:c1].copy_(dkv[0])
        dv[:, c0:c1].copy_(dkv[1])
        total = dq_c if total is None else total.add_(dq_c)
    dq.copy_(total)


def attention_fwd(q, k, v, mask_add, heads, want_probs=False, want_lse=False, drop_p=0.0, seed=0):
    """q: [Bq, Sq, H*] view, k/v: [Bk, Sk, H*] views (last dim contiguous, uniform row stride, e.g. column
    slices of a fused [q|k|v] projection); mask_add: [Bk, 1, 1, Sk] or [Bk, Sk] fp32 additive, or None.
    Bq / Bk may be 1 against a larger batch (broadcast). Returns (ctx [B, Sq, H], probs|None, lse|None).
    More than MAX_KEYS keys: served chunk by chunk (no probabilities tensor there)."""
    if k.shape[1] > MAX_KEYS:
        if want_probs:
            raise RuntimeError("attention: %d keys - more than %d keys are served without the probabilities tensor only"
                               % (k.shape[1], MAX_KEYS))
        return _attention_fwd_long(q, k, v, mask_add, heads, want_lse, drop_p, seed)
    a, keep, (B, Sq, Sk, H) = _attn_args(q, k, v, mask_add, heads, drop_p, seed)
    out = torch.empty(B, Sq, H, dtype=torch.float32, device=q.device)
    probs = torch.empty(B, heads, Sq, Sk, dtype=torch.float32, device=q.device) if want_probs else None
    lse = torch.empty(B, heads, Sq, dtype=torch.float32, device=q.device) if want_lse else None
    a.O, a.ldo = out.data_ptr(), H
    a.probs = probs.data_ptr() if want_probs else None
    a.lse = lse.data_ptr() if want_lse else None
    N.check(N.lib().vb_attention_fwd(N.stream_ptr(), ctypes.byref(a)), "vb_attention_fwd")
    return out, probs, lse


MX_ATTN_MAX_ROWS = 48


def mx_attention_ok(n_q, n_k, head_dim, drop_p=0.0, other=False):
    """The MX path's attention serves this call on a bf16 q | k | v projection: up to 48 queries / keys on
    csrc/attention_mx.hip (MX context out), longer rows - the task shapes: 101 / 200 regions, up to MAX_KEYS - on the
    key-tiled bf16-MFMA kernel of the bf16 training path (csrc/attention.hip built with bf16 tensors), whose bf16 context
    the output projection quantises (vb_quantize_rows_mx_bf16)."""
    return (N.mx_enabled() and not torch.is_grad_enabled() and n_q <= MAX_KEYS and n_k <= MAX_KEYS
            and head_dim in (64, 128) and drop_p == 0.0 and not other)


def attention_fwd_mx_any(q, k, v, mask_add, heads):
    """attention_fwd_mx where its kernel fits (MxRows), else the bf16 kernel (bfloat16 context tensor)."""
    if q.shape[1] <= MX_ATTN_MAX_ROWS and k.shape[1] <= MX_ATTN_MAX_ROWS:
        return attention_fwd_mx(q, k, v, mask_add, heads)
    from . import ops16
    return ops16.attention_fwd(q, k, v, mask_add, heads, False, 0.0, 0)[0]


def attention_fwd_mx(q, k, v, mask_add, heads):
    """q [Bq, Sq, H], k / v [Bk, Sk, H]: bfloat16 row-strided views (column slices of a fused projection); returns the
    context [B, Sq, H] as MxRows (the codes the output projection consumes)."""
    Bq, Sq, H = q.shape
    Bk, Sk, _ = k.shape
    B = max(Bq, Bk)
    d = H // heads
    for t, nm in ((q, "q"), (k, "k"), (v, "v")):
        if t.dtype != torch.bfloat16 or not t.is_cuda:
            raise RuntimeError("attention (mx): %s must be a bfloat16 tensor on a HIP device" % nm)
        if t.stride(2) != 1 or (t.shape[0] > 1 and t.stride(0) != t.shape[1] * t.stride(1)):
            raise RuntimeError("attention (mx): %s must be a row-strided view" % nm)
    a = N.AttentionMxArgs()
    a.batch, a.heads, a.head_dim, a.n_q, a.n_k = B, heads, d, Sq, Sk
    a.q_batch, a.kv_batch = Bq, Bk
    a.Q, a.ldq = q.data_ptr(), q.stride(1)
    a.K, a.ldk = k.data_ptr(), k.stride(1)
    a.V, a.ldv = v.data_ptr(), v.stride(1)
    if mask_add is not None:
        mask_add = _contig(mask_add)
        if mask_add.numel() != Bk * Sk:
            raise RuntimeError("attention: mask must hold %d x %d values" % (Bk, Sk))
        a.mask_add = N.dev_f32(mask_add, "attention mask")
    a.scale = 1.0 / math.sqrt(d)
    out = MxRows(B * Sq, H, q.device, (B, Sq))
    a.Oq, a.ldo, a.o_scales, a.o_srows = out.q.data_ptr(), H, out.s.data_ptr(), out.srows
    N.check(N.lib().vb_attention_fwd_mx(N.stream_ptr(), ctypes.byref(a)), "vb_attention_fwd_mx")
    return out


def attention_bwd(d_out, q, k, v, mask_add, heads, lse, dq, dk, dv, drop_p=0.0, seed=0, dvec=None, dvec_mode=0):
    """Writes dq / dk / dv (row-strided views, e.g. column slices of a fused gradient buffer) in place.
    dvec / dvec_mode: the two passes over key chunks of a longer sequence (_attention_bwd_long)."""
    if k.shape[1] > MAX_KEYS:
        return _attention_bwd_long(_contig(d_out), q, k, v, mask_add, heads, lse, dq, dk, dv, drop_p, seed)
    a, keep, (B, Sq, Sk, H) = _attn_args(q, k, v, mask_add, heads, drop_p, seed)
    d_out = _contig(d_out)
    a.lse = N.dev_f32(lse, "attention lse")
    g = N.AttentionGrads()
    g.dO, g.lddo = N.dev_f32(d_out, "attention grad_output"), H
    g.dQ, g.lddq = N.dev_f32(dq, "attention dq"), dq.stride(1)
    g.dK, g.lddk = N.dev_f32(dk, "attention dk"), dk.stride(1)
    g.dV, g.lddv = N.dev_f32(dv, "attention dv"), dv.stride(1)
    if dvec is None:
        dvec = torch.empty(B, heads, Sq, dtype=torch.float32, device=q.device)
    g.dvec, g.dvec_mode = N.dev_f32(dvec, "attention dvec"), dvec_mode
    N.check(N.lib().vb_attention_bwd(N.stream_ptr(), ctypes.byref(a), ctypes.byref(g)), "vb_attention_bwd")


def xent_fwd(logits, labels, ignore_index):
    """Mean cross-entropy over the rows whose label != ignore_index. logits [rows, n] fp32, labels [rows]
    int64. Returns (loss [1]-element 0-dim view, lse [rows], count [1])."""
    if logits.dim() != 2 or labels.dim() != 1 or labels.shape[0] != logits.shape[0]:
        raise RuntimeError("cross_entropy: expected logits [rows, n] and labels [rows]")
    if not _row_strided(logits):
        logits = _contig(logits)
    labels = _contig(labels)
    rows, n = logits.shape
    ld = logits.stride(0) if rows > 1 else n
    dev = logits.device
    row_loss = torch.empty(rows, dtype=torch.float32, device=dev)
    lse = torch.empty(rows, dtype=torch.float32, device=dev)
    out = torch.empty(2, dtype=torch.float32, device=dev)    # {loss, count}
    N.check(N.lib().vb_xent_fwd(N.stream_ptr(), rows, n, N.dev_f32(logits, "cross_entropy logits"), ld,
                                N.dev_i64(labels, "cross_entropy labels"), ignore_index, row_loss.data_ptr(),
                                lse.data_ptr(), out.data_ptr(), out.data_ptr() + 4), "vb_xent_fwd")
    return out[0], lse, out[1:]


def xent_bwd(grad_loss, logits, labels, ignore_index, lse, count):
    if not _row_strided(logits):
        logits = _contig(logits)
    labels = _contig(labels)
    rows, n = logits.shape
    ld = logits.stride(0) if rows > 1 else n
    grad_loss = _contig(grad_loss).reshape(1)
    # the gradient keeps the (padded) row stride of the logits
    d = torch.empty(rows, ld, dtype=torch.float32, device=logits.device)[:, :n]
    N.check(N.lib().vb_xent_bwd(N.stream_ptr(), rows, n, N.dev_f32(logits, "cross_entropy logits"), ld,
                                N.dev_i64(labels, "cross_entropy labels"), ignore_index,
                                N.dev_f32(lse, "cross_entropy lse"), N.dev_f32(grad_loss, "cross_entropy grad"),
                                N.dev_f32(count, "cross_entropy count"), d.data_ptr(), ld), "vb_xent_bwd")
    return d


def _divisor(divisor):
    """(host float, device pointer or None) of a KL divisor given as a number or as a 1-element device tensor."""
    if torch.is_tensor(divisor):
        d = _contig(divisor.reshape(1))
        return 0.0, N.dev_f32(d, "kl_div divisor"), d
    return float(divisor), None, None


def kl_fwd(scores, target, divisor):
    """sum(KLDiv(log_softmax(scores, 1), target)) / divisor; scores / target [rows, n] fp32; divisor: a number or
    a 1-element fp32 DEVICE tensor (no host sync). Returns (loss 0-dim, lse [rows], tsum [rows])."""
    if scores.dim() != 2 or scores.shape != target.shape:
        raise RuntimeError("kl_div: expected scores and target of the same [rows, n] shape")
    scores, target = _contig(scores), _contig(target)
    rows, n = scores.shape
    dev = scores.device
    row_loss = torch.empty(rows, dtype=torch.float32, device=dev)
    lse = torch.empty(rows, dtype=torch.float32, device=dev)
    tsum = torch.empty(rows, dtype=torch.float32, device=dev)
    out = torch.empty(2, dtype=torch.float32, device=dev)
    dh, dd, _keep = _divisor(divisor)
    N.check(N.lib().vb_kl_fwd(N.stream_ptr(), rows, n, N.dev_f32(scores, "kl_div scores"), n,
                              N.dev_f32(target, "kl_div target"), n, dh, row_loss.data_ptr(),
                              lse.data_ptr(), tsum.data_ptr(), out.data_ptr(), dd), "vb_kl_fwd")
    return out[0], lse, tsum


def kl_bwd(grad_loss, scores, target, lse, tsum, divisor):
    scores, target = _contig(scores), _contig(target)
    rows, n = scores.shape
    grad_loss = _contig(grad_loss).reshape(1)
    d = torch.empty_like(scores)
    dh, dd, _keep = _divisor(divisor)
    N.check(N.lib().vb_kl_bwd(N.stream_ptr(), rows, n, N.dev_f32(scores, "kl_div scores"), n,
                              N.dev_f32(target, "kl_div target"), n, N.dev_f32(lse, "kl_div lse"),
                              N.dev_f32(tsum, "kl_div tsum"), N.dev_f32(grad_loss, "kl_div grad"), dh,
                              d.data_ptr(), n, dd), "vb_kl_bwd")
    return d
