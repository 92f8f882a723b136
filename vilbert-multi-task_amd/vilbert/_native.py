"""ctypes binding of libvilbert_hip.so (C ABI: include/vilbert_hip.h).

The library is the product's only compute path: there is NO CPU or eager-PyTorch fallback.
If the shared object is missing, or a tensor is not on a HIP device, the call raises.
PyTorch is used for device memory (``torch.empty``) and the current stream only.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(os.path.dirname(_HERE), "csrc", "libvilbert_hip.so")

VB_MAX_SEGMENTS = 4
ACT_NONE, ACT_GELU, ACT_RELU = 0, 1, 2
ACT_SWISH = 3
ACT_CODES = {None: ACT_NONE, "none": ACT_NONE, "gelu": ACT_GELU, "relu": ACT_RELU, "swish": ACT_SWISH}

_c_f32p = ctypes.c_void_p  # device pointers are passed as plain addresses


class LinearArgs(ctypes.Structure):
    """vb_linear_args"""
    _fields_ = [
        ("M", ctypes.c_int32), ("K", ctypes.c_int32),
        ("nseg", ctypes.c_int32), ("seg_n", ctypes.c_int32),
        ("A", _c_f32p), ("lda", ctypes.c_int64),
        ("W", _c_f32p * VB_MAX_SEGMENTS), ("ldw", ctypes.c_int64),
        ("bias", _c_f32p * VB_MAX_SEGMENTS),
        ("C", _c_f32p), ("ldc", ctypes.c_int64),
        ("residual", _c_f32p), ("ldr", ctypes.c_int64),
        ("preact", _c_f32p), ("ldp", ctypes.c_int64),
        ("act_grad", _c_f32p), ("ldg", ctypes.c_int64),
        ("act", ctypes.c_int32),
        ("dropout_p", ctypes.c_float),
        ("seed", ctypes.c_uint64),
    ]


class LinearFp8Args(ctypes.Structure):
    """vb_linear_fp8_args"""
    _fields_ = [
        ("A", ctypes.c_void_p), ("lda", ctypes.c_int64),
        ("a_scale", _c_f32p),
        ("W", ctypes.c_void_p), ("ldw", ctypes.c_int64),
        ("w_scale", _c_f32p),
        ("bias", _c_f32p),
        ("C", _c_f32p), ("ldc", ctypes.c_int64),
        ("residual", _c_f32p), ("ldr", ctypes.c_int64),
        ("preact", _c_f32p), ("ldp", ctypes.c_int64),
        ("act_grad", _c_f32p), ("ldg", ctypes.c_int64),
        ("M", ctypes.c_int32), ("N", ctypes.c_int32), ("K", ctypes.c_int32),
        ("act", ctypes.c_int32),
        ("dropout_p", ctypes.c_float),
        ("seed", ctypes.c_uint64),
    ]


class LinearMxArgs(ctypes.Structure):
    """vb_linear_mx_args"""
    _fields_ = [
        ("A", ctypes.c_void_p), ("lda", ctypes.c_int64),
        ("a_scales", ctypes.c_void_p), ("a_srows", ctypes.c_int64),
        ("W", ctypes.c_void_p), ("ldw", ctypes.c_int64),
        ("w_scales", ctypes.c_void_p), ("w_srows", ctypes.c_int64),
        ("bias", _c_f32p),
        ("residual", _c_f32p), ("ldr", ctypes.c_int64),
        ("residual_bf16", ctypes.c_void_p), ("ldr16", ctypes.c_int64),
        ("C", _c_f32p), ("ldc", ctypes.c_int64),
        ("Cb", ctypes.c_void_p), ("ldb16", ctypes.c_int64),
        ("Cq", ctypes.c_void_p), ("ldq", ctypes.c_int64),
        ("c_scales", ctypes.c_void_p), ("c_srows", ctypes.c_int64),
        ("M", ctypes.c_int64), ("N", ctypes.c_int64), ("K", ctypes.c_int64),
        ("act", ctypes.c_int32),
    ]


class LinearBf16Args(ctypes.Structure):
    """vb_linear_bf16_args"""
    _fields_ = [
        ("A", ctypes.c_void_p), ("lda", ctypes.c_int64),
        ("W", ctypes.c_void_p), ("ldw", ctypes.c_int64),
        ("bias", _c_f32p * VB_MAX_SEGMENTS), ("bias_segments", ctypes.c_int32),
        ("C", ctypes.c_void_p), ("ldc", ctypes.c_int64),
        ("C32", _c_f32p), ("ldc32", ctypes.c_int64),
        ("residual", ctypes.c_void_p), ("ldr", ctypes.c_int64),
        ("mul", ctypes.c_void_p), ("ldm", ctypes.c_int64),
        ("act_grad", ctypes.c_void_p), ("ldg", ctypes.c_int64),
        ("M", ctypes.c_int64), ("N", ctypes.c_int64), ("K", ctypes.c_int64),
        ("act", ctypes.c_int32),
        ("dropout_p", ctypes.c_float),
        ("seed", ctypes.c_uint64),
    ]


class WgradBf16Args(ctypes.Structure):
    """vb_wgrad_bf16_args"""
    _fields_ = [
        ("dY", ctypes.c_void_p), ("ldy", ctypes.c_int64),
        ("X", ctypes.c_void_p), ("ldx", ctypes.c_int64),
        ("dW", _c_f32p * VB_MAX_SEGMENTS), ("ldw", ctypes.c_int64),
        ("dbias", _c_f32p * VB_MAX_SEGMENTS),
        ("M", ctypes.c_int64), ("K", ctypes.c_int64),
        ("nseg", ctypes.c_int32), ("seg_n", ctypes.c_int32),
        ("n_valid", ctypes.c_int32), ("reserved", ctypes.c_int32),
    ]


class LayerLinear(ctypes.Structure):
    """vb_layer_linear"""
    _fields_ = [
        ("nseg", ctypes.c_int32), ("seg_n", ctypes.c_int32), ("K", ctypes.c_int32),
        ("w", _c_f32p * VB_MAX_SEGMENTS),
        ("bias", _c_f32p * VB_MAX_SEGMENTS),
        ("w16", ctypes.c_void_p),
        ("wt16", ctypes.c_void_p),
        ("dw", _c_f32p * VB_MAX_SEGMENTS),
        ("dbias", _c_f32p * VB_MAX_SEGMENTS),
    ]


class LayerNormP(ctypes.Structure):
    """vb_layer_norm"""
    _fields_ = [("gamma", _c_f32p), ("beta", _c_f32p), ("dgamma", _c_f32p), ("dbeta", _c_f32p)]


class FfnBlock(ctypes.Structure):
    """vb_ffn_block"""
    _fields_ = [
        ("M", ctypes.c_int64),
        ("Hc", ctypes.c_int32), ("H", ctypes.c_int32), ("I", ctypes.c_int32),
        ("ctx", ctypes.c_void_p),
        ("x", ctypes.c_void_p),
        ("o", LayerLinear), ("f1", LayerLinear), ("f2", LayerLinear),
        ("ln1", LayerNormP), ("ln2", LayerNormP),
        ("eps", ctypes.c_float), ("p_o", ctypes.c_float), ("p_f", ctypes.c_float),
        ("seed_o", ctypes.c_uint64), ("seed_f", ctypes.c_uint64),
        ("sum1", ctypes.c_void_p), ("a1", ctypes.c_void_p), ("h", ctypes.c_void_p), ("dact", ctypes.c_void_p),
        ("sum2", ctypes.c_void_p), ("y", ctypes.c_void_p),
        ("mean1", _c_f32p), ("rstd1", _c_f32p), ("mean2", _c_f32p), ("rstd2", _c_f32p),
        ("dy", ctypes.c_void_p),
        ("d_sum2", ctypes.c_void_p), ("d_sum2_drop", ctypes.c_void_p), ("d_pre", ctypes.c_void_p), ("d_a1", ctypes.c_void_p),
        ("d_sum1", ctypes.c_void_p), ("d_sum1_drop", ctypes.c_void_p), ("d_ctx", ctypes.c_void_p),
        ("ln_ws", _c_f32p),
    ]


class AttnBlock(ctypes.Structure):
    """vb_attn_block"""
    _fields_ = [
        ("batch", ctypes.c_int32), ("heads", ctypes.c_int32), ("head_dim", ctypes.c_int32),
        ("n1", ctypes.c_int32), ("n2", ctypes.c_int32),
        ("x1", ctypes.c_void_p), ("x2", ctypes.c_void_p),
        ("mask1", _c_f32p), ("mask2", _c_f32p),
        ("qkv1", LayerLinear), ("qkv2", LayerLinear),
        ("p1", ctypes.c_float), ("p2", ctypes.c_float),
        ("seed1", ctypes.c_uint64), ("seed2", ctypes.c_uint64),
        ("qkv1_out", ctypes.c_void_p), ("qkv2_out", ctypes.c_void_p),
        ("lse1", _c_f32p), ("lse2", _c_f32p),
        ("ctx1", ctypes.c_void_p), ("ctx2", ctypes.c_void_p),
        ("d_ctx1", ctypes.c_void_p), ("d_ctx2", ctypes.c_void_p),
        ("dqkv1", ctypes.c_void_p), ("dqkv2", ctypes.c_void_p),
        ("dvec", _c_f32p),
        ("dres1", ctypes.c_void_p), ("dres2", ctypes.c_void_p),
        ("dx1", ctypes.c_void_p), ("dx2", ctypes.c_void_p),
    ]


class LayerArgs(ctypes.Structure):
    """vb_layer_args"""
    _fields_ = [
        ("dtype", ctypes.c_int32), ("training", ctypes.c_int32),
        ("wgrad_stream", ctypes.c_void_p),
        ("attn", AttnBlock),
        ("s1", FfnBlock),
        ("s2", FfnBlock),
    ]


class AttentionMxArgs(ctypes.Structure):
    """vb_attention_mx_args"""
    _fields_ = [
        ("batch", ctypes.c_int32), ("heads", ctypes.c_int32), ("head_dim", ctypes.c_int32),
        ("n_q", ctypes.c_int32), ("n_k", ctypes.c_int32),
        ("q_batch", ctypes.c_int32), ("kv_batch", ctypes.c_int32),
        ("Q", ctypes.c_void_p), ("ldq", ctypes.c_int64),
        ("K", ctypes.c_void_p), ("ldk", ctypes.c_int64),
        ("V", ctypes.c_void_p), ("ldv", ctypes.c_int64),
        ("mask_add", _c_f32p),
        ("scale", ctypes.c_float),
        ("Oq", ctypes.c_void_p), ("ldo", ctypes.c_int64),
        ("o_scales", ctypes.c_void_p), ("o_srows", ctypes.c_int64),
    ]


class AttentionArgs(ctypes.Structure):
    """vb_attention_args (and vb_attention_bf16_args: the same layout, bf16 tensors behind Q / K / V / O)"""
    _fields_ = [
        ("batch", ctypes.c_int32), ("heads", ctypes.c_int32), ("head_dim", ctypes.c_int32),
        ("n_q", ctypes.c_int32), ("n_k", ctypes.c_int32),
        ("q_batch", ctypes.c_int32), ("kv_batch", ctypes.c_int32),
        ("Q", _c_f32p), ("ldq", ctypes.c_int64),
        ("K", _c_f32p), ("ldk", ctypes.c_int64),
        ("V", _c_f32p), ("ldv", ctypes.c_int64),
        ("mask_add", _c_f32p),
        ("O", _c_f32p), ("ldo", ctypes.c_int64),
        ("probs", _c_f32p),
        ("lse", _c_f32p),
        ("scale", ctypes.c_float),
        ("dropout_p", ctypes.c_float),
        ("seed", ctypes.c_uint64),
    ]


DVEC_COMPUTE, DVEC_ACCUMULATE, DVEC_GIVEN = 0, 1, 2      # include/vilbert_hip.h VB_DVEC_*


class AttentionGrads(ctypes.Structure):
    """vb_attention_grads (and vb_attention_bf16_grads)"""
    _fields_ = [
        ("dO", _c_f32p), ("lddo", ctypes.c_int64),
        ("dQ", _c_f32p), ("lddq", ctypes.c_int64),
        ("dK", _c_f32p), ("lddk", ctypes.c_int64),
        ("dV", _c_f32p), ("lddv", ctypes.c_int64),
        ("dvec", _c_f32p),
        ("dvec_mode", ctypes.c_int32),
    ]


class LinearBwdInputArgs(ctypes.Structure):
    """vb_linear_bwd_input_args"""
    _fields_ = [
        ("M", ctypes.c_int32), ("K", ctypes.c_int32),
        ("nseg", ctypes.c_int32), ("seg_n", ctypes.c_int32),
        ("dY", _c_f32p), ("ldy", ctypes.c_int64),
        ("W", _c_f32p * VB_MAX_SEGMENTS), ("ldw", ctypes.c_int64),
        ("dX", _c_f32p), ("ldx", ctypes.c_int64),
        ("accumulate", ctypes.c_int32),
        ("residual", _c_f32p), ("ldr", ctypes.c_int64),
        ("mul", _c_f32p), ("ldm", ctypes.c_int64),
    ]


class LinearBwdWeightArgs(ctypes.Structure):
    """vb_linear_bwd_weight_args"""
    _fields_ = [
        ("M", ctypes.c_int32), ("K", ctypes.c_int32),
        ("nseg", ctypes.c_int32), ("seg_n", ctypes.c_int32),
        ("dY", _c_f32p), ("ldy", ctypes.c_int64),
        ("X", _c_f32p), ("ldx", ctypes.c_int64),
        ("dW", _c_f32p * VB_MAX_SEGMENTS), ("ldw", ctypes.c_int64),
        ("dbias", _c_f32p * VB_MAX_SEGMENTS),
        ("accumulate", ctypes.c_int32),
    ]


class AdamWTensor(ctypes.Structure):
    """vb_adamw_tensor (64 bytes)"""
    _fields_ = [
        ("param", _c_f32p), ("grad", _c_f32p), ("exp_avg", _c_f32p), ("exp_avg_sq", _c_f32p),
        ("numel", ctypes.c_int64),
        ("step_size", ctypes.c_float), ("beta1", ctypes.c_float), ("beta2", ctypes.c_float),
        ("eps", ctypes.c_float), ("decay", ctypes.c_float), ("reserved", ctypes.c_float),
    ]


class ConcapBatch(ctypes.Structure):
    """vb_concap_batch"""
    _fields_ = [
        ("batch", ctypes.c_int32), ("regions", ctypes.c_int32), ("tokens", ctypes.c_int32),
        ("feat_dim", ctypes.c_int32), ("objective", ctypes.c_int32),
        ("image_feat", _c_f32p), ("image_loc", _c_f32p), ("image_mask", ctypes.c_void_p),
        ("masked_label", ctypes.c_void_p), ("is_next", ctypes.c_void_p), ("image_label", ctypes.c_void_p),
        ("lm_label_ids", ctypes.c_void_p),
        ("out_image_feat", _c_f32p), ("out_image_loc", _c_f32p), ("out_image_mask", ctypes.c_void_p),
        ("out_image_label", ctypes.c_void_p), ("out_lm_label_ids", ctypes.c_void_p),
    ]


# name -> (restype, argtypes); mirrors include/vilbert_hip.h one to one (checked by
# tests/test_abi.py against the header text).
_I32, _I64, _F32, _P, _U64 = ctypes.c_int32, ctypes.c_int64, ctypes.c_float, ctypes.c_void_p, ctypes.c_uint64
SIGNATURES = {
    "vb_abi_version": (ctypes.c_int, []),
    "vb_error_string": (ctypes.c_char_p, [ctypes.c_int]),
    "vb_set_gemm_mode": (ctypes.c_int, [ctypes.c_int]),
    "vb_set_gemm_tile": (ctypes.c_int, [ctypes.c_int]),
    "vb_set_gemm_v4": (ctypes.c_int, [ctypes.c_int]),
    "vb_set_deterministic": (ctypes.c_int, [ctypes.c_int, _P, _I64]),
    "vb_deterministic_fallbacks": (_I64, []),
    "vb_set_seed_epoch": (ctypes.c_int, [_P]),
    "vb_bump_counter": (ctypes.c_int, [_P, _P]),
    "vb_linear_fwd": (ctypes.c_int, [_P, ctypes.POINTER(LinearArgs)]),
    "vb_linear_bwd_input": (ctypes.c_int, [_P, ctypes.POINTER(LinearBwdInputArgs)]),
    "vb_linear_bwd_weight": (ctypes.c_int, [_P, ctypes.POINTER(LinearBwdWeightArgs)]),
    "vb_quantize_rows_fp8": (ctypes.c_int, [_P, _I64, _I32, _P, _I64, _P, _I64, _P]),
    "vb_linear_fwd_fp8": (ctypes.c_int, [_P, ctypes.POINTER(LinearFp8Args)]),
    "vb_layernorm_fwd_fp8": (ctypes.c_int, [_P, _I64, _I32, _P, _P, _P, _P, _F32, _P, _P, _I64, _P]),
    "vb_quantize_rows_mx": (ctypes.c_int, [_P, _I64, _I32, _P, _I64, _P, _I64, _P, _I64]),
    "vb_quantize_rows_mx_bf16": (ctypes.c_int, [_P, _I64, _I32, _P, _I64, _P, _I64, _P, _I64]),
    "vb_linear_fwd_mx": (ctypes.c_int, [_P, ctypes.POINTER(LinearMxArgs)]),
    "vb_layernorm_fwd_mx": (ctypes.c_int, [_P, _I64, _I32, _P, _P, _P, _P, _F32, _P, _P, _I64, _P, _I64]),
    "vb_attention_fwd_mx": (ctypes.c_int, [_P, ctypes.POINTER(AttentionMxArgs)]),
    "vb_layernorm_fwd_mx16": (ctypes.c_int, [_P, _I64, _I32, _P, _P, _P, _F32, _P, _P, _I64, _P, _I64]),
    "vb_act_bwd": (ctypes.c_int, [_P, _I64, _I32, _P, _P, _P]),
    "vb_dropout": (ctypes.c_int, [_P, _I64, _P, _P, _P, _F32, _U64]),
    "vb_layernorm_fwd": (ctypes.c_int, [_P, _I64, _I32, _P, _P, _P, _P, _F32, _P, _P, _P]),
    "vb_layernorm_bwd_workspace": (ctypes.c_int64, [_I64, _I32]),
    "vb_layernorm_bwd": (ctypes.c_int, [_P, _I64, _I32, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "vb_layernorm_bwd_drop": (ctypes.c_int, [_P, _I64, _I32, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _F32, _U64]),
    "vb_text_embed_ln_fwd": (ctypes.c_int, [_P, _I32, _I32, _I32, _I32, _I32, _I32, _P, _P, _I32, _P, _P, _P, _P, _P, _P, _P,
                                             _F32, _P, _P, _P, _P]),
    "vb_text_embed_bwd": (ctypes.c_int, [_P, _I32, _I32, _I32, _I32, _I32, _I32, _P, _P, _P, _P, _P, _P, _P, _P]),
    "vb_image_embed_ln_fwd": (ctypes.c_int, [_P, _I64, _I32, _P, _P, _P, _P, _P, _P, _F32, _P, _P, _P, _P]),
    "vb_additive_mask": (ctypes.c_int, [_P, _I64, _P, _I32, _P]),
    "vb_attention_fwd": (ctypes.c_int, [_P, ctypes.POINTER(AttentionArgs)]),
    "vb_attention_bwd": (ctypes.c_int, [_P, ctypes.POINTER(AttentionArgs), ctypes.POINTER(AttentionGrads)]),
    "vb_adamw_step": (ctypes.c_int, [_P, _I32, _P, _P, _P, _I32]),
    "vb_xent_fwd": (ctypes.c_int, [_P, _I64, _I32, _P, _I64, _P, _I64, _P, _P, _P, _P]),
    "vb_xent_bwd": (ctypes.c_int, [_P, _I64, _I32, _P, _I64, _P, _I64, _P, _P, _P, _P, _I64]),
    "vb_kl_fwd": (ctypes.c_int, [_P, _I64, _I32, _P, _I64, _P, _I64, _F32, _P, _P, _P, _P, _P]),
    "vb_kl_bwd": (ctypes.c_int, [_P, _I64, _I32, _P, _I64, _P, _I64, _P, _P, _P, _F32, _P, _I64, _P]),
    "vb_concap_finish_batch": (ctypes.c_int, [_P, ctypes.POINTER(ConcapBatch)]),
    "vb_attention_fwd_bf16": (ctypes.c_int, [_P, ctypes.POINTER(AttentionArgs)]),
    "vb_attention_bwd_bf16": (ctypes.c_int, [_P, ctypes.POINTER(AttentionArgs), ctypes.POINTER(AttentionGrads)]),
    "vb_linear_bf16": (ctypes.c_int, [_P, ctypes.POINTER(LinearBf16Args)]),
    "vb_wgrad_bf16": (ctypes.c_int, [_P, ctypes.POINTER(WgradBf16Args)]),
    "vb_colsum_bf16_workspace": (_I64, [_I32]),
    "vb_colsum_bf16": (ctypes.c_int, [_P, _I64, _I32, _P, _I64, _P, _P]),
    "vb_weight_shadow_bf16": (ctypes.c_int, [_P, _I32, _I32, _P, _I64, _P, _I64, _P, _I64]),
    "vb_weight_shadow_multi": (ctypes.c_int, [_P, _I32, _P, _I64]),
    "vb_cast_f32_bf16": (ctypes.c_int, [_P, _I64, _P, _P]),
    "vb_cast_rows_f32_bf16": (ctypes.c_int, [_P, _I64, _I32, _P, _I64, _P, _I64]),
    "vb_cast_bf16_f32": (ctypes.c_int, [_P, _I64, _P, _P]),
    "vb_layernorm_fwd_bf16": (ctypes.c_int, [_P, _I64, _I32, _P, _P, _P, _F32, _P, _P, _P]),
    "vb_layernorm_bwd_bf16_workspace": (_I64, [_I64, _I32]),
    "vb_layernorm_bwd_bf16": (ctypes.c_int, [_P, _I64, _I32, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _F32, _U64]),
    "vb_layer_fwd": (ctypes.c_int, [_P, ctypes.POINTER(LayerArgs)]),
    "vb_layer_bwd": (ctypes.c_int, [_P, ctypes.POINTER(LayerArgs)]),
}

_lib = None


def lib():
    """Load (once) and return the shared library; raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.isfile(LIB_PATH):
            raise RuntimeError(
                "libvilbert_hip.so not found at %s - build it with `python -c 'import __graft_entry__ as g; "
                "g.build()'` (or `make -C vilbert-multi-task_amd/csrc`). There is no fallback path." % LIB_PATH)
        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)  # AttributeError if the .so is stale
            fn.restype, fn.argtypes = res, args
        if handle.vb_abi_version() != 18:
            raise RuntimeError("libvilbert_hip.so ABI version mismatch")
        _lib = handle
        if os.environ.get("VB_GEMM_MODE") in ("fp8", "mxfp8"):
            _FP8["on"] = True
            _FP8["mx"] = os.environ.get("VB_GEMM_MODE") == "mxfp8"
        if os.environ.get("VB_GEMM_MODE") == "bf16":
            _BF16["stream"] = os.environ.get("VB_BF16_STREAM", "1") != "0"
    return _lib


def set_gemm_tile(code):
    """Tile selection of the fp32 GEMM: 0 = cost model, 22 | 33 | 34 | 43 | 44 = force a tile of the
    second-generation kernel, -1 = round-1 kernel only. Returns the previous code."""
    return lib().vb_set_gemm_tile(int(code))


# "ws": device index -> the workspace registered for that device; "parked": device index -> the workspace that was
# registered when the setting was switched off - re-enabling re-registers the SAME tensor (an off / on toggle, e.g. the
# bench's --deterministic A/B or a test fixture, costs no new 2 GiB allocation). A workspace that is REPLACED by a larger
# one is dropped: a HIP graph captured while it was registered has its address and the stream -> slice assignment baked
# in, which is why GraphedTrainStep holds its own reference (`_det_workspace`, vilbert/graphed.py) - the memory stays live
# exactly as long as something can still write into it.
_DET = {"ws": {}, "parked": {}, "wanted": os.environ.get("VB_DETERMINISTIC", "1") != "0", "mb": 2048}


def set_deterministic(on, workspace_mb=None, device=None):
    """Deterministic split-K weight gradients (include/vilbert_hip.h: vb_set_deterministic): the K splits of a launch
    store their partial products to a device workspace and a second kernel adds them in split order - bit-identical
    gradients from run to run, and measured FASTER than the fp32-atomics path (profiles/r03_deterministic_cost.txt), so
    it is the default (VB_DETERMINISTIC=0 or set_deterministic(False) = atomics). One workspace PER DEVICE (default
    2 GiB = 8 per-stream slices of 256 MiB: the largest split launch of the models - a 16-way split of W[3072, 768] -
    needs ~150 MB), allocated when that device first runs a split launch (`ensure_deterministic`) or here for `device`
    (default: the current one). A launch that finds no slice (ninth stream of a device) or does not fit falls back to the
    atomics and is counted (`deterministic_fallbacks()`). Returns the previous setting."""
    import torch
    if on:
        if workspace_mb is not None:
            _DET["mb"] = int(workspace_mb)
        prev = _DET["wanted"]
        _DET["wanted"] = True
        _register_workspace(torch.device(device or "cuda"))
        return bool(prev)
    prev = lib().vb_set_deterministic(0, None, 0)
    _DET["parked"].update(_DET["ws"])
    _DET["ws"], _DET["wanted"] = {}, False
    return bool(prev)


def _register_workspace(device):
    import torch
    idx = device.index if device.index is not None else torch.cuda.current_device()
    ws = _DET["ws"].get(idx)
    if ws is None:
        ws = _DET["parked"].pop(idx, None)          # switched off earlier: the same buffer again
    if ws is None or ws.numel() * 4 < _DET["mb"] * (1 << 20):
        ws = torch.empty(_DET["mb"] * (1 << 20) // 4, dtype=torch.float32, device=torch.device("cuda", idx))
    rc = lib().vb_set_deterministic(1, ws.data_ptr(), ws.numel() * 4)
    if rc < 0:
        check(rc, "vb_set_deterministic")
    _DET["ws"][idx] = ws
    return ws


def ensure_deterministic(device):
    """Called by the split-K launchers: registers the workspace of `device` on its first split launch (the setting is on
    by default, the allocation needs the device)."""
    if _DET["wanted"]:
        idx = device.index if device.index is not None else current_device()
        if idx not in _DET["ws"]:
            _register_workspace(device)


def deterministic_enabled():
    """The setting itself (VB_DETERMINISTIC / set_deterministic): ordered split-K reduces wanted."""
    return bool(_DET["wanted"])


def deterministic_workspace(device=None):
    """The workspace tensor registered for `device` (None if none): GraphedTrainStep holds it while its graph lives."""
    import torch
    d = torch.device(device or "cuda")
    return _DET["ws"].get(d.index if d.index is not None else current_device())


def deterministic_fallbacks():
    """Split launches since the last registration that wanted the ordered reduce but ran with atomics."""
    return int(lib().vb_deterministic_fallbacks())


def set_gemm_v4(mode):
    """Persistent one-block-per-CU GEMM kernel (csrc/gemm_v4.h): 0 = never, 1 = wherever its tiles fill whole
    rounds of the 256 CUs (default), 2 = every eligible launch. Returns the previous mode."""
    return lib().vb_set_gemm_v4(int(mode))


GEMM_MODES = {"f32": 0, "bf16x6": 3, "bf16x3": 2, "bf16": 1}
_FP8 = {"on": False, "mx": False}
_BF16 = {"stream": False}


def set_gemm_mode(mode):
    """Select the GEMM arithmetic; returns the previous name.
    "f32" exact-fp32 MFMA | "bf16x6" | "bf16x3" | "bf16": arithmetic of every vb_linear_* launch (C side);
    "fp8": FORWARD linears whose shape allows it run on quantised e4m3 operands (vb_linear_fwd_fp8, host-side weight
    cache in ops.py); everything else - backward GEMMs, ineligible shapes - stays exact fp32;
    "fp8+bf16": fp8 forward as above, every other GEMM (backward, ineligible shapes) in the bf16 mode;
    "mxfp8": as "fp8" with the MX block-scaled kernels wherever K % 128 == 0 and N % 128 == 0 (round 4; inference).
    "bf16" (round 5) additionally switches the MODEL to the bf16 training / inference stream (`bf16_stream()`): the encoder's
    activations, saved tensors and activation gradients are torch.bfloat16 tensors served by csrc/gemm_bf16.hip
    (ops16.py) - the reference's `model.half()` mode; launches that still see fp32 tensors (heads, embeddings, ineligible
    shapes) run on the C side's bf16-operand kernels as before. VB_BF16_STREAM=0 keeps the round-2 behaviour (fp32
    tensors everywhere, operands rounded on their way into LDS)."""
    if mode not in GEMM_MODES and mode not in ("fp8", "fp8+bf16", "mxfp8"):
        raise KeyError("unknown GEMM mode %r (f32 | bf16x6 | bf16x3 | bf16 | fp8 | fp8+bf16 | mxfp8)" % (mode,))
    prev_fp8, prev_mx = _FP8["on"], _FP8["mx"]
    _FP8["on"] = mode in ("fp8", "fp8+bf16", "mxfp8")
    _FP8["mx"] = mode == "mxfp8"
    _BF16["stream"] = mode == "bf16" and os.environ.get("VB_BF16_STREAM", "1") != "0"
    prev = lib().vb_set_gemm_mode(GEMM_MODES["f32" if mode in ("fp8", "mxfp8") else "bf16" if mode == "fp8+bf16" else mode])
    prev_name = {v: k for k, v in GEMM_MODES.items()}[prev]
    if prev_mx:
        return "mxfp8"
    if prev_fp8:
        return "fp8+bf16" if prev_name == "bf16" else "fp8"
    return prev_name


def fp8_enabled():
    return _FP8["on"]


def bf16_stream():
    """The model keeps its encoder activations in bfloat16 (set_gemm_mode("bf16"), round 5)."""
    return _BF16["stream"]


def mx_enabled():
    """MX e4m3 forward mode (set_gemm_mode("mxfp8")): linears whose shape allows it run on block-scaled operands
    (csrc/mx8.hip), LayerNorm and GEMM epilogues emit the codes the next linear consumes; the remaining eligible shapes use
    the row-scaled fp8 kernel, everything else stays fp32."""
    return _FP8["mx"]


# Parameters are rewritten through raw pointers by the native optimizer, behind torch's version counters; caches of
# derived weights (the fp8 codes in ops.py) compare this counter.
WEIGHTS_EPOCH = [0]


def weights_changed():
    WEIGHTS_EPOCH[0] += 1


# ANY optimizer announces its step (round 6, advisor finding): an optimizer that writes through `.data` - the reference's RAdam
# does, /root/reference/vilbert/optimization.py:98,174 `p.data.copy_(...)`, selected by train_tasks.py --optim RAdam - bumps
# neither torch's version counters nor (not being the native AdamW) this epoch, and the bf16 shadows / fp8 / MX weight caches
# would keep serving the initial weights while the fp32 masters move. torch calls global post-step hooks for every
# torch.optim.Optimizer subclass, whatever its step() does.
try:
    from torch.optim.optimizer import register_optimizer_step_post_hook as _register_post_step
    _register_post_step(lambda _opt, _args, _kwargs: weights_changed())
except ImportError:      # pragma: no cover  (a torch without global optimizer hooks: the native AdamW still announces itself)
    pass


def check(code, what):
    if code != 0:
        msg = lib().vb_error_string(code)
        raise RuntimeError("%s failed: [%d] %s" % (what, code, msg.decode() if msg else "?"))


# torch.cuda.current_stream() builds a Stream object through several layers of Python (~3.5 us, and an eager step at
# 64 samples asks for the stream ~1,400 times: tools/host_profile.py); these two go straight to the C++ getters.
# (private torch entry points, present in the PyTorch 2.x builds this package targets; a build without them gets the
# public API - same results, slower)
_cuda_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_cuda_device = getattr(torch._C, "_cuda_getDevice", None)
_cuda_set_stream = getattr(torch._C, "_cuda_setStream", None)
if _cuda_raw_stream is None or _cuda_device is None:
    _cuda_device = torch.cuda.current_device

    def _cuda_raw_stream(device_index):
        return torch.cuda.current_stream(device_index).cuda_stream


def current_device():
    return _cuda_device()


def raw_stream(device_index=None):
    """hipStream_t (as an int) torch currently launches on, for `device_index` (default: the current device)."""
    return _cuda_raw_stream(_cuda_device() if device_index is None else device_index)


def set_stream(st):
    """Make the torch.cuda.Stream `st` current (torch.cuda.set_stream without its Python layers)."""
    if _cuda_set_stream is None:
        torch.cuda.set_stream(st)
    else:
        _cuda_set_stream(stream_id=st.stream_id, device_index=st.device_index, device_type=st.device_type)


def stream_ptr():
    return ctypes.c_void_p(_cuda_raw_stream(_cuda_device()))


def dev_f32(t, what):
    """Validate a device fp32 tensor and return its address (0 for None)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("%s: expected a tensor on a HIP device, got %s - the MI355X-native path has no "
                           "CPU fallback" % (what, t.device))
    if t.dtype != torch.float32:
        raise RuntimeError("%s: expected float32, got %s" % (what, t.dtype))
    return t.data_ptr()


def dev_i64(t, what):
    if t is None:
        return None
    if not t.is_cuda or t.dtype != torch.int64 or not t.is_contiguous():
        raise RuntimeError("%s: expected a contiguous int64 tensor on a HIP device" % what)
    return t.data_ptr()
