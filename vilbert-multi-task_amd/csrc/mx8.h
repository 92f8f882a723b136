// Device helpers of the MX (OCP microscaling) e4m3 format shared by the kernels that emit it (mx8.hip: quantiser and
// GEMM epilogue; rowops.hip: LayerNorm). Format and layout: see the head of mx8.hip / include/vilbert_hip.h.
#pragma once
#include "common.h"

// E8M0 byte of a block with maximum magnitude amax >= 0: e + 127 with e = ceil(log2(amax / 448)), from the bits of amax
// (448 = 1.75 x 2^8: amax = 1.f x 2^(E - 127) needs e = E - 127 - 8, one more when 1.f > 1.75)
__device__ __forceinline__ unsigned mx_scale_byte(float amax) {
    const unsigned u = __float_as_uint(amax);
    const int b = (int)(u >> 23) - 8 + ((u & 0x7fffffu) > 0x600000u ? 1 : 0);
    return (unsigned)max(b, 0);
}
// 2^-e = 2^(127 - byte)
__device__ __forceinline__ float mx_inv_scale(unsigned byte) { return __uint_as_float((254u - byte) << 23); }

__device__ __forceinline__ unsigned mx_pack4(const f32x4 v, float inv) {
    const int w = __builtin_amdgcn_cvt_pk_fp8_f32(v[0] * inv, v[1] * inv, 0, false);
    return (unsigned)__builtin_amdgcn_cvt_pk_fp8_f32(v[2] * inv, v[3] * inv, w, true);
}

__device__ __forceinline__ float amax4(const f32x4 v) {
    return fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3])));
}

// ---------------------------------------------------------------------------------------------------------------
// Row quantiser: one wave per row, 4 rows per block; a lane owns 4 consecutive columns of every 256-column chunk, so
// a 32-column block = 8 consecutive lanes (three xor steps), the four blocks of a K tile = 32 lanes.
// NV > 0: K = 256 NV, the whole row is loaded before the first use.
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void mx_quant_chunk(const f32x4 v, bool ok, int lane, int kt, int nkt, unsigned* __restrict__ dst,
                                               unsigned* __restrict__ sword) {
    float amax = ok ? amax4(v) : 0.f;
    amax = fmaxf(amax, __shfl_xor(amax, 1));
    amax = fmaxf(amax, __shfl_xor(amax, 2));
    amax = fmaxf(amax, __shfl_xor(amax, 4));
    const unsigned b = mx_scale_byte(amax);
    if (ok) *dst = mx_pack4(v, mx_inv_scale(b));
    unsigned w = b << (8 * ((lane >> 3) & 3));
    w |= (unsigned)__shfl_xor((int)w, 8);
    w |= (unsigned)__shfl_xor((int)w, 16);
    if ((lane & 31) == 0 && kt < nkt) *sword = w;
}


// GELU of the MX epilogues: x Phi(x) with Phi from an odd polynomial of degree 11 on the clamped argument,
//   Phi(t) ~ 0.5 + t Q(t^2),  t = clamp(x, -3.5, 3.5)     (minimax fit, |Phi error| <= 1.6e-4; restated in oracle/fp8_oracle.py)
// -> |gelu error| <= 4e-4 |x| + 6e-4: two orders below the e4m3 step (2^-4 relative) the value is rounded to right after.
// 10 full-rate VALU instructions; the erf form of the fp32 path (common.h gelu_parts: rcp + exp + ~14) made the GELU epilogue
// of the 3072-wide up-projection VALU-bound (40 us of a 78 us launch at 18,432 rows, tools/mx_lab).
__device__ __forceinline__ float mx_gelu(float x) {
    const float t = __builtin_amdgcn_fmed3f(x, -3.5f, 3.5f);
    const float u = t * t;
    float q = fmaf(-8.3218473e-07f, u, 3.89366778e-05f);
    q = fmaf(q, u, -0.000774126121f);
    q = fmaf(q, u, 0.00876406186f);
    q = fmaf(q, u, -0.0648922966f);
    q = fmaf(q, u, 0.398325773f);
    return x * fmaf(t, q, 0.5f);
}
