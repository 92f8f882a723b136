// Shared device helpers for the gfx950 kernels (wave = 64 lanes everywhere).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/vilbert_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define VB_LAUNCH_CHECK()                         \
    do {                                          \
        hipError_t e__ = hipGetLastError();       \
        if (e__ != hipSuccess) return (int)e__;   \
    } while (0)

// device pointer registered with vb_set_seed_epoch (or null): every dropout launch passes it to its kernel
const uint64_t* vb_seed_epoch();

static inline bool vb_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// GELU (vilbert.py:117  x * 0.5 * (1 + erf(x / sqrt(2)))) and its derivative from ONE exponential:
//   z = |x| / sqrt 2,  E = exp(-z^2) = exp(-x^2 / 2),  erfc(z) = poly(t) t E with t = 1 / (1 + 0.3275911 z)
//   (Abramowitz & Stegun 7.1.26, |error| <= 1.5e-7 - three orders below the 1e-4 parity bar),
//   Phi(x) = x >= 0 ? 1 - erfc(z) / 2 : erfc(z) / 2 (no cancellation on the negative side),
//   gelu = x Phi,  gelu' = Phi + x E / sqrt(2 pi).
// ~20 VALU instructions for both values; libdevice's erff + expf cost ~55, which showed as 15 % of the FFN
// up-projection GEMM (28 M outputs per launch) and as 58 us of a 160 us fp8 GEMM.
__device__ __forceinline__ void gelu_parts(float x, float& phi, float& e) {
    const float z = fabsf(x) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
    e = __expf(-z * z);
    float poly = fmaf(1.061405429f, t, -1.453152027f);
    poly = fmaf(poly, t, 1.421413741f);
    poly = fmaf(poly, t, -0.284496736f);
    poly = fmaf(poly, t, 0.254829592f);
    const float half_erfc = 0.5f * poly * t * e;
    phi = x >= 0.f ? 1.0f - half_erfc : half_erfc;
}

__device__ __forceinline__ float gelu_erf(float x) {
    float phi, e;
    gelu_parts(x, phi, e);
    return x * phi;
}

__device__ __forceinline__ float gelu_grad(float x) {
    // d/dx [x Phi(x)] = Phi(x) + x exp(-x^2 / 2) / sqrt(2 pi)
    float phi, e;
    gelu_parts(x, phi, e);
    return fmaf(x * 0.39894228040143267794f, e, phi);
}

// both at once (the FFN up-projection epilogue in training stores the activation and its derivative)
__device__ __forceinline__ void gelu_and_grad(float x, float& y, float& d) {
    float phi, e;
    gelu_parts(x, phi, e);
    y = x * phi;
    d = fmaf(x * 0.39894228040143267794f, e, phi);
}

__device__ __forceinline__ float swish_act(float x) {
    // vilbert.py:120-121  x * sigmoid(x)
    return x / (1.0f + expf(-x));
}

__device__ __forceinline__ float swish_grad(float x) {
    const float s = 1.0f / (1.0f + expf(-x));
    return s + x * s * (1.0f - s);
}
