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

__device__ __forceinline__ float gelu_erf(float x) {
    // vilbert.py:117  x * 0.5 * (1 + erf(x / sqrt(2)))
    return x * 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
}

__device__ __forceinline__ float gelu_grad(float x) {
    // d/dx [x * 0.5 * (1 + erf(x / sqrt 2))] = 0.5 (1 + erf(x / sqrt 2)) + x * exp(-x^2 / 2) / sqrt(2 pi)
    const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
    return cdf + x * 0.39894228040143267794f * expf(-0.5f * x * x);
}

__device__ __forceinline__ float swish_act(float x) {
    // vilbert.py:120-121  x * sigmoid(x)
    return x / (1.0f + expf(-x));
}

__device__ __forceinline__ float swish_grad(float x) {
    const float s = 1.0f / (1.0f + expf(-x));
    return s + x * s * (1.0f - s);
}
