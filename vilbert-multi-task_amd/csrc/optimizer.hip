// Multi-tensor AdamW step: ONE launch updates every parameter tensor of the model (the reference builds
// ~530 parameter groups, one per tensor: train_tasks.py:400-420, train_concap.py:420-440).
// Arithmetic = pytorch-transformers 1.0.0 `AdamW.step` (requirements.txt:1; the package is not vendored in
// the reference tree - restated in oracle/adamw_oracle.py):
//   m = beta1 m + (1 - beta1) g;  v = beta2 v + (1 - beta2) g^2;  p -= step_size * m / (sqrt(v) + eps)
//   then decoupled weight decay on the UPDATED value:  p -= lr * wd * p
// with step_size = lr * sqrt(1 - beta2^t) / (1 - beta1^t) when correct_bias else lr (computed on the host).
// HBM-bound: 16 B read + 12 B written per parameter; blocks walk fixed-size chunks listed in a table.
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void adamw_kernel(const vb_adamw_tensor* __restrict__ tab,
                                                    const int32_t* __restrict__ chunk_tensor,
                                                    const int64_t* __restrict__ chunk_off, int chunk_elems) {
    const vb_adamw_tensor t = tab[chunk_tensor[blockIdx.x]];
    const long off = chunk_off[blockIdx.x];
    const long end = min((long)t.numel, off + chunk_elems);
    const float b1 = t.beta1, b2 = t.beta2, c1 = 1.0f - t.beta1, c2 = 1.0f - t.beta2;
    float* __restrict__ p = t.param;
    const float* __restrict__ g = t.grad;
    float* __restrict__ m = t.exp_avg;
    float* __restrict__ v = t.exp_avg_sq;
    // 16-byte accesses need all four base pointers 16-byte aligned (chunk offsets are multiples of 4 elements);
    // a tensor that is an oddly offset view takes the scalar loop for its whole chunk
    const bool vec_ok = ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(m) |
                          reinterpret_cast<uintptr_t>(v)) & 15u) == 0;
    const long n4 = vec_ok ? (end - off) >> 2 : 0;
    for (long i = threadIdx.x; i < n4; i += 256) {
        const long e = off + 4 * i;
        f32x4 pp = *reinterpret_cast<f32x4*>(p + e);
        const f32x4 gg = *reinterpret_cast<const f32x4*>(g + e);
        f32x4 mm = *reinterpret_cast<f32x4*>(m + e), vv = *reinterpret_cast<f32x4*>(v + e);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            mm[k] = mm[k] * b1 + c1 * gg[k];
            vv[k] = vv[k] * b2 + c2 * gg[k] * gg[k];
            pp[k] -= t.step_size * (mm[k] / (sqrtf(vv[k]) + t.eps));
            if (t.decay > 0.f) pp[k] -= t.decay * pp[k];
        }
        *reinterpret_cast<f32x4*>(p + e) = pp;
        *reinterpret_cast<f32x4*>(m + e) = mm;
        *reinterpret_cast<f32x4*>(v + e) = vv;
    }
    for (long e = off + 4 * n4 + threadIdx.x; e < end; e += 256) {
        const float gg = g[e];
        const float mm = m[e] * b1 + c1 * gg, vv = v[e] * b2 + c2 * gg * gg;
        float pp = p[e] - t.step_size * (mm / (sqrtf(vv) + t.eps));
        if (t.decay > 0.f) pp -= t.decay * pp;
        p[e] = pp; m[e] = mm; v[e] = vv;
    }
}

}  // namespace

extern "C" int vb_adamw_step(void* stream, int32_t n_chunks, const vb_adamw_tensor* table,
                             const int32_t* chunk_tensor, const int64_t* chunk_off, int32_t chunk_elems) {
    if (table == nullptr || chunk_tensor == nullptr || chunk_off == nullptr || n_chunks <= 0) return VB_E_BADARG;
    if (chunk_elems <= 0 || chunk_elems % 4 != 0) return VB_E_ALIGN;
    hipLaunchKernelGGL(adamw_kernel, dim3((unsigned)n_chunks), dim3(256), 0, static_cast<hipStream_t>(stream), table,
                       chunk_tensor, chunk_off, chunk_elems);
    VB_LAUNCH_CHECK();
    return 0;
}
