// Elementwise kernels of the training path (HBM-bound, 16-byte accesses, grid-stride):
//   activation backward  dpre = dy * act'(pre)
//   dropout              y = x * keep(seed, i) / (1 - p)   (the same launch serves forward and backward)
#include "common.h"
#include "rng.h"

namespace {

template <int ACT>
__global__ __launch_bounds__(256) void act_bwd_kernel(long n4, const f32x4* __restrict__ dy,
                                                      const f32x4* __restrict__ pre, f32x4* __restrict__ dx) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const f32x4 g = dy[i], x = pre[i];
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e)
            o[e] = ACT == VB_ACT_GELU ? g[e] * gelu_grad(x[e])
                   : (ACT == VB_ACT_SWISH ? g[e] * swish_grad(x[e]) : (x[e] > 0.f ? g[e] : 0.f));
        dx[i] = o;
    }
}

__global__ __launch_bounds__(256) void dropout_kernel(long n, const float* __restrict__ x,
                                                      const float* __restrict__ res, float* __restrict__ y,
                                                      float p, float scale, uint64_t seed_in,
                                                      const uint64_t* __restrict__ epoch) {
    const uint64_t seed = vb_seed_with_epoch(seed_in, epoch);
    const long n4 = n >> 2;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const f32x4 v = reinterpret_cast<const f32x4*>(x)[i];
        f32x4 o = res != nullptr ? reinterpret_cast<const f32x4*>(res)[i] : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (vb_keep(seed, (uint64_t)(4 * i + e), p)) o[e] += v[e] * scale;
        reinterpret_cast<f32x4*>(y)[i] = o;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const long i = (n4 << 2) + threadIdx.x;
        y[i] = (res != nullptr ? res[i] : 0.f) + (vb_keep(seed, (uint64_t)i, p) ? x[i] * scale : 0.f);
    }
}

inline unsigned grid_for(long work_items) {
    long blocks = (work_items + 255) / 256;
    if (blocks > 2048) blocks = 2048;  // ~8 blocks per CU, grid-stride the rest
    if (blocks < 1) blocks = 1;
    return (unsigned)blocks;
}

}  // namespace

extern "C" int vb_act_bwd(void* stream, int64_t n, int32_t act, const float* dy, const float* preact, float* dx) {
    if (dy == nullptr || preact == nullptr || dx == nullptr || n <= 0) return VB_E_BADARG;
    if (n % 4 != 0 || !vb_aligned16(dy) || !vb_aligned16(preact) || !vb_aligned16(dx)) return VB_E_ALIGN;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const long n4 = n / 4;
    if (act == VB_ACT_GELU)
        hipLaunchKernelGGL(act_bwd_kernel<VB_ACT_GELU>, dim3(grid_for(n4)), dim3(256), 0, st, n4,
                           reinterpret_cast<const f32x4*>(dy), reinterpret_cast<const f32x4*>(preact),
                           reinterpret_cast<f32x4*>(dx));
    else if (act == VB_ACT_RELU)
        hipLaunchKernelGGL(act_bwd_kernel<VB_ACT_RELU>, dim3(grid_for(n4)), dim3(256), 0, st, n4,
                           reinterpret_cast<const f32x4*>(dy), reinterpret_cast<const f32x4*>(preact),
                           reinterpret_cast<f32x4*>(dx));
    else if (act == VB_ACT_SWISH)
        hipLaunchKernelGGL(act_bwd_kernel<VB_ACT_SWISH>, dim3(grid_for(n4)), dim3(256), 0, st, n4,
                           reinterpret_cast<const f32x4*>(dy), reinterpret_cast<const f32x4*>(preact),
                           reinterpret_cast<f32x4*>(dx));
    else
        return VB_E_BADARG;
    VB_LAUNCH_CHECK();
    return 0;
}

extern "C" int vb_dropout(void* stream, int64_t n, const float* x, const float* residual, float* y, float p,
                          uint64_t seed) {
    if (x == nullptr || y == nullptr || n <= 0 || !(p >= 0.f && p < 1.f)) return VB_E_BADARG;
    if (!vb_aligned16(x) || !vb_aligned16(y) || (residual != nullptr && !vb_aligned16(residual))) return VB_E_ALIGN;
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(dropout_kernel, dim3(grid_for(n / 4 + 1)), dim3(256), 0, st, (long)n, x, residual, y, p,
                       1.0f / (1.0f - p), seed, vb_seed_epoch());
    VB_LAUNCH_CHECK();
    return 0;
}

namespace {
__global__ void bump_counter_kernel(uint64_t* c) { *c += 1; }
}  // namespace

extern "C" int vb_bump_counter(void* stream, uint64_t* device_counter) {
    if (device_counter == nullptr) return VB_E_BADARG;
    hipLaunchKernelGGL(bump_counter_kernel, dim3(1), dim3(1), 0, static_cast<hipStream_t>(stream), device_counter);
    VB_LAUNCH_CHECK();
    return 0;
}

// ---- post-passes of the round-1 GEMM kernel (ragged / unaligned launches only; the second-generation kernel fuses
// both into its epilogue) ------------------------------------------------------------------------------------------
namespace {

// d[r][c] = act'(d[r][c]) in place: d holds the pre-activation on entry
__global__ __launch_bounds__(256) void act_grad_inplace_kernel(long rows, int cols, float* __restrict__ d, long ld, int act) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= rows * cols) return;
    float* q = d + (i / cols) * ld + (i % cols);
    const float v = *q;
    *q = act == VB_ACT_GELU ? gelu_grad(v)
         : (act == VB_ACT_RELU ? (v > 0.f ? 1.f : 0.f) : (act == VB_ACT_SWISH ? swish_grad(v) : 1.f));
}

__global__ __launch_bounds__(256) void mul_inplace_kernel(long rows, int cols, float* __restrict__ c, long ldc,
                                                          const float* __restrict__ m, long ldm) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= rows * cols) return;
    c[(i / cols) * ldc + (i % cols)] *= m[(i / cols) * ldm + (i % cols)];
}

}  // namespace

namespace vbgemm {

int launch_act_grad_inplace(hipStream_t st, long rows, int cols, float* d, long ld, int act) {
    hipLaunchKernelGGL(act_grad_inplace_kernel, dim3((unsigned)((rows * cols + 255) / 256)), dim3(256), 0, st, rows, cols,
                       d, ld, act);
    VB_LAUNCH_CHECK();
    return 0;
}

int launch_mul_inplace(hipStream_t st, long rows, int cols, float* c, long ldc, const float* m, long ldm) {
    hipLaunchKernelGGL(mul_inplace_kernel, dim3((unsigned)((rows * cols + 255) / 256)), dim3(256), 0, st, rows, cols, c,
                       ldc, m, ldm);
    VB_LAUNCH_CHECK();
    return 0;
}

}  // namespace vbgemm
