// Row kernels of the bf16 training path (gemm_bf16.hip): BertLayerNorm forward / backward on bf16 rows
// (reference vilbert.py:313-317 and its autograd). One 64-lane wave owns a row; a lane holds 4 consecutive columns of every
// 256-column chunk (8-byte loads / stores); the statistics, the normalisation and the gradient sums are fp32 exactly as in
// rowops.hip - only what crosses HBM is bf16. dgamma / dbeta: per-block partial rows to a workspace, then a second kernel
// adds them in block order (deterministic).
#include "common.h"
#include "rng.h"

namespace {

__device__ __forceinline__ unsigned short bf16_rne(float v) {
    const unsigned u = __float_as_uint(v);
    return (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}
__device__ __forceinline__ f32x4 load4(const unsigned short* p) {
    const uint2 w = *reinterpret_cast<const uint2*>(p);
    return f32x4{__uint_as_float(w.x << 16), __uint_as_float(w.x & 0xffff0000u), __uint_as_float(w.y << 16),
                 __uint_as_float(w.y & 0xffff0000u)};
}
__device__ __forceinline__ f32x4 unpack4(const uint2 w) {
    return f32x4{__uint_as_float(w.x << 16), __uint_as_float(w.x & 0xffff0000u), __uint_as_float(w.y << 16),
                 __uint_as_float(w.y & 0xffff0000u)};
}
__device__ __forceinline__ void store4(unsigned short* p, const f32x4 v) {
    *reinterpret_cast<uint2*>(p) = uint2{(unsigned)bf16_rne(v[0]) | ((unsigned)bf16_rne(v[1]) << 16),
                                         (unsigned)bf16_rne(v[2]) | ((unsigned)bf16_rne(v[3]) << 16)};
}

template <int R>
__device__ __forceinline__ void wave_sum_rows(float (&a)[R]) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        float t[R];
#pragma unroll
        for (int r = 0; r < R; ++r) t[r] = __shfl_xor(a[r], off, 64);
#pragma unroll
        for (int r = 0; r < R; ++r) a[r] += t[r];
    }
}

// A wave owns R consecutive rows and runs their dependent chains (load -> sum -> wave reduction -> squared deviations -> wave
// reduction -> store) side by side: with one row per wave the launch is bound by that chain's latency and the number of block
// rounds, not by HBM (11.4 us for 28 MB at 9,216 x 768: 2,304 blocks for the 2,048 places of the chip). The rows stay packed
// in registers between the phases (widened where used), so that R = 4 still runs at full occupancy.
template <int NV, int R>
__global__ __launch_bounds__(256) void layernorm16_fwd_kernel(long rows, int n_cols, const unsigned short* __restrict__ x,
                                                              const float* __restrict__ gamma, const float* __restrict__ beta,
                                                              float eps, unsigned short* __restrict__ y, float* __restrict__ mean,
                                                              float* __restrict__ rstd) {
    const int lane = threadIdx.x & 63;
    const long row0 = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * R;
    if (row0 >= rows) return;
    long rrow[R];
    bool have[R];                                       // (wave-uniform)
#pragma unroll
    for (int r = 0; r < R; ++r) {
        have[r] = row0 + r < rows;
        rrow[r] = have[r] ? row0 + r : row0;
    }
    uint2 w[R][NV];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int col = (i * 64 + lane) * 4;
            w[r][i] = uint2{0u, 0u};
            if (col < n_cols) w[r][i] = *reinterpret_cast<const uint2*>(x + rrow[r] * n_cols + col);
        }
    auto opaque = [&]() {   // the compiler must not keep the widened copies of one phase alive for the next
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int i = 0; i < NV; ++i) asm volatile("" : "+v"(w[r][i].x), "+v"(w[r][i].y));
    };
    float mu[R], q[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        mu[r] = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const f32x4 v = unpack4(w[r][i]);
            mu[r] += (v[0] + v[1]) + (v[2] + v[3]);
        }
    }
    wave_sum_rows(mu);
    if (R > 1) opaque();
#pragma unroll
    for (int r = 0; r < R; ++r) {
        mu[r] /= (float)n_cols;
        q[r] = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int col = (i * 64 + lane) * 4;
            if (col < n_cols) {
                const f32x4 d = unpack4(w[r][i]) - mu[r];
                q[r] += (d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]);
            }
        }
    }
    wave_sum_rows(q);
    if (R > 1) opaque();
#pragma unroll
    for (int r = 0; r < R; ++r) {
        q[r] = 1.0f / sqrtf(q[r] / (float)n_cols + eps);
        if (lane == 0 && have[r]) {
            if (mean != nullptr) mean[rrow[r]] = mu[r];
            if (rstd != nullptr) rstd[rrow[r]] = q[r];
        }
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int col = (i * 64 + lane) * 4;
        if (col < n_cols) {
            const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + col), b = *reinterpret_cast<const f32x4*>(beta + col);
#pragma unroll
            for (int r = 0; r < R; ++r)
                if (have[r]) store4(y + rrow[r] * n_cols + col, g * ((unpack4(w[r][i]) - mu[r]) * q[r]) + b);
        }
    }
}

constexpr int LN16_ROWS_PER_WAVE = 4;

template <int NV>
__global__ __launch_bounds__(256) void layernorm16_bwd_kernel(long rows, int n_cols, const unsigned short* __restrict__ dy,
                                                              const unsigned short* __restrict__ x, const float* __restrict__ mean,
                                                              const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                              unsigned short* __restrict__ dx, float* __restrict__ ws,
                                                              unsigned short* __restrict__ dxd, float drop_p, float drop_scale,
                                                              uint64_t seed_in, const uint64_t* __restrict__ epoch) {
    // dxd (optional): dx under the dropout mask of the dense layer in FRONT of this LayerNorm (the gradient its backward
    // GEMMs consume), written in the same pass - element index = row * n_cols + col as in that layer's forward epilogue
    const uint64_t seed = dxd != nullptr ? vb_seed_with_epoch(seed_in, epoch) : 0;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long row_begin = ((long)blockIdx.x * 4 + wave) * LN16_ROWS_PER_WAVE;
    f32x4 gam[NV], dg[NV], db[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int col = (i * 64 + lane) * 4;
        gam[i] = col < n_cols ? *reinterpret_cast<const f32x4*>(gamma + col) : f32x4{0.f, 0.f, 0.f, 0.f};
        dg[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        db[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    // the NEXT row's dy / x (packed) and statistics are requested before this row's two wave reductions: a wave's four rows
    // were four strictly sequential load -> reduce -> store chains (23.8 us for 42 MB at 9,216 x 768)
    uint2 ndy[NV], nx[NV];
    float nmu = 0.f, nrs = 0.f;
    auto fetch = [&](long row) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int col = (i * 64 + lane) * 4;
            ndy[i] = uint2{0u, 0u};
            nx[i] = uint2{0u, 0u};
            if (col < n_cols) {
                ndy[i] = *reinterpret_cast<const uint2*>(dy + row * n_cols + col);
                nx[i] = *reinterpret_cast<const uint2*>(x + row * n_cols + col);
            }
        }
        nmu = mean[row];
        nrs = rstd[row];
    };
    if (row_begin < rows) fetch(row_begin);
    for (int rr = 0; rr < LN16_ROWS_PER_WAVE; ++rr) {
        const long row = row_begin + rr;
        if (row >= rows) break;
        const float mu = nmu, rs = nrs;
        f32x4 xh[NV], g[NV], dcur[NV];
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            dcur[i] = unpack4(ndy[i]);
            xh[i] = unpack4(nx[i]);
        }
        if (rr + 1 < LN16_ROWS_PER_WAVE && row + 1 < rows) fetch(row + 1);
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int col = (i * 64 + lane) * 4;
            g[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (col < n_cols) {
                const f32x4 d = dcur[i];
                xh[i] = (xh[i] - mu) * rs;
                g[i] = d * gam[i];
                dg[i] += d * xh[i];
                db[i] += d;
                s1 += (g[i][0] + g[i][1]) + (g[i][2] + g[i][3]);
                s2 += (g[i][0] * xh[i][0] + g[i][1] * xh[i][1]) + (g[i][2] * xh[i][2] + g[i][3] * xh[i][3]);
            }
        }
        const float m1 = wave_sum(s1) / (float)n_cols, m2 = wave_sum(s2) / (float)n_cols;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int col = (i * 64 + lane) * 4;
            if (col < n_cols) {
                const f32x4 d = (g[i] - m1 - xh[i] * m2) * rs;
                store4(dx + row * n_cols + col, d);
                if (dxd != nullptr) {
                    f32x4 dd;
                    const uint64_t idx = (uint64_t)(row * n_cols + col);
#pragma unroll
                    for (int e = 0; e < 4; ++e) dd[e] = vb_keep(seed, idx + e, drop_p) ? d[e] * drop_scale : 0.f;
                    store4(dxd + row * n_cols + col, dd);
                }
            }
        }
    }
    __shared__ f32x4 red[3 * 2 * NV * 64];
    if (wave > 0) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            red[((wave - 1) * 2 * NV + i) * 64 + lane] = dg[i];
            red[((wave - 1) * 2 * NV + NV + i) * 64 + lane] = db[i];
        }
    }
    __syncthreads();
    if (wave > 0) return;
#pragma unroll
    for (int w2 = 0; w2 < 3; ++w2)
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            dg[i] += red[(w2 * 2 * NV + i) * 64 + lane];
            db[i] += red[(w2 * 2 * NV + NV + i) * 64 + lane];
        }
    float* w = ws + (long)blockIdx.x * 2 * n_cols;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int col = (i * 64 + lane) * 4;
        if (col < n_cols) {
            *reinterpret_cast<f32x4*>(w + col) = dg[i];
            *reinterpret_cast<f32x4*>(w + n_cols + col) = db[i];
        }
    }
}

// second stage: column c of [dgamma | dbeta] = sum over the blocks' partial rows, 64 columns per block, 16 row lanes each
__global__ __launch_bounds__(1024) void ln16_colreduce_kernel(long parts, int width, const float* __restrict__ ws,
                                                              float* __restrict__ dgamma, float* __restrict__ dbeta, int n_cols) {
    __shared__ float red[16][65];
    const int c = threadIdx.x & 63, r = threadIdx.x >> 6;
    const int col = blockIdx.x * 64 + c;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (col < width) {
        long p = r;
        for (; p + 48 < parts; p += 64) {          // four loads in flight per thread
            s0 += ws[p * width + col];
            s1 += ws[(p + 16) * width + col];
            s2 += ws[(p + 32) * width + col];
            s3 += ws[(p + 48) * width + col];
        }
        for (; p < parts; p += 16) s0 += ws[p * width + col];
    }
    red[r][c] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (r == 0 && col < width) {
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) t += red[i][c];
        if (col < n_cols) dgamma[col] = t;
        else dbeta[col - n_cols] = t;
    }
}

inline long ln16_blocks(long rows) { return (rows + 4 * LN16_ROWS_PER_WAVE - 1) / (4 * LN16_ROWS_PER_WAVE); }

}  // namespace

extern "C" int vb_layernorm_fwd_bf16(void* stream, int64_t rows, int32_t n_cols, const uint16_t* x, const float* gamma,
                                     const float* beta, float eps, uint16_t* y, float* mean, float* rstd) {
    if (x == nullptr || gamma == nullptr || beta == nullptr || y == nullptr || rows <= 0) return VB_E_BADARG;
    if (n_cols <= 0 || n_cols % 4 != 0 || n_cols > 1024) return VB_E_RANGE;
    if ((reinterpret_cast<uintptr_t>(x) & 7u) != 0 || (reinterpret_cast<uintptr_t>(y) & 7u) != 0 || !vb_aligned16(gamma) ||
        !vb_aligned16(beta))
        return VB_E_ALIGN;
    hipStream_t st = static_cast<hipStream_t>(stream);
    // rows per wave: 1 for small launches (more blocks than rows / 8 would leave CUs idle), else the smallest of 2 / 4 whose
    // blocks fit the chip in one round (8 blocks of 4 waves per CU)
    const int rpw = rows < 4096 ? 1 : (rows + 7) / 8 <= 2048 ? 2 : 4;
    const dim3 grid((unsigned)((rows + 4 * rpw - 1) / (4 * rpw))), block(256);
#define VB_LN16F(NV, R) hipLaunchKernelGGL((layernorm16_fwd_kernel<NV, R>), grid, block, 0, st, (long)rows, n_cols, x, gamma, beta, eps, y, mean, rstd)
#define VB_LN16F_R(NV) do { if (rpw == 1) VB_LN16F(NV, 1); else if (rpw == 2) VB_LN16F(NV, 2); else VB_LN16F(NV, 4); } while (0)
    switch ((n_cols + 255) / 256) {
        case 1: VB_LN16F_R(1); break;
        case 2: VB_LN16F_R(2); break;
        case 3: VB_LN16F_R(3); break;
        default: VB_LN16F_R(4); break;
    }
#undef VB_LN16F_R
#undef VB_LN16F
    VB_LAUNCH_CHECK();
    return 0;
}

extern "C" int64_t vb_layernorm_bwd_bf16_workspace(int64_t rows, int32_t n_cols) {
    return ln16_blocks(rows) * 2 * (int64_t)n_cols;
}

extern "C" int vb_layernorm_bwd_bf16(void* stream, int64_t rows, int32_t n_cols, const uint16_t* dy, const uint16_t* x,
                                     const float* mean, const float* rstd, const float* gamma, uint16_t* dx, float* dgamma,
                                     float* dbeta, float* workspace, uint16_t* dx_dropped, float dropout_p, uint64_t seed) {
    if (dy == nullptr || x == nullptr || mean == nullptr || rstd == nullptr || gamma == nullptr || dx == nullptr ||
        dgamma == nullptr || dbeta == nullptr || workspace == nullptr || rows <= 0)
        return VB_E_BADARG;
    if (n_cols <= 0 || n_cols % 4 != 0 || n_cols > 1024) return VB_E_RANGE;
    if (dx_dropped != nullptr && !(dropout_p > 0.f && dropout_p < 1.f)) return VB_E_BADARG;
    for (const void* ptr : {(const void*)dy, (const void*)x, (const void*)dx, (const void*)dx_dropped})
        if ((reinterpret_cast<uintptr_t>(ptr) & 7u) != 0) return VB_E_ALIGN;
    if (!vb_aligned16(gamma) || !vb_aligned16(workspace)) return VB_E_ALIGN;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const long blocks = ln16_blocks(rows);
    const float dscale = dx_dropped != nullptr ? 1.0f / (1.0f - dropout_p) : 1.0f;
    const uint64_t* epoch = vb_seed_epoch();
    const dim3 grid((unsigned)blocks), block(256);
#define VB_LN16B(NV) hipLaunchKernelGGL(layernorm16_bwd_kernel<NV>, grid, block, 0, st, (long)rows, n_cols, dy, x, mean, rstd, gamma, dx, \
                                        workspace, dx_dropped, dropout_p, dscale, seed, epoch)
    switch ((n_cols + 255) / 256) {
        case 1: VB_LN16B(1); break;
        case 2: VB_LN16B(2); break;
        case 3: VB_LN16B(3); break;
        default: VB_LN16B(4); break;
    }
#undef VB_LN16B
    VB_LAUNCH_CHECK();
    const int width = 2 * n_cols;
    hipLaunchKernelGGL(ln16_colreduce_kernel, dim3((unsigned)((width + 63) / 64)), dim3(1024), 0, st, blocks, width, workspace,
                       dgamma, dbeta, n_cols);
    VB_LAUNCH_CHECK();
    return 0;
}
