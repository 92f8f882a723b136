// Row kernels (HBM-bound): LayerNorm, fused text embedding gather + LayerNorm, fused image
// location projection + LayerNorm, additive-mask conversion.
//
// One 64-lane wave owns one row (768 / 1024 / 2048 columns = 3 / 4 / 8 float4 per lane, read and
// written as 16-byte coalesced accesses); the row stays in registers between the statistics
// passes, so every element is read from HBM once and written once. Statistics are two-pass
// (mean, then the mean of squared deviations) exactly as the reference computes them.
#include "common.h"
#include "mx8.h"
#include "rng.h"

namespace {

constexpr int ROWS_PER_BLOCK = 4;  // 4 waves

// y = gamma * (x - mean) * rstd + beta, TF style (biased variance, eps inside the sqrt)
// reference vilbert.py:313-317
template <int NV>
__device__ __forceinline__ void ln_finish(f32x4 (&x)[NV], int n_cols, int lane, const float* gamma,
                                          const float* beta, float eps, float* yrow, float* mean_out,
                                          float* rstd_out, float* presum_row = nullptr,
                                          unsigned char* qrow = nullptr, float* qscale = nullptr,
                                          unsigned* mx_words = nullptr, long mx_rows = 0) {
    if (presum_row != nullptr) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int col = (i * 64 + lane) * 4;
            if (col < n_cols) *reinterpret_cast<f32x4*>(presum_row + col) = x[i];
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int col = (i * 64 + lane) * 4;
        if (col < n_cols) s += (x[i][0] + x[i][1]) + (x[i][2] + x[i][3]);
    }
    const float mean = wave_sum(s) / (float)n_cols;
    float v = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int col = (i * 64 + lane) * 4;
        if (col < n_cols) {
            x[i] -= mean;
            v += (x[i][0] * x[i][0] + x[i][1] * x[i][1]) + (x[i][2] * x[i][2] + x[i][3] * x[i][3]);
        }
    }
    const float var = wave_sum(v) / (float)n_cols;
    const float rstd = 1.0f / sqrtf(var + eps);
    if (lane == 0) {
        if (mean_out != nullptr) *mean_out = mean;
        if (rstd_out != nullptr) *rstd_out = rstd;
    }
    float amax = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int col = (i * 64 + lane) * 4;
        if (col < n_cols) {
            const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + col);
            const f32x4 b = *reinterpret_cast<const f32x4*>(beta + col);
            x[i] = g * (x[i] * rstd) + b;
            *reinterpret_cast<f32x4*>(yrow + col) = x[i];
            if (qrow != nullptr)
                amax = fmaxf(amax, fmaxf(fmaxf(fabsf(x[i][0]), fabsf(x[i][1])), fmaxf(fabsf(x[i][2]), fabsf(x[i][3]))));
        }
    }
    if (qrow != nullptr && mx_words != nullptr) {
        // MX codes of the row (mx8.h): chunk i = columns 256 i .. 256 i + 255, a 32-column block = 8 consecutive lanes;
        // mx_words = scale plane base + this row, plane stride mx_rows words (bit-identical to vb_quantize_rows_mx on y)
        const int nkt = n_cols >> 7;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int col = (i * 64 + lane) * 4;
            const bool ok = col < n_cols;
            const int kt = 2 * i + (lane >> 5);
            mx_quant_chunk(x[i], ok, lane, kt, nkt, reinterpret_cast<unsigned*>(qrow + (ok ? col : 0)),
                           mx_words + (long)(kt < nkt ? kt : 0) * mx_rows);
        }
    } else if (qrow != nullptr) {
        // the row's e4m3 codes + scale for the fp8 linears that consume it (same recipe, same bits as
        // vb_quantize_rows_fp8 applied to the stored row - csrc/fp8.hip)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o));
        const bool zero = !(amax > 0.f);
        const float inv = zero ? 1.f : 448.0f / amax;
        if (lane == 0) *qscale = zero ? 1.f : amax / 448.0f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int col = (i * 64 + lane) * 4;
            if (col < n_cols) {
                int w = __builtin_amdgcn_cvt_pk_fp8_f32(x[i][0] * inv, x[i][1] * inv, 0, false);
                w = __builtin_amdgcn_cvt_pk_fp8_f32(x[i][2] * inv, x[i][3] * inv, w, true);
                *reinterpret_cast<unsigned*>(qrow + col) = (unsigned)w;
            }
        }
    }
}

template <int NV>
__global__ __launch_bounds__(256) void layernorm_kernel(long rows, int n_cols, const float* __restrict__ x,
                                                        const float* __restrict__ x2,
                                                        const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, float eps,
                                                        float* __restrict__ y, float* mean, float* rstd,
                                                        unsigned char* __restrict__ q, long ldq,
                                                        float* __restrict__ qscale, unsigned* __restrict__ mxs = nullptr,
                                                        long mxs_rows = 0) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* xr = x + row * n_cols;
    f32x4 v[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int col = (i * 64 + lane) * 4;
        v[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (col < n_cols) {
            v[i] = *reinterpret_cast<const f32x4*>(xr + col);
            if (x2 != nullptr) v[i] += *reinterpret_cast<const f32x4*>(x2 + row * n_cols + col);
        }
    }
    ln_finish<NV>(v, n_cols, lane, gamma, beta, eps, y + row * n_cols, mean ? mean + row : nullptr,
                  rstd ? rstd + row : nullptr, nullptr, q ? q + row * ldq : nullptr, (q && qscale) ? qscale + row : nullptr,
                  mxs ? mxs + row : nullptr, mxs_rows);
}

// reference vilbert.py:346-367
template <int NV>
__global__ __launch_bounds__(256) void text_embed_kernel(int batch, int n_tok, int hidden, int vocab, int n_types,
                                                         int n_tasks, const int64_t* __restrict__ ids,
                                                         const int64_t* __restrict__ seg, int pos_offset,
                                                         const float* __restrict__ word,
                                                         const float* __restrict__ pos,
                                                         const float* __restrict__ type,
                                                         const int64_t* __restrict__ task_ids,
                                                         const float* __restrict__ task_emb,
                                                         const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, float eps,
                                                         float* __restrict__ out, float* mean, float* rstd,
                                                         float* presum) {
    const int lane = threadIdx.x & 63;
    const int n_out = n_tok + (task_ids != nullptr ? 1 : 0);
    const long row = (long)blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6);
    if (row >= (long)batch * n_out) return;
    const int b = (int)(row / n_out), t_out = (int)(row % n_out);
    // with task tokens: output 0 <- token 0, output 1 <- task embedding, output t <- token t - 1
    const bool is_task = task_ids != nullptr && t_out == 1;
    const int t = (task_ids != nullptr && t_out >= 2) ? t_out - 1 : t_out;
    // ids outside their table (the reference's nn.Embedding raises a device assert there) read nothing: the
    // row contributes zeros instead of whatever lies past the allocation
    const float *w = nullptr, *pp = nullptr, *ty = nullptr;
    if (is_task) {
        const int64_t k = task_ids[b];
        if (k >= 0 && k < n_tasks) w = task_emb + k * hidden;
    } else {
        const int64_t id = ids[(long)b * n_tok + t], sg = seg[(long)b * n_tok + t];
        if (id >= 0 && id < vocab) w = word + id * hidden;
        pp = pos + (long)(t + pos_offset) * hidden;
        if (sg >= 0 && sg < n_types) ty = type + sg * hidden;
    }
    f32x4 v[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int col = (i * 64 + lane) * 4;
        v[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (col < hidden) {
            if (w != nullptr) v[i] = *reinterpret_cast<const f32x4*>(w + col);
            if (!is_task) {
                // words + position + token_type, in the reference's order (vilbert.py:355)
                v[i] += *reinterpret_cast<const f32x4*>(pp + col);
                if (ty != nullptr) v[i] += *reinterpret_cast<const f32x4*>(ty + col);
            }
        }
    }
    ln_finish<NV>(v, hidden, lane, gamma, beta, eps, out + row * hidden, mean ? mean + row : nullptr,
                  rstd ? rstd + row : nullptr, presum ? presum + row * hidden : nullptr);
}

// reference vilbert.py:1421-1432 (the 5 -> hidden location projection, the sum and the LayerNorm)
template <int NV>
__global__ __launch_bounds__(256) void image_embed_kernel(long rows, int hidden,
                                                          const float* __restrict__ feat_proj,
                                                          const float* __restrict__ loc,
                                                          const float* __restrict__ w_loc,
                                                          const float* __restrict__ b_loc,
                                                          const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, float eps,
                                                          float* __restrict__ out, float* mean, float* rstd,
                                                          float* presum) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6);
    if (row >= rows) return;
    float l[5];
#pragma unroll
    for (int j = 0; j < 5; ++j) l[j] = loc[row * 5 + j];
    f32x4 v[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int col = (i * 64 + lane) * 4;
        v[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (col < hidden) {
            f32x4 lp = *reinterpret_cast<const f32x4*>(b_loc + col);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float* wr = w_loc + (long)(col + e) * 5;
                float a = 0.f;
#pragma unroll
                for (int j = 0; j < 5; ++j) a = fmaf(l[j], wr[j], a);
                lp[e] += a;
            }
            v[i] = *reinterpret_cast<const f32x4*>(feat_proj + row * hidden + col) + lp;
        }
    }
    ln_finish<NV>(v, hidden, lane, gamma, beta, eps, out + row * hidden, mean ? mean + row : nullptr,
                  rstd ? rstd + row : nullptr, presum ? presum + row * hidden : nullptr);
}

template <typename T>
__global__ void additive_mask_kernel(long n, const T* __restrict__ mask, float* __restrict__ out) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    // reference vilbert.py:1353,1362: (1.0 - mask) * -10000.0
    if (i < n) out[i] = (1.0f - (float)mask[i]) * -10000.0f;
}

inline int nv_for(int n_cols) { return (n_cols + 255) / 256; }

#define VB_NV_DISPATCH(nv, CALL)             \
    switch (nv) {                            \
        case 1: { constexpr int NV = 1; CALL; } break; \
        case 2: { constexpr int NV = 2; CALL; } break; \
        case 3: { constexpr int NV = 3; CALL; } break; \
        case 4: { constexpr int NV = 4; CALL; } break; \
        case 5: case 6: case 7: case 8: { constexpr int NV = 8; CALL; } break; \
        case 9: case 10: case 11: case 12: case 13: case 14: case 15: case 16: \
                { constexpr int NV = 16; CALL; } break; \
        default: { constexpr int NV = 32; CALL; } break; \
    }

inline int check_cols(int n_cols) {
    if (n_cols <= 0) return VB_E_BADARG;
    if (n_cols % 4 != 0) return VB_E_ALIGN;
    if (n_cols > VB_MAX_LN_COLS) return VB_E_RANGE;
    return 0;
}

}  // namespace

extern "C" int vb_layernorm_fwd(void* stream, int64_t rows, int32_t n_cols, const float* x, const float* x2,
                                const float* gamma, const float* beta, float eps, float* y, float* mean,
                                float* rstd) {
    if (x == nullptr || gamma == nullptr || beta == nullptr || y == nullptr || rows <= 0) return VB_E_BADARG;
    if (int e = check_cols(n_cols)) return e;
    if (!vb_aligned16(x) || !vb_aligned16(y) || !vb_aligned16(gamma) || !vb_aligned16(beta) ||
        (x2 != nullptr && !vb_aligned16(x2)))
        return VB_E_ALIGN;
    hipStream_t st = static_cast<hipStream_t>(stream);
    dim3 grid((unsigned)((rows + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK)), block(256);
    VB_NV_DISPATCH(nv_for(n_cols), hipLaunchKernelGGL((layernorm_kernel<NV>), grid, block, 0, st, (long)rows,
                                                      n_cols, x, x2, gamma, beta, eps, y, mean, rstd,
                                                      (unsigned char*)nullptr, 0L, (float*)nullptr));
    VB_LAUNCH_CHECK();
    return 0;
}

// LayerNorm forward that also emits the row's e4m3 codes + scale (inference in fp8 mode: the consumer GEMMs read the
// codes, the fp32 row stays for the residual path). Bit-identical to vb_layernorm_fwd followed by vb_quantize_rows_fp8.
extern "C" int vb_layernorm_fwd_fp8(void* stream, int64_t rows, int32_t n_cols, const float* x, const float* x2,
                                    const float* gamma, const float* beta, float eps, float* y, uint8_t* q,
                                    int64_t ldq, float* qscale) {
    if (x == nullptr || gamma == nullptr || beta == nullptr || y == nullptr || q == nullptr || qscale == nullptr ||
        rows <= 0)
        return VB_E_BADARG;
    if (int e = check_cols(n_cols)) return e;
    if (!vb_aligned16(x) || !vb_aligned16(y) || !vb_aligned16(gamma) || !vb_aligned16(beta) ||
        (x2 != nullptr && !vb_aligned16(x2)) || ldq < n_cols || ldq % 4 != 0 || (reinterpret_cast<uintptr_t>(q) & 3u) != 0)
        return VB_E_ALIGN;
    hipStream_t st = static_cast<hipStream_t>(stream);
    dim3 grid((unsigned)((rows + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK)), block(256);
    VB_NV_DISPATCH(nv_for(n_cols), hipLaunchKernelGGL((layernorm_kernel<NV>), grid, block, 0, st, (long)rows,
                                                      n_cols, x, x2, gamma, beta, eps, y, (float*)nullptr,
                                                      (float*)nullptr, q, (long)ldq, qscale));
    VB_LAUNCH_CHECK();
    return 0;
}

// LayerNorm forward that also emits its output rows in the MX e4m3 format (mx8.hip) for the linears consuming it; the
// fp32 row stays for the residual path. Bit-identical to vb_layernorm_fwd followed by vb_quantize_rows_mx on y.
extern "C" int vb_layernorm_fwd_mx(void* stream, int64_t rows, int32_t n_cols, const float* x, const float* x2,
                                   const float* gamma, const float* beta, float eps, float* y, uint8_t* q, int64_t ldq,
                                   uint32_t* scales, int64_t scale_rows) {
    if (x == nullptr || gamma == nullptr || beta == nullptr || y == nullptr || q == nullptr || scales == nullptr || rows <= 0)
        return VB_E_BADARG;
    if (int e = check_cols(n_cols)) return e;
    if (n_cols % 128 != 0 || scale_rows < rows) return VB_E_RANGE;
    if (!vb_aligned16(x) || !vb_aligned16(y) || !vb_aligned16(gamma) || !vb_aligned16(beta) ||
        (x2 != nullptr && !vb_aligned16(x2)) || ldq < n_cols || ldq % 4 != 0 || (reinterpret_cast<uintptr_t>(q) & 3u) != 0 ||
        (reinterpret_cast<uintptr_t>(scales) & 3u) != 0)
        return VB_E_ALIGN;
    hipStream_t st = static_cast<hipStream_t>(stream);
    dim3 grid((unsigned)((rows + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK)), block(256);
    VB_NV_DISPATCH(nv_for(n_cols), hipLaunchKernelGGL((layernorm_kernel<NV>), grid, block, 0, st, (long)rows,
                                                      n_cols, x, x2, gamma, beta, eps, y, (float*)nullptr,
                                                      (float*)nullptr, q, (long)ldq, (float*)nullptr, scales, (long)scale_rows));
    VB_LAUNCH_CHECK();
    return 0;
}

namespace {
// LayerNorm of the MX path's bf16 residual stream: bf16 row in, bf16 row + MX codes out (lane = 4 consecutive columns of
// every 256-column chunk, as layernorm_kernel). A wave owns TWO rows and runs their dependent chains (load -> sum -> wave
// reduction -> squared deviations -> wave reduction -> codes) side by side: with one row per wave the launch was bound by
// that chain's latency times the number of block rounds (29.5 us for 70 MB at 18,432 x 768), not by HBM. LN16_RPW = 2 or 4:
// the launcher picks 4 when two rows per wave would need more blocks than fit the chip at once (8 per CU) - at batch 512
// (18,432 rows) the 2,304 blocks of the two-row form ran as one full round plus a 12 % tail round of the same latency.
template <int R>
__device__ __forceinline__ void wave_sum_n(float (&a)[R]) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        float t[R];
#pragma unroll
        for (int r = 0; r < R; ++r) t[r] = __shfl_xor(a[r], off, 64);
#pragma unroll
        for (int r = 0; r < R; ++r) a[r] += t[r];
    }
}

template <int NV, int LN16_RPW>
__global__ __launch_bounds__(256) void layernorm16_mx_kernel(long rows, int n_cols, const unsigned short* __restrict__ x,
                                                             const float* __restrict__ gamma, const float* __restrict__ beta,
                                                             float eps, unsigned short* __restrict__ y,
                                                             unsigned char* __restrict__ q, long ldq, unsigned* __restrict__ mxs,
                                                             long mxs_rows) {
    const int lane = threadIdx.x & 63;
    const long row0 = ((long)blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6)) * LN16_RPW;
    if (row0 >= rows) return;
    long rrow[LN16_RPW];                                // rows past the end: row0 again, every store masked
    bool have[LN16_RPW];                                // (wave-uniform)
#pragma unroll
    for (int r = 0; r < LN16_RPW; ++r) {
        have[r] = row0 + r < rows;
        rrow[r] = have[r] ? row0 + r : row0;
    }
    // the rows stay PACKED in registers (2 per 4 values) and are widened wherever they are used: four rows of 1,024 columns
    // in 32 registers keep the occupancy at 8 waves per SIMD (as fp32 they took 64 and the kernel 112)
    uint2 w[LN16_RPW][NV];
    auto wide = [](const uint2 t) {
        return f32x4{__uint_as_float(t.x << 16), __uint_as_float(t.x & 0xffff0000u), __uint_as_float(t.y << 16),
                     __uint_as_float(t.y & 0xffff0000u)};
    };
#pragma unroll
    for (int r = 0; r < LN16_RPW; ++r)
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int col = (i * 64 + lane) * 4;
            w[r][i] = uint2{0u, 0u};
            if (col < n_cols) w[r][i] = *reinterpret_cast<const uint2*>(x + rrow[r] * n_cols + col);
        }
    float mean[LN16_RPW], var[LN16_RPW];
#pragma unroll
    for (int r = 0; r < LN16_RPW; ++r) {
        mean[r] = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const f32x4 v = wide(w[r][i]);
            mean[r] += (v[0] + v[1]) + (v[2] + v[3]);   // (columns past n_cols hold 0)
        }
    }
    wave_sum_n(mean);
    auto opaque = [&]() {   // the compiler must not keep the widened copies of one phase for the next (that is the 64 registers)
#pragma unroll
        for (int r = 0; r < LN16_RPW; ++r)
#pragma unroll
            for (int i = 0; i < NV; ++i) asm volatile("" : "+v"(w[r][i].x), "+v"(w[r][i].y));
    };
    opaque();
#pragma unroll
    for (int r = 0; r < LN16_RPW; ++r) {
        mean[r] /= (float)n_cols;
        var[r] = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int col = (i * 64 + lane) * 4;
            if (col < n_cols) {
                const f32x4 d = wide(w[r][i]) - mean[r];
                var[r] += (d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]);
            }
        }
    }
    wave_sum_n(var);
    opaque();
    const int nkt = n_cols >> 7;
    auto bf = [](float f) -> unsigned { const unsigned u = __float_as_uint(f); return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16; };
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int col = (i * 64 + lane) * 4;
        const bool ok = col < n_cols;
        f32x4 g = f32x4{0.f, 0.f, 0.f, 0.f}, b = f32x4{0.f, 0.f, 0.f, 0.f};
        if (ok) {
            g = *reinterpret_cast<const f32x4*>(gamma + col);
            b = *reinterpret_cast<const f32x4*>(beta + col);
        }
        const int kt = 2 * i + (lane >> 5);
#pragma unroll
        for (int r = 0; r < LN16_RPW; ++r) {
            const float rstd = 1.0f / sqrtf(var[r] / (float)n_cols + eps);
            const bool live = ok && have[r];
            f32x4 v = wide(w[r][i]);
            if (ok) v = g * ((v - mean[r]) * rstd) + b;
            if (live)
                *reinterpret_cast<uint2*>(y + rrow[r] * n_cols + col) =
                    uint2{bf(v[0]) | (bf(v[1]) << 16), bf(v[2]) | (bf(v[3]) << 16)};
            // (the cross-lane steps inside run for every lane; the stores of the rows that do not exist are masked)
            mx_quant_chunk(v, live, lane, have[r] ? kt : nkt, nkt, reinterpret_cast<unsigned*>(q + rrow[r] * ldq + (ok ? col : 0)),
                           mxs + (long)(kt < nkt ? kt : 0) * mxs_rows + rrow[r]);
        }
    }
}
}  // namespace

extern "C" int vb_layernorm_fwd_mx16(void* stream, int64_t rows, int32_t n_cols, const uint16_t* x, const float* gamma,
                                     const float* beta, float eps, uint16_t* y, uint8_t* q, int64_t ldq, uint32_t* scales,
                                     int64_t scale_rows) {
    if (x == nullptr || gamma == nullptr || beta == nullptr || y == nullptr || q == nullptr || scales == nullptr || rows <= 0)
        return VB_E_BADARG;
    if (int e = check_cols(n_cols)) return e;
    if (n_cols % 128 != 0 || scale_rows < rows) return VB_E_RANGE;
    if ((reinterpret_cast<uintptr_t>(x) & 7u) != 0 || (reinterpret_cast<uintptr_t>(y) & 7u) != 0 || !vb_aligned16(gamma) ||
        !vb_aligned16(beta) || ldq < n_cols || ldq % 4 != 0 || (reinterpret_cast<uintptr_t>(q) & 3u) != 0 ||
        (reinterpret_cast<uintptr_t>(scales) & 3u) != 0)
        return VB_E_ALIGN;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const dim3 block(256);
    if ((rows + 2 * ROWS_PER_BLOCK - 1) / (2 * ROWS_PER_BLOCK) <= 2048 || n_cols > 1024) {
        const dim3 grid((unsigned)((rows + 2 * ROWS_PER_BLOCK - 1) / (2 * ROWS_PER_BLOCK)));
        VB_NV_DISPATCH(nv_for(n_cols), hipLaunchKernelGGL((layernorm16_mx_kernel<NV, 2>), grid, block, 0, st, (long)rows, n_cols, x,
                                                          gamma, beta, eps, y, q, (long)ldq, scales, (long)scale_rows));
    } else {   // (rows of at most 1,024 columns: 4 x 4 float4 registers per lane)
        const dim3 grid((unsigned)((rows + 4 * ROWS_PER_BLOCK - 1) / (4 * ROWS_PER_BLOCK)));
        switch (nv_for(n_cols)) {
#define VB_LN16_4(NV) hipLaunchKernelGGL((layernorm16_mx_kernel<NV, 4>), grid, block, 0, st, (long)rows, n_cols, x, gamma, beta, eps, y, \
                                         q, (long)ldq, scales, (long)scale_rows)
            case 1: VB_LN16_4(1); break;
            case 2: VB_LN16_4(2); break;
            case 3: VB_LN16_4(3); break;
            default: VB_LN16_4(4); break;
#undef VB_LN16_4
        }
    }
    VB_LAUNCH_CHECK();
    return 0;
}

extern "C" int vb_text_embed_ln_fwd(void* stream, int32_t batch, int32_t n_tok, int32_t hidden, int32_t vocab,
                                    int32_t n_types, int32_t n_tasks,
                                    const int64_t* ids, const int64_t* seg, int32_t pos_offset,
                                    const float* word_emb, const float* pos_emb, const float* type_emb,
                                    const int64_t* task_ids, const float* task_emb, const float* gamma,
                                    const float* beta, float eps, float* out, float* mean, float* rstd,
                                    float* presum) {
    if (ids == nullptr || seg == nullptr || word_emb == nullptr || pos_emb == nullptr || type_emb == nullptr ||
        gamma == nullptr || beta == nullptr || out == nullptr || batch <= 0 || n_tok <= 0 || vocab <= 0 ||
        n_types <= 0)
        return VB_E_BADARG;
    if (task_ids != nullptr && (task_emb == nullptr || n_tasks <= 0)) return VB_E_BADARG;
    if (int e = check_cols(hidden)) return e;
    if (!vb_aligned16(word_emb) || !vb_aligned16(pos_emb) || !vb_aligned16(type_emb) || !vb_aligned16(out) ||
        !vb_aligned16(gamma) || !vb_aligned16(beta) || (task_emb != nullptr && !vb_aligned16(task_emb)))
        return VB_E_ALIGN;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const long rows = (long)batch * (n_tok + (task_ids != nullptr ? 1 : 0));
    dim3 grid((unsigned)((rows + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK)), block(256);
    VB_NV_DISPATCH(nv_for(hidden),
                   hipLaunchKernelGGL((text_embed_kernel<NV>), grid, block, 0, st, batch, n_tok, hidden, vocab, n_types,
                                      n_tasks, ids, seg,
                                      pos_offset, word_emb, pos_emb, type_emb, task_ids, task_emb, gamma, beta,
                                      eps, out, mean, rstd, presum));
    VB_LAUNCH_CHECK();
    return 0;
}

extern "C" int vb_image_embed_ln_fwd(void* stream, int64_t rows, int32_t hidden, const float* feat_proj,
                                     const float* loc, const float* w_loc, const float* b_loc,
                                     const float* gamma, const float* beta, float eps, float* out, float* mean,
                                     float* rstd, float* presum) {
    if (feat_proj == nullptr || loc == nullptr || w_loc == nullptr || b_loc == nullptr || gamma == nullptr ||
        beta == nullptr || out == nullptr || rows <= 0)
        return VB_E_BADARG;
    if (int e = check_cols(hidden)) return e;
    if (!vb_aligned16(feat_proj) || !vb_aligned16(out) || !vb_aligned16(b_loc) || !vb_aligned16(gamma) ||
        !vb_aligned16(beta))
        return VB_E_ALIGN;
    hipStream_t st = static_cast<hipStream_t>(stream);
    dim3 grid((unsigned)((rows + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK)), block(256);
    VB_NV_DISPATCH(nv_for(hidden),
                   hipLaunchKernelGGL((image_embed_kernel<NV>), grid, block, 0, st, (long)rows, hidden, feat_proj,
                                      loc, w_loc, b_loc, gamma, beta, eps, out, mean, rstd, presum));
    VB_LAUNCH_CHECK();
    return 0;
}

extern "C" int vb_additive_mask(void* stream, int64_t n, const void* mask, int32_t mask_is_f32, float* out) {
    if (mask == nullptr || out == nullptr || n <= 0) return VB_E_BADARG;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const dim3 grid((unsigned)((n + 255) / 256)), block(256);
    if (mask_is_f32)
        hipLaunchKernelGGL(additive_mask_kernel<float>, grid, block, 0, st, (long)n,
                           static_cast<const float*>(mask), out);
    else
        hipLaunchKernelGGL(additive_mask_kernel<int64_t>, grid, block, 0, st, (long)n,
                           static_cast<const int64_t*>(mask), out);
    VB_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------------
// LayerNorm backward. Per row (xhat = (x - mean) rstd, g = dy * gamma):
//   dx = rstd * (g - mean(g) - xhat * mean(g * xhat))
//   dgamma = sum_rows dy * xhat,  dbeta = sum_rows dy
// Stage 1: one wave walks LNB_ROWS_PER_WAVE rows keeping its column partials of dgamma / dbeta in
// registers; the four waves of a block combine theirs through LDS (rows of <= 1024 columns) and write ONE
// registers and writes them to a workspace row; stage 2 sums the workspace rows column-wise. No
// atomics: the result is deterministic.
// ------------------------------------------------------------------------------------------------
namespace {

constexpr int LNB_ROWS_PER_WAVE = 4;   // few rows per wave: 9216 rows -> 2304 waves keep the chip's 1024 SIMDs busy
constexpr int LNB_ROWS_PER_BLOCK = 4 * LNB_ROWS_PER_WAVE;

template <int NV>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(long rows, int n_cols, const float* __restrict__ dy,
                                                            const float* __restrict__ x,
                                                            const float* __restrict__ mean,
                                                            const float* __restrict__ rstd,
                                                            const float* __restrict__ gamma,
                                                            float* __restrict__ dx, float* __restrict__ ws,
                                                            float* __restrict__ dxd, float drop_p, float drop_scale,
                                                            uint64_t seed_in, const uint64_t* __restrict__ epoch) {
    // dxd (optional): dx with the dropout mask of the layer in FRONT of this LayerNorm applied - the gradient that layer's
    // backward GEMMs consume - written in the same pass (saves the separate vb_dropout launch over dx)
    const uint64_t seed = dxd != nullptr ? vb_seed_with_epoch(seed_in, epoch) : 0;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long part = (long)blockIdx.x * 4 + wave;
    const long row_begin = part * LNB_ROWS_PER_WAVE;
    f32x4 gam[NV], dg[NV], db[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int col = (i * 64 + lane) * 4;
        gam[i] = col < n_cols ? *reinterpret_cast<const f32x4*>(gamma + col) : f32x4{0.f, 0.f, 0.f, 0.f};
        dg[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        db[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    for (int rr = 0; rr < LNB_ROWS_PER_WAVE; ++rr) {
        const long row = row_begin + rr;
        if (row >= rows) break;
        const float mu = mean[row], rs = rstd[row];
        f32x4 xh[NV], g[NV];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int col = (i * 64 + lane) * 4;
            xh[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            g[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (col < n_cols) {
                const f32x4 d = *reinterpret_cast<const f32x4*>(dy + row * n_cols + col);
                xh[i] = (*reinterpret_cast<const f32x4*>(x + row * n_cols + col) - mu) * rs;
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
                *reinterpret_cast<f32x4*>(dx + row * n_cols + col) = d;
                if (dxd != nullptr) {
                    f32x4 dd;
                    const uint64_t idx = (uint64_t)(row * n_cols + col);
#pragma unroll
                    for (int e = 0; e < 4; ++e) dd[e] = vb_keep(seed, idx + e, drop_p) ? d[e] * drop_scale : 0.f;
                    *reinterpret_cast<f32x4*>(dxd + row * n_cols + col) = dd;
                }
            }
        }
    }
    constexpr bool BLOCK_REDUCE = NV <= 4;
    __shared__ f32x4 red[BLOCK_REDUCE ? 3 * 2 * NV * 64 : 1];
    if (BLOCK_REDUCE) {
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
    }
    float* w = ws + (BLOCK_REDUCE ? (long)blockIdx.x : part) * 2 * n_cols;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int col = (i * 64 + lane) * 4;
        if (col < n_cols) {
            *reinterpret_cast<f32x4*>(w + col) = dg[i];
            *reinterpret_cast<f32x4*>(w + n_cols + col) = db[i];
        }
    }
}

// Rows wider than 4096 columns (nothing in the two-stream models; the C ABI allows up to VB_MAX_LN_COLS): one wave per
// row, the row walked twice in 256-column chunks (pass 1: the two row means, pass 2: dx and this row's dgamma / dbeta
// terms, written straight to the workspace: one partial per row) - no per-column accumulators in registers, so no
// scratch (the register-resident variant above spilled 940 bytes per lane at NV = 32).
__global__ __launch_bounds__(256) void layernorm_bwd_wide_kernel(long rows, int n_cols, const float* __restrict__ dy,
                                                                 const float* __restrict__ x,
                                                                 const float* __restrict__ mean,
                                                                 const float* __restrict__ rstd,
                                                                 const float* __restrict__ gamma,
                                                                 float* __restrict__ dx, float* __restrict__ ws) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float mu = mean[row], rs = rstd[row];
    const float* __restrict__ dyr = dy + row * n_cols;
    const float* __restrict__ xr = x + row * n_cols;
    float s1 = 0.f, s2 = 0.f;
    for (int col = lane * 4; col < n_cols; col += 256) {
        const f32x4 d = *reinterpret_cast<const f32x4*>(dyr + col);
        const f32x4 xh = (*reinterpret_cast<const f32x4*>(xr + col) - mu) * rs;
        const f32x4 g = d * *reinterpret_cast<const f32x4*>(gamma + col);
        s1 += (g[0] + g[1]) + (g[2] + g[3]);
        s2 += (g[0] * xh[0] + g[1] * xh[1]) + (g[2] * xh[2] + g[3] * xh[3]);
    }
    const float m1 = wave_sum(s1) / (float)n_cols, m2 = wave_sum(s2) / (float)n_cols;
    float* __restrict__ w = ws + row * 2 * n_cols;
    for (int col = lane * 4; col < n_cols; col += 256) {
        const f32x4 d = *reinterpret_cast<const f32x4*>(dyr + col);
        const f32x4 xh = (*reinterpret_cast<const f32x4*>(xr + col) - mu) * rs;
        const f32x4 g = d * *reinterpret_cast<const f32x4*>(gamma + col);
        *reinterpret_cast<f32x4*>(dx + row * n_cols + col) = (g - m1 - xh * m2) * rs;
        *reinterpret_cast<f32x4*>(w + col) = d * xh;
        *reinterpret_cast<f32x4*>(w + n_cols + col) = d;
    }
}

// Column sums of `parts` workspace rows of width 2 * n_cols. Block = 1024 threads = 16 row groups x
// 64 columns; each group strides over the parts, then an LDS tree over the 16 groups.
__global__ __launch_bounds__(1024) void colreduce_kernel(long parts, int width, const float* __restrict__ ws,
                                                         float* __restrict__ out0, float* __restrict__ out1,
                                                         int n_cols) {
    __shared__ float red[16][64];
    const int c = threadIdx.x & 63, pg = threadIdx.x >> 6;
    const int col = blockIdx.x * 64 + c;
    float acc = 0.f;
    if (col < width)
        for (long r = pg; r < parts; r += 16) acc += ws[r * width + col];
    red[pg][c] = acc;
    __syncthreads();
    if (pg == 0 && col < width) {
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) t += red[i][c];
        if (col < n_cols) out0[col] = t;
        else out1[col - n_cols] = t;
    }
}

// Scatter-add of embedding-row gradients (fp32 atomics into the zero-filled tables).
// reference vilbert.py:353-362 backward; word row 0 is padding_idx (no gradient from the gather).
template <int NV>
__global__ __launch_bounds__(256) void text_embed_scatter_kernel(int batch, int n_tok, int hidden, int vocab,
                                                                 int n_tasks, const int64_t* __restrict__ ids,
                                                                 const int64_t* __restrict__ seg,
                                                                 const int64_t* __restrict__ task_ids,
                                                                 const float* __restrict__ dx,
                                                                 float* __restrict__ dword, float* __restrict__ dpos,
                                                                 float* __restrict__ dtype, float* __restrict__ dtask) {
    const int lane = threadIdx.x & 63;
    const int n_out = n_tok + (task_ids != nullptr ? 1 : 0);
    const long row = (long)blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6);
    if (row >= (long)batch * n_out) return;
    const int b = (int)(row / n_out), t_out = (int)(row % n_out);
    const bool is_task = task_ids != nullptr && t_out == 1;
    const int t = (task_ids != nullptr && t_out >= 2) ? t_out - 1 : t_out;
    float* w = nullptr;
    if (is_task) {
        const int64_t k = task_ids[b];
        w = (k >= 0 && k < n_tasks) ? dtask + k * hidden : nullptr;
    } else {
        const int64_t id = ids[(long)b * n_tok + t];
        w = (id > 0 && id < vocab) ? dword + id * hidden : nullptr;
        // position / token-type rows are shared by every sample (36 + 2 rows for 9216 tokens): they are
        // reduced by pos_type_grad_kernel instead of 9216-way contended atomics
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int col = (i * 64 + lane) * 4;
        if (col < hidden) {
            const f32x4 d = *reinterpret_cast<const f32x4*>(dx + row * hidden + col);
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (w != nullptr) unsafeAtomicAdd(w + col + e, d[e]);
        }
    }
}

// dpos[t] += sum_b dx[b, t] and dtype[s] += sum over the tokens of type s: one block per token
// position walks the batch (coalesced rows), so dpos needs no atomics and dtype one per position.
__global__ __launch_bounds__(256) void pos_type_grad_kernel(int batch, int n_tok, int hidden, int n_types,
                                                            const int64_t* __restrict__ seg,
                                                            const int64_t* __restrict__ task_ids,
                                                            const float* __restrict__ dx,
                                                            float* __restrict__ dpos, float* __restrict__ dtype) {
    const int n_out = n_tok + (task_ids != nullptr ? 1 : 0);
    const int t_out = blockIdx.x;
    if (task_ids != nullptr && t_out == 1) return;  // the task-token row has no position / type
    const int t = (task_ids != nullptr && t_out >= 2) ? t_out - 1 : t_out;
    for (int col = threadIdx.x * 4; col < hidden; col += 256 * 4) {
        f32x4 ap = {0.f, 0.f, 0.f, 0.f}, a0 = ap, a1 = ap;
        for (int b = 0; b < batch; ++b) {
            const f32x4 d = *reinterpret_cast<const f32x4*>(dx + ((long)b * n_out + t_out) * hidden + col);
            const int64_t ty = seg[(long)b * n_tok + t];
            ap += d;
            if (ty == 0) a0 += d;
            else if (ty == 1) a1 += d;
            else if (ty > 1 && ty < n_types) {
#pragma unroll
                for (int e = 0; e < 4; ++e) unsafeAtomicAdd(dtype + ty * hidden + col + e, d[e]);
            }
        }
        // the type table may hold a single row (roberta_base_6layer_6connect.json: type_vocab_size = 1)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            unsafeAtomicAdd(dpos + (long)t * hidden + col + e, ap[e]);
            if (n_types > 0) unsafeAtomicAdd(dtype + col + e, a0[e]);
            if (n_types > 1) unsafeAtomicAdd(dtype + hidden + col + e, a1[e]);
        }
    }
}

}  // namespace

extern "C" int64_t vb_layernorm_bwd_workspace(int64_t rows, int32_t n_cols) {
    if (rows <= 0 || n_cols <= 0) return 0;
    if (nv_for(n_cols) > 16) return rows * 2 * n_cols;   // wide rows: one partial per row (layernorm_bwd_wide_kernel)
    const int64_t parts = (rows + LNB_ROWS_PER_WAVE - 1) / LNB_ROWS_PER_WAVE;
    const int64_t parts_padded = (parts + 3) / 4 * 4;  // whole blocks write
    return parts_padded * 2 * n_cols;
}

namespace {
int layernorm_bwd_impl(void* stream, int64_t rows, int32_t n_cols, const float* dy, const float* x, const float* mean,
                       const float* rstd, const float* gamma, float* dx, float* dgamma, float* dbeta, float* workspace,
                       float* dxd, float drop_p, uint64_t seed) {
    if (dy == nullptr || x == nullptr || mean == nullptr || rstd == nullptr || gamma == nullptr || dx == nullptr ||
        dgamma == nullptr || dbeta == nullptr || workspace == nullptr || rows <= 0)
        return VB_E_BADARG;
    if (int e = check_cols(n_cols)) return e;
    if (!vb_aligned16(dy) || !vb_aligned16(x) || !vb_aligned16(dx) || !vb_aligned16(gamma) || !vb_aligned16(workspace))
        return VB_E_ALIGN;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const long blocks = (rows + LNB_ROWS_PER_BLOCK - 1) / LNB_ROWS_PER_BLOCK;
    const int width = 2 * n_cols;
    long parts = nv_for(n_cols) <= 4 ? blocks : blocks * 4;  // one partial per block / per wave
    const float dscale = dxd != nullptr ? 1.0f / (1.0f - drop_p) : 1.0f;
    const uint64_t* epoch = dxd != nullptr ? vb_seed_epoch() : nullptr;
    if (nv_for(n_cols) > 16) {
        if (dxd != nullptr) return VB_E_RANGE;    // (the fused mask is not offered for rows wider than 4096)
        hipLaunchKernelGGL(layernorm_bwd_wide_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, (long)rows, n_cols, dy,
                           x, mean, rstd, gamma, dx, workspace);
        parts = rows;
    } else {
        switch (nv_for(n_cols)) {
            case 1: hipLaunchKernelGGL((layernorm_bwd_kernel<1>), dim3((unsigned)blocks), dim3(256), 0, st, (long)rows, n_cols, dy, x, mean, rstd, gamma, dx, workspace, dxd, drop_p, dscale, seed, epoch); break;
            case 2: hipLaunchKernelGGL((layernorm_bwd_kernel<2>), dim3((unsigned)blocks), dim3(256), 0, st, (long)rows, n_cols, dy, x, mean, rstd, gamma, dx, workspace, dxd, drop_p, dscale, seed, epoch); break;
            case 3: hipLaunchKernelGGL((layernorm_bwd_kernel<3>), dim3((unsigned)blocks), dim3(256), 0, st, (long)rows, n_cols, dy, x, mean, rstd, gamma, dx, workspace, dxd, drop_p, dscale, seed, epoch); break;
            case 4: hipLaunchKernelGGL((layernorm_bwd_kernel<4>), dim3((unsigned)blocks), dim3(256), 0, st, (long)rows, n_cols, dy, x, mean, rstd, gamma, dx, workspace, dxd, drop_p, dscale, seed, epoch); break;
            case 5: case 6: case 7: case 8:
                hipLaunchKernelGGL((layernorm_bwd_kernel<8>), dim3((unsigned)blocks), dim3(256), 0, st, (long)rows, n_cols, dy, x, mean, rstd, gamma, dx, workspace, dxd, drop_p, dscale, seed, epoch); break;
            default:
                hipLaunchKernelGGL((layernorm_bwd_kernel<16>), dim3((unsigned)blocks), dim3(256), 0, st, (long)rows, n_cols, dy, x, mean, rstd, gamma, dx, workspace, dxd, drop_p, dscale, seed, epoch); break;
        }
    }
    VB_LAUNCH_CHECK();
    hipLaunchKernelGGL(colreduce_kernel, dim3((unsigned)((width + 63) / 64)), dim3(1024), 0, st, parts, width,
                       workspace, dgamma, dbeta, n_cols);
    VB_LAUNCH_CHECK();
    return 0;
}
}  // namespace

extern "C" int vb_layernorm_bwd(void* stream, int64_t rows, int32_t n_cols, const float* dy, const float* x,
                                const float* mean, const float* rstd, const float* gamma, float* dx,
                                float* dgamma, float* dbeta, float* workspace) {
    return layernorm_bwd_impl(stream, rows, n_cols, dy, x, mean, rstd, gamma, dx, dgamma, dbeta, workspace, nullptr, 0.f, 0);
}

extern "C" int vb_layernorm_bwd_drop(void* stream, int64_t rows, int32_t n_cols, const float* dy, const float* x,
                                     const float* mean, const float* rstd, const float* gamma, float* dx,
                                     float* dgamma, float* dbeta, float* workspace, float* dx_dropped, float dropout_p,
                                     uint64_t seed) {
    if (dx_dropped == nullptr || !(dropout_p > 0.f && dropout_p < 1.f)) return VB_E_BADARG;
    if (!vb_aligned16(dx_dropped)) return VB_E_ALIGN;
    return layernorm_bwd_impl(stream, rows, n_cols, dy, x, mean, rstd, gamma, dx, dgamma, dbeta, workspace, dx_dropped,
                              dropout_p, seed);
}

extern "C" int vb_text_embed_bwd(void* stream, int32_t batch, int32_t n_tok, int32_t hidden, int32_t vocab,
                                 int32_t n_types, int32_t n_tasks, const int64_t* ids, const int64_t* seg,
                                 const int64_t* task_ids, const float* dx, float* dword, float* dpos, float* dtype,
                                 float* dtask) {
    if (ids == nullptr || seg == nullptr || dx == nullptr || dword == nullptr || dpos == nullptr ||
        dtype == nullptr || batch <= 0 || n_tok <= 0 || vocab <= 0 || n_types <= 0)
        return VB_E_BADARG;
    if (task_ids != nullptr && (dtask == nullptr || n_tasks <= 0)) return VB_E_BADARG;
    if (int e = check_cols(hidden)) return e;
    if (!vb_aligned16(dx)) return VB_E_ALIGN;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const long rows = (long)batch * (n_tok + (task_ids != nullptr ? 1 : 0));
    dim3 grid((unsigned)((rows + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK)), block(256);
    VB_NV_DISPATCH(nv_for(hidden), hipLaunchKernelGGL((text_embed_scatter_kernel<NV>), grid, block, 0, st, batch,
                                                      n_tok, hidden, vocab, n_tasks, ids, seg, task_ids, dx, dword, dpos,
                                                      dtype, dtask));
    VB_LAUNCH_CHECK();
    hipLaunchKernelGGL(pos_type_grad_kernel, dim3((unsigned)(n_tok + (task_ids != nullptr ? 1 : 0))), dim3(256), 0,
                       st, batch, n_tok, hidden, n_types, seg, task_ids, dx, dpos, dtype);
    VB_LAUNCH_CHECK();
    return 0;
}
