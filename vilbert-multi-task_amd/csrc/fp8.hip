// FP8 (OCP e4m3fn) forward GEMM path - BASELINE config 5 ("fp8 MFMA co-attention path"). Not in the reference
// (its reduced-precision option is apex fp16, train_tasks.py --fp16); the numerics contract is this file's own and is
// restated on the CPU by oracle/fp8_oracle.py:
//
//   quantise   q[r][k] = e4m3_rne(x[r][k] * (448 / amax_r)),  scale[r] = amax_r / 448   (amax_r = max_k |x[r][k]|;
//              an all-zero row gets scale 1), one scale per ROW of the activation and per OUT-FEATURE of the weight
//   product    y[m][n] = (sum_k q_a[m][k] * q_w[n][k]) * scale_a[m] * scale_w[n] + bias[n], fp32 accumulation on
//              v_mfma_scale_f32_32x32x64_f8f6f4 with every block scale = 2^0 (the only fp8 MFMA of gfx950 that runs
//              at twice the bf16 rate; the carried-forward 32x32x16 fp8 form runs at the bf16 rate)
//   epilogue   the same fused bias / GELU / residual / dropout epilogues as the fp32 kernels (gemm_core.h).
//
// The operands live in HBM as bytes: a K-contiguous [rows][K] e4m3 matrix per operand (4x less operand traffic than
// fp32), K a multiple of 128.
#include "gemm_core.h"
#include <type_traits>

namespace {

using namespace vbgemm;

typedef int v8i __attribute__((ext_vector_type(8)));
typedef int v4i __attribute__((ext_vector_type(4)));

constexpr float E4M3_MAX = 448.0f;

// ---------------------------------------------------------------------------------------------------------------
// Row quantiser: one wave per row, 4 rows per block. NV > 0: the row (K = 256 NV floats, the widths of the encoder:
// 768, 1024, 2048, 3072, 4096) is read ONCE into NV float4 registers per lane - amax, scale, convert, 256-byte stores.
// NV == 0: any K % 4 == 0 in two sweeps (the second one hits L2: a row is at most 16 KB).
// ---------------------------------------------------------------------------------------------------------------
template <int NV>
__global__ __launch_bounds__(256) void quant_rows_kernel(long rows, int K, const float* __restrict__ x, long ldx,
                                                         unsigned char* __restrict__ q, long ldq,
                                                         float* __restrict__ scale) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const f32x4* __restrict__ src = reinterpret_cast<const f32x4*>(x + row * ldx);
    unsigned* __restrict__ dst = reinterpret_cast<unsigned*>(q + row * ldq);
    const int nv = K >> 2;
    f32x4 reg[NV > 0 ? NV : 1];
    float amax = 0.f;
    if (NV > 0) {
#pragma unroll
        for (int i = 0; i < NV; ++i) reg[i] = src[lane + 64 * i];
#pragma unroll
        for (int i = 0; i < NV; ++i)
            amax = fmaxf(amax, fmaxf(fmaxf(fabsf(reg[i][0]), fabsf(reg[i][1])), fmaxf(fabsf(reg[i][2]), fabsf(reg[i][3]))));
    } else {
        for (int i = lane; i < nv; i += 64) {
            const f32x4 v = src[i];
            amax = fmaxf(amax, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o));
    const bool zero = !(amax > 0.f);
    const float inv = zero ? 1.f : E4M3_MAX / amax;
    if (lane == 0) scale[row] = zero ? 1.f : amax / E4M3_MAX;
    auto pack = [&](const f32x4 v) {
        int w = __builtin_amdgcn_cvt_pk_fp8_f32(v[0] * inv, v[1] * inv, 0, false);
        return (unsigned)__builtin_amdgcn_cvt_pk_fp8_f32(v[2] * inv, v[3] * inv, w, true);
    };
    if (NV > 0) {
#pragma unroll
        for (int i = 0; i < NV; ++i) dst[lane + 64 * i] = pack(reg[i]);
    } else {
        for (int i = lane; i < nv; i += 64) dst[i] = pack(src[i]);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// GEMM: 128 x 128 block tile, K step 128 bytes, 4 waves (2 x 2), 64 x 64 per wave = 2 x 2 MFMA 32x32x64 tiles.
// LDS per stage and operand: [128 rows][128 bytes]; the eight 16-byte chunks of a row are XOR-swizzled with
// (row >> 1) & 7, so that the 16 lanes of a ds_read_b128 group (16 consecutive rows, same logical chunk) cover all
// 64 banks once, and the 8 lanes of a ds_write_b128 group (one row, 8 chunks) a contiguous 128 bytes.
// Staging global -> registers -> LDS through THREE register sets (tiles kt + 1 .. kt + 3 in flight while tile kt is
// multiplied), one barrier per K tile (= 8 MFMAs of 64 cycles per wave). The epilogue issues all of its global reads
// (row / column scales, bias, residual) before the first dependent use.
// ---------------------------------------------------------------------------------------------------------------
// abl (tools/fp8_lab.py, VB_FP8_ABL): 1 = no global stores / residual reads in the epilogue, 2 = one K tile only
struct Fp8X {
    const unsigned char* A; long lda;   // [M][K] bytes
    const unsigned char* B; long ldb;   // [N][K] bytes
    const float* sa;                    // [M]
    const float* sb;                    // [N]
    int nk;                             // K / 128
    int n_tiles;
    int abl;
};

constexpr int F8_BK = 128;
constexpr int F8_OPER = 128 * F8_BK;        // bytes
constexpr int F8_STAGE = 2 * F8_OPER;
constexpr int F8_LDS = 2 * F8_STAGE;        // 65,536 B: two blocks per CU

// MODE: compile-time epilogue of the common cases (runtime checks per element serialise the 64 stores of a lane):
// 1 = bias, 2 = bias + GELU, 3 = bias + residual, 0 = everything at run time (pre-activation output, dropout, ReLU,
// swish and their combinations).
template <int MODE>
__global__ __launch_bounds__(256, 2) void gemm_fp8_kernel(const GemmP p, const Fp8X x) {
    extern __shared__ __attribute__((aligned(16))) char smem8[];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    const int t = xcd_swizzle(blockIdx.x, x.n_tiles);
    const int m0 = (t / p.tiles_n) * 128, n0 = (t % p.tiles_n) * 128;

    // staging: thread owns the 16-byte chunk c of the rows (tid >> 3) + 32 i
    const int c = tid & 7, r0 = tid >> 3;
    const unsigned char* ag[4];
    const unsigned char* bg[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        ag[i] = x.A + (long)min(m0 + r0 + 32 * i, p.M - 1) * x.lda + c * 16;   // rows past the edge: clamped, results unused
        bg[i] = x.B + (long)min(n0 + r0 + 32 * i, p.N - 1) * x.ldb + c * 16;
    }
    const int st_off = r0 * F8_BK + ((c ^ ((r0 >> 1) & 7)) << 4);   // + 32 i rows = + 4096 i bytes (same swizzle key)
    // fragments: lane reads row l31 (+ 32 per MFMA tile) of its wave's 64 rows, 32 bytes = chunks 4 s + 2 hi + {0, 1}
    const int key = (l31 >> 1) & 7;
    const int fa_off = (wm * 64 + l31) * F8_BK, fb_off = F8_OPER + (wn * 64 + l31) * F8_BK;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // Three register sets: the tiles kt + 1 .. kt + 3 are in flight while tile kt is multiplied. These GEMMs are bound
    // by HBM / L2 latency, not by the matrix pipe (8 MFMAs = 512 cycles per K tile against > 1 us of memory
    // latency), so what matters is bytes in flight: 3 tiles x 32 KB x 2 blocks per CU.
    v4i ra[3][4], rb[3][4];
    auto load = [&](auto S_, int kt) {
        constexpr int S = decltype(S_)::value;
#pragma unroll
        for (int i = 0; i < 4; ++i) ra[S][i] = *reinterpret_cast<const v4i*>(ag[i] + (long)kt * F8_BK);
#pragma unroll
        for (int i = 0; i < 4; ++i) rb[S][i] = *reinterpret_cast<const v4i*>(bg[i] + (long)kt * F8_BK);
    };
    auto store = [&](auto S_, char* stage) {
        constexpr int S = decltype(S_)::value;
#pragma unroll
        for (int i = 0; i < 4; ++i) *reinterpret_cast<v4i*>(stage + st_off + 4096 * i) = ra[S][i];
#pragma unroll
        for (int i = 0; i < 4; ++i) *reinterpret_cast<v4i*>(stage + F8_OPER + st_off + 4096 * i) = rb[S][i];
    };
    auto frag = [&](const char* base, int s) -> v8i {
        const int c0 = 4 * s + 2 * hi;
        const v4i lo = *reinterpret_cast<const v4i*>(base + (((c0) ^ key) << 4));
        const v4i up = *reinterpret_cast<const v4i*>(base + (((c0 + 1) ^ key) << 4));
        return __builtin_shufflevector(lo, up, 0, 1, 2, 3, 4, 5, 6, 7);
    };
    auto compute = [&](const char* stage) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            v8i a[2], b[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) a[i] = frag(stage + fa_off + i * 32 * F8_BK, s);
#pragma unroll
            for (int j = 0; j < 2; ++j) b[j] = frag(stage + fb_off + j * 32 * F8_BK, s);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    // formats 0 / 0 = e4m3 x e4m3; block scales: E8M0 127 = 2^0 for every 32-element block
                    acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[i], b[j], acc[i][j], 0, 0,
                                                                               0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
        }
    };

    // The staging registers are loaded / stored unconditionally (past the end: the last K tile again, stored into the
    // stage nobody reads any more) - a conditional load makes hipcc keep the register arrays in scratch.
    const int last = x.nk - 1;
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    // K tile kt (kt % 3 == S): issue tile kt + 3 into the free set S, multiply stage kt & 1, then move tile kt + 1
    // (set S + 1, the oldest load in flight) into the other stage.
    auto step = [&](auto S_, auto S1_, int kt) {
        load(S_, min(kt + 3, last));
        compute(smem8 + (kt & 1) * F8_STAGE);
        store(S1_, smem8 + ((kt & 1) ^ 1) * F8_STAGE);
        __syncthreads();
    };
    load(I0{}, 0);
    store(I0{}, smem8);
    load(I1{}, min(1, last));
    load(I2{}, min(2, last));
    __syncthreads();
    int kt = 0;
    for (; kt + 2 < x.nk; kt += 3) {
        step(I0{}, I1{}, kt);
        step(I1{}, I2{}, kt + 1);
        step(I2{}, I0{}, kt + 2);
    }
    if (kt < x.nk) step(I0{}, I1{}, kt);
    if (kt + 1 < x.nk) step(I1{}, I2{}, kt + 1);

    // Epilogue. These GEMMs are short (K / 128 = 6 .. 24 steps) and their fp32 output is 4x the operand bytes, so the
    // epilogue is a large part of the kernel: every global read it needs (row scales, bias, residual tile) is issued
    // up front - 64 + 34 loads in flight per lane - before the first dependent use, then one pass of arithmetic and
    // stores. Accumulator map (32x32): col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5).
    const int row_w = m0 + wm * 64 + 4 * hi, col_w = n0 + wn * 64 + l31;
    const bool interior = m0 + 128 <= p.M && n0 + 128 <= p.N;
    float sbv[2], bv[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int col = min(col_w + j * 32, p.N - 1);
        sbv[j] = x.sb[col];
        bv[j] = p.bias[0] != nullptr ? p.bias[0][col] : 0.f;
    }
    const bool has_r = (MODE == 0 ? p.R != nullptr : MODE == 3) && x.abl != 1;
    const uint64_t seed = (MODE == 0 && p.drop_p > 0.f) ? vb_seed_with_epoch(p.seed, p.epoch) : 0;
    // two halves of 32 rows: 16 scale + 32 residual loads in flight per lane and half (64 at once spill)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        float sav[16], rv[2][16];
#pragma unroll
        for (int r = 0; r < 16; ++r) sav[r] = x.sa[min(row_w + i * 32 + (r & 3) + 8 * (r >> 2), p.M - 1)];
        if (has_r) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                // clamped, never predicated: a per-element condition makes hipcc branch around (and wait for) each load
                const int row = min(row_w + i * 32 + (r & 3) + 8 * (r >> 2), p.M - 1);
#pragma unroll
                for (int j = 0; j < 2; ++j) rv[j][r] = p.R[(long)row * p.ldr + min(col_w + j * 32, p.N - 1)];
            }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = row_w + i * 32 + (r & 3) + 8 * (r >> 2);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int col = col_w + j * 32;
                float v = acc[i][j][r] * (sav[r] * sbv[j]) + bv[j];
                const bool inside = interior || (row < p.M && col < p.N);
                if (MODE == 0) {
                    if (p.P != nullptr && inside) p.P[(long)row * p.ldp + col] = v;
                    if (p.act == VB_ACT_GELU) v = gelu_erf(v);
                    else if (p.act == VB_ACT_RELU) v = fmaxf(v, 0.f);
                    else if (p.act == VB_ACT_SWISH) v = swish_act(v);
                    if (p.drop_p > 0.f)
                        v = vb_keep(seed, (uint64_t)((long)row * p.N + col), p.drop_p) ? v * p.drop_scale : 0.f;
                }
                if (MODE == 2) v = gelu_erf(v);
                if (has_r) v += rv[j][r];
                if (inside && (x.abl != 1 || v == 12345.678f)) p.C[0][(long)row * p.ldc + col] = v;
            }
        }
    }
}

}  // namespace

// q[rows][K] (bytes, row stride ldq) = e4m3(x / scale), scale[rows] = amax_row / 448
extern "C" int vb_quantize_rows_fp8(void* stream, int64_t rows, int32_t K, const float* x, int64_t ldx, uint8_t* q, int64_t ldq,
                                    float* scale) {
    if (x == nullptr || q == nullptr || scale == nullptr || rows <= 0 || K <= 0) return VB_E_BADARG;
    if (K % 4 != 0 || ldx % 4 != 0 || ldq % 4 != 0 || ldq < K || ldx < K || !vb_aligned16(x) ||
        (reinterpret_cast<uintptr_t>(q) & 3u) != 0)
        return VB_E_ALIGN;
    const dim3 grid((unsigned)((rows + 3) / 4)), block(256);
    hipStream_t st = static_cast<hipStream_t>(stream);
#define VB_QUANT(NV) hipLaunchKernelGGL(quant_rows_kernel<NV>, grid, block, 0, st, rows, K, x, ldx, q, ldq, scale)
    switch (K) {
        case 768: VB_QUANT(3); break;
        case 1024: VB_QUANT(4); break;
        case 2048: VB_QUANT(8); break;
        case 3072: VB_QUANT(12); break;
        case 4096: VB_QUANT(16); break;
        default: VB_QUANT(0); break;
    }
#undef VB_QUANT
    VB_LAUNCH_CHECK();
    return 0;
}

// C[M][N] = act((A8 . W8^T) * a_scale[m] * w_scale[n] + bias) (+ dropout) (+ residual)
extern "C" int vb_linear_fwd_fp8(void* stream, const vb_linear_fp8_args* a) {
    if (a == nullptr || a->A == nullptr || a->W == nullptr || a->C == nullptr || a->a_scale == nullptr ||
        a->w_scale == nullptr)
        return VB_E_BADARG;
    if (a->M <= 0 || a->N <= 0 || a->K <= 0) return VB_E_BADARG;
    if (a->act < VB_ACT_NONE || a->act > VB_ACT_SWISH) return VB_E_BADARG;
    if (a->K % F8_BK != 0 || a->lda % 16 != 0 || a->ldw % 16 != 0 || a->lda < a->K || a->ldw < a->K ||
        !vb_aligned16(a->A) || !vb_aligned16(a->W))
        return VB_E_ALIGN;
    if (!(a->dropout_p >= 0.f && a->dropout_p < 1.f)) return VB_E_BADARG;
    if (a->dropout_p > 0.f && a->ldc != a->N) return VB_E_ALIGN;
    hipStream_t st = static_cast<hipStream_t>(stream);
    GemmP p{};
    p.M = a->M; p.N = a->N; p.K = a->K;
    p.bseg = a->N; p.bias[0] = a->bias;
    p.C[0] = a->C; p.ldc = a->ldc; p.cseg = (p.M + 127) / 128 * 128;
    p.R = a->residual; p.ldr = a->ldr;
    p.act = a->act;
    p.drop_p = a->dropout_p; p.drop_scale = 1.0f / (1.0f - a->dropout_p); p.seed = a->seed;
    p.epoch = a->dropout_p > 0.f ? vb_seed_epoch() : nullptr;
    // the activation derivative is produced as in the round-1 fp32 kernel: pre-activation stored by the epilogue,
    // turned into act'(.) in place by a post-pass
    if (a->act_grad != nullptr && a->preact != nullptr) return VB_E_BADARG;
    if (a->act_grad != nullptr) { p.P = a->act_grad; p.ldp = a->ldg; }
    if (a->preact != nullptr) { p.P = a->preact; p.ldp = a->ldp; }
    if (a->dropout_p > 0.f)
        p.epi = (a->act == VB_ACT_NONE && p.P == nullptr && a->residual != nullptr) ? EPI_RES_DROP : EPI_GENERIC;
    else if (a->act == VB_ACT_NONE && p.P == nullptr) p.epi = a->residual != nullptr ? EPI_RES : EPI_STORE;
    else if (a->act == VB_ACT_GELU && a->residual == nullptr) p.epi = p.P != nullptr ? EPI_PRE_GELU : EPI_GELU;
    else p.epi = EPI_GENERIC;
    p.tiles_n = (p.N + 127) / 128;
    Fp8X x{};
    x.A = a->A; x.lda = a->lda; x.B = a->W; x.ldb = a->ldw; x.sa = a->a_scale; x.sb = a->w_scale;
    x.nk = a->K / F8_BK;
    static const int abl = [] { const char* e = getenv("VB_FP8_ABL"); return e ? atoi(e) : 0; }();
    x.abl = abl;
    if (abl == 2) x.nk = 1;
    x.n_tiles = ((p.M + 127) / 128) * p.tiles_n;
    const bool plain = p.P == nullptr && a->dropout_p == 0.f;
    const int mode = !plain ? 0
                     : (a->act == VB_ACT_NONE && a->residual == nullptr) ? 1
                     : (a->act == VB_ACT_GELU && a->residual == nullptr) ? 2
                     : (a->act == VB_ACT_NONE && a->residual != nullptr) ? 3 : 0;
    static const bool attr = [] {
        bool ok = true;
        ok = ok && hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_fp8_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, F8_LDS) == hipSuccess;
        ok = ok && hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_fp8_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, F8_LDS) == hipSuccess;
        ok = ok && hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_fp8_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, F8_LDS) == hipSuccess;
        ok = ok && hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_fp8_kernel<3>), hipFuncAttributeMaxDynamicSharedMemorySize, F8_LDS) == hipSuccess;
        return ok;
    }();
    if (!attr) return VB_E_RANGE;
    const dim3 grid(x.n_tiles), block(256);
    switch (mode) {
        case 1: hipLaunchKernelGGL(gemm_fp8_kernel<1>, grid, block, F8_LDS, st, p, x); break;
        case 2: hipLaunchKernelGGL(gemm_fp8_kernel<2>, grid, block, F8_LDS, st, p, x); break;
        case 3: hipLaunchKernelGGL(gemm_fp8_kernel<3>, grid, block, F8_LDS, st, p, x); break;
        default: hipLaunchKernelGGL(gemm_fp8_kernel<0>, grid, block, F8_LDS, st, p, x); break;
    }
    VB_LAUNCH_CHECK();
    if (a->act_grad != nullptr)
        if (int e = launch_act_grad_inplace(st, p.M, p.N, a->act_grad, a->ldg, a->act)) return e;
    return 0;
}
