// fp32 GEMM family on v_mfma_f32_32x32x2_f32 (exact-fp32 matrix cores, 157.3 TFLOP/s peak on
// gfx950) with fused epilogues. One kernel template covers the three operand layouts the
// encoder needs:
//   forward  (NT)  C[M,N] = A[M,K] . W[N,K]^T        A k-contiguous,  B k-contiguous
//   dgrad    (NN)  dX[M,K'] = dY[M,N'] . W[N',K']    A k-contiguous,  B j-contiguous
//   wgrad    (TN)  dW[N',K'] = dY[M,N']^T . X[M,K']  A i-contiguous,  B j-contiguous
//
// Block = 256 threads = 4 waves (one per SIMD), block tile 128x128, K step 32, each wave owns a
// 64x64 sub-tile = 2x2 MFMA tiles of 32x32 (64 accumulator VGPRs). LDS is double buffered
// (2 x 36 KiB -> two blocks per CU, i.e. two waves per SIMD so one block's barrier / epilogue is
// covered by the other's MFMAs). Global -> register -> LDS staging: the loads for K tile t+1 are
// issued before the 64 MFMAs (4096 matrix-pipe cycles) of tile t and written to the other LDS
// buffer after them; one barrier per K tile.
//
// LDS layouts (floats):
//   k-contiguous operand: [128 rows][36]  (32 + 4 pad): 16-lane ds_read_b128 groups hit 16
//       distinct 16-B slots (row stride 36 dwords = 9 slots, odd) -> conflict free.
//   row-contiguous operand: [32 k][132]: ds_read_b32, lanes = consecutive rows -> conflict free.
// MFMA operand convention (32x32x2): lane l supplies A[i = l&31][k = l>>5], B[k = l>>5][j = l&31].
// The contraction order inside a K step is permuted (lane half `hi` owns k = 8c + 4hi + e) so that
// a k-contiguous operand is fetched with one ds_read_b128 per four MFMAs; A and B use the same
// permutation, which only reorders the fp32 summation.
//
// Workgroup -> tile map is XCD aware: block b runs on XCD b % 8, so XCD x gets a contiguous run
// of logical tiles (N fastest) and the blocks sharing an A panel share one L2.
#include "common.h"

namespace {

constexpr int BM = 128, BN = 128;
constexpr int RC_LD = 132;   // row-contiguous LDS row stride ([BK][132])

// K-step dependent geometry. k-contiguous LDS rows hold BK + 4 floats: (BK + 4) / 4 is odd for
// BK = 16 / 32, so the 16-lane ds_read_b128 groups fall on 16 distinct 16-byte slots (conflict free).
template <int BK> struct Geo {
    static constexpr int KC_LD = BK + 4;
    static constexpr int OPER_SZ = 128 * KC_LD;      // >= BK * RC_LD
    static constexpr int STAGE_SZ = 2 * OPER_SZ;     // A + B
    static constexpr int LDS_BYTES = 2 * STAGE_SZ * 4;  // double buffered: 73,728 B (BK 32), 40,960 B (BK 16)
    static constexpr int NLD = BK / 8;               // float4 loads per thread per operand tile
    static constexpr int KQ = BK / 4;                // float4 per k-contiguous row
};

struct GemmP {
    int M, N, K;
    const float* A; long lda;
    const float* B[VB_MAX_SEGMENTS]; long ldb; int bseg;
    const float* bias[VB_MAX_SEGMENTS];
    float* C; long ldc;
    const float* R; long ldr;
    float* P; long ldp;
    int act;
    int accumulate;       // C += result
    int tiles_m, tiles_n;
    int ktiles_per_split; // split-K (gridDim.y > 1): atomicAdd into C
    float* colsum;        // row-contiguous A only: colsum[i] += sum_k A[i][k] (bias gradient), may be null
    int epi;              // EPI_* fast path of interior tiles (EPI_GENERIC = none)
    int flags;            // tuning knobs (VB_GEMM_FLAGS): 1 = raise wave priority around the MFMA block
};

// Staging of one 128 x 32 operand tile into registers (4 float4 per thread).
// k-contiguous operand (global [rows][ld]): thread t owns rows (t >> 3) + 32 it, it = 0..3, and the
// four k values 4 (t & 7) .. +3 of every K tile, so the row base pointers are computed once per
// block (this is also where a row is mapped to its weight segment).
template <bool VEC, int NLD>
__device__ __forceinline__ void load_tile_kc(f32x4 (&reg)[NLD], const float* const (&rowp)[NLD], int k, int K) {
#pragma unroll
    for (int it = 0; it < NLD; ++it) {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (rowp[it] != nullptr) {
            const float* g = rowp[it] + k;
            if (VEC) {
                if (k < K) v = *reinterpret_cast<const f32x4*>(g);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (k + e < K) v[e] = g[e];
            }
        }
        reg[it] = v;
    }
}

// row-contiguous operand (global [k][ld], rows contiguous): thread t owns k = (t >> 5) + 8 it and the
// four rows row0 + 4 (t & 31) .. +3.
template <bool VEC, int NLD>
__device__ __forceinline__ void load_tile_rc(f32x4 (&reg)[NLD], const float* __restrict__ base, long ld,
                                             int row0, int nrows, int k0, int K, int tid) {
#pragma unroll
    for (int it = 0; it < NLD; ++it) {
        const int f = tid + 256 * it;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        const int k = k0 + (f >> 5), row = row0 + (f & 31) * 4;
        if (k < K) {
            const float* g = base + (long)k * ld + row;
            if (VEC) {
                if (row < nrows) v = *reinterpret_cast<const f32x4*>(g);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (row + e < nrows) v[e] = g[e];
            }
        }
        reg[it] = v;
    }
}

template <bool KC, int BK>
__device__ __forceinline__ void store_tile(float* __restrict__ s, const f32x4 (&reg)[BK / 8], int tid) {
    constexpr int KQ = BK / 4;
#pragma unroll
    for (int it = 0; it < BK / 8; ++it) {
        const int f = tid + 256 * it;
        const int off = KC ? (f / KQ) * (BK + 4) + (f % KQ) * 4 : (f >> 5) * RC_LD + (f & 31) * 4;
        *reinterpret_cast<f32x4*>(s + off) = reg[it];
    }
}

enum { EPI_GENERIC = 0, EPI_STORE, EPI_GELU, EPI_RES, EPI_PRE_GELU, EPI_ACCUM, EPI_ATOMIC };

// Branch-free epilogue of a full interior tile. MODE: STORE c = v; GELU c = gelu(v); RES c = v + R;
// PRE_GELU P = v, c = gelu(v); ACCUM c += v; ATOMIC atomicAdd(c, v)   with v = acc + bias.
template <int MODE>
__device__ __forceinline__ void epilogue_full(const GemmP& p, const f32x16 (&acc)[2][2], const float (&bv)[2],
                                              int row0, int col0) {
    float* __restrict__ cbase = p.C + (long)row0 * p.ldc + col0;
    const float* __restrict__ rbase = MODE == EPI_RES ? p.R + (long)row0 * p.ldr + col0 : nullptr;
    float* __restrict__ pbase = MODE == EPI_PRE_GELU ? p.P + (long)row0 * p.ldp + col0 : nullptr;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int dr = i * 32 + (r & 3) + 8 * (r >> 2);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                float v = acc[i][j][r] + bv[j];
                float* c = cbase + (long)dr * p.ldc + j * 32;
                if (MODE == EPI_PRE_GELU) pbase[(long)dr * p.ldp + j * 32] = v;
                if (MODE == EPI_GELU || MODE == EPI_PRE_GELU) v = gelu_erf(v);
                if (MODE == EPI_RES) v += rbase[(long)dr * p.ldr + j * 32];
                if (MODE == EPI_ATOMIC) unsafeAtomicAdd(c, v);
                else if (MODE == EPI_ACCUM) *c += v;
                else *c = v;
            }
        }
    }
}

template <bool A_KC, bool B_KC, bool VEC, int BK>
__global__ __launch_bounds__(256, 2) void gemm_f32_kernel(const GemmP p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    using G = Geo<BK>;
    constexpr int KC_LD = G::KC_LD, OPER_SZ = G::OPER_SZ, STAGE_SZ = G::STAGE_SZ, NLD = G::NLD, KQ = G::KQ;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;

    // XCD-aware bijective remap of the linear block id (guide T1).
    const int nb = p.tiles_m * p.tiles_n;
    int logical;
    {
        const int b = blockIdx.x, q = nb >> 3, r = nb & 7, xcd = b & 7, idx = b >> 3;
        logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int m0 = (logical / p.tiles_n) * BM;
    const int n0 = (logical % p.tiles_n) * BN;

    const int kt_total = (p.K + BK - 1) / BK;
    const int kt_begin = blockIdx.y * p.ktiles_per_split;
    const int kt_end = min(kt_total, kt_begin + p.ktiles_per_split);
    if (kt_begin >= kt_end) return;

    // Row base pointers of the k-contiguous operands. B rows (= output columns) are mapped to their
    // weight segment here: the segments are stacked along N (q | k | v projections in one launch).
    const float* arow[NLD];
    const float* brow[NLD];
    const int kq = (tid % KQ) * 4;
#pragma unroll
    for (int it = 0; it < NLD; ++it) {
        arow[it] = nullptr;
        brow[it] = nullptr;
        const int r = tid / KQ + (256 / KQ) * it;
        if (A_KC && m0 + r < p.M) arow[it] = p.A + (long)(m0 + r) * p.lda;
        if (B_KC && n0 + r < p.N) {
            const int n = n0 + r, sg = n / p.bseg;
            brow[it] = p.B[sg] + (long)(n - sg * p.bseg) * p.ldb;
        }
    }

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    f32x4 ra[NLD], rb[NLD];
    float csum = 0.f;

    auto load_ab = [&](int kt) {
        const int k0 = kt * BK;
        if (A_KC) load_tile_kc<VEC, NLD>(ra, arow, k0 + kq, p.K);
        else load_tile_rc<VEC, NLD>(ra, p.A, p.lda, m0, p.M, k0, p.K, tid);
        if (B_KC) {
            load_tile_kc<VEC, NLD>(rb, brow, k0 + kq, p.K);
        } else {
            // segments stacked along K (dgrad through stacked weights); bseg is a multiple of BK
            const int sg = k0 / p.bseg;
            load_tile_rc<VEC, NLD>(rb, p.B[sg], p.ldb, n0, p.N, k0 - sg * p.bseg, min(p.bseg, p.K - sg * p.bseg), tid);
        }
    };

    load_ab(kt_begin);
    store_tile<A_KC, BK>(smem, ra, tid);
    store_tile<B_KC, BK>(smem + OPER_SZ, rb, tid);
    __syncthreads();

    for (int kt = kt_begin; kt < kt_end; ++kt) {
        const int cur = (kt - kt_begin) & 1;
        const float* sA = smem + cur * STAGE_SZ;
        const float* sB = sA + OPER_SZ;
        const bool more = kt + 1 < kt_end;
        if (more) load_ab(kt + 1);

        if (p.flags & 1) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kc = 0; kc < BK / 8; ++kc) {
            f32x4 af[2], bf[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                if (A_KC) {
                    af[t] = *reinterpret_cast<const f32x4*>(
                        sA + (wm * 64 + t * 32 + l31) * KC_LD + kc * 8 + hi * 4);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        af[t][e] = sA[(kc * 8 + hi * 4 + e) * RC_LD + wm * 64 + t * 32 + l31];
                }
                if (B_KC) {
                    bf[t] = *reinterpret_cast<const f32x4*>(
                        sB + (wn * 64 + t * 32 + l31) * KC_LD + kc * 8 + hi * 4);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        bf[t][e] = sB[(kc * 8 + hi * 4 + e) * RC_LD + wn * 64 + t * 32 + l31];
                }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][e], bf[j][e],
                                                                        acc[i][j], 0, 0, 0);
        }

        if (p.flags & 1) __builtin_amdgcn_s_setprio(0);
        if (!A_KC && p.colsum != nullptr && n0 == 0 && tid < BM) {
            // bias gradient fused into wgrad: A = dY^T, so the sum over this K tile of row i = tid
#pragma unroll
            for (int kk = 0; kk < BK; ++kk) csum += sA[kk * RC_LD + tid];
        }

        if (more) {
            float* dA = smem + (cur ^ 1) * STAGE_SZ;
            store_tile<A_KC, BK>(dA, ra, tid);
            store_tile<B_KC, BK>(dA + OPER_SZ, rb, tid);
        }
        __syncthreads();
    }

    if (!A_KC && p.colsum != nullptr && n0 == 0 && tid < BM && m0 + tid < p.M)
        unsafeAtomicAdd(p.colsum + m0 + tid, csum);

    // Epilogue. Accumulator map (32x32): col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5).
    // Interior tiles with one of the common epilogues take a branch-free specialised path (the generic
    // predicated loop costs ~2k VALU instructions per wave, during which the matrix pipe starves when
    // the co-resident blocks reach their epilogues together).
    const bool lead = blockIdx.y == 0;  // bias / residual are added by one split only
    float bv[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int col = n0 + wn * 64 + j * 32 + l31;
        bv[j] = 0.f;
        if (col < p.N && lead) {
            const int sg = B_KC ? col / p.bseg : 0;  // bias follows the N segmentation of a k-contiguous B
            const float* bp = p.bias[sg];
            if (bp != nullptr) bv[j] = bp[col - sg * p.bseg * (B_KC ? 1 : 0)];
        }
    }
    const int row0 = m0 + wm * 64 + 4 * hi, col0 = n0 + wn * 64 + l31;
    if (m0 + BM <= p.M && n0 + BN <= p.N && p.epi != EPI_GENERIC) {
        switch (p.epi) {
            case EPI_STORE: epilogue_full<EPI_STORE>(p, acc, bv, row0, col0); break;
            case EPI_GELU: epilogue_full<EPI_GELU>(p, acc, bv, row0, col0); break;
            case EPI_RES: epilogue_full<EPI_RES>(p, acc, bv, row0, col0); break;
            case EPI_PRE_GELU: epilogue_full<EPI_PRE_GELU>(p, acc, bv, row0, col0); break;
            case EPI_ACCUM: epilogue_full<EPI_ACCUM>(p, acc, bv, row0, col0); break;
            default: epilogue_full<EPI_ATOMIC>(p, acc, bv, row0, col0); break;
        }
        return;
    }
    const bool split = gridDim.y > 1;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int col = col0 + j * 32;
        if (col >= p.N) continue;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = row0 + i * 32 + (r & 3) + 8 * (r >> 2);
                if (row >= p.M) continue;
                float v = acc[i][j][r] + bv[j];
                if (p.P != nullptr) p.P[(long)row * p.ldp + col] = v;
                if (p.act == VB_ACT_GELU) v = gelu_erf(v);
                else if (p.act == VB_ACT_RELU) v = fmaxf(v, 0.f);
                if (p.R != nullptr && lead) v += p.R[(long)row * p.ldr + col];
                float* c = p.C + (long)row * p.ldc + col;
                if (split) unsafeAtomicAdd(c, v);
                else if (p.accumulate) *c += v;
                else *c = v;
            }
        }
    }
}

constexpr int BK_DEFAULT = 16;

inline int gemm_bk() {
    // K step of the GEMM kernels; VB_GEMM_BK=32 selects the 72 KiB double buffer (tuning knob).
    static int bk = [] {
        const char* e = getenv("VB_GEMM_BK");
        const int v = e ? atoi(e) : BK_DEFAULT;
        return v == 32 ? 32 : 16;
    }();
    return bk;
}

template <bool A_KC, bool B_KC, bool VEC, int BK>
int launch_gemm_bk(hipStream_t st, const GemmP& p, int splits) {
    dim3 grid(p.tiles_m * p.tiles_n, splits), block(256);
    auto k = gemm_f32_kernel<A_KC, B_KC, VEC, BK>;
    static bool attr_done = false;  // > 64 KiB of dynamic LDS needs the attribute once per kernel
    if (Geo<BK>::LDS_BYTES > 65536 && !attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, Geo<BK>::LDS_BYTES);
        if (e != hipSuccess) return (int)e;
        attr_done = true;
    }
    hipLaunchKernelGGL(k, grid, block, Geo<BK>::LDS_BYTES, st, p);
    VB_LAUNCH_CHECK();
    return 0;
}

template <bool A_KC, bool B_KC>
int launch_gemm(hipStream_t st, GemmP p, bool vec, int splits) {
    static const int flags = [] { const char* e = getenv("VB_GEMM_FLAGS"); return e ? atoi(e) : 0; }();
    p.flags = flags;
    if (gemm_bk() == 32)
        return vec ? launch_gemm_bk<A_KC, B_KC, true, 32>(st, p, splits)
                   : launch_gemm_bk<A_KC, B_KC, false, 32>(st, p, splits);
    return vec ? launch_gemm_bk<A_KC, B_KC, true, 16>(st, p, splits)
               : launch_gemm_bk<A_KC, B_KC, false, 16>(st, p, splits);
}

}  // namespace

extern "C" int vb_linear_fwd(void* stream, const vb_linear_args* a) {
    if (a == nullptr || a->A == nullptr || a->C == nullptr) return VB_E_BADARG;
    if (a->M <= 0 || a->K <= 0 || a->seg_n <= 0) return VB_E_BADARG;
    if (a->nseg < 1 || a->nseg > VB_MAX_SEGMENTS) return VB_E_SEGMENT;
    if (a->act < VB_ACT_NONE || a->act > VB_ACT_RELU) return VB_E_BADARG;
    GemmP p{};
    p.M = a->M; p.K = a->K; p.N = a->nseg * a->seg_n;
    p.A = a->A; p.lda = a->lda;
    p.ldb = a->ldw; p.bseg = a->seg_n;
    bool vec = (a->K % 4 == 0) && (a->lda % 4 == 0) && (a->ldw % 4 == 0) && vb_aligned16(a->A);
    for (int s = 0; s < a->nseg; ++s) {
        if (a->W[s] == nullptr) return VB_E_SEGMENT;
        p.B[s] = a->W[s];
        p.bias[s] = a->bias[s];
        vec = vec && vb_aligned16(a->W[s]);
    }
    p.C = a->C; p.ldc = a->ldc;
    p.R = a->residual; p.ldr = a->ldr;
    p.P = a->preact; p.ldp = a->ldp;
    p.act = a->act; p.accumulate = 0;
    if (a->act == VB_ACT_NONE && a->preact == nullptr) p.epi = a->residual != nullptr ? EPI_RES : EPI_STORE;
    else if (a->act == VB_ACT_GELU && a->residual == nullptr) p.epi = a->preact != nullptr ? EPI_PRE_GELU : EPI_GELU;
    else p.epi = EPI_GENERIC;
    p.tiles_m = (p.M + BM - 1) / BM;
    p.tiles_n = (p.N + BN - 1) / BN;
    p.ktiles_per_split = (p.K + gemm_bk() - 1) / gemm_bk();
    return launch_gemm<true, true>(static_cast<hipStream_t>(stream), p, vec, 1);
}

// dX[M,K] (+)= dY[M, nseg*seg_n] . stack(W)   - nn.Linear backward w.r.t. its input
extern "C" int vb_linear_bwd_input(void* stream, const vb_linear_bwd_input_args* a) {
    if (a == nullptr || a->dY == nullptr || a->dX == nullptr) return VB_E_BADARG;
    if (a->M <= 0 || a->K <= 0 || a->seg_n <= 0) return VB_E_BADARG;
    if (a->nseg < 1 || a->nseg > VB_MAX_SEGMENTS) return VB_E_SEGMENT;
    hipStream_t st = static_cast<hipStream_t>(stream);
    // a K tile of the contraction (over out-features) must not straddle two weight segments; otherwise
    // run one launch per segment, accumulating.
    const int BK = gemm_bk();
    const bool fused = a->nseg == 1 || (a->seg_n % BK) == 0;
    const int launches = fused ? 1 : a->nseg;
    for (int l = 0; l < launches; ++l) {
        GemmP p{};
        p.M = a->M; p.N = a->K;  // output is [M, in_features]
        p.K = fused ? a->nseg * a->seg_n : a->seg_n;
        p.A = a->dY + (fused ? 0 : (long)l * a->seg_n); p.lda = a->ldy;
        p.ldb = a->ldw; p.bseg = a->seg_n;
        bool vec = (a->K % 4 == 0) && (a->seg_n % 4 == 0) && (a->ldy % 4 == 0) && (a->ldw % 4 == 0) &&
                   vb_aligned16(p.A);
        for (int s = 0; s < (fused ? a->nseg : 1); ++s) {
            const float* w = a->W[fused ? s : l];
            if (w == nullptr) return VB_E_SEGMENT;
            p.B[s] = w;
            vec = vec && vb_aligned16(w);
        }
        p.C = a->dX; p.ldc = a->ldx;
        p.act = VB_ACT_NONE;
        p.accumulate = (a->accumulate || l > 0) ? 1 : 0;
        p.epi = p.accumulate ? EPI_ACCUM : EPI_STORE;
        p.tiles_m = (p.M + BM - 1) / BM;
        p.tiles_n = (p.N + BN - 1) / BN;
        p.ktiles_per_split = (p.K + BK - 1) / BK;
        if (int e = launch_gemm<true, false>(st, p, vec, 1)) return e;
    }
    return 0;
}

// dW[n,K] (+)= dY[:, :n]^T . X[M,K] and dbias[n] (+)= column sums of dY   - nn.Linear backward w.r.t.
// weight and bias. The contraction runs over the M rows: split over gridDim.y with fp32 atomics.
extern "C" int vb_linear_bwd_weight(void* stream, const vb_linear_bwd_weight_args* a) {
    if (a == nullptr || a->dY == nullptr || a->X == nullptr || a->dW == nullptr) return VB_E_BADARG;
    if (a->M <= 0 || a->K <= 0 || a->n <= 0) return VB_E_BADARG;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (!a->accumulate) {
        hipError_t e = hipMemset2DAsync(a->dW, a->ldw * sizeof(float), 0, (size_t)a->K * sizeof(float), a->n, st);
        if (e != hipSuccess) return (int)e;
        if (a->dbias != nullptr) {
            e = hipMemsetAsync(a->dbias, 0, (size_t)a->n * sizeof(float), st);
            if (e != hipSuccess) return (int)e;
        }
    }
    GemmP p{};
    p.M = a->n; p.N = a->K; p.K = a->M;
    p.A = a->dY; p.lda = a->ldy;
    p.B[0] = a->X; p.ldb = a->ldx; p.bseg = a->M;
    p.C = a->dW; p.ldc = a->ldw;
    p.act = VB_ACT_NONE; p.accumulate = 1;
    p.colsum = a->dbias;
    p.tiles_m = (p.M + BM - 1) / BM;
    p.tiles_n = (p.N + BN - 1) / BN;
    const int BK = gemm_bk();
    const int kt_total = (p.K + BK - 1) / BK;
    const int tiles = p.tiles_m * p.tiles_n;
    int splits = (1024 + tiles - 1) / tiles;  // aim at ~4 blocks per CU
    if (splits > kt_total) splits = kt_total;
    if (splits < 1) splits = 1;
    p.ktiles_per_split = (kt_total + splits - 1) / splits;
    splits = (kt_total + p.ktiles_per_split - 1) / p.ktiles_per_split;
    const bool vec = (a->n % 4 == 0) && (a->K % 4 == 0) && (a->ldy % 4 == 0) && (a->ldx % 4 == 0) &&
                     vb_aligned16(a->dY) && vb_aligned16(a->X);
    p.epi = splits > 1 ? EPI_ATOMIC : EPI_ACCUM;
    return launch_gemm<false, false>(st, p, vec, splits);
}
