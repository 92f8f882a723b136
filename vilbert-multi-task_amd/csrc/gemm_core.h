// Shared pieces of the GEMM kernels (gemm.hip): launch parameters, tile planning, XCD-aware block map and
// the fused epilogues.
#pragma once
#include "common.h"
#include "rng.h"

namespace vbgemm {

constexpr int BK = 16;
constexpr int KC_LD = BK + 4;
constexpr int OPER_SZ = 128 * KC_LD;         // 2560 floats >= 16 * 132 (row-contiguous big tile)
constexpr int STAGE_SZ = 2 * OPER_SZ;        // A + B
constexpr int GEMM_LDS_BYTES = 2 * STAGE_SZ * 4;  // 40,960 B

// EPI_DGELU: c = gelu(v), D = gelu'(v) (the activation derivative saved for backward); EPI_MUL: c = v * mul
// (the saved derivative applied to the incoming gradient in the dgrad epilogue)
enum { EPI_GENERIC = 0, EPI_STORE, EPI_GELU, EPI_RES, EPI_PRE_GELU, EPI_ACCUM, EPI_ATOMIC, EPI_RES_DROP, EPI_DGELU,
       EPI_MUL };

struct GemmP {
    int M, N, K;
    const float* A; long lda;
    const float* B[VB_MAX_SEGMENTS]; long ldb; int bseg;   // B row segments (stacked weights)
    const float* bias[VB_MAX_SEGMENTS];
    float* C[VB_MAX_SEGMENTS]; long ldc; int cseg;          // C row segments (wgrad of stacked weights)
    float* colsum[VB_MAX_SEGMENTS];  // row-contiguous A only: colsum[i] += sum_k A[i][k] (bias gradient)
    const float* R; long ldr;
    float* P; long ldp;
    float* D; long ldd;          // activation derivative act'(pre-activation) (may be null)
    const float* mul; long ldmul;  // elementwise multiplier of the result (may be null)
    int act;
    int accumulate;       // C += result
    int tiles_n;          // big-tile grid columns
    int n_big, n_small;   // blocks [0, n_big): big tiles; [n_big, n_big + n_small): small tiles
    int m_split;          // second-generation kernel: rows [0, m_split) are cut into the taller tiles
    int ktiles_per_split; // split-K (gridDim.y > 1): atomicAdd into C
    int epi;              // EPI_* fast path of interior tiles
    int flags;            // tuning knobs (VB_GEMM_FLAGS): 1 = raise wave priority around the MFMA block
    float drop_p, drop_scale;  // dropout on the activated value, before the residual (0 = off)
    uint64_t seed;
    const uint64_t* epoch;     // device step counter mixed into the seed (vb_set_seed_epoch), may be null
    unsigned long long* dbg;   // lab only (vblab_gemm_cycles): block 0 stores its shader-clock span here
    // deterministic split-K (vb_set_deterministic): split s stores its partial product to det_ws + s * det_stride as a
    // plain [M, N] matrix (and its bias-gradient partial to det_cs + s * M) instead of adding into C with atomics; a
    // second kernel sums the partials in split order (splitk_reduce_kernel)
    float* det_ws; float* det_cs; long det_stride;
    int det_cs_parts;          // bias-gradient partials per (split, row): 1, or 2 for the bf16-plane kernels (two threads per row)
};

// XCD-aware bijective remap of a linear block id over `nb` blocks (guide T1).
__device__ __forceinline__ int xcd_swizzle(int b, int nb) {
    const int q = nb >> 3, r = nb & 7, xcd = b & 7, idx = b >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

// Branch-free epilogue of a full interior tile. MODE: STORE c = v; GELU c = gelu(v); RES c = v + R;
// PRE_GELU P = v, c = gelu(v); ACCUM c += v; ATOMIC atomicAdd(c, v); RES_DROP c = dropout(v) + R
// with v = acc + bias.
template <int MODE, int TM, int TN>
__device__ __forceinline__ void epilogue_full(const GemmP& p, float* __restrict__ cptr, const f32x16 (&acc)[TM][TN],
                                              const float (&bv)[TN], int row0, int col0) {
    const float* __restrict__ rbase =
        (MODE == EPI_RES || MODE == EPI_RES_DROP) ? p.R + (long)row0 * p.ldr + col0 : nullptr;
    float* __restrict__ pbase = MODE == EPI_PRE_GELU ? p.P + (long)row0 * p.ldp + col0 : nullptr;
    const uint64_t seed = MODE == EPI_RES_DROP ? vb_seed_with_epoch(p.seed, p.epoch) : 0;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int dr = i * 32 + (r & 3) + 8 * (r >> 2);
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                float v = acc[i][j][r] + bv[j];
                float* c = cptr + (long)dr * p.ldc + j * 32;
                if (MODE == EPI_PRE_GELU) pbase[(long)dr * p.ldp + j * 32] = v;
                if (MODE == EPI_GELU || MODE == EPI_PRE_GELU) v = gelu_erf(v);
                if (MODE == EPI_RES_DROP) {
                    const uint64_t idx = (uint64_t)((long)(row0 + dr) * p.N + col0 + j * 32);
                    v = vb_keep(seed, idx, p.drop_p) ? v * p.drop_scale : 0.f;
                }
                if (MODE == EPI_RES || MODE == EPI_RES_DROP) v += rbase[(long)dr * p.ldr + j * 32];
                if (MODE == EPI_ATOMIC) unsafeAtomicAdd(c, v);
                else if (MODE == EPI_ACCUM) *c += v;
                else *c = v;
            }
        }
    }
}

// Epilogue of one (64 TM) x (64 TN) tile: fused bias gradient (column sums), bias, activation, residual,
// pre-activation store, plain / accumulating / atomic stores.
template <int TM, int TN, bool A_KC, bool B_KC>
__device__ __forceinline__ void tile_epilogue(const GemmP& p, const f32x16 (&acc)[TM][TN], const int m0, const int n0,
                                              const bool want_colsum, const float csum, const int crow,
                                              const int cpart = 0) {
    constexpr int RA = 64 * TM, RB = 64 * TN;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    // C row segment of this tile (tiles never straddle segments: cseg is a multiple of the tile rows)
    const int cs = m0 / p.cseg;
    const int mloc = m0 - cs * p.cseg;  // row of the tile inside its segment
    // crow = the tile row whose (partial) sum over this block's K range the thread holds
    if (p.det_ws != nullptr) {
        // deterministic split-K (weight gradients): this split's partial tile goes to its workspace slice as a plain
        // [M, N] matrix, its bias-gradient partial to det_cs; splitk_reduce_kernel adds the slices in split order
        if (want_colsum && mloc + crow < p.cseg && m0 + crow < p.M)
            p.det_cs[((long)blockIdx.y * p.det_cs_parts + cpart) * p.M + m0 + crow] = csum;
        float* __restrict__ w = p.det_ws + (long)blockIdx.y * p.det_stride;
        const int r0d = m0 + wm * 32 * TM + 4 * hi, c0d = n0 + wn * 32 * TN + l31;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = c0d + j * 32;
            if (col >= p.N) continue;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = r0d + i * 32 + (r & 3) + 8 * (r >> 2);
                    if (row < p.M) w[(long)row * p.N + col] = acc[i][j][r];
                }
            }
        }
        return;
    }
    if (want_colsum && mloc + crow < p.cseg && m0 + crow < p.M) unsafeAtomicAdd(p.colsum[cs] + mloc + crow, csum);

    // Epilogue. Accumulator map (32x32): col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5).
    // Interior tiles with one of the common epilogues take a branch-free specialised path (the generic
    // predicated loop costs ~2k VALU instructions per wave, during which the matrix pipe starves when
    // the co-resident blocks reach their epilogues together).
    const bool lead = blockIdx.y == 0;  // bias / residual are added by one split only
    float bv[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int col = n0 + wn * 32 * TN + j * 32 + l31;
        bv[j] = 0.f;
        if (col < p.N && lead) {
            const int sg = B_KC ? col / p.bseg : 0;  // bias follows the N segmentation of a k-contiguous B
            const float* bp = p.bias[sg];
            if (bp != nullptr) bv[j] = bp[col - sg * p.bseg * (B_KC ? 1 : 0)];
        }
    }
    const int row0 = m0 + wm * 32 * TM + 4 * hi, col0 = n0 + wn * 32 * TN + l31;
    float* cptr = p.C[cs] + (long)(row0 - cs * p.cseg) * p.ldc + col0;
    if (m0 + RA <= p.M && n0 + RB <= p.N && p.epi != EPI_GENERIC) {
        switch (p.epi) {
            case EPI_STORE: epilogue_full<EPI_STORE, TM, TN>(p, cptr, acc, bv, row0, col0); break;
            case EPI_GELU: epilogue_full<EPI_GELU, TM, TN>(p, cptr, acc, bv, row0, col0); break;
            case EPI_RES: epilogue_full<EPI_RES, TM, TN>(p, cptr, acc, bv, row0, col0); break;
            case EPI_PRE_GELU: epilogue_full<EPI_PRE_GELU, TM, TN>(p, cptr, acc, bv, row0, col0); break;
            case EPI_ACCUM: epilogue_full<EPI_ACCUM, TM, TN>(p, cptr, acc, bv, row0, col0); break;
            case EPI_RES_DROP: epilogue_full<EPI_RES_DROP, TM, TN>(p, cptr, acc, bv, row0, col0); break;
            default: epilogue_full<EPI_ATOMIC, TM, TN>(p, cptr, acc, bv, row0, col0); break;
        }
        return;
    }
    const bool split = gridDim.y > 1;
    const uint64_t seed = p.drop_p > 0.f ? vb_seed_with_epoch(p.seed, p.epoch) : 0;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int col = col0 + j * 32;
        if (col >= p.N) continue;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int dr = i * 32 + (r & 3) + 8 * (r >> 2);
                const int row = row0 + dr;
                if (row >= p.M) continue;
                float v = acc[i][j][r] + bv[j];
                if (p.P != nullptr) p.P[(long)row * p.ldp + col] = v;
                if (p.act == VB_ACT_GELU) v = gelu_erf(v);
                else if (p.act == VB_ACT_RELU) v = fmaxf(v, 0.f);
                else if (p.act == VB_ACT_SWISH) v = swish_act(v);
                if (p.drop_p > 0.f) v = vb_keep(seed, (uint64_t)((long)row * p.N + col), p.drop_p) ? v * p.drop_scale : 0.f;
                if (p.R != nullptr && lead) v += p.R[(long)row * p.ldr + col];
                float* c = cptr + (long)dr * p.ldc + j * 32;
                if (split) unsafeAtomicAdd(c, v);
                else if (p.accumulate) *c += v;
                else *c = v;
            }
        }
    }
}

// Tile plan: full rounds of 256 big tiles, leftover as small tiles when that shortens the tail.
inline void plan_tiles(GemmP& p, int splits, bool planes_mode) {
    const int tiles_m = (p.M + 127) / 128;
    p.tiles_n = (p.N + 127) / 128;
    const int total = tiles_m * p.tiles_n;
    static const int hybrid = [] { const char* e = getenv("VB_GEMM_HYBRID"); return e ? atoi(e) : 1; }();
    const int left = total % 256;
    // 4 * left small tiles cost ceil(4 left / 256) quarter-rounds vs one full big round (= 4). A small tile
    // runs at ~3/4 of a big tile's MFMA efficiency in the fp32 kernel (re-cut when < 4 quarter-rounds) but
    // at ~1/2 in the bf16-planes kernel, whose per-thread split work does not shrink with the tile
    // (re-cut only when the tail fits ONE quarter-round).
    const int limit = planes_mode ? 2 : 4;
    const bool recut = hybrid && splits == 1 && left > 0 && (4 * left + 255) / 256 < limit && (p.cseg % 64) == 0;
    p.n_big = recut ? total - left : total;
    p.n_small = recut ? 4 * left : 0;
}

// Staging of one R x 16 operand tile into registers (R / 64 float4 per thread).
// k-contiguous operand (global [rows][ld]): thread t owns rows (t >> 2) + 64 it and the four k values
// 4 (t & 3) .. +3 of every K tile, so the row base pointers are computed once per block (this is also
// where a row is mapped to its weight segment).
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


// post-passes of the round-1 kernel for the two epilogues only the second-generation kernel fuses (elementwise.hip)
int launch_act_grad_inplace(hipStream_t st, long rows, int cols, float* d, long ld, int act);
int launch_mul_inplace(hipStream_t st, long rows, int cols, float* c, long ldc, const float* m, long ldm);

// second-generation fp32 kernels (gemm_v2.hip, one object per operand layout): launch tile (32 tm) x (32 tn)
// (p.n_big blocks of (32 tm1) x (32 tn) over rows [0, p.m_split), then tiles - p.n_big blocks of (32 tm2) x (32 tn))
int launch_gemm_v2_nt(hipStream_t st, const GemmP& p, int tm1, int tm2, int tn, int tiles, int splits);
int launch_gemm_v2_nn(hipStream_t st, const GemmP& p, int tm1, int tm2, int tn, int tiles, int splits);
int launch_gemm_v2_tn(hipStream_t st, const GemmP& p, int tm1, int tm2, int tn, int tiles, int splits);
// persistent one-block-per-CU kernels (gemm_v4.h): 288 x (32 tn) tiles, forward / dgrad layouts, no split-K
int launch_gemm_v4_nt(hipStream_t st, const GemmP& p, int tn);
int launch_gemm_v4_nn(hipStream_t st, const GemmP& p, int tn);
// persistent weight-gradient kernel (gemm_v4w.h); the launch parameters are filled by plan_v4w (gemm.hip)
int launch_gemm_v4_tn(hipStream_t st, const GemmP& p, int cfg);

// Deterministic split-K workspace (gemm.hip, vb_set_deterministic) for the kernels of other translation units (gemm_bf16.hip):
// det_on() = the setting; det_slice(stream, &bytes) = the slice of (current device, stream) or nullptr (no workspace / all
// slices taken); det_fallback() counts a launch that ran with atomics although the setting is on.
bool det_on();
float* det_slice(hipStream_t st, size_t* slice_bytes);
void det_fallback();
// bf16-planes kernels (gemm_planes.hip, one object per plane count): launch for operand layouts (a_kc, b_kc)
int launch_gemm_planes3(hipStream_t st, const GemmP& p, bool vec, int splits, bool a_kc, bool b_kc);
int launch_gemm_planes2(hipStream_t st, const GemmP& p, bool vec, int splits, bool a_kc, bool b_kc);
int launch_gemm_planes1(hipStream_t st, const GemmP& p, bool vec, int splits, bool a_kc, bool b_kc);

}  // namespace vbgemm
