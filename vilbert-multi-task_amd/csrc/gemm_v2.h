// Second-generation fp32 GEMM main loop (v_mfma_f32_16x16x4_f32): the kernel every aligned launch takes.
//
// Why a second main loop (round-1 kernel = gemm_tile in gemm.hip, kept as the generic / ragged fallback):
//   * tile quantisation - the model's GEMMs have M = batch * 36 rows and N in {768, 1024, 2304, 3072}; with
//     128 x 128 tiles that is 432 / 576 tiles for 256 CUs (1.7 / 2.25 blocks per CU: the last round runs on a
//     quarter of the chip). 16 x 16 MFMA tiles make 96-row / 96-column block tiles possible: 9216 x 768 =
//     768 tiles of 96 x 96 (exactly 3 per CU), 9216 x 1024 = 768 tiles of 96 x 128, 9216 x 3072 = 2304 tiles of
//     96 x 128 (exactly 9 per CU) - see plan_tiles_v2;
//   * a wave must keep its SIMD's matrix pipe fed on its own: fragments of the NEXT K step are fetched from LDS
//     while the MFMAs of the current one run (register double buffer across the barrier, possible because the LDS
//     ring has 3 stages: the stage read next was published by the PREVIOUS barrier), global loads get a whole K
//     step of latency (issued in step t, written to LDS at the top of step t + 1), one barrier per K step and
//     nothing but the barrier skew is exposed;
//   * the MFMA is issued transposed (D^T = B . A^T) so that a lane owns 4 consecutive output COLUMNS of one row:
//     the epilogue moves float4 (bias, residual, mask, store) - 4x fewer memory instructions than the 32 x 32 map.
//
// Block = 256 threads = 4 waves (2 x 2); wave tile = (16 TM) x (16 TN), block tile BM x BN = (32 TM) x (32 TN),
// TM, TN in {2, 3, 4}. K step 16.
// LDS (floats), per stage [A tile | B tile]:
//   k-contiguous operand  [rows][16], 16-byte slot index XOR-swizzled by the row: slot' = slot ^ ((-(row >> 2)) & 3)
//       -> the ds_read_b128 of a fragment (lane = row & 15, slot = lane >> 4) and the staging ds_write_b128 are both
//       bank-conflict free without padding;
//   row-contiguous operand [16 k][rows + 4]: ds_read_b32, lanes = consecutive rows.
// MFMA operand convention (16x16x4): lane l supplies A[i = l & 15][k = l >> 4], B[k = l >> 4][j = l & 15]. Inside a
// K step lane group g = l >> 4 owns k = 4 g + e (e = 0..3, one MFMA each), for A and B alike, so a k-contiguous
// operand needs one ds_read_b128 per fragment per K step; this only permutes the fp32 summation order.
#pragma once
#include <type_traits>

#include "gemm_core.h"

namespace vbgemm {

constexpr int V2_BK = 16;
constexpr int V2_STAGES = 3;

template <bool KC, int R>
constexpr int v2_oper_floats() { return KC ? R * 16 : 16 * (R + 4); }

template <int TM, int TN, bool A_KC, bool B_KC>
struct V2Cfg {
    static constexpr int BM = 32 * TM, BN = 32 * TN;
    static constexpr int A_SZ = v2_oper_floats<A_KC, BM>();
    static constexpr int B_SZ = v2_oper_floats<B_KC, BN>();
    static constexpr int STAGE = A_SZ + B_SZ;
    // + one 16-byte dump slot per thread: staging slots past the end of a 96-row tile (1.5 float4 per thread) store
    // there instead of branching, so the K loop stays one basic block
    static constexpr int DUMP = V2_STAGES * STAGE;
    static constexpr int LDS_BYTES = (V2_STAGES * STAGE + 256 * 4) * 4;
    // co-resident blocks per CU the kernel is compiled for (register budget 512 / OCC per lane)
    // (a row-contiguous operand tile is padded: 4 blocks of 96 x 96 no longer fit the 160 KiB of LDS)
    static constexpr int OCC = (TM * TN <= 9 && A_KC && B_KC) ? 4 : 3;
};

__device__ __forceinline__ int v2_swz(int row) { return (-(row >> 2)) & 3; }

// Epilogue of the transposed accumulator map: lane (l15 = lane & 15, g = lane >> 4), register r of tile (i, j)
// holds C[row = tile_m + l15][col = tile_n + 4 g + r]. MODE as in gemm_core.h plus EPI_MUL (c = v * R: the
// saved activation derivative applied to the incoming gradient).
template <int MODE, int TM, int TN, bool BIAS_SEG, int EXT_DEPTH = 2>
__device__ __forceinline__ void epilogue_v2(const GemmP& p, float* __restrict__ cbase, const f32x4 (&acc)[TM][TN],
                                            int row0, int col0, bool lead, bool full) {
    // row0 / col0: this lane's first row / column (global indices); cbase + row * ldc addresses global row `row`
    const uint64_t seed = MODE == EPI_RES_DROP ? vb_seed_with_epoch(p.seed, p.epoch) : 0;
    constexpr bool EXT = EXT_DEPTH > 0 && (MODE == EPI_RES || MODE == EPI_RES_DROP || MODE == EPI_MUL || MODE == EPI_ACCUM);
    if (EXT && full) {
        // Interior tile of an epilogue that READS a second [M, N] operand (residual, activation derivative, C itself):
        // all of the wave tile's loads are issued before the first store. The generic loop below interleaves
        // load -> use -> store per fragment behind per-fragment bounds branches, i.e. TM x TN dependent trips to memory
        // per tile (round 3: the image stream's dgrads, which all carry such an operand, ran at 99 TF against 113 TF for
        // its forward launches - profiles/r03_bench_train_b256_gemm_breakdown.txt).
        const float* ebase = MODE == EPI_MUL ? p.mul : (MODE == EPI_ACCUM ? cbase : p.R);
        const long eld = MODE == EPI_MUL ? p.ldmul : (MODE == EPI_ACCUM ? (long)p.ldc : (long)p.ldr);
        const bool use = MODE == EPI_MUL || MODE == EPI_ACCUM || lead;
        // rolling by output column block (EXT_DEPTH 2): the TM loads of block j + 1 are in flight while block j is finished
        // and stored; EXT_DEPTH 1 (the persistent kernels, which run at a hard 128-register budget): the TM loads of a
        // block together, then its stores
        f32x4 ext[EXT_DEPTH > 0 ? EXT_DEPTH : 1][TM];
        auto fetch = [&](int j, f32x4 (&dst)[TM]) {
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                dst[i] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (use) dst[i] = *reinterpret_cast<const f32x4*>(ebase + (long)(row0 + i * 16) * eld + col0 + j * 16);
            }
        };
        if (EXT_DEPTH == 2) fetch(0, ext[0]);
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            if (EXT_DEPTH == 1) fetch(j, ext[0]);
            else if (j + 1 < TN) fetch(j + 1, ext[(j + 1) & (EXT_DEPTH > 1 ? 1 : 0)]);
            const int col = col0 + j * 16;
            f32x4 bv = {0.f, 0.f, 0.f, 0.f};
            if (lead) {
                const int sg = BIAS_SEG ? col / p.bseg : 0;
                const float* bp = p.bias[sg];
                if (bp != nullptr) bv = *reinterpret_cast<const f32x4*>(bp + (col - (BIAS_SEG ? sg * p.bseg : 0)));
            }
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int row = row0 + i * 16;
                f32x4 v = acc[i][j] + bv;
                if (MODE == EPI_RES_DROP) {
                    const uint64_t idx = (uint64_t)((long)row * p.N + col);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = vb_keep(seed, idx + e, p.drop_p) ? v[e] * p.drop_scale : 0.f;
                }
                if (MODE == EPI_MUL) v *= ext[j & (EXT_DEPTH > 1 ? 1 : 0)][i];
                else v += ext[j & (EXT_DEPTH > 1 ? 1 : 0)][i];
                *reinterpret_cast<f32x4*>(cbase + (long)row * p.ldc + col) = v;
            }
        }
        return;
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int col = col0 + j * 16;
        if (!full && col >= p.N) continue;
        f32x4 bv = {0.f, 0.f, 0.f, 0.f};
        if (lead) {
            const int sg = BIAS_SEG ? col / p.bseg : 0;   // the bias follows the N segmentation of a k-contiguous B
            const float* bp = p.bias[sg];
            if (bp != nullptr) bv = *reinterpret_cast<const f32x4*>(bp + (col - (BIAS_SEG ? sg * p.bseg : 0)));
        }
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int row = row0 + i * 16;
            if (!full && row >= p.M) continue;
            f32x4 v = acc[i][j] + bv;
            float* c = cbase + (long)row * p.ldc + col;
            if (MODE == EPI_DGELU) {
                // forward of an FFN up-projection in training: store the activation AND its derivative (the
                // backward then needs one multiply in a dgrad epilogue - no erf / exp, no separate pass)
                f32x4 d;
#pragma unroll
                for (int e = 0; e < 4; ++e) { float y, dd; gelu_and_grad(v[e], y, dd); v[e] = y; d[e] = dd; }
                *reinterpret_cast<f32x4*>(p.D + (long)row * p.ldd + col) = d;
            } else if (MODE == EPI_GELU) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = gelu_erf(v[e]);
            }
            if (MODE == EPI_RES_DROP) {
                const uint64_t idx = (uint64_t)((long)row * p.N + col);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = vb_keep(seed, idx + e, p.drop_p) ? v[e] * p.drop_scale : 0.f;
            }
            if (MODE == EPI_RES || MODE == EPI_RES_DROP) {
                if (lead) v += *reinterpret_cast<const f32x4*>(p.R + (long)row * p.ldr + col);
            }
            if (MODE == EPI_MUL) v *= *reinterpret_cast<const f32x4*>(p.mul + (long)row * p.ldmul + col);
            if (MODE == EPI_ATOMIC) {
#pragma unroll
                for (int e = 0; e < 4; ++e) unsafeAtomicAdd(c + e, v[e]);
            } else if (MODE == EPI_ACCUM) {
                *reinterpret_cast<f32x4*>(c) = *reinterpret_cast<const f32x4*>(c) + v;
            } else if (p.flags & 64) {
                __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(c));   // lab: streaming stores of the C tile
            } else {
                *reinterpret_cast<f32x4*>(c) = v;
            }
        }
    }
}

// Epilogue of the NATURAL accumulator map (wgrad): register r of tile (i, j) holds C[row = tile_m + 4 g + r][col =
// tile_n + l15] - one instruction covers 4 rows x 64 contiguous bytes, which is what the split-K atomics want (the
// transposed map would scatter each atomic instruction over 16 rows).
template <int MODE, int TM, int TN>
__device__ __forceinline__ void epilogue_v2_nat(const GemmP& p, float* __restrict__ cbase, const f32x4 (&acc)[TM][TN],
                                                int row0, int col0, bool full) {
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int col = col0 + j * 16;
        if (!full && col >= p.N) continue;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = row0 + i * 16 + r;
                if (!full && row >= p.M) continue;
                float* c = cbase + (long)row * p.ldc + col;
                if (MODE == EPI_ATOMIC) unsafeAtomicAdd(c, acc[i][j][r]);
                else if (MODE == EPI_ACCUM) *c += acc[i][j][r];
                else *c = acc[i][j][r];
            }
        }
    }
}

template <int TM, int TN, bool A_KC, bool B_KC, int ABL>
__device__ __forceinline__ void gemm_tile_v2(const GemmP& p, float* __restrict__ smem, const int m0, const int n0) {
    using Cfg = V2Cfg<TM, TN, A_KC, B_KC>;
    constexpr int BM = Cfg::BM, BN = Cfg::BN;
    constexpr int SA = (BM * 4 + 255) / 256, SB = (BN * 4 + 255) / 256;   // float4 staging slots per thread
    constexpr int TMa = (TM + 1) / 2;                                      // first-half tile rows
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, g = lane >> 4;
    const int wm = wave >> 1, wn = wave & 1;

    const int kt_total = p.K / V2_BK;
    const int kt_begin = blockIdx.y * p.ktiles_per_split;
    const int nk = min(kt_total, kt_begin + p.ktiles_per_split) - kt_begin;
    if (nk <= 0) return;

    // ---- staging: global source pointers + LDS destinations of this thread's float4 slots ---------------------
    const float* ga[SA];
    const float* gb[SB];
    int la[SA], lb[SB];
    long b_off[SB];   // row-contiguous B: offset inside the (segment) k-row block
    const long a_step = A_KC ? V2_BK : (long)V2_BK * p.lda;
#pragma unroll
    for (int s = 0; s < SA; ++s) {
        const int f = min(tid + 256 * s, BM * 4 - 1);
        if (A_KC) {
            const int row = f >> 2, quad = f & 3;
            ga[s] = p.A + (long)min(m0 + row, p.M - 1) * p.lda + (long)kt_begin * V2_BK + quad * 4;
            la[s] = row * 16 + ((quad ^ v2_swz(row)) << 2);
        } else {
            const int k = f / (BM / 4), c4 = f % (BM / 4);
            int col = m0 + c4 * 4;
            if (col >= p.M) col = 0;   // rows past the matrix: any in-bounds address (their outputs are never stored)
            ga[s] = p.A + ((long)kt_begin * V2_BK + k) * p.lda + col;
            la[s] = k * (BM + 4) + c4 * 4;
        }
    }
    int b_seg = 0, b_krem = 0;   // row-contiguous B: running (segment, k inside segment) of the NEXT tile to load
#pragma unroll
    for (int s = 0; s < SB; ++s) {
        const int f = min(tid + 256 * s, BN * 4 - 1);
        if (B_KC) {
            const int row = f >> 2, quad = f & 3;
            const int n = min(n0 + row, p.N - 1);
            const int sg = n / p.bseg;
            gb[s] = p.B[sg] + (long)(n - sg * p.bseg) * p.ldb + (long)kt_begin * V2_BK + quad * 4;
            lb[s] = row * 16 + ((quad ^ v2_swz(row)) << 2);
            b_off[s] = 0;
        } else {
            const int k = f / (BN / 4), c4 = f % (BN / 4);
            int col = n0 + c4 * 4;
            if (col >= p.N) col = 0;
            gb[s] = nullptr;
            b_off[s] = (long)k * p.ldb + col;
            lb[s] = k * (BN + 4) + c4 * 4;
        }
    }
    long a_stepv = a_step;
    if (ABL == 7) {
        // timing experiment only (wrong results): every lane of the block loads the SAME 16 bytes
#pragma unroll
        for (int s = 0; s < SA; ++s) ga[s] = p.A + (long)(blockIdx.x % 64) * 65536;
#pragma unroll
        for (int s = 0; s < SB; ++s)
            if (B_KC) gb[s] = p.B[0] + (long)(blockIdx.x % 4) * 65536;
        a_stepv = 0;
    }
    if (ABL == 6) {
        // timing experiment only (wrong results): fully contiguous 4 KiB-per-slot loads instead of 64-byte row pieces
#pragma unroll
        for (int s = 0; s < SA; ++s) ga[s] = p.A + ((long)(blockIdx.x % 64) * 65536 + s * 1024 + tid * 4);
#pragma unroll
        for (int s = 0; s < SB; ++s)
            if (B_KC) gb[s] = p.B[0] + ((long)(blockIdx.x % 4) * 65536 + s * 1024 + tid * 4);
        a_stepv = 2048;
    }
    if (!B_KC) {
        const int k0 = kt_begin * V2_BK;
        b_seg = k0 / p.bseg;
        b_krem = k0 - b_seg * p.bseg;
    }

    // Staging registers: NSET sets of one K tile each. A tile loaded in step t is written to LDS in step t + NSET
    // (its set is then reloaded), i.e. the loads of tile t + NSET + 2 are issued in step t: NSET = 1 gives a global
    // load one K step of latency, NSET = 2 two steps (up to 96 x 128 tiles, where the registers allow it).
    constexpr int NSET = 1;   // (2 measured: no gain where it fits the registers - the loads are not latency-exposed)
    f32x4 ra[NSET][SA], rb[NSET][SB];
    // staging, one float4 slot at a time (u < SA: A slots, then B slots) so that the K loop can place every memory
    // instruction by hand between two MFMAs
    auto load_slot = [&](int u, int set) {
        if (u < SA) {
            ra[set][u] = *reinterpret_cast<const f32x4*>(ga[u]);
            if (ABL != 5) ga[u] += a_stepv;
        } else if (B_KC) {
            rb[set][u - SA] = *reinterpret_cast<const f32x4*>(gb[u - SA]);
            if (ABL != 5 && ABL != 7) gb[u - SA] += ABL == 6 ? 2048 : V2_BK;
        } else {
            // segments stacked along K (dgrad through stacked weights); bseg is a multiple of 16
            const float* __restrict__ bb = p.B[b_seg] + (long)b_krem * p.ldb;
            rb[set][u - SA] = *reinterpret_cast<const f32x4*>(bb + b_off[u - SA]);
            if (u == SA + SB - 1) {
                b_krem += V2_BK;
                if (b_krem >= p.bseg) { b_krem = 0; ++b_seg; }
            }
        }
    };
    const bool a_last_ok = (BM * 4) % 256 == 0 || tid + 256 * (SA - 1) < BM * 4;
    const bool b_last_ok = (BN * 4) % 256 == 0 || tid + 256 * (SB - 1) < BN * 4;
    auto store_slot = [&](int u, float* __restrict__ st, int set) {
        if (ABL == 3) {   // lab: wait for the staged data where the LDS write would be, write nothing
            if (u < SA) asm volatile("" ::"v"(ra[set][u]));
            else asm volatile("" ::"v"(rb[set][u - SA]));
            return;
        }
        // slots past the end of a 96-row tile (1.5 float4 per thread) go to the dump area instead of branching
        if (u < SA) {
            float* dst = (u + 1 < SA || a_last_ok) ? st + la[u] : smem + Cfg::DUMP + tid * 4;
            *reinterpret_cast<f32x4*>(dst) = ra[set][u];
        } else {
            float* dst = (u + 1 < SA + SB || b_last_ok) ? st + Cfg::A_SZ + lb[u - SA] : smem + Cfg::DUMP + tid * 4;
            *reinterpret_cast<f32x4*>(dst) = rb[set][u - SA];
        }
    };
    auto load_tile = [&](int set) {
#pragma unroll
        for (int u = 0; u < SA + SB; ++u) load_slot(u, set);
    };
    auto store_tile = [&](float* __restrict__ st, int set) {
#pragma unroll
        for (int u = 0; u < SA + SB; ++u) store_slot(u, st, set);
    };

    // ---- fragment reads ----------------------------------------------------------------------------------------
    const int a_frag = A_KC ? (wm * 16 * TM + l15) * 16 + ((g ^ v2_swz(l15)) << 2) : (4 * g) * (BM + 4) + wm * 16 * TM + l15;
    const int b_frag = Cfg::A_SZ + (B_KC ? (wn * 16 * TN + l15) * 16 + ((g ^ v2_swz(l15)) << 2)
                                         : (4 * g) * (BN + 4) + wn * 16 * TN + l15);
    auto read_a = [&](const float* __restrict__ st, int i) -> f32x4 {
        if (A_KC) return *reinterpret_cast<const f32x4*>(st + a_frag + i * 256);
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = st[a_frag + e * (BM + 4) + i * 16];
        return v;
    };
    auto read_b = [&](const float* __restrict__ st, int j) -> f32x4 {
        if (B_KC) return *reinterpret_cast<const f32x4*>(st + b_frag + j * 256);
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = st[b_frag + e * (BN + 4) + j * 16];
        return v;
    };

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

#ifdef VB_GEMM_LAB
    // lab timeline (tools/gemm_lab LAB_TIMELINE=1): per block {realtime start, after prologue, after K loop, after
    // epilogue (stores drained), shader cycles start, end, hardware id}; realtime = s_memrealtime, 100 MHz
    unsigned long long* const tl = p.dbg != nullptr ? p.dbg + 8 * ((long)blockIdx.y * gridDim.x + blockIdx.x) : nullptr;
    if (tl != nullptr && tid == 0) { tl[0] = wall_clock64(); tl[4] = __builtin_readcyclecounter(); }
#endif
    // ---- prologue: tiles 0, 1 into stages 0, 1; tiles 2 .. NSET + 1 in flight ----------------------------------
    load_tile(0);
    store_tile(smem, 0);
    if (nk > 1) {
        load_tile(0);
        store_tile(smem + Cfg::STAGE, 0);
    }
    if (nk > 2) load_tile(0);
    if (NSET == 2 && nk > 3) load_tile(1);
    __syncthreads();
#ifdef VB_GEMM_LAB
    if (tl != nullptr && tid == 0) tl[1] = wall_clock64();
#endif

    // fragment registers: first-half A tiles and all B tiles are double buffered (set = K step parity), the
    // second-half A tiles are read at the top of their own step
    constexpr int TMb = TM - TMa;
    f32x4 afa[2][TMa], afb[TMb > 0 ? TMb : 1], bfr[2][TN];
#pragma unroll
    for (int i = 0; i < TMa; ++i) afa[0][i] = read_a(smem, i);
#pragma unroll
    for (int j = 0; j < TN; ++j) bfr[0][j] = read_b(smem, j);

    const bool want_colsum = !A_KC && n0 == 0 && tid < BM && p.colsum[0] != nullptr;
    float csum = 0.f;
    int cur = 0;   // stage of K tile t

    auto mfma_at = [&](int m, int P) {
        // MFMA number m of a step: first half = tile rows [0, TMa), second half = [TMa, TM); inside a half the
        // contraction index e is outermost so that consecutive MFMAs never share an accumulator
        constexpr int NA = 4 * TMa * TN;
        const int h = m < NA ? 0 : 1;
        const int mm = h ? m - NA : m;
        const int rows = h ? TMb : TMa;
        const int e = mm / (rows * TN), r = mm % (rows * TN);
        const int i = r / TN, j = r % TN;
        // forward / dgrad: transposed product (lane = 4 consecutive columns of one row); wgrad: natural
        if (A_KC) {
            if (!h) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(bfr[P][j][e], afa[P][i][e], acc[i][j], 0, 0, 0);
            else acc[TMa + i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(bfr[P][j][e], afb[i][e], acc[TMa + i][j], 0, 0, 0);
        } else {
            if (!h) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(afa[P][i][e], bfr[P][j][e], acc[i][j], 0, 0, 0);
            else acc[TMa + i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(afb[i][e], bfr[P][j][e], acc[TMa + i][j], 0, 0, 0);
        }
    };

    // Steady-state K step t (parity P), hand-scheduled: every MFMA is followed by at most one memory "unit" and a
    // scheduling fence, so the instruction order is exactly this source order -
    //   LDS writes of tile t + 2 (loaded during step t - 1; its registers are reloaded right after),
    //   global loads of tile t + 3, LDS reads of the fragments of step t + 1 (stage published by the last barrier).
    auto full_step = [&](auto parity) {
        constexpr int P = decltype(parity)::value;
        const float* __restrict__ sc = smem + cur * Cfg::STAGE;
        const int nxt = cur == V2_STAGES - 1 ? 0 : cur + 1;
        const int nn = nxt == V2_STAGES - 1 ? 0 : nxt + 1;
        float* __restrict__ sw = smem + nn * Cfg::STAGE;
        const float* __restrict__ sn = smem + nxt * Cfg::STAGE;
#pragma unroll
        for (int i = 0; i < TMb; ++i) afb[i] = read_a(sc, TMa + i);
        __builtin_amdgcn_sched_barrier(0);
        constexpr int NU = SA + SB;
        // memory units of a step, in issue order: NU LDS writes, NU global loads, TMa + TN fragment reads; one unit
        // after every SPREAD-th MFMA (all waves of a CU run this code in near lockstep: back-to-back memory
        // instructions from 16 waves queue up in the texture / LDS address paths and the waves stall at issue)
        constexpr int UNITS = 2 * NU + TMa + TN;
        constexpr int SPREAD = (4 * TM * TN) / UNITS > 0 ? (4 * TM * TN) / UNITS : 1;
#pragma unroll
        for (int m = 0; m < 4 * TM * TN; ++m) {
            mfma_at(m, P);
            const int u = m / SPREAD;
            if (m % SPREAD == 0 && u < UNITS) {
                if (u < NU) { if (ABL != 1) store_slot(u, sw, NSET == 2 ? P : 0); }
                else if (u < 2 * NU) { if (ABL != 1 && ABL != 4) load_slot(u - NU, NSET == 2 ? P : 0); }
                else if (u < 2 * NU + TMa) afa[P ^ 1][u - 2 * NU] = read_a(sn, u - 2 * NU);
                else bfr[P ^ 1][u - 2 * NU - TMa] = read_b(sn, u - 2 * NU - TMa);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (want_colsum) {
#pragma unroll
            for (int kk = 0; kk < V2_BK; ++kk) csum += sc[kk * (BM + 4) + tid];
        }
        if (ABL != 2) __syncthreads();
        cur = nxt;
    };
    // Tail steps (the last <= 4 of a tile): same data flow with run-time conditions, fragment set 0 is current.
    auto tail_step = [&](bool do_store, bool do_load, bool do_next, auto set_c) {
        constexpr int SET = decltype(set_c)::value;
        const float* __restrict__ sc = smem + cur * Cfg::STAGE;
        const int nxt = cur == V2_STAGES - 1 ? 0 : cur + 1;
        const int nn = nxt == V2_STAGES - 1 ? 0 : nxt + 1;
        if (do_store) store_tile(smem + nn * Cfg::STAGE, SET);
        if (do_load) load_tile(SET);
#pragma unroll
        for (int i = 0; i < TMb; ++i) afb[i] = read_a(sc, TMa + i);
        if (do_next) {
            const float* __restrict__ sn = smem + nxt * Cfg::STAGE;
#pragma unroll
            for (int i = 0; i < TMa; ++i) afa[1][i] = read_a(sn, i);
#pragma unroll
            for (int j = 0; j < TN; ++j) bfr[1][j] = read_b(sn, j);
        }
#pragma unroll
        for (int m = 0; m < 4 * TM * TN; ++m) mfma_at(m, 0);
        if (want_colsum) {
#pragma unroll
            for (int kk = 0; kk < V2_BK; ++kk) csum += sc[kk * (BM + 4) + tid];
        }
        __syncthreads();
        if (do_next) {
#pragma unroll
            for (int i = 0; i < TMa; ++i) afa[0][i] = afa[1][i];
#pragma unroll
            for (int j = 0; j < TN; ++j) bfr[0][j] = bfr[1][j];
        }
        cur = nxt;
    };
    int t = 0;
    for (; t + 3 + NSET < nk; t += 2) {   // both steps issue loads: tile (t + 1) + NSET + 2 must exist
        full_step(std::integral_constant<int, 0>{});
        full_step(std::integral_constant<int, 1>{});
    }
    for (; t < nk; ++t) {
        if (NSET == 2 && (t & 1)) tail_step(t + 2 < nk, t + 2 + NSET < nk, t + 1 < nk, std::integral_constant<int, NSET - 1>{});
        else tail_step(t + 2 < nk, t + 2 + NSET < nk, t + 1 < nk, std::integral_constant<int, 0>{});
    }

#ifdef VB_GEMM_LAB
    if (tl != nullptr && tid == 0) tl[2] = wall_clock64();
    struct LabEnd {
        unsigned long long* tl; int tid;
        __device__ ~LabEnd() {
            if (tl == nullptr) return;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (tid == 0) {
                tl[3] = wall_clock64();
                tl[5] = __builtin_readcyclecounter();
                unsigned hw, xcc;
                asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
                asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
                tl[6] = ((unsigned long long)xcc << 32) | hw;
            }
        }
    } lab_end{tl, tid};
#endif
    // ---- epilogue ----------------------------------------------------------------------------------------------
    const int cs = m0 / p.cseg;                 // C row segment of this tile (tiles never straddle segments)
    const int mloc = m0 - cs * p.cseg;
    const bool full = m0 + BM <= p.M && n0 + BN <= p.N;
    if (p.det_ws != nullptr) {
        // deterministic split-K: this split's partial tile as a plain store into its own [M, N] workspace slice
        if (want_colsum && m0 + tid < p.M) p.det_cs[(long)blockIdx.y * p.M + m0 + tid] = csum;
        GemmP q = p;
        q.ldc = p.N;
        float* wbase = p.det_ws + (long)blockIdx.y * p.det_stride;
        if (!A_KC) epilogue_v2_nat<EPI_STORE, TM, TN>(q, wbase, acc, m0 + wm * 16 * TM + 4 * g, n0 + wn * 16 * TN + l15, full);
        else epilogue_v2<EPI_STORE, TM, TN, false>(q, wbase, acc, m0 + wm * 16 * TM + l15, n0 + wn * 16 * TN + 4 * g, false, full);
        return;
    }
    if (want_colsum && mloc + tid < p.cseg && m0 + tid < p.M) unsafeAtomicAdd(p.colsum[cs] + mloc + tid, csum);
    const bool lead = blockIdx.y == 0;          // bias / residual are added by one split only
    float* cbase = p.C[cs] - (long)cs * p.cseg * p.ldc;   // so that cbase + row * ldc addresses global row `row`
    if (!A_KC) {
        const int r0 = m0 + wm * 16 * TM + 4 * g, c0 = n0 + wn * 16 * TN + l15;
        if (p.epi == EPI_ATOMIC) epilogue_v2_nat<EPI_ATOMIC, TM, TN>(p, cbase, acc, r0, c0, full);
        else if (p.epi == EPI_ACCUM) epilogue_v2_nat<EPI_ACCUM, TM, TN>(p, cbase, acc, r0, c0, full);
        else epilogue_v2_nat<EPI_STORE, TM, TN>(p, cbase, acc, r0, c0, full);
        return;
    }
    const int row0 = m0 + wm * 16 * TM + l15, col0 = n0 + wn * 16 * TN + 4 * g;
    // only the epilogues a layout can be launched with are instantiated: forward (NT) = store / gelu / residual /
    // dropout; dgrad (NN) = store / residual / accumulate / multiply / atomic (split-K of a small output); wgrad (TN) = store / accumulate / atomic
    constexpr bool FWD = A_KC && B_KC, DGRAD = A_KC && !B_KC;
    if (FWD && p.epi == EPI_GELU) epilogue_v2<EPI_GELU, TM, TN, FWD>(p, cbase, acc, row0, col0, lead, full);
    else if (FWD && p.epi == EPI_DGELU) epilogue_v2<EPI_DGELU, TM, TN, FWD>(p, cbase, acc, row0, col0, lead, full);
    else if (FWD && p.epi == EPI_RES_DROP) epilogue_v2<EPI_RES_DROP, TM, TN, FWD>(p, cbase, acc, row0, col0, lead, full);
    else if ((FWD || DGRAD) && p.epi == EPI_RES) epilogue_v2<EPI_RES, TM, TN, FWD>(p, cbase, acc, row0, col0, lead, full);
    else if (DGRAD && p.epi == EPI_MUL) epilogue_v2<EPI_MUL, TM, TN, FWD>(p, cbase, acc, row0, col0, lead, full);
    else if (DGRAD && p.epi == EPI_ACCUM) epilogue_v2<EPI_ACCUM, TM, TN, FWD>(p, cbase, acc, row0, col0, lead, full);
    else if (DGRAD && p.epi == EPI_ATOMIC) epilogue_v2<EPI_ATOMIC, TM, TN, FWD>(p, cbase, acc, row0, col0, lead, full);
    else epilogue_v2<EPI_STORE, TM, TN, FWD>(p, cbase, acc, row0, col0, lead, full);
}

}  // namespace vbgemm
