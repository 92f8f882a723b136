// Weight-gradient GEMM on the persistent one-block-per-CU design of gemm_v4.h:
//     dW[n_out, k_in] += dY[:, n_out]^T . X[:, k_in]      (contraction over the M token rows, split over the launch)
// Both operands are row-contiguous in memory (a K tile = 16 token rows of dY / of X), the MFMA is issued in its natural
// orientation (a lane owns 4 consecutive rows of one output column: the split-K atomics then cover 64-byte row pieces,
// gemm_v2.h), the work unit of the persistent loop is (output tile, K split): all units have the same even number of K
// steps and form one stream through the 4-stage LDS ring exactly as the output tiles of gemm_v4.h do.
//   tile = 384 x 96 (6 x 2 waves of 64 x 48): n_out = 768 / 2304 / 3072 and k_in = 768 / 3072 of the text stream divide
//          evenly; W[768, 768] = 16 tiles x 16 splits of 36 K steps = 256 units (M = 9216 rows), W[3072, 768] = 64 x 4.
//   LDS   = [16 k][384] | [16 k][96] per stage, unpadded and unswizzled: the LDS-DMA then needs no masked lanes and
//          exactly 24 + 6 instructions per K step; the price is a 2-way bank conflict on the ds_read_b32 fragment reads
//          (lane groups g = 0 / 1 read k rows 4 apart at the same row offsets), i.e. ~1,350 of the 4,608 LDS cycles of a
//          K step instead of ~670 - the LDS is not the binding resource here.
//   bias gradient (column sums of dY) fused as in gemm_v2.h: the units with n0 == 0 add up their A tiles from LDS.
#pragma once
#include "gemm_v4.h"

namespace vbgemm {

constexpr int Cfg_threads_guard(int mfma_waves) { return 64 * mfma_waves; }

// WM = wave rows of the MFMA-wave grid (always 2 wave columns): 6 -> 12 MFMA waves, 384-row tiles with TM = 4 (weights with
// 768 / 2304 / 3072 rows); 4 -> 8 MFMA waves (two per SIMD, 170 registers), 256-row tiles (the 1024 / 4096-row weights of
// the image stream, the connection layers and bert_large).
template <int WM, int TM, int TN>
struct V4WCfg {
    static constexpr int MFMA_WAVES = WM * V4_WN, THREADS = 64 * (MFMA_WAVES + 1);
    static constexpr int BM = 16 * TM * WM, BN = 16 * TN * V4_WN;
    static constexpr int A_SZ = 16 * BM, B_SZ = 16 * BN;
    static constexpr int STAGE = A_SZ + B_SZ;
    static constexpr int LDS_BYTES = V4_STAGES * STAGE * 4;
    static constexpr int NA = A_SZ / 4 / 64, NB = B_SZ / 4 / 64, NI = NA + NB;
    static_assert(A_SZ % 256 == 0 && B_SZ % 256 == 0, "whole DMA instructions");
    static_assert(2 * NI <= 63, "vmcnt is a 6-bit counter");
    static_assert(2 * LDS_BYTES > 160 * 1024 && LDS_BYTES <= 160 * 1024, "exactly one block per CU");
    static_assert(BM <= Cfg_threads_guard(MFMA_WAVES), "the bias-gradient sums use one MFMA-wave thread per tile row");
};

template <int WM, int TM, int TN>
__device__ __forceinline__ void v4w_loader(const GemmP& p, const unsigned lds0, const int lane, const int nk,
                                           const int units, const int rounds) {
    using Cfg = V4WCfg<WM, TM, TN>;
    constexpr int BM = Cfg::BM, BN = Cfg::BN, NA = Cfg::NA, NB = Cfg::NB, NI = Cfg::NI, S = V4_STAGES;
    unsigned oa[NA], ob[NB];          // per-lane byte offsets inside a K tile (constant for the whole launch)
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int s = 64 * i + lane, kk = s / (BM / 4), c = s % (BM / 4);
        oa[i] = (unsigned)(((long)kk * p.lda + 4 * c) * 4);
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const int s = 64 * i + lane, kk = s / (BN / 4), c = s % (BN / 4);
        ob[i] = (unsigned)(((long)kk * p.ldb + 4 * c) * 4);
    }
    const float* abase = nullptr;     // dY + k * ldy + m0   (+ 16 rows per K step)
    const float* bbase = nullptr;     // X  + k * ldx + n0
    const int tiles = p.n_small;      // output tiles (units = tiles x splits)
    auto set_unit = [&](int u) {
        const int split = u / tiles, t = u - split * tiles;
        const int m0 = (t / p.tiles_n) * BM, n0 = (t % p.tiles_n) * BN;
        const long k0 = (long)split * nk * V2_BK;
        abase = p.A + k0 * p.lda + m0;
        bbase = p.B[0] + k0 * p.ldb + n0;
    };
    int it = 0, kt = 0, stage_w = 0;
    const int b = blockIdx.x;
    auto issue_next = [&]() {
        const unsigned la = lds0 + (unsigned)stage_w * (Cfg::STAGE * 4);
        const unsigned lb = la + Cfg::A_SZ * 4;
#pragma unroll
        for (int i = 0; i < NA; ++i) v4_glds16(oa[i], abase, la + 1024u * i);
#pragma unroll
        for (int i = 0; i < NB; ++i) v4_glds16(ob[i], bbase, lb + 1024u * i);
        abase += (long)V2_BK * p.lda;
        bbase += (long)V2_BK * p.ldb;
        stage_w = stage_w == S - 1 ? 0 : stage_w + 1;
        if (++kt == nk) {
            kt = 0;
            ++it;
            if (it < rounds) set_unit(v4_tile_of(b, it, gridDim.x, units));
        }
    };
    auto wait_pending = [&](int k_tiles) {
        if (k_tiles >= 2) v4_wait_vm<2 * NI>();
        else if (k_tiles == 1) v4_wait_vm<NI>();
        else v4_wait_vm<0>();
    };
    const int total = rounds * nk;
    set_unit(v4_tile_of(b, 0, gridDim.x, units));
    __builtin_amdgcn_s_setprio(2);
    for (int s = 0; s < S && s < total; ++s) issue_next();
    wait_pending(min(total, S) - 2);
    __builtin_amdgcn_s_barrier();            // P0
    __builtin_amdgcn_s_barrier();            // P1
    for (int g = 0, kq = 0; g < total; ++g) {
        if (++kq > nk) { kq = 1; __builtin_amdgcn_s_barrier(); }   // X: unit boundary (gemm_v4.h)
        if (g + S < total) issue_next();
        wait_pending(min(total - 1, g + S) - (g + 2));
        __builtin_amdgcn_s_barrier();
    }
}

template <int WM, int TM, int TN>
__device__ __forceinline__ void gemm_block_v4w(const GemmP& p, float* __restrict__ smem) {
    using Cfg = V4WCfg<WM, TM, TN>;
    constexpr int BM = Cfg::BM, BN = Cfg::BN, S = V4_STAGES;
    const int units = p.n_big, tiles = p.n_small;
    const int nk = p.ktiles_per_split;       // even, the same for every unit
    const int b = blockIdx.x, grid = gridDim.x;
    int rounds = 0;
    while (rounds * grid < units && v4_tile_of(b, rounds, grid, units) >= 0) ++rounds;
    if (rounds == 0) return;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (wave == Cfg::MFMA_WAVES) {
        const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) float*)smem;
        v4w_loader<WM, TM, TN>(p, __builtin_amdgcn_readfirstlane(lds0), threadIdx.x & 63, nk, units, rounds);
        return;
    }
    f32x4 acc[TM][TN], afr[2][TM], bfr[2][TN];
    int a_frag, b_frag;
    {
        const int lane = threadIdx.x & 63, l15 = lane & 15, g = lane >> 4;
        const int wm = wave >> 1, wn = wave & 1;
        a_frag = (4 * g) * BM + wm * 16 * TM + l15;
        b_frag = Cfg::A_SZ + (4 * g) * BN + wn * 16 * TN + l15;
    }
    auto read_a = [&](const float* __restrict__ st, int i) -> f32x4 {
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = st[a_frag + e * BM + i * 16];
        return v;
    };
    auto read_b = [&](const float* __restrict__ st, int j) -> f32x4 {
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = st[b_frag + e * BN + j * 16];
        return v;
    };
    auto mfma_at = [&](int m, int P) {
        const int e = m / (TM * TN), r = m % (TM * TN);
        const int i = r / TN, j = r % TN;
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(afr[P][i][e], bfr[P][j][e], acc[i][j], 0, 0, 0);
    };
    // bias gradient: thread t < BM of a unit with n0 == 0 sums row t of every A tile of the unit (the tile of K step
    // g + 1 is summed during step g, like the fragments are fetched)
    bool want_colsum = false;
    float csum = 0.f;
    auto colsum_of = [&](const float* __restrict__ st) {
        if (want_colsum) {
#pragma unroll
            for (int kk = 0; kk < V2_BK; ++kk) csum += st[kk * BM + threadIdx.x];
        }
    };
    int nxt = 1;
    auto step = [&](auto parity, bool has_next) {
        constexpr int P = decltype(parity)::value;
        if (has_next) {
            const float* __restrict__ sn = smem + nxt * Cfg::STAGE;
            constexpr int UNITS = TM + TN;
            constexpr int SPREAD = (4 * TM * TN) / UNITS;
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int m = 0; m < 4 * TM * TN; ++m) {
                mfma_at(m, P);
                const int u = m / SPREAD;
                if (m % SPREAD == 0 && u < UNITS) {
                    if (u < TM) afr[P ^ 1][u] = read_a(sn, u);
                    else bfr[P ^ 1][u - TM] = read_b(sn, u - TM);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            colsum_of(sn);
        } else {
#pragma unroll
            for (int m = 0; m < 4 * TM * TN; ++m) mfma_at(m, P);
        }
        __syncthreads();
        nxt = nxt == S - 1 ? 0 : nxt + 1;
    };

    __builtin_amdgcn_s_barrier();   // P0
    for (int it = 0; it < rounds; ++it) {
        const int u = v4_tile_of(b, it, grid, units);
        const int split = u / tiles, t = u - split * tiles;
        const int m0 = (t / p.tiles_n) * BM, n0 = (t % p.tiles_n) * BN;
        want_colsum = n0 == 0 && p.colsum[0] != nullptr && threadIdx.x < BM;
        csum = 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        {
            // fragments (and column sums) of the unit's first K tile: stage 0 at the start, otherwise the stage before nxt
            const float* __restrict__ s0 = smem + (it == 0 ? 0 : (nxt == 0 ? S - 1 : nxt - 1)) * Cfg::STAGE;
#pragma unroll
            for (int i = 0; i < TM; ++i) afr[0][i] = read_a(s0, i);
#pragma unroll
            for (int j = 0; j < TN; ++j) bfr[0][j] = read_b(s0, j);
            colsum_of(s0);
            __syncthreads();        // P1 / X
        }
        for (int kt = 0; kt + 2 < nk; kt += 2) {
            step(std::integral_constant<int, 0>{}, true);
            step(std::integral_constant<int, 1>{}, true);
        }
        step(std::integral_constant<int, 0>{}, true);
        step(std::integral_constant<int, 1>{}, false);
        // ---- epilogue: the unit's partial product is ADDED to dW (zero-filled or holding an earlier contribution) ----
        int tid2;
        asm volatile("v_mov_b32 %0, %1" : "=v"(tid2) : "v"(threadIdx.x));
        const int lane = tid2 & 63, l15 = lane & 15, g = lane >> 4, w2 = tid2 >> 6;
        const int wm = w2 >> 1, wn = w2 & 1;
        const int cs = m0 / p.cseg;                 // C row segment (stacked weights: one dW tensor per segment)
        const int mloc = m0 - cs * p.cseg;
        const int r0 = m0 + wm * 16 * TM + 4 * g, c0 = n0 + wn * 16 * TN + l15;
        if (p.det_ws != nullptr) {     // deterministic split-K: plain store of the unit's partial into its split's slice
            if (want_colsum) p.det_cs[(long)split * p.M + m0 + tid2] = csum;
            GemmP q = p;
            q.ldc = p.N;
            epilogue_v2_nat<EPI_STORE, TM, TN>(q, p.det_ws + (long)split * p.det_stride, acc, r0, c0, true);
            continue;
        }
        if (want_colsum) unsafeAtomicAdd(p.colsum[cs] + mloc + tid2, csum);
        float* cbase = p.C[cs] - (long)cs * p.cseg * p.ldc;
        if (p.epi == EPI_ATOMIC) epilogue_v2_nat<EPI_ATOMIC, TM, TN>(p, cbase, acc, r0, c0, true);
        else epilogue_v2_nat<EPI_ACCUM, TM, TN>(p, cbase, acc, r0, c0, true);
    }
}

}  // namespace vbgemm
