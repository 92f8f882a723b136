// Fourth-generation fp32 GEMM: ONE persistent, wave-specialised block per compute unit.
//
//   block  = 13 waves: 12 MFMA waves in a 6 x 2 grid (3 per SIMD) + 1 loader wave (LDS-DMA only)
//   tile   = (6 . 16 TM) x (2 . 16 TN) = 288 x 96 (TM = TN = 3) or 288 x 128 (TN = 4): the batch-256 text stream has
//            M = 9216 = 32 x 288 rows, so N = 768 / 1024 give exactly 256 tiles (one per CU) and N = 2304 / 3072 exactly
//            768 (three per CU, run back to back by the same block)
//   ring   = 4 LDS stages of [A tile | B tile] (24.5 / 26.6 KB each), filled by the loader two K steps ahead
//   loop   = the K tiles of ALL output tiles of a block form one stream: the loader keeps filling the ring across
//            output-tile boundaries, so the next tile's first K tiles land while the MFMA waves run the epilogue of the
//            previous one (prologue hidden), one barrier per K step (+ one per output-tile boundary), nothing else.
//
// What this buys over the 4-wave blocks of gemm_v2.h (measured with the per-block timeline of tools/gemm_lab, round 3):
//   * three co-resident independent blocks finish one after the other (the oldest wave wins the issue arbitration): the
//     last one runs alone on its CU for the final quarter of a launch with one wave per SIMD; twelve waves in ONE
//     barrier domain finish together;
//   * the A tile is shared by what used to be three blocks: (288 + 96) / (288 x 96) = 1/72 operand bytes per MFMA
//     flop-pair instead of 1/48 (-33 % vector-memory / L2 traffic), -38 % with 288 x 128;
//   * the MFMA waves issue no vector-memory instruction and no LDS store at all (v3 rationale);
//   * exactly one block per CU by construction (LDS): no dependence on how the dispatcher spreads blocks.
// Layouts: forward (NT) and dgrad (NN); the A operand is always k-contiguous. No split-K (launches that need it stay
// on gemm_v2.h). Requires K % 32 == 0 (an even number of K steps per tile keeps the fragment-set parity static).
#pragma once
#include "gemm_v2.h"

namespace vbgemm {

// Round 4: the kernel is a template over the wave grid and the wave tile - WM x 2 MFMA waves (WM = 6: 12 waves, three per
// SIMD, 128 registers; WM = 4: 8 waves, two per SIMD, 168 registers) of (16 TM) x (16 TN) outputs each, + the loader
// wave. Block tile = (16 TM WM) x (32 TN):
//   WM = 6, TM = 3: 288 x 96 / 288 x 128   the round-3 kernel (M = 9216, 18432: batch-256 / 512 text stream, bert_large)
//   WM = 4, TM = 5 | 4 MIXED in one launch: 20 row tiles of 320 rows + 12 of 256 rows = 9472 rows = the 37-region image
//           stream at batch 256 (M = 2^8 x 37 has no equal tiling into <= 256 tiles of 16-row fragments on a wave grid whose
//           wave count is a multiple of 4; 32 row tiles x 8 column tiles of 128 = exactly 256 tiles, one per CU, 0.925 of
//           the tallest tile's time useful)
//   small-M menu (the per-GPU batch 64 of BASELINE configs[2]: M = 2304 / 2368): WM = 4 / 6 with TM = 1, 2 - see plan_v4
constexpr int V4_STAGES = 4;
constexpr int V4_WN = 2;            // wave columns of every persistent configuration

template <int N>
__device__ __forceinline__ void v4_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int WM, int TM, int TN, bool B_KC>
struct V4Cfg {
    static constexpr int MFMA_WAVES = WM * 2, THREADS = 64 * (MFMA_WAVES + 1);
    static constexpr int BM = 16 * TM * WM, BN = 16 * TN * 2;
    static constexpr int A_SZ = BM * 16;                              // k-contiguous [rows][16], slot-swizzled
    static constexpr int B_SZ = B_KC ? BN * 16 : 16 * (BN + 4);       // or row-contiguous [16 k][cols + 4]
    static constexpr int STAGE = A_SZ + B_SZ;
    static constexpr int RING_BYTES = V4_STAGES * STAGE * 4;
    // exactly one block per CU is part of the design: small tiles ask for more LDS than they use
    static constexpr int LDS_BYTES = RING_BYTES > 82 * 1024 ? RING_BYTES : 82 * 1024;
    static constexpr int A_SLOTS = A_SZ / 4, B_SLOTS = B_SZ / 4;
    static constexpr int NA = (A_SLOTS + 63) / 64, NB = (B_SLOTS + 63) / 64, NI = NA + NB;
    static_assert(2 * NI <= 63, "vmcnt is a 6-bit counter");
    static_assert(LDS_BYTES <= 160 * 1024, "ring does not fit the CU");
};

// tile of block `b` in round `it` of a persistent launch over `tiles` output tiles on `grid` blocks: the blocks of one
// XCD (b % 8) work on a contiguous run of tiles (N fastest) at any time, so they share A / W panels in their L2
__device__ __forceinline__ int v4_tile_of(int b, int it, int grid, int tiles) {
    const int base = it * grid;
    const int n = min(grid, tiles - base);          // tiles of this round
    if ((n & 7) != 0) return b < n ? base + b : -1;
    const int per = n >> 3, x = b & 7, j = b >> 3;
    return j < per ? base + x * per + j : -1;
}

// one LDS-DMA, lean form for the loader's inner loop: LDS[lds + 16 lane] <- *(base + off[lane]). The source address is
// a wave-uniform 64-bit base (SGPR pair, advanced once per K step) plus a per-lane 32-bit byte offset that is constant
// for a whole output tile, so a K step costs the loader 3 instructions per DMA and no vector ALU work at all. (The
// first version bumped a 64-bit pointer per lane per DMA and saved / restored M0 around each one: 3,658 cycles per K step
// to issue 24 DMAs - more than the 3,456 matrix-pipe cycles of the step, measured with the lab's loader counters.)
// M0 is not preserved: nothing else in the loader wave uses it.
__device__ __forceinline__ void v4_glds16(unsigned off, const float* base, unsigned lds) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(off), "s"(base), "s"(lds) : "memory");
}

// (row, column) of output tile `t`. The 32 tiles an XCD works on at any time (v4_tile_of) should share as few A / W
// panels as possible: where the tile grid allows it they form a 4 x 8 patch (4 A panels + 8 W panels per XCD and round
// instead of 1.3 + 24 for a 24-column grid walked row by row: measured 7.0x -> see profiles/r03_gemm_pmc.txt for
// the operand bytes fetched through the fabric per launch); other grids are walked row by row (N fastest).
__device__ __forceinline__ void v4_tile_rc(int t, int tiles, int tiles_n, int& r, int& c) {
    const int tiles_m = tiles / tiles_n;
    if ((tiles_n & 7) == 0 && (tiles_m & 3) == 0) {
        const int patch = t >> 5, w = t & 31, pcols = tiles_n >> 3;
        r = (patch / pcols) * 4 + (w >> 3);
        c = (patch % pcols) * 8 + (w & 7);
    } else {
        r = t / tiles_n;
        c = t % tiles_n;
    }
}

// first row / column of the output tile block `b` works on in round `it`; false = no tile (the block is done).
// MIXED (two tile heights in one launch; p.n_small = number of TALL row tiles, p.m_split = the rows they cover, 32 row tiles
// in all, tiles_n a multiple of 8): block b = XCD x (b & 7), slot j (b >> 3); the XCD owns row tiles 4x .. 4x + 3, the slot
// picks one of them (j >> 3) and a column (j & 7) inside the round's group of 8 columns - so a block keeps its row tile (and
// with it its height class and its A panel) across the rounds, and an XCD works on a 4 x 8 patch as in the uniform map.
// BM is the CALLER's tile height: tall blocks only ever see tall row tiles, short blocks short ones.
template <int BM, int BN, bool MIXED>
__device__ __forceinline__ bool v4_origin(const GemmP& p, int b, int it, int grid, int tiles, int& m0, int& n0) {
    if (MIXED) {
        const int x = b & 7, j = b >> 3;
        const int r = 4 * x + (j >> 3), c = it * 8 + (j & 7);
        if (c >= p.tiles_n) return false;
        m0 = r < p.n_small ? r * BM : p.m_split + (r - p.n_small) * BM;
        n0 = c * BN;
        return true;
    }
    const int t = v4_tile_of(b, it, grid, tiles);
    if (t < 0) return false;
    int tr, tc;
    v4_tile_rc(t, tiles, p.tiles_n, tr, tc);
    m0 = tr * BM;
    n0 = tc * BN;
    return true;
}

template <int WM, int TM, int TN, bool B_KC, bool MIXED>
__device__ __forceinline__ void v4_loader(const GemmP& p, const unsigned lds0, const int lane, const int nk,
                                          const int tiles, const int rounds) {
    using Cfg = V4Cfg<WM, TM, TN, B_KC>;
    constexpr int BM = Cfg::BM, BN = Cfg::BN, NA = Cfg::NA, NB = Cfg::NB, NI = Cfg::NI, S = V4_STAGES;
    unsigned oa[NA], ob[NB];          // per-lane byte offsets from the tile's scalar bases (constant per output tile)
    bool okb[NB];
    const float* abase = nullptr;     // A + m0 * lda + k   (wave-uniform, + 16 floats per K step)
    const float* bbase = nullptr;     // k-contiguous B: W_seg + n_local0 * ldb + k; row-contiguous B: W_seg + k_local * ldb + n0
    int b_seg = 0, b_krem = 0, n0_cur = 0;
    // source addressing of one output tile (A rows clamped to the matrix: rows past M are computed but never stored)
    auto set_tile = [&](int round) {
        int m0 = 0, n0 = 0;
        v4_origin<BM, BN, MIXED>(p, blockIdx.x, round, gridDim.x, tiles, m0, n0);
        abase = p.A + (long)m0 * p.lda;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int s = 64 * i + lane, row = s >> 2, q = s & 3;
            oa[i] = (unsigned)(((long)min(row, p.M - 1 - m0) * p.lda + ((q ^ v2_swz(row)) << 2)) * 4);
        }
        if (B_KC) {
            // a tile never straddles two weight segments (bseg % BN == 0, checked by the planner)
            const int sg = n0 / p.bseg;
            bbase = p.B[sg] + (long)(n0 - sg * p.bseg) * p.ldb;
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                const int s = 64 * i + lane, row = s >> 2, q = s & 3;
                okb[i] = true;
                ob[i] = (unsigned)(((long)min(row, p.N - 1 - n0) * p.ldb + ((q ^ v2_swz(row)) << 2)) * 4);
            }
        } else {
            n0_cur = n0;
            b_seg = 0;
            b_krem = 0;
            bbase = p.B[0] + n0;
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                const int s = 64 * i + lane;
                const int kk = s / (BN / 4 + 1), c = s % (BN / 4 + 1);
                okb[i] = s < Cfg::B_SLOTS && c < BN / 4;
                int col = c * 4;
                if (n0 + col >= p.N || !okb[i]) col = 0;
                ob[i] = (unsigned)(((long)min(kk, V2_BK - 1) * p.ldb + col) * 4);
            }
        }
    };
    int it = 0, kt = 0, stage_w = 0;   // next K tile to issue: (round, kt), into stage stage_w
    auto issue_next = [&]() {
        const unsigned la = lds0 + (unsigned)stage_w * (Cfg::STAGE * 4);
        const unsigned lb = la + Cfg::A_SZ * 4;
#pragma unroll
        for (int i = 0; i < NA; ++i) v4_glds16(oa[i], abase, la + 1024u * i);
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            if (B_KC) v4_glds16(ob[i], bbase, lb + 1024u * i);
            else if (okb[i]) v4_glds16(ob[i], bbase, lb + 1024u * i);
        }
        abase += V2_BK;
        if (B_KC) {
            bbase += V2_BK;
        } else {
            b_krem += V2_BK;                   // weight segments stacked along K
            if (b_krem >= p.bseg) { b_krem = 0; ++b_seg; bbase = p.B[b_seg < VB_MAX_SEGMENTS ? b_seg : 0] + n0_cur; }
            else bbase += (long)V2_BK * p.ldb;
        }
        stage_w = stage_w == S - 1 ? 0 : stage_w + 1;
        if (++kt == nk) {
            kt = 0;
            ++it;
            if (it < rounds) set_tile(it);
        }
    };
    auto wait_pending = [&](int k_tiles) {   // at most k_tiles of the most recently issued K tiles still in flight
        if (k_tiles >= 2) v4_wait_vm<2 * NI>();
        else if (k_tiles == 1) v4_wait_vm<NI>();
        else v4_wait_vm<0>();
    };
    const int total = rounds * nk;           // K tiles of this block's stream
    set_tile(0);
    __builtin_amdgcn_s_setprio(2);
    for (int s = 0; s < S && s < total; ++s) issue_next();
    wait_pending(min(total, S) - 2);         // K tiles 0, 1 have landed
    __builtin_amdgcn_s_barrier();            // P0
    __builtin_amdgcn_s_barrier();            // P1: stage 0 has been read completely
#ifdef VB_GEMM_LAB
    // lab: where the loader's time goes (shader cycles): issuing, waiting for the DMA, waiting at the barrier;
    // VB_GEMM_FLAGS bit 4 (16): the loader issues nothing (garbage results: isolates the MFMA + barrier loop)
    unsigned long long c_issue = 0, c_vm = 0, c_bar = 0;
    const bool no_dma = (p.flags & 16) != 0;
    for (int g = 0, kq = 0; g < total; ++g) {
        if (++kq > nk) { kq = 1; __builtin_amdgcn_s_barrier(); }   // X (output-tile boundary, see the MFMA side)
        const unsigned long long t0 = __builtin_readcyclecounter();
        if (g + S < total && !no_dma) issue_next();
        const unsigned long long t1 = __builtin_readcyclecounter();
        wait_pending(min(total - 1, g + S) - (g + 2));
        const unsigned long long t2 = __builtin_readcyclecounter();
        __builtin_amdgcn_s_barrier();
        const unsigned long long t3 = __builtin_readcyclecounter();
        c_issue += t1 - t0; c_vm += t2 - t1; c_bar += t3 - t2;
    }
    if (p.dbg != nullptr && lane == 0) {
        unsigned long long* st = p.dbg + 8 * 65536 + 16 * (long)blockIdx.x;
        st[0] = c_issue; st[1] = c_vm; st[2] = c_bar; st[3] = (unsigned long long)total;
    }
#else
    for (int g = 0, kq = 0; g < total; ++g) {
        // X: at an output-tile boundary the MFMA waves read the first fragments of the new tile AFTER their epilogue;
        // the stage those live in is the one the next issue overwrites
        if (++kq > nk) { kq = 1; __builtin_amdgcn_s_barrier(); }
        if (g + S < total) issue_next();
        wait_pending(min(total - 1, g + S) - (g + 2));   // K tile g + 2 has landed
        __builtin_amdgcn_s_barrier();
    }
#endif
}

template <int WM, int TM, int TN, bool B_KC, bool MIXED>
__device__ __forceinline__ void gemm_block_v4(const GemmP& p, float* __restrict__ smem) {
    using Cfg = V4Cfg<WM, TM, TN, B_KC>;
    constexpr int BM = Cfg::BM, BN = Cfg::BN, S = V4_STAGES, V4_MFMA_WAVES = Cfg::MFMA_WAVES;
    const int tiles = p.n_big;               // output tiles of the launch
    const int nk = p.K / V2_BK;              // even (K % 32 == 0)
    const int b = blockIdx.x, grid = gridDim.x;
    // rounds this block takes part in (a block whose tile index falls off the end of the last round stops earlier)
    int rounds = 0;
    if (MIXED) rounds = p.tiles_n >> 3;
    else while (rounds * grid < tiles && v4_tile_of(b, rounds, grid, tiles) >= 0) ++rounds;
    if (rounds == 0) return;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
#ifdef VB_GEMM_LAB
    unsigned long long* const tl = p.dbg != nullptr ? p.dbg + 8 * (long)blockIdx.x : nullptr;
    if (tl != nullptr && threadIdx.x == 0) { tl[0] = wall_clock64(); tl[4] = __builtin_readcyclecounter(); }
#endif
    if (wave == V4_MFMA_WAVES) {
        const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) float*)smem;
        v4_loader<WM, TM, TN, B_KC, MIXED>(p, __builtin_amdgcn_readfirstlane(lds0), threadIdx.x & 63, nk, tiles, rounds);
        return;
    }
    f32x4 acc[TM][TN], afr[2][TM], bfr[2][TN];
    int a_frag, b_frag;
    {
        const int lane = threadIdx.x & 63, l15 = lane & 15, g = lane >> 4;
        const int wm = wave >> 1, wn = wave & 1;
        a_frag = (wm * 16 * TM + l15) * 16 + ((g ^ v2_swz(l15)) << 2);
        b_frag = Cfg::A_SZ + (B_KC ? (wn * 16 * TN + l15) * 16 + ((g ^ v2_swz(l15)) << 2) : (4 * g) * (BN + 4) + wn * 16 * TN + l15);
    }
    auto read_a = [&](const float* __restrict__ st, int i) -> f32x4 {
        return *reinterpret_cast<const f32x4*>(st + a_frag + i * 256);
    };
    auto read_b = [&](const float* __restrict__ st, int j) -> f32x4 {
        if (B_KC) return *reinterpret_cast<const f32x4*>(st + b_frag + j * 256);
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = st[b_frag + e * (BN + 4) + j * 16];
        return v;
    };
#ifdef VB_GEMM_LAB
    const bool no_mfma = (p.flags & 32) != 0;   // lab: VB_GEMM_FLAGS bit 5: fragment reads + barriers only
#endif
    auto mfma_at = [&](int m, int P) {
        // contraction index e outermost (consecutive MFMAs never share an accumulator); transposed product: a lane
        // owns 4 consecutive output columns of one row (float4 epilogue)
#ifdef VB_GEMM_LAB
        if (no_mfma) return;
#endif
        const int e = m / (TM * TN), r = m % (TM * TN);
        const int i = r / TN, j = r % TN;
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(bfr[P][j][e], afr[P][i][e], acc[i][j], 0, 0, 0);
    };
#ifdef VB_GEMM_LAB
    unsigned long long* const lab_steps = (p.dbg != nullptr && blockIdx.x == 0 && wave == 0) ? p.dbg + 8 * 65536 + 16 * 256 : nullptr;
    int lab_n = 0;
#endif
    int nxt = 1;   // ring stage of the NEXT K tile of the stream
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
        } else {
#pragma unroll
            for (int m = 0; m < 4 * TM * TN; ++m) mfma_at(m, P);
        }
#ifdef VB_GEMM_LAB
        unsigned long long lab_t0 = 0;
        if (lab_steps != nullptr) lab_t0 = __builtin_readcyclecounter();
#endif
        __syncthreads();   // s_waitcnt lgkmcnt(0) + barrier: the reads of stage nxt are complete
#ifdef VB_GEMM_LAB
        if (lab_steps != nullptr && lab_n < 1000) {   // block 0, wave 0: per step {cycles at barrier entry, cycles after, realtime}
            const unsigned long long t1 = __builtin_readcyclecounter();
            if (threadIdx.x == 0) { lab_steps[3 * lab_n] = lab_t0; lab_steps[3 * lab_n + 1] = t1; lab_steps[3 * lab_n + 2] = wall_clock64(); }
            ++lab_n;
        }
#endif
        nxt = nxt == S - 1 ? 0 : nxt + 1;
    };

    __builtin_amdgcn_s_barrier();   // P0
#ifdef VB_GEMM_LAB
    if (tl != nullptr && threadIdx.x == 0) tl[1] = wall_clock64();
#endif
#pragma unroll
    for (int i = 0; i < TM; ++i) afr[0][i] = read_a(smem, i);
#pragma unroll
    for (int j = 0; j < TN; ++j) bfr[0][j] = read_b(smem, j);
    __syncthreads();                // P1

    for (int it = 0; it < rounds; ++it) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        const bool last_round = it + 1 == rounds;
        (void)last_round;
        if (it > 0) {
            // fragments of this tile's first K tile (its stage landed before the previous barrier). Read here and not
            // under the last MFMAs of the previous tile, so that no fragment register is live across the epilogue.
            const float* __restrict__ s0 = smem + (nxt == 0 ? S - 1 : nxt - 1) * Cfg::STAGE;
#pragma unroll
            for (int i = 0; i < TM; ++i) afr[0][i] = read_a(s0, i);
#pragma unroll
            for (int j = 0; j < TN; ++j) bfr[0][j] = read_b(s0, j);
            __syncthreads();   // X: only now may the loader overwrite that stage (it holds back one K tile for this)
        }
        for (int kt = 0; kt + 2 < nk; kt += 2) {
            step(std::integral_constant<int, 0>{}, true);
            step(std::integral_constant<int, 1>{}, true);
        }
        step(std::integral_constant<int, 0>{}, true);
        step(std::integral_constant<int, 1>{}, false);
#ifdef VB_GEMM_LAB
        if (last_round && tl != nullptr && threadIdx.x == 0) tl[2] = wall_clock64();
        if (tl != nullptr && threadIdx.x == 0 && it < 6) p.dbg[8 * 65536 + 16 * (long)blockIdx.x + 4 + 2 * it] = wall_clock64();
#endif
        // ---- epilogue of this output tile (the loader is already filling the ring with the next tile's K tiles) ---
        // everything the epilogue needs is recomputed here from the thread index (through an opaque copy, so that the
        // compiler cannot keep it alive across the K loop: the loop runs at the 128-register budget of 13 waves per CU)
        int tid2;
        asm volatile("v_mov_b32 %0, %1" : "=v"(tid2) : "v"(threadIdx.x));
        const int lane = tid2 & 63, l15 = lane & 15, g = lane >> 4, w2 = tid2 >> 6;
        const int wm = w2 >> 1, wn = w2 & 1;
        int m0 = 0, n0 = 0;
        v4_origin<BM, BN, MIXED>(p, b, it, grid, tiles, m0, n0);
        const bool full = m0 + BM <= p.M && n0 + BN <= p.N;
        float* cbase = p.C[0];
        const int row0 = m0 + wm * 16 * TM + l15, col0 = n0 + wn * 16 * TN + 4 * g;
        constexpr bool FWD = B_KC, DGRAD = !B_KC;
        // batched loads of an epilogue's second operand (gemm_v2.h): dgrad only - the forward variants (+ bias, + mask hash)
        // spill at this kernel's 128 registers with them
        constexpr int XD = FWD ? 0 : 1;
        if (FWD && p.epi == EPI_GELU) epilogue_v2<EPI_GELU, TM, TN, FWD, XD>(p, cbase, acc, row0, col0, true, full);
        else if (FWD && p.epi == EPI_DGELU) epilogue_v2<EPI_DGELU, TM, TN, FWD, XD>(p, cbase, acc, row0, col0, true, full);
        else if (FWD && p.epi == EPI_RES_DROP) epilogue_v2<EPI_RES_DROP, TM, TN, FWD, XD>(p, cbase, acc, row0, col0, true, full);
        else if (p.epi == EPI_RES) epilogue_v2<EPI_RES, TM, TN, FWD, XD>(p, cbase, acc, row0, col0, true, full);
        else if (DGRAD && p.epi == EPI_MUL) epilogue_v2<EPI_MUL, TM, TN, FWD, XD>(p, cbase, acc, row0, col0, true, full);
        else if (DGRAD && p.epi == EPI_ACCUM) epilogue_v2<EPI_ACCUM, TM, TN, FWD, XD>(p, cbase, acc, row0, col0, true, full);
        else epilogue_v2<EPI_STORE, TM, TN, FWD, XD>(p, cbase, acc, row0, col0, true, full);
#ifdef VB_GEMM_LAB
        if (tl != nullptr && threadIdx.x == 0 && it < 6) p.dbg[8 * 65536 + 16 * (long)blockIdx.x + 5 + 2 * it] = wall_clock64();
#endif
    }
#ifdef VB_GEMM_LAB
    if (tl != nullptr) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (threadIdx.x == 0) {
            tl[3] = wall_clock64();
            tl[5] = __builtin_readcyclecounter();
            unsigned hw, xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            tl[6] = ((unsigned long long)xcc << 32) | hw;
        }
    }
#endif
}

}  // namespace vbgemm
