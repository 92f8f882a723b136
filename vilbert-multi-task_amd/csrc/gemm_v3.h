// Third-generation fp32 GEMM main loop: WAVE-SPECIALISED. Same tiles, LDS layouts, MFMA order and epilogues as
// gemm_v2.h, but a block has FIVE waves: waves 0-3 only read fragments from LDS and issue MFMAs, wave 4 (the loader)
// moves the operand tiles global -> LDS with the LDS-DMA instruction (global_load_lds_dwordx4: no staging registers,
// no ds_write pass) and does nothing else.
//
// Why (round 3, profiles/r03_gemm_lab_timeline*.txt + the round-2 ablations): in the v2 loop every wave issues
// 3-4 global loads + 3-4 LDS stores per K step between its MFMAs. The ablations showed that the loss of the real
// loop against a loop without staging (12-15 %) appears as soon as the loads are ISSUED AND WAITED FOR by the MFMA
// waves (no LDS write needed), and disappears when every lane loads the same address - i.e. the cost sits in the
// vector-memory path the MFMA waves queue behind (address processing / return), not in latency or bandwidth. A wave
// is in-order: while it is stuck at a vector-memory instruction (issue back-pressure) or at the s_waitcnt in front of
// its LDS stores, it cannot issue the MFMAs behind it. Taking the memory instructions out of the MFMA waves'
// instruction streams removes those stalls from the matrix pipe's feeders; the loader wave may stall as long as it
// likes, it only has to stay one K step ahead.
//
// Ring protocol (shown for 3 stages; with 4 the loader runs one more tile ahead, stage = [A tile | B tile] as in gemm_v2.h; NI = DMA instructions per K tile):
//   loader                                            MFMA waves
//   issue tiles 0, 1, 2 -> stages 0, 1, 2
//   wait until tiles 0, 1 have landed
//   barrier P0 ------------------------------------- barrier P0
//                                                     fragments(0) <- stage 0   (ALL of them, both halves)
//   barrier P1 ------------------------------------- barrier P1   (stage 0 is dead from here on)
//   step t = 0 .. nk-1:
//     issue tile t+3 -> stage t % 3                   MFMAs of step t on fragment set t & 1, interleaved with the
//     wait until tile t+2 has landed                  LDS reads of fragments(t+1) <- stage (t+1) % 3 (landed before
//       (s_waitcnt vmcnt(NI): only tile t+3 pending)  barrier t-1) into set (t+1) & 1; s_waitcnt lgkmcnt(0)
//     barrier t ------------------------------------- barrier t
// so a tile gets one full K step of memory latency (as in v2) and the three stages are always in three different
// roles: being overwritten (t), being read (t+1), landing (t+2).
//
// LDS-DMA writes lane-linear: lane l of instruction i fills the 16-byte slot 64 i + l of the operand tile. The layouts
// of gemm_v2.h are kept by choosing the SOURCE address per lane: k-contiguous operand = slot (row, q') holds source
// quad q' ^ swz(row) of that row; row-contiguous operand [16 k][rows + 4] = the pad slot of each k row is skipped by
// masking the lane (exec).
#pragma once
#include "gemm_v2.h"

namespace vbgemm {

// one LDS-DMA of 16 bytes per active lane: LDS[lds + 16 lane] <- *g. M0 holds the LDS base for the instruction and is
// compiler-reserved: written and restored inside the same statement (cdna_hip_programming.md, LDS-DMA recipe).
__device__ __forceinline__ void v3_glds16(const float* g, unsigned lds) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(g), "s"(lds) : "memory");
}

template <int N>
__device__ __forceinline__ void v3_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int TM, int TN, bool A_KC, bool B_KC>
struct V3Cfg {
    using C2 = V2Cfg<TM, TN, A_KC, B_KC>;
    static constexpr int BM = C2::BM, BN = C2::BN;
    static constexpr int A_SLOTS = C2::A_SZ / 4, B_SLOTS = C2::B_SZ / 4;     // 16-byte slots per operand tile
    static constexpr int NA = (A_SLOTS + 63) / 64, NB = (B_SLOTS + 63) / 64;  // DMA instructions per operand tile
    static constexpr int NI = NA + NB;
    // (the last DMA of a row-contiguous tile may run past the tile: masked lanes write nothing)
    // ring depth: 4 stages where three blocks still fit a CU's 160 KiB (a tile then gets two K steps of memory latency)
    static constexpr int STAGES = 3 * 4 * C2::STAGE * 4 <= 160 * 1024 ? 4 : 3;
    static constexpr int LDS_BYTES = STAGES * C2::STAGE * 4;
    // blocks per CU: 5 waves each, <= 16 waves per CU at 128 registers (4 per SIMD); the 16-tile shapes need more
    // registers (full fragment double buffer) and run 2 blocks per CU
    static constexpr int OCC = TM * TN <= 12 ? 3 : 2;
    static constexpr int MIN_WAVES_PER_SIMD = TM * TN <= 12 ? 4 : 3;
};

template <int TM, int TN, bool A_KC, bool B_KC>
__device__ __forceinline__ void v3_loader(const GemmP& p, const unsigned lds0, const int m0, const int n0, const int lane,
                                          const int kt_begin, const int nk) {
    using Cfg = V3Cfg<TM, TN, A_KC, B_KC>;
    using C2 = typename Cfg::C2;
    constexpr int BM = Cfg::BM, BN = Cfg::BN, NA = Cfg::NA, NB = Cfg::NB, NI = Cfg::NI;
    const float* ga[NA];
    bool oka[NA], okb[NB];
    const float* gb[NB];    // k-contiguous B: source pointer; row-contiguous B: unused
    long b_off[NB];         // row-contiguous B: offset inside the (segment) k-row block
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int s = 64 * i + lane;
        if (A_KC) {
            const int row = s >> 2, q = s & 3;
            oka[i] = s < Cfg::A_SLOTS;
            ga[i] = p.A + (long)min(m0 + min(row, BM - 1), p.M - 1) * p.lda + (long)kt_begin * V2_BK + ((q ^ v2_swz(row)) << 2);
        } else {
            const int kk = s / (BM / 4 + 1), c = s % (BM / 4 + 1);
            oka[i] = s < Cfg::A_SLOTS && c < BM / 4;
            int col = m0 + c * 4;
            if (col >= p.M || !oka[i]) col = 0;   // rows past the matrix: any in-bounds address (never stored)
            ga[i] = p.A + ((long)kt_begin * V2_BK + min(kk, V2_BK - 1)) * p.lda + col;
        }
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const int s = 64 * i + lane;
        if (B_KC) {
            const int row = min(s >> 2, BN - 1), q = s & 3;
            okb[i] = s < Cfg::B_SLOTS;
            const int n = min(n0 + row, p.N - 1);
            const int sg = n / p.bseg;
            gb[i] = p.B[sg] + (long)(n - sg * p.bseg) * p.ldb + (long)kt_begin * V2_BK + ((q ^ v2_swz(row)) << 2);
            b_off[i] = 0;
        } else {
            const int kk = s / (BN / 4 + 1), c = s % (BN / 4 + 1);
            okb[i] = s < Cfg::B_SLOTS && c < BN / 4;
            int col = n0 + c * 4;
            if (col >= p.N || !okb[i]) col = 0;
            gb[i] = nullptr;
            b_off[i] = (long)min(kk, V2_BK - 1) * p.ldb + col;
        }
    }
    int b_seg = 0, b_krem = 0;   // row-contiguous B: (segment, k inside segment) of the next tile to load
    if (!B_KC) {
        const int k0 = kt_begin * V2_BK;
        b_seg = k0 / p.bseg;
        b_krem = k0 - b_seg * p.bseg;
    }
    const long a_step = A_KC ? V2_BK : (long)V2_BK * p.lda;
    auto issue = [&](int stage) {
        const unsigned la = lds0 + (unsigned)stage * (C2::STAGE * 4);
        const unsigned lb = la + C2::A_SZ * 4;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            if (oka[i]) v3_glds16(ga[i], la + 1024u * i);
            ga[i] += a_step;
        }
        if (B_KC) {
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                if (okb[i]) v3_glds16(gb[i], lb + 1024u * i);
                gb[i] += V2_BK;
            }
        } else {
            const float* __restrict__ bb = p.B[b_seg] + (long)b_krem * p.ldb;   // segments stacked along K
#pragma unroll
            for (int i = 0; i < NB; ++i)
                if (okb[i]) v3_glds16(bb + b_off[i], lb + 1024u * i);
            b_krem += V2_BK;
            if (b_krem >= p.bseg) { b_krem = 0; ++b_seg; }
        }
    };
    constexpr int S = Cfg::STAGES;
    // wait until at most `tiles` of the most recently issued K tiles are still in flight (vmcnt is an immediate)
    auto wait_pending = [&](int tiles) {
        if (tiles >= 2 && S >= 4) v3_wait_vm<2 * NI>();
        else if (tiles >= 1) v3_wait_vm<NI>();
        else v3_wait_vm<0>();
    };
    __builtin_amdgcn_s_setprio(2);   // the loader's few instructions per step should never wait for issue slots
    issue(0);
    if (nk > 1) issue(1);
    if (nk > 2) issue(2);
    if (S >= 4 && nk > 3) issue(3);
    wait_pending(min(nk, S) - 2);    // tiles 0, 1 have landed
    __builtin_amdgcn_s_barrier();    // P0
    __builtin_amdgcn_s_barrier();    // P1: stage 0 has been read completely
    int stage = 0;
    for (int t = 0; t < nk; ++t) {
        if (t + S < nk) issue(stage);
        wait_pending(min(nk - 1, t + S) - (t + 2));   // tile t + 2 has landed
        __builtin_amdgcn_s_barrier();
        stage = stage == S - 1 ? 0 : stage + 1;
    }
}

template <int TM, int TN, bool A_KC, bool B_KC>
__device__ __forceinline__ void gemm_tile_v3(const GemmP& p, float* __restrict__ smem, const int m0, const int n0) {
    using Cfg = V3Cfg<TM, TN, A_KC, B_KC>;
    using C2 = typename Cfg::C2;
    constexpr int BM = Cfg::BM, BN = Cfg::BN;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    const int kt_total = p.K / V2_BK;
    const int kt_begin = blockIdx.y * p.ktiles_per_split;
    const int nk = min(kt_total, kt_begin + p.ktiles_per_split) - kt_begin;
    if (nk <= 0) return;

#ifdef VB_GEMM_LAB
    unsigned long long* const tl = p.dbg != nullptr ? p.dbg + 8 * ((long)blockIdx.y * gridDim.x + blockIdx.x) : nullptr;
    if (tl != nullptr && tid == 0) { tl[0] = wall_clock64(); tl[4] = __builtin_readcyclecounter(); }
#endif
    if (wave == 4) {
        const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) float*)smem;
        v3_loader<TM, TN, A_KC, B_KC>(p, __builtin_amdgcn_readfirstlane(lds0), m0, n0, lane, kt_begin, nk);
        return;
    }
    const int l15 = lane & 15, g = lane >> 4;
    const int wm = wave >> 1, wn = wave & 1;

    // ---- fragment reads (layouts of gemm_v2.h) ------------------------------------------------------------------
    const int a_frag = A_KC ? (wm * 16 * TM + l15) * 16 + ((g ^ v2_swz(l15)) << 2) : (4 * g) * (BM + 4) + wm * 16 * TM + l15;
    const int b_frag = C2::A_SZ + (B_KC ? (wn * 16 * TN + l15) * 16 + ((g ^ v2_swz(l15)) << 2)
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
    f32x4 afr[2][TM], bfr[2][TN];

    const bool want_colsum = !A_KC && n0 == 0 && tid < BM && p.colsum[0] != nullptr;
    float csum = 0.f;
    auto colsum_of = [&](const float* __restrict__ st) {
        if (want_colsum) {
#pragma unroll
            for (int kk = 0; kk < V2_BK; ++kk) csum += st[kk * (BM + 4) + tid];
        }
    };

    __builtin_amdgcn_s_barrier();   // P0 (nothing of this wave is in flight yet)
#ifdef VB_GEMM_LAB
    if (tl != nullptr && tid == 0) tl[1] = wall_clock64();
#endif
#pragma unroll
    for (int i = 0; i < TM; ++i) afr[0][i] = read_a(smem, i);
#pragma unroll
    for (int j = 0; j < TN; ++j) bfr[0][j] = read_b(smem, j);
    colsum_of(smem);
    __syncthreads();                // P1 (with s_waitcnt lgkmcnt(0): the reads of stage 0 are complete)

    auto mfma_at = [&](int m, int P) {
        // contraction index e outermost: consecutive MFMAs never share an accumulator
        const int e = m / (TM * TN), r = m % (TM * TN);
        const int i = r / TN, j = r % TN;
        // forward / dgrad: transposed product (a lane owns 4 consecutive columns of one row); wgrad: natural
        if (A_KC) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(bfr[P][j][e], afr[P][i][e], acc[i][j], 0, 0, 0);
        else acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(afr[P][i][e], bfr[P][j][e], acc[i][j], 0, 0, 0);
    };
    int nxt = 1;   // stage of K tile t + 1
    // step with a successor: the fragment reads of step t + 1 are spread over the MFMAs of step t
    auto full_step = [&](auto parity) {
        constexpr int P = decltype(parity)::value;
        const float* __restrict__ sn = smem + nxt * C2::STAGE;
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
        __syncthreads();
        nxt = nxt == Cfg::STAGES - 1 ? 0 : nxt + 1;
    };
    auto last_step = [&](auto parity) {
        constexpr int P = decltype(parity)::value;
#pragma unroll
        for (int m = 0; m < 4 * TM * TN; ++m) mfma_at(m, P);
        __syncthreads();
    };
    int t = 0;
    for (; t + 2 < nk; t += 2) {
        full_step(std::integral_constant<int, 0>{});
        full_step(std::integral_constant<int, 1>{});
    }
    if (t + 1 < nk) {
        full_step(std::integral_constant<int, 0>{});
        last_step(std::integral_constant<int, 1>{});
    } else {
        last_step(std::integral_constant<int, 0>{});
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
    // ---- epilogue (gemm_v2.h) -------------------------------------------------------------------------------------
    const int cs = m0 / p.cseg;
    const int mloc = m0 - cs * p.cseg;
    if (want_colsum && mloc + tid < p.cseg && m0 + tid < p.M) unsafeAtomicAdd(p.colsum[cs] + mloc + tid, csum);
    const bool lead = blockIdx.y == 0;
    const bool full = m0 + BM <= p.M && n0 + BN <= p.N;
    float* cbase = p.C[cs] - (long)cs * p.cseg * p.ldc;
    if (!A_KC) {
        const int r0 = m0 + wm * 16 * TM + 4 * g, c0 = n0 + wn * 16 * TN + l15;
        if (p.epi == EPI_ATOMIC) epilogue_v2_nat<EPI_ATOMIC, TM, TN>(p, cbase, acc, r0, c0, full);
        else if (p.epi == EPI_ACCUM) epilogue_v2_nat<EPI_ACCUM, TM, TN>(p, cbase, acc, r0, c0, full);
        else epilogue_v2_nat<EPI_STORE, TM, TN>(p, cbase, acc, r0, c0, full);
        return;
    }
    const int row0 = m0 + wm * 16 * TM + l15, col0 = n0 + wn * 16 * TN + 4 * g;
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
