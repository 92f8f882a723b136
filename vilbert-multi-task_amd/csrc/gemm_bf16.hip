// bf16 TRAINING path, round 5: the reduced-precision mode the reference reaches with `model.half()` + apex FP16_Optimizer
// (/root/reference/train_concap.py:443-461,504-505), re-designed for gfx950 as bf16 tensors in HBM + fp32 master weights:
// activations, saved tensors and activation gradients are bf16 (2 bytes per element through HBM and LDS), every product runs
// on v_mfma_f32_32x32x16_bf16 with fp32 accumulation, weight gradients are accumulated in fp32 straight into the gradient
// arena, AdamW keeps updating fp32 parameters, and a bf16 shadow of every weight (row-major AND transposed) is refreshed once
// per optimizer step.
//
//   gemm_bf16_kernel   C[M, N] = epilogue(A[M, K] W[N, K]^T)      forward (W = the shadow) and dgrad (dX = dY Wt^T with the
//                      TRANSPOSED shadow Wt[K, N] as the "weight": the same kernel, both operands contraction-contiguous)
//   wgrad_bf16_kernel  dW[N, K] += dY[M, N]^T X[M, K]              contraction over the ROWS of both operands: the tiles land
//                      in LDS row-major (LDS-DMA) and the MFMA fragments are fetched with ds_read_b64_tr_b16, gfx950's
//                      transposing LDS read (4 x 16 block per 16 lanes -> each lane 4 consecutive rows of one column)
//   weight_shadow_kernel, cast kernels, colsum (bias gradient)
//
// Both GEMM kernels are the persistent skeleton of mx8.hip (its byte geometry is identical: a K tile of 64 bf16 = 128 bytes
// per row): ONE block per CU = 8 MFMA waves in a 4 x 2 grid (wave tile 64 x 64 = 2 x 2 MFMA tiles of 32 x 32, block tile
// 256 x 128) + 2 loader waves that only issue LDS-DMA (global_load_lds_dwordx4, 1 KiB each: 32 for the A tile, 16 for the W
// tile per K tile); 3-stage ring of 48 KiB; the K tiles of ALL output tiles of a block form one stream (the next tile's
// operands land under the epilogue); ONE barrier per K tile. A K tile = four MFMA steps (K = 16 each) on two static fragment
// register sets: step s multiplies set s & 1 while set (s + 1) & 1 is read from LDS; the barrier sits between steps 2 and 3,
// after the last read of the stage.
#include "gemm_core.h"
#include <type_traits>

namespace {

using namespace vbgemm;

typedef int v4i __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;

constexpr int HB_BM = 256, HB_BN = 128, HB_BK = 64, HB_S = 3;
constexpr int HB_ROWB = 2 * HB_BK;              // bytes of one tile row (NT kernel): 128
constexpr int HB_A = HB_BM * HB_ROWB;           // 32,768 bytes
constexpr int HB_B = HB_BN * HB_ROWB;           // 16,384
constexpr int HB_STAGE = HB_A + HB_B;           // 49,152
constexpr int HB_LDS = HB_S * HB_STAGE;         // 147,456
// loader waves per operand: ONE wave keeps at most 64 KiB of LDS-DMA in flight (vmcnt is a 6-bit counter) - at the ~1 us
// an L2-side request takes under load that is ~30 GB/s per CU, a third of what the matrix pipe consumes at these tile sizes
// (rocprofv3 of the first version: 1.5 us per K tile against 0.43 us of MFMA time, profiles/r05_bf16_lab_first_version.txt)
constexpr int HB_LW = 2;
constexpr int HB_MFMA_WAVES = 8, HB_THREADS = 64 * (HB_MFMA_WAVES + 2 * HB_LW);

// Block shapes of the NT kernel. HbFull: the geometry above, ONE block per CU. HbHalf: 128 x 128 tiles, 4 MFMA waves (2 x 2) + one
// loader wave per operand, a 2-stage ring of 32 KiB stages = 64 KiB, so that TWO independent blocks share a CU. Built to hide
// one block's epilogue under the other's MFMAs; measured, that only pays for SMALL launches (see launch_hb).
template <int BM_, int S_, int LW_>
struct HbCfg {
    static constexpr int BM = BM_, S = S_, LW = LW_;
    static constexpr int A = BM * HB_ROWB, B = HB_BN * HB_ROWB, STAGE = A + B, LDS = S * STAGE;
    static constexpr int MW = BM / 32;                       // MFMA waves: (BM / 64) x 2
    static constexpr int THREADS = 64 * (MW + 2 * LW);
    static constexpr int PER_CU = BM == 256 ? 1 : 2;
};
using HbFull = HbCfg<256, 3, 2>;
using HbHalf = HbCfg<128, 2, 1>;

__device__ __forceinline__ unsigned short bf16_rne(float v) {
    const unsigned u = __float_as_uint(v);
    return (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}
__device__ __forceinline__ unsigned pack_bf16(float lo, float hi) { return (unsigned)bf16_rne(lo) | ((unsigned)bf16_rne(hi) << 16); }
__device__ __forceinline__ float bf16_lo(unsigned w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf16_hi(unsigned w) { return __uint_as_float(w & 0xffff0000u); }

// one LDS-DMA: LDS[lds + 16 lane] <- *(base + off[lane]), wave-uniform 64-bit base + per-lane 32-bit byte offset
__device__ __forceinline__ void hb_glds16(unsigned off, const void* base, unsigned lds) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(off), "s"(base), "s"(lds) : "memory");
}
template <int N>
__device__ __forceinline__ void hb_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// tile of block `b` in round `it`: the 32 blocks of an XCD (b % 8) work on a 4 x 8 patch of tiles where the tile grid allows
// it, so that they share A / W panels in their L2 (mx8.hip: mx_tile_of / mx_origin)
__device__ __forceinline__ int hb_tile_of(int b, int it, int grid, int tiles) {
    const int base = it * grid;
    const int n = min(grid, tiles - base);
    if (n <= 0) return -1;
    if ((n & 7) != 0) return b < n ? base + b : -1;
    const int per = n >> 3, x = b & 7, j = b >> 3;
    return j < per ? base + x * per + j : -1;
}

// ---------------------------------------------------------------------------------------------------------------
// NT kernel
// ---------------------------------------------------------------------------------------------------------------
// EPI (compile time): what happens to v = acc + bias before it is stored
enum { HB_PLAIN = 0,   // v
       HB_GELU = 1,    // gelu(v), and gelu'(v) to D (the derivative the dgrad epilogue multiplies by - no erf in backward)
       HB_RES = 2,     // v + R
       HB_DROPRES = 3, // dropout(v) + R
       HB_MUL = 4,     // v * MUL            (dgrad through an activation: MUL = the saved derivative)
       HB_RELU = 5 };  // max(v, 0)
enum { HB_OUT_BF16 = 0, HB_OUT_F32 = 1 };

struct HbP {
    int M, N, K;
    const unsigned short* A; long lda;     // elements
    const unsigned short* B; long ldb;
    const float* bias[VB_MAX_SEGMENTS]; int bseg;   // bias of output columns [s bseg, (s + 1) bseg); null = none
    const unsigned short* R; long ldr;     // residual or multiplier, bf16 [M, N]
    unsigned short* C; long ldc;
    float* C32; long ldc32;
    unsigned short* D; long ldd;           // gelu'(pre) out (HB_GELU), may be null
    int tiles_n, tiles;
    int bm;      // rows of an output tile (256 or 128)
    float drop_p, drop_scale;
    uint64_t seed;
    const uint64_t* epoch;
    int flags;   // laboratory (VB_BF16_FLAGS, timing only - results are garbage): 1 = no DMA, 2 = no MFMA, 4 = no fragment reads,
                 // 8 = no epilogue
};

__device__ __forceinline__ bool hb_origin(const HbP& p, int b, int it, int grid, int& m0, int& n0) {
    const int t = hb_tile_of(b, it, grid, p.tiles);
    if (t < 0) return false;
    const int tiles_m = p.tiles / p.tiles_n;
    int r, c;
    if ((p.tiles_n & 7) == 0 && (tiles_m & 3) == 0) {
        const int patch = t >> 5, w = t & 31, pcols = p.tiles_n >> 3;
        r = (patch / pcols) * 4 + (w >> 3);
        c = (patch % pcols) * 8 + (w & 7);
    } else {
        r = t / p.tiles_n;
        c = t % p.tiles_n;
    }
    m0 = r * p.bm;
    n0 = c * HB_BN;
    return true;
}

// Loader wave of one operand of the NT kernel: ND DMAs (8 tile rows of 128 bytes each) per K tile. LDS rows of 128 bytes,
// the eight 16-byte chunks of a row XOR-swizzled with (row >> 1) & 7 (fragment ds_read_b128 conflict-free); the DMA writes
// lane-linear, so the swizzle is applied on the SOURCE address.
template <class Cfg, bool IS_A>
__device__ __forceinline__ void hb_loader(const HbP& p, const unsigned lds0, const int lane, const int nk, const int rounds,
                                          const int widx) {
    constexpr int HB_LW = Cfg::LW, HB_S = Cfg::S, HB_STAGE = Cfg::STAGE, HB_A = Cfg::A;      // (shadow the full-size constants)
    constexpr int ND = (IS_A ? Cfg::A : Cfg::B) / 1024 / HB_LW;             // this wave's share: DMAs widx, widx + HB_LW, ...
    const unsigned short* const mat = IS_A ? p.A : p.B;
    const long ld = IS_A ? p.lda : p.ldb;
    const int nrows = IS_A ? p.M : p.N;
    constexpr unsigned REG = IS_A ? 0u : (unsigned)HB_A;
    static_assert(ND <= 63, "vmcnt is a 6-bit counter");
    unsigned off[ND];
    const unsigned short* base = nullptr;
    auto set_tile = [&](int round) {
        int m0 = 0, n0 = 0;
        hb_origin(p, blockIdx.x, round, gridDim.x, m0, n0);
        const int r0 = IS_A ? m0 : n0;
        base = mat + (long)r0 * ld;
#pragma unroll
        for (int i = 0; i < ND; ++i) {
            const int row = 8 * (HB_LW * i + widx) + (lane >> 3), slot = lane & 7;     // LDS row of this lane's 16 bytes
            const int chunk = slot ^ ((row >> 1) & 7);
            off[i] = (unsigned)((long)min(row, nrows - 1 - r0) * ld * 2 + 16 * chunk);    // rows past the matrix: clamped, never stored
        }
    };
    int it = 0, kt = 0, stage_w = 0;
    auto issue_next = [&]() {
        const unsigned l = lds0 + (unsigned)stage_w * HB_STAGE;
        if (!(p.flags & 1)) {
#pragma unroll
            for (int i = 0; i < ND; ++i) hb_glds16(off[i], base, l + REG + 1024u * (HB_LW * i + widx));
        }
        base += HB_BK;
        stage_w = stage_w == HB_S - 1 ? 0 : stage_w + 1;
        if (++kt == nk) {
            kt = 0;
            ++it;
            if (it < rounds) set_tile(it);
        }
    };
    const int total = rounds * nk;
    set_tile(0);
    __builtin_amdgcn_s_setprio(2);
    for (int s = 0; s < HB_S && s < total; ++s) issue_next();
    // K tile 0 has landed (S - 2 of the newer tiles may still be pending)
    if (total >= 2) hb_wait_vm<(HB_S - 2) * ND>(); else hb_wait_vm<0>();
    __builtin_amdgcn_s_barrier();                                 // P0
    for (int g = 0; g < total; ++g) {
        // K tile g + 1 has landed: issued so far = min(total, g + S) tiles, so at most S - 2 newer tiles may be pending
        if (g + HB_S <= total) hb_wait_vm<(HB_S - 2) * ND>(); else hb_wait_vm<0>();
        __builtin_amdgcn_s_barrier();                             // B_g: stage g has been read completely
        if (g + HB_S < total) issue_next();
    }
}

template <int OUT, int EPI, class Cfg>
__global__ __launch_bounds__(Cfg::THREADS, Cfg::PER_CU == 1 ? 1 : 3) void gemm_bf16_kernel(const HbP p) {
    constexpr int HB_MFMA_WAVES = Cfg::MW, HB_LW = Cfg::LW, HB_S = Cfg::S, HB_STAGE = Cfg::STAGE, HB_A = Cfg::A;   // (shadow)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int nk = p.K / HB_BK;
    const int b = blockIdx.x, grid = gridDim.x;
    int rounds = 0;
    while (rounds * grid < p.tiles && hb_tile_of(b, rounds, grid, p.tiles) >= 0) ++rounds;
    if (rounds == 0) return;
    if (Cfg::PER_CU == 2 && (p.flags & 16) && b >= grid / 2) {
        // laboratory: the second block of a CU starts half an output tile late, so that the two are never in their epilogues
        // at the same time (timing experiments only; they de-phase on their own after the first tile)
        for (int i = 0; i < nk * 8; ++i) __builtin_amdgcn_s_sleep(1);
    }
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (wave >= HB_MFMA_WAVES) {
        const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) char*)smem);
        const int lw = wave - HB_MFMA_WAVES;
        if (lw < HB_LW) hb_loader<Cfg, true>(p, lds0, threadIdx.x & 63, nk, rounds, lw);
        else hb_loader<Cfg, false>(p, lds0, threadIdx.x & 63, nk, rounds, lw - HB_LW);
        return;
    }
    const int lane = threadIdx.x & 63, l31 = lane & 31, hi = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    const int key = (l31 >> 1) & 7;
    const int fa_off = (wm * 64 + l31) * HB_ROWB, fb_off = HB_A + (wn * 64 + l31) * HB_ROWB;

    f32x16 acc[2][2];
    bf16x8 fa[2][2], fb[2][2];          // [register set][tile]

    // v_mfma_f32_32x32x16_bf16: lane (l31, hi) supplies 8 contraction elements of row / column l31 - the 16 bytes of chunk
    // 2 s + hi of the tile row for step s. Both operands are fetched with the same (lane, element) -> k map, so the order in
    // which the instruction consumes its 16 k values cannot matter.
    auto read_set = [&](auto S_, const char* stage, int s) {
        constexpr int S = decltype(S_)::value;
        if (p.flags & 4) return;
        const int co = ((2 * s + hi) ^ key) << 4;
#pragma unroll
        for (int i = 0; i < 2; ++i) fa[S][i] = *reinterpret_cast<const bf16x8*>(stage + fa_off + i * 32 * HB_ROWB + co);
#pragma unroll
        for (int j = 0; j < 2; ++j) fb[S][j] = *reinterpret_cast<const bf16x8*>(stage + fb_off + j * 32 * HB_ROWB + co);
    };
    auto mfmas = [&](auto S_) {
        constexpr int S = decltype(S_)::value;
        if (p.flags & 2) return;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                // TRANSPOSED product: rows of the W fragment -> accumulator registers (output columns), rows of the A fragment
                // -> lane & 31 (output row): a lane owns ONE output row and 4-column groups of it
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[S][j], fa[S][i], acc[i][j], 0, 0, 0);
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;

    __builtin_amdgcn_s_barrier();   // P0: K tile 0 has landed
    int st = 0;                     // ring stage of the current K tile
    for (int it = 0; it < rounds; ++it) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        read_set(I0{}, smem + st * HB_STAGE, 0);   // first step of this output tile (landed before the previous barrier)
        for (int kt = 0; kt + 1 < nk; ++kt) {
            const char* cur = smem + st * HB_STAGE;
            st = st == HB_S - 1 ? 0 : st + 1;
            const char* nxt = smem + st * HB_STAGE;
            read_set(I1{}, cur, 1);
            __builtin_amdgcn_sched_barrier(0);   // the LDS reads go out first, the MFMAs cover their latency
            mfmas(I0{});
            __builtin_amdgcn_sched_barrier(0);
            read_set(I0{}, cur, 2);
            __builtin_amdgcn_sched_barrier(0);
            mfmas(I1{});
            __builtin_amdgcn_sched_barrier(0);
            read_set(I1{}, cur, 3);
            __builtin_amdgcn_sched_barrier(0);
            mfmas(I0{});
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // B_g: every read of stage g is complete
            read_set(I0{}, nxt, 0);
            __builtin_amdgcn_sched_barrier(0);
            mfmas(I1{});
            __builtin_amdgcn_sched_barrier(0);
        }
        {   // last K tile of this output tile: no successor to prefetch (the next output tile reads its first step above, so
            // that no fragment register is live across the epilogue)
            const char* cur = smem + st * HB_STAGE;
            st = st == HB_S - 1 ? 0 : st + 1;
            read_set(I1{}, cur, 1);
            __builtin_amdgcn_sched_barrier(0);
            mfmas(I0{});
            __builtin_amdgcn_sched_barrier(0);
            read_set(I0{}, cur, 2);
            __builtin_amdgcn_sched_barrier(0);
            mfmas(I1{});
            __builtin_amdgcn_sched_barrier(0);
            read_set(I1{}, cur, 3);
            __builtin_amdgcn_sched_barrier(0);
            mfmas(I0{});
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            mfmas(I1{});
        }
        int m0 = 0, n0 = 0;
        hb_origin(p, b, it, grid, m0, n0);
        if (p.flags & 8) continue;
        // ---- Epilogue on the transposed map: lane (l31, hi) holds output row m0 + wm 64 + 32 i + l31 of tile (i, j) and its
        // columns nw + 32 j + 8 q + 4 hi + e (e = 0..3) in acc[i][j][4 q + e]. The epilogue of a persistent block is bound by
        // store ISSUE (~7 B/clk/CU with 4-byte stores: cdna guide T21), not by bandwidth - so the bf16 results leave as 16-byte
        // stores: the two half-waves of a row trade their 4-column pieces of two adjacent 8-column groups
        // (v_permlane32_swap), after which every lane owns 8 consecutive columns. The second [M, N] operand (residual /
        // multiplier) arrives the same way in reverse: one 16-byte load per two groups, then the same swap.
        const int nw = n0 + wn * 64;
        const uint64_t seed = EPI == HB_DROPRES ? vb_seed_with_epoch(p.seed, p.epoch) : 0;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int m = m0 + wm * 64 + 32 * i + l31;
            const bool live = m < p.M;
            const long mr = live ? m : p.M - 1;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int nb = nw + 32 * j;
                const int bs = nb / p.bseg;                                   // (a 32-column block never straddles segments)
                const float* __restrict__ bp = p.bias[bs] != nullptr ? p.bias[bs] + (nb - bs * p.bseg) : nullptr;
                uint2 rq[4];
                if (EPI == HB_RES || EPI == HB_DROPRES || EPI == HB_MUL) {
#pragma unroll
                    for (int kq = 0; kq < 4; kq += 2) {
                        // lanes 0-31: columns 8 kq .. 8 kq + 7, lanes 32-63: the next eight
                        uint4 t = *reinterpret_cast<const uint4*>(p.R + mr * p.ldr + nb + 8 * (kq + hi));
                        const auto sx = __builtin_amdgcn_permlane32_swap(t.x, t.z, false, false);
                        const auto sy = __builtin_amdgcn_permlane32_swap(t.y, t.w, false, false);
                        rq[kq] = uint2{sx[0], sy[0]};
                        rq[kq + 1] = uint2{sx[1], sy[1]};
                    }
                }
                uint2 o2[4], d2[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    f32x4 bv = {0.f, 0.f, 0.f, 0.f};
                    if (bp != nullptr) bv = *reinterpret_cast<const f32x4*>(bp + 8 * q + 4 * hi);
                    f32x4 v, d;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float x = acc[i][j][4 * q + e] + bv[e];
                        if (EPI == HB_GELU) {
                            float gy, gd;
                            gelu_and_grad(x, gy, gd);
                            x = gy;
                            d[e] = gd;
                        }
                        if (EPI == HB_RELU) x = fmaxf(x, 0.f);
                        if (EPI == HB_DROPRES) {
                            const uint64_t idx = (uint64_t)((long)m * p.N + nb + 8 * q + 4 * hi + e);
                            x = vb_keep(seed, idx, p.drop_p) ? x * p.drop_scale : 0.f;
                        }
                        v[e] = x;
                    }
                    if (EPI == HB_RES || EPI == HB_DROPRES)
                        v += f32x4{bf16_lo(rq[q].x), bf16_hi(rq[q].x), bf16_lo(rq[q].y), bf16_hi(rq[q].y)};
                    if (EPI == HB_MUL) v *= f32x4{bf16_lo(rq[q].x), bf16_hi(rq[q].x), bf16_lo(rq[q].y), bf16_hi(rq[q].y)};
                    if (OUT == HB_OUT_F32) {
                        if (live) *reinterpret_cast<f32x4*>(p.C32 + (long)m * p.ldc32 + nb + 8 * q + 4 * hi) = v;
                    } else {
                        o2[q] = uint2{pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3])};
                        if (EPI == HB_GELU) d2[q] = uint2{pack_bf16(d[0], d[1]), pack_bf16(d[2], d[3])};
                    }
                }
                if (OUT == HB_OUT_BF16) {
#pragma unroll
                    for (int kq = 0; kq < 4; kq += 2) {
                        const auto sx = __builtin_amdgcn_permlane32_swap(o2[kq].x, o2[kq + 1].x, false, false);
                        const auto sy = __builtin_amdgcn_permlane32_swap(o2[kq].y, o2[kq + 1].y, false, false);
                        if (live) *reinterpret_cast<uint4*>(p.C + (long)m * p.ldc + nb + 8 * (kq + hi)) = uint4{sx[0], sy[0], sx[1], sy[1]};
                        if (EPI == HB_GELU) {
                            if (p.D != nullptr) {
                                const auto dx = __builtin_amdgcn_permlane32_swap(d2[kq].x, d2[kq + 1].x, false, false);
                                const auto dy = __builtin_amdgcn_permlane32_swap(d2[kq].y, d2[kq + 1].y, false, false);
                                if (live) *reinterpret_cast<uint4*>(p.D + (long)m * p.ldd + nb + 8 * (kq + hi)) = uint4{dx[0], dy[0], dx[1], dy[1]};
                            }
                        }
                    }
                }
            }
        }
    }
}

// persistent blocks per launch (VB_BF16_GRID, a multiple of 8, default = the 256 CUs): with fewer, two launches of
// different streams share the chip side by side instead of one after the other
inline int hb_grid_limit() {
    static const int g = [] { const char* e = getenv("VB_BF16_GRID"); const int v = e ? atoi(e) : 256; return v >= 8 && v <= 256 ? v / 8 * 8 : 256; }();
    return g;
}

template <int OUT, int EPI, class Cfg>
int launch_hb_cfg(hipStream_t st, HbP p) {
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_bf16_kernel<OUT, EPI, Cfg>),
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS);
    if (attr != hipSuccess) return (int)attr;
    p.bm = Cfg::BM;
    p.tiles = ((p.M + Cfg::BM - 1) / Cfg::BM) * p.tiles_n;
    const int slots = hb_grid_limit() * Cfg::PER_CU;
    const int grid = p.tiles < slots ? p.tiles : slots;
    hipLaunchKernelGGL((gemm_bf16_kernel<OUT, EPI, Cfg>), dim3(grid), dim3(Cfg::THREADS), Cfg::LDS, st, p);
    VB_LAUNCH_CHECK();
    return 0;
}

// VB_BF16_HALF: 0 = always the full-size block, 2 = always two half-size blocks per CU, 1 (default) = the half-size blocks for
// launches of at most 128 full-size tiles (the per-GPU batch 64 shapes: 80 tiles on 256 CUs become 160 blocks - 23.5 -> 16.6 us
// at 2368 x 1024 x 1024). Everywhere else the half-size block LOSES (profiles/r05_bf16_half_blocks.txt): a 128 x 128 tile
// reads a third more operand bytes per FLOP through L2 / LDS-DMA, and the main loop alone falls from ~1.0 PF to 0.62 - 0.74 PF -
// more than the overlapped epilogues win back (q|k|v forward 44.8 -> 67.7 us).
template <int OUT, int EPI>
int launch_hb(hipStream_t st, const HbP& p) {
    static const int half = [] { const char* e = getenv("VB_BF16_HALF"); return e ? atoi(e) : 1; }();
    const int tiles_full = ((p.M + 255) / 256) * p.tiles_n;
    const bool use_half = half == 2 || (half == 1 && tiles_full <= 128);
    return use_half ? launch_hb_cfg<OUT, EPI, HbHalf>(st, p) : launch_hb_cfg<OUT, EPI, HbFull>(st, p);
}

// ---------------------------------------------------------------------------------------------------------------
// TN kernel (weight gradient): dW[n, k] += sum_m dY[m, n] X[m, k]
// ---------------------------------------------------------------------------------------------------------------
// Output tile 256 (n) x 128 (k), contraction tile 64 rows of m. Both operand tiles land in LDS ROW-major as the DMA
// delivers them - [64 m][256 n] (512-byte rows) and [64 m][128 k] (256-byte rows) - and the MFMA fragments (8 consecutive
// m for one n / k per lane) are fetched with ds_read_b64_tr_b16: the 16 lanes of a group read a 4 (m) x 16 (n) block,
// lane 4 r + q supplying the address of row r, columns 4 q .. 4 q + 3, and lane i receives column i of the block = 4
// consecutive m. Two such reads (m = 8 hi + 4 t + 0..3, t = 0, 1) make one operand register set; A and B use the same
// (lane, element) -> m map, so the instruction's internal order of its 16 contraction values cannot matter.
// Bank conflicts: the 32 lanes of a half-wave read 4 rows x 64 bytes; rows are 2 (A) / 1 (B) whole 256-byte bank rows
// apart, so the 16-byte chunks of tile row m are stored XORed with (m & 3) << 2 (again on the DMA's SOURCE side): the four
// rows of a read fall into the four different 64-byte quarters of the bank row.
// Work unit = (output tile, contraction split); the units are dealt to the persistent blocks round-robin, all tiles of
// one split first (the 256 concurrent units then share the same m range of dY and X in L2). Partial sums are ADDED to dW
// with fp32 atomics (the gradient arena is zero-filled once per backward pass).
constexpr int HW_ROWA = 2 * HB_BM;   // 512 bytes: one m row of the dY tile (256 n)
constexpr int HW_ROWB = 2 * HB_BN;   // 256 bytes: one m row of the X tile (128 k)

struct HwP {
    int M, N, K;
    const unsigned short* Y; long ldy;
    const unsigned short* X; long ldx;
    float* C[VB_MAX_SEGMENTS]; long ldc; int cseg;
    float* bias[VB_MAX_SEGMENTS];   // bias gradient of the segment (column sums of dY), ADDED into; may be null
    int tiles_k, tiles;             // tiles = (N / 256) (K / 128)
    int nkt, kt_per_split, splits;  // contraction tiles in total / per unit, units per tile
    int units;
    int flags;                      // laboratory (VB_BF16_FLAGS), as HbP
    // deterministic form (vb_set_deterministic, the default): a unit STORES its partial tile to ws + (split tiles + tile) 32768
    // floats in accumulator order - [wave][MFMA tile][register quad][lane] float4: every store instruction writes 1 KiB of
    // consecutive addresses - and its bias-gradient partial to ws_b + split N; wgrad_bf16_reduce_kernel adds the splits in
    // order into dW / db. nullptr = fp32 atomics straight into dW / db (run-to-run differences in the last bits).
    float* ws; float* ws_b;
    int n_valid;                    // rows of a segment's dW (and entries of its db) that exist: the tiles cover cseg >= n_valid rows
};

// unit of block `b` in its i-th round: the 32 blocks of an XCD (b % 8) take CONSECUTIVE units - the same contraction split and
// neighbouring output tiles - so that the row range of dY / X they stream is shared in their L2 (hb_tile_of's map)
// (grid = a multiple of 8; any unit count: XCD x takes the units [x per, (x + 1) per) of the round, per = ceil(n / 8))
__device__ __forceinline__ int hw_unit_of(int b, int i, int grid, int units) {
    const int base = i * grid;
    const int n = min(grid, units - base);
    if (n <= 0) return -1;
    const int per = (n + 7) >> 3, x = b & 7, j = b >> 3;
    const int idx = x * per + j;
    return (j < per && idx < n) ? base + idx : -1;
}
__device__ __forceinline__ void hw_unit(const HwP& p, int b, int i, int grid, int& n0, int& k0, int& kt0, int& nk) {
    const int u = hw_unit_of(b, i, grid, p.units);
    const int split = u / p.tiles, t = u - split * p.tiles;
    n0 = (t / p.tiles_k) * HB_BM;
    k0 = (t % p.tiles_k) * HB_BN;
    kt0 = split * p.kt_per_split;
    nk = min(p.kt_per_split, p.nkt - kt0);
}

template <int ND_ALL, bool IS_A>
__device__ __forceinline__ void hw_loader(const HwP& p, const unsigned lds0, const int lane, const int n_units, const int widx) {
    constexpr int ND = ND_ALL / HB_LW;
    const unsigned short* const mat = IS_A ? p.Y : p.X;
    const long ld = IS_A ? p.ldy : p.ldx;
    constexpr unsigned REG = IS_A ? 0u : (unsigned)HB_A;
    constexpr int RPD = IS_A ? 2 : 4;              // tile rows per DMA
    constexpr int CPR = IS_A ? 32 : 16;            // 16-byte chunks per tile row
    unsigned off[ND];
    const int sub = lane / CPR, cp = lane % CPR;   // row inside the DMA's rows, physical chunk
#pragma unroll
    for (int i = 0; i < ND; ++i) {
        const int m = RPD * (HB_LW * i + widx) + sub;
        off[i] = (unsigned)((long)m * ld * 2 + 16 * (cp ^ ((m & 3) << 2)));
    }
    const unsigned short* base = nullptr;
    int u_idx = 0, kt = 0, nk = 0, kt0 = 0, stage_w = 0;
    long total = 0;
    for (int i = 0; i < n_units; ++i) {
        int n0, k0, a, c;
        hw_unit(p, blockIdx.x, i, gridDim.x, n0, k0, a, c);
        total += c;
    }
    auto set_unit = [&](int i) {
        int n0, k0;
        hw_unit(p, blockIdx.x, i, gridDim.x, n0, k0, kt0, nk);
        base = mat + (long)kt0 * HB_BK * ld + (IS_A ? n0 : k0);
        kt = 0;
    };
    auto issue_next = [&]() {
        const unsigned l = lds0 + (unsigned)stage_w * HB_STAGE;
        const int rows_left = p.M - (kt0 + kt) * HB_BK;
        if (p.flags & 1) {
        } else if (rows_left >= HB_BK) {
#pragma unroll
            for (int i = 0; i < ND; ++i) hb_glds16(off[i], base, l + REG + 1024u * (HB_LW * i + widx));
        } else {
            // the last contraction tile of a ragged M: rows past the end are fetched from the last real row (finite
            // data; the MFMA waves zero the dY fragment elements of those rows)
#pragma unroll
            for (int i = 0; i < ND; ++i) {
                const int m = RPD * (HB_LW * i + widx) + sub;
                const unsigned o = (unsigned)((long)min(m, rows_left - 1) * ld * 2 + 16 * (cp ^ ((m & 3) << 2)));
                hb_glds16(o, base, l + REG + 1024u * (HB_LW * i + widx));
            }
        }
        base += (long)HB_BK * ld;
        stage_w = stage_w == HB_S - 1 ? 0 : stage_w + 1;
        if (++kt == nk) {
            ++u_idx;
            if (u_idx < n_units) set_unit(u_idx);
        }
    };
    set_unit(0);
    __builtin_amdgcn_s_setprio(2);
    for (int s = 0; s < HB_S && s < total; ++s) issue_next();
    if (total >= 2) hb_wait_vm<ND>(); else hb_wait_vm<0>();
    __builtin_amdgcn_s_barrier();                                 // P0
    for (long g = 0; g < total; ++g) {
        if (g + 3 <= total) hb_wait_vm<ND>(); else hb_wait_vm<0>();
        __builtin_amdgcn_s_barrier();                             // B_g
        if (g + 3 < total) issue_next();
    }
}

__global__ __launch_bounds__(HB_THREADS) void wgrad_bf16_kernel(const HwP p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int b = blockIdx.x, grid = gridDim.x;
    int n_units = 0;
    while (n_units * grid < p.units && hw_unit_of(b, n_units, grid, p.units) >= 0) ++n_units;
    if (n_units == 0) return;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (wave >= HB_MFMA_WAVES) {
        const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) char*)smem);
        const int lw = wave - HB_MFMA_WAVES;
        if (lw < HB_LW) hw_loader<HB_A / 1024, true>(p, lds0, threadIdx.x & 63, n_units, lw);
        else hw_loader<HB_B / 1024, false>(p, lds0, threadIdx.x & 63, n_units, lw - HB_LW);
        return;
    }
    const int lane = threadIdx.x & 63, l31 = lane & 31, hi = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;          // wave tile: n rows wm 64 .., k columns wn 64 ..
    // transposing read: lane = 16 g4 + 4 r + q reads row (8 hi + 4 t + r) of the step, columns 16 nb + 4 q .. + 3
    const int r = (lane >> 2) & 3, q = lane & 3, nb = (lane >> 4) & 1;
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    unsigned a_addr[2], b_addr[2];                    // byte address inside a stage of (step 0, t = 0) for MFMA tile i / j
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int ca = wm * 8 + 4 * i + 2 * nb + (q >> 1);
        a_addr[i] = (unsigned)((8 * hi + r) * HW_ROWA + ((ca ^ (r << 2)) << 4) + (q & 1) * 8);
        const int cb = wn * 8 + 4 * i + 2 * nb + (q >> 1);
        b_addr[i] = (unsigned)(HB_A + (8 * hi + r) * HW_ROWB + ((cb ^ (r << 2)) << 4) + (q & 1) * 8);
    }

    f32x16 acc[2][2];
    bf16x8 fa[2][2], fb[2][2];
    auto tr = [&](unsigned addr) -> bf16x4 {
        return __builtin_amdgcn_ds_read_tr16_b64_v4bf16(reinterpret_cast<lds_bf16x4*>(addr));
    };
    auto read_set = [&](auto S_, unsigned stage, int s) {
        constexpr int S = decltype(S_)::value;
        if (p.flags & 4) return;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const bf16x4 lo = tr(stage + a_addr[i] + (unsigned)(16 * s) * HW_ROWA);
            const bf16x4 up = tr(stage + a_addr[i] + (unsigned)(16 * s + 4) * HW_ROWA);
            fa[S][i] = __builtin_shufflevector(lo, up, 0, 1, 2, 3, 4, 5, 6, 7);
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const bf16x4 lo = tr(stage + b_addr[j] + (unsigned)(16 * s) * HW_ROWB);
            const bf16x4 up = tr(stage + b_addr[j] + (unsigned)(16 * s + 4) * HW_ROWB);
            fb[S][j] = __builtin_shufflevector(lo, up, 0, 1, 2, 3, 4, 5, 6, 7);
        }
    };
    // ragged M: zero the dY fragment elements whose row is past the end (element e of the set = row 16 s + 8 hi + 4 (e >> 2)
    // + (e & 3) of the contraction tile); the X side then multiplies finite values by exact zeros
    auto mask_set = [&](auto S_, int s, int rows_left) {
        constexpr int S = decltype(S_)::value;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            v4i w = __builtin_bit_cast(v4i, fa[S][i]);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int m = 16 * s + 8 * hi + 4 * (e >> 2) + (e & 3);
                if (m >= rows_left) w[e >> 1] &= (e & 1) ? 0x0000ffff : 0xffff0000;
            }
            fa[S][i] = __builtin_bit_cast(bf16x8, w);
        }
    };
    auto mfmas = [&](auto S_) {
        constexpr int S = decltype(S_)::value;
        if (p.flags & 2) return;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[S][i], fb[S][j], acc[i][j], 0, 0, 0);
    };
    // Bias gradient = column sums of dY over this unit's rows, for free: the waves of k column 0 (wn == 0) of the units of
    // k tile 0 already hold every dY element of their 64 columns in their A fragments - lane (l31, hi) the 8 rows of column
    // l31 of a step - and add them up with VALU instructions that issue beside the MFMAs.
    float bsum[2] = {0.f, 0.f};
    auto bias_acc = [&](auto S_) {
        constexpr int S = decltype(S_)::value;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const v4i w = __builtin_bit_cast(v4i, fa[S][i]);
            float t = 0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) t += bf16_lo((unsigned)w[e]) + bf16_hi((unsigned)w[e]);
            bsum[i] += t;
        }
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;

    __builtin_amdgcn_s_barrier();   // P0
    int st = 0;
    for (int ui = 0; ui < n_units; ++ui) {
        int n0, k0, kt0, nk;
        hw_unit(p, b, ui, grid, n0, k0, kt0, nk);
        const bool do_bias = k0 == 0 && wn == 0 && p.bias[n0 / p.cseg] != nullptr;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int rr = 0; rr < 16; ++rr) acc[i][j][rr] = 0.f;
        // rows of the LAST contraction tile of this unit that exist (64 = all)
        const int last_rows = min(HB_BK, p.M - (kt0 + nk - 1) * HB_BK);
        const bool ragged = last_rows < HB_BK;
        read_set(I0{}, lds_base + st * HB_STAGE, 0);
        if (ragged && nk == 1) mask_set(I0{}, 0, last_rows);
        for (int kt = 0; kt + 1 < nk; ++kt) {
            const unsigned cur = lds_base + st * HB_STAGE;
            st = st == HB_S - 1 ? 0 : st + 1;
            const unsigned nxt = lds_base + st * HB_STAGE;
            read_set(I1{}, cur, 1);
            __builtin_amdgcn_sched_barrier(0);
            mfmas(I0{});
            if (do_bias) bias_acc(I0{});
            __builtin_amdgcn_sched_barrier(0);
            read_set(I0{}, cur, 2);
            __builtin_amdgcn_sched_barrier(0);
            mfmas(I1{});
            if (do_bias) bias_acc(I1{});
            __builtin_amdgcn_sched_barrier(0);
            read_set(I1{}, cur, 3);
            __builtin_amdgcn_sched_barrier(0);
            mfmas(I0{});
            if (do_bias) bias_acc(I0{});
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            read_set(I0{}, nxt, 0);
            if (ragged && kt + 2 == nk) mask_set(I0{}, 0, last_rows);
            __builtin_amdgcn_sched_barrier(0);
            mfmas(I1{});
            if (do_bias) bias_acc(I1{});
            __builtin_amdgcn_sched_barrier(0);
        }
        {
            const unsigned cur = lds_base + st * HB_STAGE;
            st = st == HB_S - 1 ? 0 : st + 1;
            read_set(I1{}, cur, 1);
            if (ragged) mask_set(I1{}, 1, last_rows);
            __builtin_amdgcn_sched_barrier(0);
            mfmas(I0{});
            if (do_bias) bias_acc(I0{});
            __builtin_amdgcn_sched_barrier(0);
            read_set(I0{}, cur, 2);
            if (ragged) mask_set(I0{}, 2, last_rows);
            __builtin_amdgcn_sched_barrier(0);
            mfmas(I1{});
            if (do_bias) bias_acc(I1{});
            __builtin_amdgcn_sched_barrier(0);
            read_set(I1{}, cur, 3);
            if (ragged) mask_set(I1{}, 3, last_rows);
            __builtin_amdgcn_sched_barrier(0);
            mfmas(I0{});
            if (do_bias) bias_acc(I0{});
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            mfmas(I1{});
            if (do_bias) bias_acc(I1{});
        }
        // acc[i][j][rr]: n = n0 + wm 64 + 32 i + 4 hi + (rr & 3) + 8 (rr >> 2), k = k0 + wn 64 + 32 j + l31
        if (p.flags & 8) continue;
        const int seg = n0 / p.cseg;
        if (p.ws != nullptr) {
            const int u = hw_unit_of(b, ui, grid, p.units);      // = split * tiles + tile
            f32x4* __restrict__ w = reinterpret_cast<f32x4*>(p.ws) + ((long)u * 8 + wave) * (4 * 4 * 64) + lane;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int qq = 0; qq < 4; ++qq)
                        w[((2 * i + j) * 4 + qq) * 64] =
                            f32x4{acc[i][j][4 * qq], acc[i][j][4 * qq + 1], acc[i][j][4 * qq + 2], acc[i][j][4 * qq + 3]};
            if (do_bias) {
                const int split = u / p.tiles;
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const float t = bsum[i] + __shfl_xor(bsum[i], 32);
                    if (hi == 0) p.ws_b[(long)split * p.N + n0 + wm * 64 + 32 * i + l31] = t;
                    bsum[i] = 0.f;
                }
            }
            continue;
        }
        const int nl0 = n0 - seg * p.cseg + wm * 64 + 4 * hi;       // this lane's first row inside the segment
        float* __restrict__ cb = p.C[seg] + (long)nl0 * p.ldc + k0 + wn * 64 + l31;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int rr = 0; rr < 16; ++rr)
                    if (nl0 + 32 * i + (rr & 3) + 8 * (rr >> 2) < p.n_valid)
                        unsafeAtomicAdd(cb + (long)(32 * i + (rr & 3) + 8 * (rr >> 2)) * p.ldc + 32 * j, acc[i][j][rr]);
        if (do_bias) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const float t = bsum[i] + __shfl_xor(bsum[i], 32);
                if (hi == 0 && n0 - seg * p.cseg + wm * 64 + 32 * i + l31 < p.n_valid)
                    unsafeAtomicAdd(p.bias[seg] + (n0 - seg * p.cseg + wm * 64 + 32 * i + l31), t);
                bsum[i] = 0.f;
            }
        }
    }
}

// Second pass of the deterministic weight gradient: dW += sum over the splits, IN SPLIT ORDER, of the partial tiles the units
// stored in accumulator order; db likewise. One thread per float4 of a tile (the four values are four ROWS of one column: lane
// l31 -> consecutive k, so a half-wave reads and writes 128 consecutive bytes of a dW row), the last blocks take the bias.
__global__ __launch_bounds__(256) void wgrad_bf16_reduce_kernel(const HwP p) {
    const long n4 = (long)p.tiles * 8192;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx < n4) {
        const int t = (int)(idx >> 13), in = (int)(idx & 8191);
        const int wave = in >> 10, ij = (in >> 8) & 3, qq = (in >> 6) & 3, lane = in & 63;
        const int l31 = lane & 31, hi = lane >> 5, wm = wave >> 1, wn = wave & 1, i = ij >> 1, j = ij & 1;
        const f32x4* __restrict__ w = reinterpret_cast<const f32x4*>(p.ws) + idx;
        f32x4 a = w[0];
        for (int s = 1; s < p.splits; ++s) a += w[(long)s * n4];
        const int n0 = (t / p.tiles_k) * HB_BM, k0 = (t % p.tiles_k) * HB_BN;
        const int seg = n0 / p.cseg;
        const int row = n0 - seg * p.cseg + wm * 64 + 32 * i + 4 * hi + 8 * qq;
        float* __restrict__ c = p.C[seg] + (long)row * p.ldc + k0 + wn * 64 + 32 * j + l31;
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (row + e < p.n_valid) c[(long)e * p.ldc] += a[e];
    } else {
        const long n = idx - n4;
        if (n >= p.N) return;
        const int seg = (int)(n / p.cseg);
        if (p.bias[seg] == nullptr || n - (long)seg * p.cseg >= p.n_valid) return;
        float a = p.ws_b[n];
        for (int s = 1; s < p.splits; ++s) a += p.ws_b[(long)s * p.N + n];
        p.bias[seg][n - (long)seg * p.cseg] += a;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// small kernels
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void cast_f32_bf16_kernel(long n8, const float* __restrict__ x, unsigned short* __restrict__ y,
                                                            long n) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n8) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(x + 8 * i), c = *reinterpret_cast<const f32x4*>(x + 8 * i + 4);
        *reinterpret_cast<v4i*>(y + 8 * i) = v4i{(int)pack_bf16(a[0], a[1]), (int)pack_bf16(a[2], a[3]), (int)pack_bf16(c[0], c[1]),
                                                 (int)pack_bf16(c[2], c[3])};
    } else if (i == n8) {
        for (long e = 8 * n8; e < n; ++e) y[e] = bf16_rne(x[e]);
    }
}

// fp32 [rows][n] (row stride ldx) -> bf16 [rows][ldy], columns n .. ldy - 1 zero-filled: the padded bf16 operand of a weight
// gradient whose output width is not a tile multiple (the 30,522-wide MLM decoder: 30,720 = 120 x 256)
__global__ __launch_bounds__(256) void cast_rows_f32_bf16_kernel(long rows, int n, const float* __restrict__ x, long ldx,
                                                                 unsigned short* __restrict__ y, long ldy) {
    const long chunks = ldy / 8;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= rows * chunks) return;
    const long r = i / chunks;
    const int c = (int)(i % chunks) * 8;
    const float* __restrict__ xp = x + r * ldx + c;
    f32x4 a = {0.f, 0.f, 0.f, 0.f}, b = {0.f, 0.f, 0.f, 0.f};
    if (c + 8 <= n) {
        a = *reinterpret_cast<const f32x4*>(xp);
        b = *reinterpret_cast<const f32x4*>(xp + 4);
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (c + e < n) a[e] = xp[e];
            if (c + 4 + e < n) b[e] = xp[4 + e];
        }
    }
    *reinterpret_cast<v4i*>(y + r * ldy + c) = v4i{(int)pack_bf16(a[0], a[1]), (int)pack_bf16(a[2], a[3]), (int)pack_bf16(b[0], b[1]),
                                                   (int)pack_bf16(b[2], b[3])};
}

__global__ __launch_bounds__(256) void cast_bf16_f32_kernel(long n8, const unsigned short* __restrict__ x, float* __restrict__ y,
                                                            long n) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n8) {
        const v4i w = *reinterpret_cast<const v4i*>(x + 8 * i);
        *reinterpret_cast<f32x4*>(y + 8 * i) = f32x4{bf16_lo(w[0]), bf16_hi(w[0]), bf16_lo(w[1]), bf16_hi(w[1])};
        *reinterpret_cast<f32x4*>(y + 8 * i + 4) = f32x4{bf16_lo(w[2]), bf16_hi(w[2]), bf16_lo(w[3]), bf16_hi(w[3])};
    } else if (i == n8) {
        for (long e = 8 * n8; e < n; ++e) y[e] = __uint_as_float((unsigned)x[e] << 16);
    }
}

// fp32 master weight [rows, cols] -> bf16 shadow rows (w16 [rows, ld16]) AND its transpose (wt16 [cols, ldt], written at
// column offset col_off = the row offset of this segment inside a stacked weight); 64 x 64 tiles through LDS
__global__ __launch_bounds__(256) void weight_shadow_kernel(int rows, int cols, const float* __restrict__ w, long ldw,
                                                            unsigned short* __restrict__ w16, long ld16,
                                                            unsigned short* __restrict__ wt16, long ldt) {
    __shared__ unsigned short tile[64][66];
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    const int tr = threadIdx.x >> 4, tc = (threadIdx.x & 15) * 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = tr + 16 * i;
        const f32x4 v = *reinterpret_cast<const f32x4*>(w + (long)(r0 + row) * ldw + c0 + tc);
        const unsigned lo = pack_bf16(v[0], v[1]), hi = pack_bf16(v[2], v[3]);
        if (w16 != nullptr) *reinterpret_cast<uint2*>(w16 + (long)(r0 + row) * ld16 + c0 + tc) = uint2{lo, hi};
        tile[row][tc] = (unsigned short)lo; tile[row][tc + 1] = (unsigned short)(lo >> 16);
        tile[row][tc + 2] = (unsigned short)hi; tile[row][tc + 3] = (unsigned short)(hi >> 16);
    }
    __syncthreads();
    if (wt16 == nullptr) return;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int col = tr + 16 * i;      // row of the transposed tile
        const unsigned lo = (unsigned)tile[tc][col] | ((unsigned)tile[tc + 1][col] << 16);
        const unsigned hi = (unsigned)tile[tc + 2][col] | ((unsigned)tile[tc + 3][col] << 16);
        *reinterpret_cast<uint2*>(wt16 + (long)(c0 + col) * ldt + r0 + tc) = uint2{lo, hi};
    }
}

// the same for EVERY registered weight in one launch (once per optimizer step): block b finds its segment in the table by
// bisection over the segments' first tile
__global__ __launch_bounds__(256) void weight_shadow_multi_kernel(int n_segs, const vb_shadow_seg* __restrict__ tab) {
    __shared__ unsigned short tile[64][66];
    int lo = 0, hi = n_segs - 1;
    const long b = blockIdx.x;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (tab[mid].tile0 <= b) lo = mid; else hi = mid - 1;
    }
    const vb_shadow_seg sg = tab[lo];
    const int t = (int)(b - sg.tile0), tiles_c = sg.cols >> 6;
    const int r0 = (t / tiles_c) * 64, c0 = (t % tiles_c) * 64;
    const int tr = threadIdx.x >> 4, tc = (threadIdx.x & 15) * 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = tr + 16 * i;
        const f32x4 v = *reinterpret_cast<const f32x4*>(sg.w + (long)(r0 + row) * sg.cols + c0 + tc);
        const unsigned lo2 = pack_bf16(v[0], v[1]), hi2 = pack_bf16(v[2], v[3]);
        *reinterpret_cast<uint2*>(sg.w16 + (long)(r0 + row) * sg.ld16 + c0 + tc) = uint2{lo2, hi2};
        tile[row][tc] = (unsigned short)lo2; tile[row][tc + 1] = (unsigned short)(lo2 >> 16);
        tile[row][tc + 2] = (unsigned short)hi2; tile[row][tc + 3] = (unsigned short)(hi2 >> 16);
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int col = tr + 16 * i;
        const unsigned lo2 = (unsigned)tile[tc][col] | ((unsigned)tile[tc + 1][col] << 16);
        const unsigned hi2 = (unsigned)tile[tc + 2][col] | ((unsigned)tile[tc + 3][col] << 16);
        *reinterpret_cast<uint2*>(sg.wt16 + (long)(c0 + col) * sg.ldt + r0 + tc) = uint2{lo2, hi2};
    }
}

// column sums of a bf16 [rows, cols] matrix (bias gradient): stage 1 - a block owns 256 columns x one row slab, a thread 4
// columns of every fourth row, waves summed through LDS, one partial row per slab; stage 2 - the slabs in order
// (deterministic). out: ADDED into (the gradient arena semantics of the weight gradients).
constexpr int CS_SLABS = 64;
__global__ __launch_bounds__(256) void colsum16_kernel(long rows, int cols, const unsigned short* __restrict__ x, long ldx,
                                                       float* __restrict__ part) {
    __shared__ f32x4 red[3][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = blockIdx.x * 256 + 4 * lane;
    const long per = (rows + CS_SLABS - 1) / CS_SLABS;
    const long lo = blockIdx.y * per, hi = min(rows, lo + per);
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    if (col < cols)
        for (long r = lo + wave; r < hi; r += 4) {
            const uint2 w = *reinterpret_cast<const uint2*>(x + r * ldx + col);
            s += f32x4{bf16_lo(w.x), bf16_hi(w.x), bf16_lo(w.y), bf16_hi(w.y)};
        }
    if (wave > 0) red[wave - 1][lane] = s;
    __syncthreads();
    if (wave == 0 && col < cols) {
        s += red[0][lane]; s += red[1][lane]; s += red[2][lane];
        *reinterpret_cast<f32x4*>(part + (long)blockIdx.y * cols + col) = s;
    }
}
__global__ __launch_bounds__(256) void colsum16_finish_kernel(int cols, const float* __restrict__ part, float* __restrict__ out) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= cols) return;
    float s = 0.f;
    for (int i = 0; i < CS_SLABS; ++i) s += part[(long)i * cols + c];
    out[c] += s;
}

}  // namespace

extern "C" int vb_linear_bf16(void* stream, const vb_linear_bf16_args* a) {
    if (a == nullptr || a->A == nullptr || a->W == nullptr) return VB_E_BADARG;
    if ((a->C != nullptr) + (a->C32 != nullptr) != 1) return VB_E_BADARG;
    if (a->M <= 0 || a->N <= 0 || a->K <= 0) return VB_E_BADARG;
    if (a->K % HB_BK != 0 || a->N % HB_BN != 0) return VB_E_ALIGN;
    if (a->lda % 8 != 0 || a->ldw % 8 != 0 || a->lda < a->K || a->ldw < a->K || !vb_aligned16(a->A) || !vb_aligned16(a->W))
        return VB_E_ALIGN;
    if (a->C != nullptr && (a->ldc % 8 != 0 || a->ldc < a->N || !vb_aligned16(a->C))) return VB_E_ALIGN;
    if (a->C32 != nullptr && (a->ldc32 % 4 != 0 || a->ldc32 < a->N || !vb_aligned16(a->C32))) return VB_E_ALIGN;
    const int nbias = a->bias_segments > 0 ? a->bias_segments : 1;
    if (nbias > VB_MAX_SEGMENTS || a->N % nbias != 0 || (a->N / nbias) % 32 != 0) return VB_E_SEGMENT;
    for (int s = 0; s < nbias; ++s)
        if (a->bias[s] != nullptr && !vb_aligned16(a->bias[s])) return VB_E_ALIGN;
    if (a->residual != nullptr && a->mul != nullptr) return VB_E_BADARG;
    const uint16_t* second = a->residual != nullptr ? a->residual : a->mul;
    const int64_t ld2 = a->residual != nullptr ? a->ldr : a->ldm;
    if (second != nullptr && (ld2 % 8 != 0 || ld2 < a->N || !vb_aligned16(second))) return VB_E_ALIGN;
    if (a->act_grad != nullptr && (a->ldg % 8 != 0 || a->ldg < a->N || !vb_aligned16(a->act_grad))) return VB_E_ALIGN;
    if (a->act != VB_ACT_NONE && a->act != VB_ACT_GELU && a->act != VB_ACT_RELU) return VB_E_BADARG;
    if (!(a->dropout_p >= 0.f && a->dropout_p < 1.f)) return VB_E_BADARG;
    if (a->lda * 2 * 256 > 0xffffffffL || a->ldw * 2 * 256 > 0xffffffffL) return VB_E_RANGE;
    const bool drop = a->dropout_p > 0.f;
    // combinations the model uses; anything else is not built
    if (a->act != VB_ACT_NONE && (second != nullptr || drop)) return VB_E_BADARG;
    if (a->act_grad != nullptr && a->act != VB_ACT_GELU) return VB_E_BADARG;
    if (drop && a->residual == nullptr) return VB_E_BADARG;
    HbP p{};
    p.M = (int)a->M; p.N = (int)a->N; p.K = (int)a->K;
    p.A = a->A; p.lda = a->lda; p.B = a->W; p.ldb = a->ldw;
    for (int s = 0; s < nbias; ++s) p.bias[s] = a->bias[s];
    p.bseg = p.N / nbias;
    p.R = second; p.ldr = ld2;
    p.C = a->C; p.ldc = a->ldc; p.C32 = a->C32; p.ldc32 = a->ldc32;
    p.D = a->act_grad; p.ldd = a->ldg;
    p.tiles_n = p.N / HB_BN;
    p.drop_p = a->dropout_p;
    p.drop_scale = drop ? 1.0f / (1.0f - a->dropout_p) : 1.0f;
    p.seed = a->seed;
    p.epoch = vb_seed_epoch();
    static const int lab_flags = [] { const char* e = getenv("VB_BF16_FLAGS"); return e ? atoi(e) : 0; }();
    p.flags = lab_flags;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (a->C32 != nullptr) {
        if (a->act == VB_ACT_GELU || drop || a->mul != nullptr) return VB_E_BADARG;
        if (a->act == VB_ACT_RELU) return launch_hb<HB_OUT_F32, HB_RELU>(st, p);
        return a->residual != nullptr ? launch_hb<HB_OUT_F32, HB_RES>(st, p) : launch_hb<HB_OUT_F32, HB_PLAIN>(st, p);
    }
    if (a->act == VB_ACT_GELU) return launch_hb<HB_OUT_BF16, HB_GELU>(st, p);
    if (a->act == VB_ACT_RELU) return launch_hb<HB_OUT_BF16, HB_RELU>(st, p);
    if (a->mul != nullptr) return launch_hb<HB_OUT_BF16, HB_MUL>(st, p);
    if (drop) return launch_hb<HB_OUT_BF16, HB_DROPRES>(st, p);
    if (a->residual != nullptr) return launch_hb<HB_OUT_BF16, HB_RES>(st, p);
    return launch_hb<HB_OUT_BF16, HB_PLAIN>(st, p);
}

extern "C" int vb_wgrad_bf16(void* stream, const vb_wgrad_bf16_args* a) {
    if (a == nullptr || a->dY == nullptr || a->X == nullptr) return VB_E_BADARG;
    if (a->M <= 0 || a->K <= 0 || a->nseg <= 0 || a->nseg > VB_MAX_SEGMENTS || a->seg_n <= 0) return VB_E_BADARG;
    const long N = (long)a->nseg * a->seg_n;
    if (a->seg_n % HB_BM != 0 || a->K % HB_BN != 0) return VB_E_ALIGN;
    if (a->ldy % 8 != 0 || a->ldx % 8 != 0 || a->ldy < N || a->ldx < a->K || !vb_aligned16(a->dY) || !vb_aligned16(a->X))
        return VB_E_ALIGN;
    if (a->ldw < a->K) return VB_E_ALIGN;
    if (a->ldy * 2 * 64 > 0xffffffffL || a->ldx * 2 * 64 > 0xffffffffL) return VB_E_RANGE;
    HwP p{};
    p.M = (int)a->M; p.N = (int)N; p.K = (int)a->K;
    p.Y = a->dY; p.ldy = a->ldy; p.X = a->X; p.ldx = a->ldx;
    for (int s = 0; s < a->nseg; ++s) {
        if (a->dW[s] == nullptr || (reinterpret_cast<uintptr_t>(a->dW[s]) & 3u) != 0) return VB_E_BADARG;
        p.C[s] = a->dW[s];
        p.bias[s] = a->dbias[s];
    }
    p.ldc = a->ldw; p.cseg = a->seg_n;
    if (a->n_valid < 0 || a->n_valid > a->seg_n || (a->n_valid != 0 && a->n_valid != a->seg_n && a->nseg != 1)) return VB_E_BADARG;
    p.n_valid = a->n_valid > 0 ? a->n_valid : a->seg_n;
    p.tiles_k = p.K / HB_BN;
    p.tiles = (p.N / HB_BM) * p.tiles_k;
    p.nkt = (p.M + HB_BK - 1) / HB_BK;
    // Contraction splits by a time model (measured with the laboratory flags, profiles/r05_bf16_lab_ablations.txt): a unit's
    // main loop costs ~1.0 us per contraction tile, its epilogue - 128 KiB of fp32 atomics that execute at the memory side,
    // ~1.7 TB/s for the whole chip - ~0.075 us per unit IN FLIGHT ANYWHERE (0.12 in the model: in the step, where other streams compete for the memory side, fewer splits measured +0.7 %); rounds of 256 units. More splits shorten the main
    // loop and lengthen the atomics: the first version's "fill two rounds" rule spent 30 - 50 % of a launch in atomics.
    static const float t_k = [] { const char* e = getenv("VB_BF16_WG_TK"); return e ? (float)atof(e) : 1.0f; }();
    static const float t_e = [] { const char* e = getenv("VB_BF16_WG_TE"); return e ? (float)atof(e) : 0.12f; }();
    // deterministic form: a unit's 128 KiB leave as plain 16-byte stores (t_d per unit), and the reduce pass reads every
    // partial once and updates dW: (splits + 2) x 4 N K bytes at ~3.5 TB/s + its launch
    static const float t_d = [] { const char* e = getenv("VB_BF16_WG_TD"); return e ? (float)atof(e) : 0.04f; }();
    hipStream_t st = static_cast<hipStream_t>(stream);
    size_t slice_bytes = 0;
    float* slice = det_on() ? det_slice(st, &slice_bytes) : nullptr;
    const size_t per_split = ((size_t)p.tiles * 32768 + (size_t)p.N) * sizeof(float);
    int best = 1;
    float best_t = 1e30f;
    for (int sp = 1; sp <= p.nkt && sp <= 64; ++sp) {
        const int per = (p.nkt + sp - 1) / sp, real = (p.nkt + per - 1) / per;
        if (real != sp) continue;
        const long units = (long)p.tiles * sp;
        float t = (float)((units + 255) / 256) * (per * t_k + 2.0f);
        if (slice != nullptr) {
            if ((size_t)sp * per_split > slice_bytes) continue;
            t += units * t_d + 3.0f + (float)(sp + 2) * (4.0f * p.N * p.K) / 3.5e6f;
        } else {
            t += units * t_e;
        }
        if (t < best_t) { best_t = t; best = sp; }
    }
    if (best_t >= 1e30f) slice = nullptr;          // (not even one split fits the slice)
    if (det_on() && slice == nullptr) det_fallback();
    p.kt_per_split = (p.nkt + best - 1) / best;
    p.splits = (p.nkt + p.kt_per_split - 1) / p.kt_per_split;
    p.units = p.tiles * p.splits;
    if (slice != nullptr) {
        p.ws = slice;
        p.ws_b = slice + (size_t)p.splits * p.tiles * 32768;
    }
    static const int lab_flags = [] { const char* e = getenv("VB_BF16_FLAGS"); return e ? atoi(e) : 0; }();
    p.flags = lab_flags;
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_bf16_kernel),
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, HB_LDS);
    if (attr != hipSuccess) return (int)attr;
    const int cus = hb_grid_limit();
    const int grid = p.units < cus ? (p.units + 7) / 8 * 8 : cus;
    hipLaunchKernelGGL(wgrad_bf16_kernel, dim3(grid), dim3(HB_THREADS), HB_LDS, st, p);
    VB_LAUNCH_CHECK();
    if (p.ws != nullptr && !(p.flags & 8)) {
        const long work = (long)p.tiles * 8192 + p.N;
        hipLaunchKernelGGL(wgrad_bf16_reduce_kernel, dim3((unsigned)((work + 255) / 256)), dim3(256), 0, st, p);
        VB_LAUNCH_CHECK();
    }
    return 0;
}

extern "C" int vb_cast_f32_bf16(void* stream, int64_t n, const float* x, uint16_t* y) {
    if (x == nullptr || y == nullptr || n <= 0) return VB_E_BADARG;
    if (!vb_aligned16(x) || !vb_aligned16(y)) return VB_E_ALIGN;
    const long n8 = n / 8;
    hipLaunchKernelGGL(cast_f32_bf16_kernel, dim3((unsigned)((n8 + 1 + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       n8, x, y, (long)n);
    VB_LAUNCH_CHECK();
    return 0;
}

extern "C" int vb_cast_rows_f32_bf16(void* stream, int64_t rows, int32_t n, const float* x, int64_t ldx, uint16_t* y, int64_t ldy) {
    if (x == nullptr || y == nullptr || rows <= 0 || n <= 0 || ldx < n || ldy < n) return VB_E_BADARG;
    if (ldx % 4 != 0 || ldy % 8 != 0 || !vb_aligned16(x) || !vb_aligned16(y)) return VB_E_ALIGN;
    const long work = rows * (ldy / 8);
    hipLaunchKernelGGL(cast_rows_f32_bf16_kernel, dim3((unsigned)((work + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       (long)rows, (int)n, x, (long)ldx, y, (long)ldy);
    VB_LAUNCH_CHECK();
    return 0;
}

extern "C" int vb_cast_bf16_f32(void* stream, int64_t n, const uint16_t* x, float* y) {
    if (x == nullptr || y == nullptr || n <= 0) return VB_E_BADARG;
    if (!vb_aligned16(x) || !vb_aligned16(y)) return VB_E_ALIGN;
    const long n8 = n / 8;
    hipLaunchKernelGGL(cast_bf16_f32_kernel, dim3((unsigned)((n8 + 1 + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       n8, x, y, (long)n);
    VB_LAUNCH_CHECK();
    return 0;
}

extern "C" int vb_weight_shadow_bf16(void* stream, int32_t rows, int32_t cols, const float* w, int64_t ldw, uint16_t* w16,
                                     int64_t ld16, uint16_t* wt16, int64_t ldt) {
    if (w == nullptr || (w16 == nullptr && wt16 == nullptr) || rows <= 0 || cols <= 0) return VB_E_BADARG;
    if (rows % 64 != 0 || cols % 64 != 0 || ldw % 4 != 0 || !vb_aligned16(w)) return VB_E_ALIGN;
    if (w16 != nullptr && (ld16 % 4 != 0 || ld16 < cols || (reinterpret_cast<uintptr_t>(w16) & 7u) != 0)) return VB_E_ALIGN;
    if (wt16 != nullptr && (ldt % 4 != 0 || ldt < rows || (reinterpret_cast<uintptr_t>(wt16) & 7u) != 0)) return VB_E_ALIGN;
    hipLaunchKernelGGL(weight_shadow_kernel, dim3(cols / 64, rows / 64), dim3(256), 0, static_cast<hipStream_t>(stream), rows, cols,
                       w, (long)ldw, w16, (long)ld16, wt16, (long)ldt);
    VB_LAUNCH_CHECK();
    return 0;
}

extern "C" int vb_weight_shadow_multi(void* stream, int32_t n_segs, const vb_shadow_seg* table, int64_t total_tiles) {
    if (table == nullptr || n_segs <= 0 || total_tiles <= 0 || total_tiles > 0x7fffffffL) return VB_E_BADARG;
    hipLaunchKernelGGL(weight_shadow_multi_kernel, dim3((unsigned)total_tiles), dim3(256), 0, static_cast<hipStream_t>(stream), n_segs,
                       table);
    VB_LAUNCH_CHECK();
    return 0;
}

extern "C" int64_t vb_colsum_bf16_workspace(int32_t cols) { return (int64_t)CS_SLABS * cols; }

extern "C" int vb_colsum_bf16(void* stream, int64_t rows, int32_t cols, const uint16_t* x, int64_t ldx, float* out,
                              float* workspace) {
    if (x == nullptr || out == nullptr || workspace == nullptr || rows <= 0 || cols <= 0) return VB_E_BADARG;
    if (cols % 4 != 0 || ldx % 4 != 0 || ldx < cols || (reinterpret_cast<uintptr_t>(x) & 7u) != 0 || !vb_aligned16(workspace))
        return VB_E_ALIGN;
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(colsum16_kernel, dim3((cols + 255) / 256, CS_SLABS), dim3(256), 0, st, (long)rows, cols, x, (long)ldx, workspace);
    VB_LAUNCH_CHECK();
    hipLaunchKernelGGL(colsum16_finish_kernel, dim3((cols + 255) / 256), dim3(256), 0, st, cols, workspace, out);
    VB_LAUNCH_CHECK();
    return 0;
}
