// MX (OCP microscaling) e4m3 forward path - BASELINE configs[4] "fp8 MFMA co-attention path", round 4.
//
// The round-2 fp8 path (fp8.hip) scales whole ROWS: a producer that only sees a tile of its output row (a GEMM epilogue,
// an attention head) cannot emit codes, so every linear paid a separate quantiser pass over an fp32 tensor, and the
// GEMM outputs had to be fp32 (4 bytes of HBM traffic per 2 K flops). gfx950's scaled MFMA takes one E8M0 scale per 32
// consecutive K elements of each operand - the MX block format - which makes quantisation LOCAL: any producer that holds
// 32 consecutive output columns of a row can write that block's codes and its scale byte. This file holds
//
//   * the MX format (numerics restated on the CPU in oracle/fp8_oracle.py, `mx_*`):
//       block      = 32 consecutive elements of a row (the contraction dimension of the consuming GEMM)
//       scale      = 2^e, e = the smallest integer with amax_block / 2^e <= 448 (e4m3 max), stored as the E8M0 byte
//                    e + 127 (clamped at 0; an all-zero block gets byte 0)
//       codes      = e4m3fn_rne(x * 2^-e)  (exact scaling, one rounding)
//       layout     = codes [rows][K] bytes, row-major; scales as uint32 words S[K / 128][srows]: word (kt, row) holds the
//                    4 scale bytes of the row's K range [128 kt, 128 kt + 128), byte b = block 4 kt + b - exactly what one
//                    K tile of the GEMM below needs per row, contiguous over the rows of an output tile (one 1 KiB DMA)
//   * quant_rows_mx_kernel: fp32 rows -> MX (weights once per optimizer step; activations that no producer quantises yet)
//   * gemm_mx_kernel: C = act(A W^T + bias) (+ residual), A and W in MX, fp32 accumulate on
//     v_mfma_scale_f32_32x32x64_f8f6f4 WITH the operands' block scales (hardware dequantisation), result written as fp32
//     and / or straight as MX codes + scales for the NEXT linear (the GELU output of an FFN never exists in fp32).
//
// Kernel design (the gemm_v4.h skeleton re-cut for byte operands; what bounded fp8.hip's 128 x 128 blocks of 4 waves was
// the LDS->register fragment traffic - 256 B/clk/CU needed for a busy matrix pipe, the whole LDS bandwidth - and the
// L2->LDS operand traffic, 64 B/clk/CU):
//   block   = ONE persistent block per CU: 8 MFMA waves in a 4 x 2 grid, wave tile 64 x 64 (2 x 2 MFMA tiles of 32 x 32),
//             block tile 256 x 128; + 2 loader waves (wave 8: the A tile + its scales, wave 9: the W tile + its scales)
//             that only issue LDS-DMA (global_load_lds_dwordx4, 1 KiB each)
//   K tile  = 128 bytes per row: A 32 KiB + W 16 KiB + 2 x 1 KiB of scale words = 50 KiB per stage, 3 stages (150 KiB)
//   traffic = fragments 128 B/clk/CU at a busy pipe (half the LDS bandwidth), operands 48 B/clk/CU from L2
//   LDS     = rows of 128 bytes; the eight 16-byte chunks of a row XOR-swizzled with (row >> 1) & 7 (fp8.hip's layout:
//             fragment ds_read_b128 conflict-free); the DMA writes lane-linear, so the swizzle is applied on the SOURCE
//             address (lane = (row-in-8, slot) loads chunk slot ^ key(row))
//   loop    = the K tiles of ALL output tiles of a block form one stream (the next tile's first K tiles land during the
//             epilogue). A K tile is two MFMA sub-steps s = 0, 1 (K = 64 each) on two static fragment register sets:
//                 (g, 0): MFMAs on set 0 | read set 1 = (g, 1) from stage g        -> barrier B_g
//                 (g, 1): MFMAs on set 1 | read set 0 = (g + 1, 0) from stage g + 1
//             so ALL reads of stage g happen before B_g: ONE barrier per K tile and none at output-tile boundaries. The
//             loaders guarantee tile g + 1 has landed before B_g and issue tile g + 3 into stage g right after it: every
//             DMA has two full K steps to land.
//   product = TRANSPOSED (first MFMA operand = W fragment): lane (l31, hi) ends up with row m = l31 of the output tile and
//             the columns 8 q + 4 hi + e of a 32-column block in its 16 accumulator registers - float4 bias / residual /
//             store, and a 32-column MX block is 16 in-lane values + ONE cross-lane exchange (lane ^ 32).
#include "gemm_core.h"
#include "mx8.h"
#include <type_traits>

namespace {

using namespace vbgemm;

typedef int v8i __attribute__((ext_vector_type(8)));
typedef int v4i __attribute__((ext_vector_type(4)));

// four consecutive elements of an fp32 or a bfloat16 row as f32x4
__device__ __forceinline__ f32x4 mx_ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ f32x4 mx_ld4(const unsigned short* p) {
    const uint2 t = *reinterpret_cast<const uint2*>(p);
    return f32x4{__uint_as_float(t.x << 16), __uint_as_float(t.x & 0xffff0000u), __uint_as_float(t.y << 16),
                 __uint_as_float(t.y & 0xffff0000u)};
}

template <int NV, class IN>
__global__ __launch_bounds__(256) void quant_rows_mx_kernel(long rows, int K, const IN* __restrict__ x, long ldx,
                                                            unsigned char* __restrict__ q, long ldq,
                                                            unsigned* __restrict__ sc, long sc_rows) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const IN* __restrict__ src = x + row * ldx;
    unsigned char* __restrict__ qrow = q + row * ldq;
    const int nkt = K >> 7;
    if (NV > 0) {
        f32x4 reg[NV > 0 ? NV : 1];
#pragma unroll
        for (int i = 0; i < NV; ++i) reg[i] = mx_ld4(src + 256 * i + 4 * lane);
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int kt = 2 * i + (lane >> 5);
            mx_quant_chunk(reg[i], true, lane, kt, nkt, reinterpret_cast<unsigned*>(qrow + 256 * i + 4 * lane),
                           sc + (long)kt * sc_rows + row);
        }
    } else {
        for (int c0 = 0; c0 < K; c0 += 256) {
            const int col = c0 + 4 * lane;
            const bool ok = col < K;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (ok) v = mx_ld4(src + col);
            const int kt = (c0 >> 7) + (lane >> 5);
            mx_quant_chunk(v, ok, lane, kt, nkt, reinterpret_cast<unsigned*>(qrow + (ok ? col : 0)),
                           sc + (long)(kt < nkt ? kt : 0) * sc_rows + row);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// GEMM
// ---------------------------------------------------------------------------------------------------------------
constexpr int MX_BM = 256, MX_BN = 128, MX_BK = 128, MX_S = 3;
constexpr int MX_A = MX_BM * MX_BK;            // 32,768 bytes
constexpr int MX_B = MX_BN * MX_BK;            // 16,384
constexpr int MX_SA = 1024, MX_SB = 1024;      // scale words of the tile's rows (W: 512 bytes used)
constexpr int MX_STAGE = MX_A + MX_B + MX_SA + MX_SB;   // 51,200
constexpr int MX_LDS = MX_S * MX_STAGE;        // 153,600
constexpr int MX_MFMA_WAVES = 8, MX_THREADS = 64 * (MX_MFMA_WAVES + 2);

struct MxP {
    int M, N, K;
    const unsigned char* A; long lda; const unsigned* sa; long sa_rows;
    const unsigned char* B; long ldb; const unsigned* sb; long sb_rows;
    const float* bias;
    const float* R; long ldr;
    const unsigned short* R16; long ldr16;   // bf16 residual (the MX inference mode keeps the residual stream in bf16)
    float* C; long ldc;
    unsigned char* Cq; long ldq; unsigned* cs; long cs_rows;
    unsigned short* Cb; long ldb16;     // bf16 output (round-to-nearest-even), may be null
    int act;
    int tiles_n, tiles;
    int flags;   // laboratory build only (VB_MX_FLAGS): 1 = the loaders issue no DMA, 2 = no MFMAs, 4 = no fragment reads,
                 // 8 = no global stores in the epilogue, 16 = no epilogue at all
};

// tile of block `b` in round `it` (gemm_v4.h: v4_tile_of / v4_tile_rc): the 32 blocks of an XCD (b % 8) work on a 4 x 8
// patch of tiles where the tile grid allows it, so they share A / W panels in their L2
__device__ __forceinline__ int mx_tile_of(int b, int it, int grid, int tiles) {
    const int base = it * grid;
    const int n = min(grid, tiles - base);
    if (n <= 0) return -1;
    if ((n & 7) != 0) return b < n ? base + b : -1;
    const int per = n >> 3, x = b & 7, j = b >> 3;
    return j < per ? base + x * per + j : -1;
}

__device__ __forceinline__ bool mx_origin(const MxP& p, int b, int it, int grid, int& m0, int& n0) {
    const int t = mx_tile_of(b, it, grid, p.tiles);
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
    m0 = r * MX_BM;
    n0 = c * MX_BN;
    return true;
}

// one LDS-DMA: LDS[lds + 16 lane] <- *(base + off[lane]), wave-uniform 64-bit base + per-lane 32-bit byte offset
__device__ __forceinline__ void mx_glds16(unsigned off, const void* base, unsigned lds) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(off), "s"(base), "s"(lds) : "memory");
}

template <int N>
__device__ __forceinline__ void mx_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// Loader wave of one operand: ND code DMAs (8 rows each) + 1 scale DMA per K tile.
// PERM (W operand of the fp32 / bf16 launches): LDS row 64 w + 32 j + k of the tile holds W row 64 w + 2 k + j - MFMA tile j of
// wave column w then multiplies the even (j = 0) / odd (j = 1) columns, so that lane k of the natural accumulator map owns
// the ADJACENT output columns 2 k, 2 k + 1: 8-byte fp32 / 4-byte bf16 stores, 256 / 128 contiguous bytes per row.
template <int ND, bool IS_A, bool PERM>
__device__ __forceinline__ void mx_loader(const MxP& p, const unsigned lds0, const int lane, const int nk, const int rounds) {
    const unsigned char* const mat = IS_A ? p.A : p.B;
    const long ld = IS_A ? p.lda : p.ldb;
    const int nrows = IS_A ? p.M : p.N;
    const unsigned* const sc = IS_A ? p.sa : p.sb;
    const long sc_rows = IS_A ? p.sa_rows : p.sb_rows;
    constexpr unsigned REG = IS_A ? 0u : (unsigned)MX_A;
    constexpr unsigned SREG = IS_A ? (unsigned)(MX_A + MX_B) : (unsigned)(MX_A + MX_B + MX_SA);
    constexpr int NI = ND + 1;
    static_assert(NI <= 63, "vmcnt is a 6-bit counter");
    unsigned off[ND];
    const unsigned char* base = nullptr;
    const unsigned* sbase = nullptr;
    // the W tile has 128 rows = 512 bytes of scale words: the upper half of the wave repeats the lower half's addresses
    const unsigned soff = IS_A ? 16u * lane : 16u * (lane & 31);
    auto set_tile = [&](int round) {
        int m0 = 0, n0 = 0;
        mx_origin(p, blockIdx.x, round, gridDim.x, m0, n0);
        const int r0 = IS_A ? m0 : n0;
        base = mat + (long)r0 * ld;
        sbase = sc + r0;
#pragma unroll
        for (int i = 0; i < ND; ++i) {
            const int row = 8 * i + (lane >> 3), slot = lane & 7;     // LDS row of this lane's 16 bytes
            const int chunk = slot ^ ((row >> 1) & 7);
#ifdef VB_GEMM_LAB
            const bool perm = PERM != ((p.flags & 64) != 0);      // lab: flip the row map (timing only, results garbage)
#else
            constexpr bool perm = PERM;
#endif
            const int grow = perm ? (row & 64) + 2 * (row & 31) + ((row >> 5) & 1) : row;   // matrix row it holds
            off[i] = (unsigned)((long)min(grow, nrows - 1 - r0) * ld + 16 * chunk);          // rows past the matrix: clamped, never stored
        }
    };
    int it = 0, kt = 0, stage_w = 0;
    auto issue_next = [&]() {
        const unsigned l = lds0 + (unsigned)stage_w * MX_STAGE;
#ifdef VB_GEMM_LAB
        if (!(p.flags & 1))
#endif
        {
#pragma unroll
            for (int i = 0; i < ND; ++i) mx_glds16(off[i], base, l + REG + 1024u * i);
            mx_glds16(soff, sbase, l + SREG);
        }
        base += MX_BK;
        sbase += sc_rows;
        stage_w = stage_w == MX_S - 1 ? 0 : stage_w + 1;
        if (++kt == nk) {
            kt = 0;
            ++it;
            if (it < rounds) set_tile(it);
        }
    };
    const int total = rounds * nk;
    set_tile(0);
    __builtin_amdgcn_s_setprio(2);
    for (int s = 0; s < MX_S && s < total; ++s) issue_next();
    if (total >= 2) mx_wait_vm<NI>(); else mx_wait_vm<0>();     // K tile 0 has landed (at most the newest tile is pending)
    __builtin_amdgcn_s_barrier();                                 // P0
    for (int g = 0; g < total; ++g) {
        // K tile g + 1 has landed: issued so far = min(total, g + 3) tiles, so at most tile g + 2 may be pending
        if (g + 3 <= total) mx_wait_vm<NI>(); else mx_wait_vm<0>();
        __builtin_amdgcn_s_barrier();                             // B_g: stage g has been read completely
        if (g + 3 < total) issue_next();
    }
}

__device__ __forceinline__ unsigned short bf16_rne(float v) {
    const unsigned u = __float_as_uint(v);
    return (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}

// KIND: what the launch writes = how the MFMA product is oriented = who owns what in the epilogue.
//   MX_OUT_F32 / MX_OUT_BF16: natural product (accumulator rows = output rows in registers, lane & 31 = output column) on a
//       column-permuted W tile (mx_loader PERM): lane k owns columns 2 k, 2 k + 1 of its wave's 64 and 16 rows per MFMA tile
//       row - one 8-byte (fp32) / 4-byte (bf16) store per row, 256 / 128 contiguous bytes per row and instruction;
//   MX_OUT_MX: transposed product (see the head of the file): lane (l31, hi) owns 32-column pieces of ONE row - the block
//       maximum is 16 in-lane values + one exchange with lane ^ 32.
// RES: 0 = no residual, 1 = fp32 residual (R), 2 = bf16 residual (R16).
// GELU / RES are compile-time too: runtime checks per element (and per-element predication of the 64 stores of a lane)
// made the epilogue the longest phase of a launch (tools/mx_lab_dbg VB_MX_FLAGS=15 vs 23: 57 us of a 140 us launch spent
// in an epilogue that stored nothing).
enum { MX_OUT_F32 = 0, MX_OUT_BF16 = 1, MX_OUT_MX = 2 };

template <int KIND, bool GELU, int RES>
__global__ __launch_bounds__(MX_THREADS) void gemm_mx_kernel(const MxP p) {
    // every launch uses the TRANSPOSED product since round 5: a lane owns ONE output row and 4-column groups of it, which
    // pairs of half-waves turn into 16-byte stores (fp32: directly; bf16: after a v_permlane32_swap) - the epilogue of a
    // persistent block is bound by store issue, and the natural map's 4- / 8-byte stores cost twice the instructions
    constexpr bool TRANS = true;
    constexpr bool HAS_R = RES != 0;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int nk = p.K / MX_BK;
    const int b = blockIdx.x, grid = gridDim.x;
    int rounds = 0;
    while (rounds * grid < p.tiles && mx_tile_of(b, rounds, grid, p.tiles) >= 0) ++rounds;
    if (rounds == 0) return;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (wave >= MX_MFMA_WAVES) {
        const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) char*)smem);
        if (wave == MX_MFMA_WAVES) mx_loader<MX_A / 1024, true, false>(p, lds0, threadIdx.x & 63, nk, rounds);
        else mx_loader<MX_B / 1024, false, !TRANS>(p, lds0, threadIdx.x & 63, nk, rounds);
        return;
    }
    const int lane = threadIdx.x & 63, l31 = lane & 31, hi = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    const int key = (l31 >> 1) & 7;
    const int fa_off = (wm * 64 + l31) * MX_BK, fb_off = MX_A + (wn * 64 + l31) * MX_BK;
    const int sa_off = MX_A + MX_B + 4 * (wm * 64 + l31);
    // scale word of the W row behind LDS row wn 64 + 32 j + l31 (PERM: matrix row wn 64 + 2 l31 + j); + sb_step per j
    const int sb_off = MX_A + MX_B + MX_SA + 4 * (wn * 64 + (TRANS ? l31 : 2 * l31));
    constexpr int sb_step = TRANS ? 128 : 4;

    f32x16 acc[2][2];
    v8i fa[2][2], fb[2][2];          // [register set = sub-step][tile]
    unsigned sca[2], scb[2];         // scale words of the current K tile (rows of the wave's two A / two W tiles)
    unsigned nsa[2], nsb[2];

    // Operand layout of v_mfma_scale_f32_32x32x64_f8f6f4 (measured with tools/mx_lab diag: per-K-block magnitudes only come
    // out right this way): registers 0-3 of lane (l31, hi) hold K = 16 hi .. 16 hi + 15 of row l31, registers 4-7 hold
    // K = 32 + 16 hi .. 32 + 16 hi + 15 - each lane carries half of BOTH 32-element scale blocks; the scale operand of
    // lanes 0-31 applies to block 0 (K 0-31) of their row, that of lanes 32-63 to block 1 (K 32-63).
    auto frag = [&](const char* base, int s) -> v8i {
        const int c0 = 4 * s + hi;
        const v4i lo = *reinterpret_cast<const v4i*>(base + ((c0 ^ key) << 4));
        const v4i up = *reinterpret_cast<const v4i*>(base + (((c0 + 2) ^ key) << 4));
        return __builtin_shufflevector(lo, up, 0, 1, 2, 3, 4, 5, 6, 7);
    };
    auto read_set = [&](auto S_, const char* stage) {
        constexpr int S = decltype(S_)::value;
#ifdef VB_GEMM_LAB
        if (p.flags & 4) return;
#endif
#pragma unroll
        for (int i = 0; i < 2; ++i) fa[S][i] = frag(stage + fa_off + i * 32 * MX_BK, S);
#pragma unroll
        for (int j = 0; j < 2; ++j) fb[S][j] = frag(stage + fb_off + j * 32 * MX_BK, S);
    };
    auto read_scales = [&](const char* stage, unsigned (&a)[2], unsigned (&bq)[2]) {
#pragma unroll
        for (int i = 0; i < 2; ++i) a[i] = *reinterpret_cast<const unsigned*>(stage + sa_off + 128 * i);
#pragma unroll
        for (int j = 0; j < 2; ++j) bq[j] = *reinterpret_cast<const unsigned*>(stage + sb_off + sb_step * j);
    };
    // the lane's scale byte for sub-step S (block 2 S + hi of the K tile), in all four byte lanes of the scale operand
    // (whatever byte the instruction's op_sel picks, it is this one)
    auto sval = [&](unsigned word, int S) -> int { return (int)(((word >> (8 * (2 * S + hi))) & 0xffu) * 0x01010101u); };
    auto mfmas = [&](auto S_) {
        constexpr int S = decltype(S_)::value;
#ifdef VB_GEMM_LAB
        if (p.flags & 2) return;
#endif
        int va[2], vb[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) va[i] = sval(sca[i], S);
#pragma unroll
        for (int j = 0; j < 2; ++j) vb[j] = sval(scb[j], S);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                // formats 0 / 0 = e4m3 x e4m3. Transposed product: first operand = W fragment (its rows -> accumulator
                // registers), second = A fragment (its rows -> lane & 31); natural: the other way round
#ifdef VB_GEMM_LAB
                if (p.flags & 32)      // lab: the other operand order (timing only)
                    acc[i][j] = !TRANS ? __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(fb[S][j], fa[S][i], acc[i][j], 0, 0, 0, vb[j],
                                                                                        0, va[i])
                                       : __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(fa[S][i], fb[S][j], acc[i][j], 0, 0, 0, va[i],
                                                                                        0, vb[j]);
                else
#endif
                acc[i][j] = TRANS ? __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(fb[S][j], fa[S][i], acc[i][j], 0, 0, 0, vb[j],
                                                                                   0, va[i])
                                  : __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(fa[S][i], fb[S][j], acc[i][j], 0, 0, 0, va[i],
                                                                                   0, vb[j]);
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
        {   // first K tile of this output tile (landed before the previous barrier); read here and not under the last MFMAs
            // of the previous tile, so that no fragment register is live across the epilogue
            const char* s0 = smem + st * MX_STAGE;
            read_set(I0{}, s0);
            read_scales(s0, sca, scb);
        }
        // steady state: every K tile but the last one of this output tile prefetches its successor
        for (int kt = 0; kt + 1 < nk; ++kt) {
            const char* cur = smem + st * MX_STAGE;
            st = st == MX_S - 1 ? 0 : st + 1;
            const char* nxt = smem + st * MX_STAGE;
            read_set(I1{}, cur);
            __builtin_amdgcn_sched_barrier(0);   // the LDS reads go out first, the MFMAs cover their latency
            mfmas(I0{});
            __builtin_amdgcn_sched_barrier(0);   // (the MFMAs must not sink below the wait for the reads)
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // B_g: every read of stage g is complete
            read_set(I0{}, nxt);
            read_scales(nxt, nsa, nsb);
            __builtin_amdgcn_sched_barrier(0);
            mfmas(I1{});
#pragma unroll
            for (int i = 0; i < 2; ++i) { sca[i] = nsa[i]; scb[i] = nsb[i]; }
        }
        {
            const char* cur = smem + st * MX_STAGE;
            st = st == MX_S - 1 ? 0 : st + 1;
            read_set(I1{}, cur);
            __builtin_amdgcn_sched_barrier(0);
            mfmas(I0{});
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            mfmas(I1{});
        }
        int m0 = 0, n0 = 0;
        mx_origin(p, b, it, grid, m0, n0);
        const int nw = n0 + wn * 64;
#ifdef VB_GEMM_LAB
        if (p.flags & 16) continue;
        const bool lab_store = !(p.flags & 8);
#else
        constexpr bool lab_store = true;
#endif
        if (KIND != MX_OUT_MX) {
            // ---- fp32 / bf16 output on the transposed map: lane (l31, hi) holds row m of tile (i, j) and its columns
            // nw + 32 j + 8 q + 4 hi + e in acc[i][j][4 q + e]
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int m = m0 + wm * 64 + 32 * i + l31;
                const bool live = m < p.M && lab_store;
                const long mr = m < p.M ? m : p.M - 1;
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int nb = nw + 32 * j;
                    f32x4 rv[4];
                    if (RES == 1) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) rv[q] = *reinterpret_cast<const f32x4*>(p.R + mr * p.ldr + nb + 8 * q + 4 * hi);
                    } else if (RES == 2) {
#pragma unroll
                        for (int kq = 0; kq < 4; kq += 2) {
                            // one 16-byte load per two 8-column groups (lanes 0-31: group kq, lanes 32-63: group kq + 1), then the
                            // half-waves trade halves so that each lane holds ITS 4 columns of both groups
                            const uint4 t = *reinterpret_cast<const uint4*>(p.R16 + mr * p.ldr16 + nb + 8 * (kq + hi));
                            const auto sx = __builtin_amdgcn_permlane32_swap(t.x, t.z, false, false);
                            const auto sy = __builtin_amdgcn_permlane32_swap(t.y, t.w, false, false);
                            rv[kq] = f32x4{__uint_as_float(sx[0] << 16), __uint_as_float(sx[0] & 0xffff0000u),
                                           __uint_as_float(sy[0] << 16), __uint_as_float(sy[0] & 0xffff0000u)};
                            rv[kq + 1] = f32x4{__uint_as_float(sx[1] << 16), __uint_as_float(sx[1] & 0xffff0000u),
                                               __uint_as_float(sy[1] << 16), __uint_as_float(sy[1] & 0xffff0000u)};
                        }
                    }
                    uint2 o2[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
                        if (p.bias != nullptr) v = *reinterpret_cast<const f32x4*>(p.bias + nb + 8 * q + 4 * hi);
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            v[e] += acc[i][j][4 * q + e];
                            if (GELU) v[e] = mx_gelu(v[e]);
                        }
                        if (HAS_R) v += rv[q];
                        if (KIND == MX_OUT_F32) {
                            if (live) *reinterpret_cast<f32x4*>(p.C + (long)m * p.ldc + nb + 8 * q + 4 * hi) = v;
                        } else {
                            o2[q] = uint2{(unsigned)bf16_rne(v[0]) | ((unsigned)bf16_rne(v[1]) << 16),
                                          (unsigned)bf16_rne(v[2]) | ((unsigned)bf16_rne(v[3]) << 16)};
                        }
                    }
                    if (KIND == MX_OUT_BF16) {
#pragma unroll
                        for (int kq = 0; kq < 4; kq += 2) {
                            const auto sx = __builtin_amdgcn_permlane32_swap(o2[kq].x, o2[kq + 1].x, false, false);
                            const auto sy = __builtin_amdgcn_permlane32_swap(o2[kq].y, o2[kq + 1].y, false, false);
                            if (live) *reinterpret_cast<uint4*>(p.Cb + (long)m * p.ldb16 + nb + 8 * (kq + hi)) = uint4{sx[0], sy[0], sx[1], sy[1]};
                        }
                    }
                }
            }
            continue;
        }
        // ---- transposed map (MX output): lane (l31, hi) holds row m of tile (i, j) and its columns 8 q + 4 hi + e, e = 0..3,
        // in acc[i][j][4 q + e]
        f32x4 bv[2][4];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                bv[j][q] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (p.bias != nullptr) bv[j][q] = *reinterpret_cast<const f32x4*>(p.bias + nw + 32 * j + 8 * q + 4 * hi);
            }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int m = m0 + wm * 64 + 32 * i + l31;
            const bool live = m < p.M;
            const long mr = live ? m : p.M - 1;
            v4i piece[2];        // per 32-column block j: this lane's 16 codes (columns 16 hi .. 16 hi + 15 of the block)
            unsigned bytes = 0;  // scale bytes of blocks j = 0, 1
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int nb = nw + 32 * j;
                f32x4 v[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[q][e] = acc[i][j][4 * q + e] + bv[j][q][e];
                }
                if (GELU) {
#pragma unroll
                    for (int q = 0; q < 4; ++q)
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[q][e] = mx_gelu(v[q][e]);
                }
                if (RES == 1) {
                    f32x4 rv[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) rv[q] = *reinterpret_cast<const f32x4*>(p.R + mr * p.ldr + nb + 8 * q + 4 * hi);
#pragma unroll
                    for (int q = 0; q < 4; ++q) v[q] += rv[q];
                } else if (RES == 2) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const uint2 w = *reinterpret_cast<const uint2*>(p.R16 + mr * p.ldr16 + nb + 8 * q + 4 * hi);
                        v[q] += f32x4{__uint_as_float(w.x << 16), __uint_as_float(w.x & 0xffff0000u), __uint_as_float(w.y << 16),
                                      __uint_as_float(w.y & 0xffff0000u)};
                    }
                }
                // the row's 32-column block = this lane's 16 values + lane ^ 32's 16
                float amax = fmaxf(fmaxf(amax4(v[0]), amax4(v[1])), fmaxf(amax4(v[2]), amax4(v[3])));
                amax = fmaxf(amax, __shfl_xor(amax, 32));
                const unsigned byte = mx_scale_byte(amax);
                const float inv = mx_inv_scale(byte);
                bytes |= byte << (8 * j);
                unsigned d[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) d[q] = mx_pack4(v[q], inv);
                // lane hi = 0 holds columns {0-3, 8-11, 16-19, 24-27}, its partner {4-7, 12-15, 20-23, 28-31}: after the
                // exchange hi = 0 owns columns 0-15 and hi = 1 columns 16-31 of the block
                const unsigned x = (unsigned)__shfl_xor((int)(hi ? d[0] : d[2]), 32);
                const unsigned y = (unsigned)__shfl_xor((int)(hi ? d[1] : d[3]), 32);
                piece[j] = hi ? v4i{(int)x, (int)d[2], (int)y, (int)d[3]} : v4i{(int)d[0], (int)x, (int)d[1], (int)y};
            }
            // Full 64-byte segments: rows of 16 lanes trade pieces (v_permlane16_swap: odd 16-lane rows of the first operand
            // <-> even rows of the second) so that the first store covers tile rows 0-15 (lanes 0-15 / 32-47: block 0, lanes
            // 16-31 / 48-63: block 1 of row lane & 15) and the second rows 16-31 - 4 lanes x 16 bytes per output row and
            // instruction instead of 2.
            v4i st0, st1;
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                const auto sw = __builtin_amdgcn_permlane16_swap((unsigned)piece[0][w], (unsigned)piece[1][w], false, false);
                st0[w] = (int)sw[0];
                st1[w] = (int)sw[1];
            }
            {
                const int blk = (l31 >> 4) & 1, rr = l31 & 15;
                const int mA = m0 + wm * 64 + 32 * i + rr, mB = mA + 16;
                if (mA < p.M && (lab_store || st0[0] == 0x12345678)) *reinterpret_cast<v4i*>(p.Cq + (long)mA * p.ldq + nw + 32 * blk + 16 * hi) = st0;
                if (mB < p.M && (lab_store || st1[0] == 0x12345678)) *reinterpret_cast<v4i*>(p.Cq + (long)mB * p.ldq + nw + 32 * blk + 16 * hi) = st1;
            }
            // scale bytes of (row m, blocks nw / 32 and nw / 32 + 1): bytes 2 wn, 2 wn + 1 of word (n0 / 128, m)
            if (live && hi == 0)
                reinterpret_cast<unsigned short*>(p.cs + (long)(n0 >> 7) * p.cs_rows + m)[wn] = (unsigned short)bytes;
        }
    }
}

template <int KIND, bool GELU, int RES>
int launch_mx(hipStream_t st, const MxP& p) {
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_mx_kernel<KIND, GELU, RES>),
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, MX_LDS);
    if (attr != hipSuccess) return (int)attr;
    const int grid = p.tiles < 256 ? p.tiles : 256;
    hipLaunchKernelGGL((gemm_mx_kernel<KIND, GELU, RES>), dim3(grid), dim3(MX_THREADS), MX_LDS, st, p);
    VB_LAUNCH_CHECK();
    return 0;
}

}  // namespace

namespace {
template <class IN>
int quantize_rows_mx(void* stream, int64_t rows, int32_t K, const IN* x, int64_t ldx, uint8_t* q, int64_t ldq, uint32_t* scales,
                     int64_t scale_rows) {
    if (x == nullptr || q == nullptr || scales == nullptr || rows <= 0 || K <= 0) return VB_E_BADARG;
    // (rows of 4 elements per lane: 16-byte loads of fp32, 8-byte loads of bfloat16)
    if (K % 128 != 0 || ldx % 4 != 0 || ldq % 4 != 0 || ldq < K || ldx < K || scale_rows < rows ||
        (reinterpret_cast<uintptr_t>(x) & (4 * sizeof(IN) - 1)) != 0 || (reinterpret_cast<uintptr_t>(q) & 3u) != 0 ||
        (reinterpret_cast<uintptr_t>(scales) & 3u) != 0)
        return VB_E_ALIGN;
    const dim3 grid((unsigned)((rows + 3) / 4)), block(256);
    hipStream_t st = static_cast<hipStream_t>(stream);
#define VB_QUANT(NV) hipLaunchKernelGGL((quant_rows_mx_kernel<NV, IN>), grid, block, 0, st, (long)rows, (int)K, x, (long)ldx, q, (long)ldq, scales, (long)scale_rows)
    switch (K) {
        case 768: VB_QUANT(3); break;
        case 1024: VB_QUANT(4); break;
        case 2048: VB_QUANT(8); break;
        case 3072: VB_QUANT(12); break;
        default: VB_QUANT(0); break;
    }
#undef VB_QUANT
    VB_LAUNCH_CHECK();
    return 0;
}
}  // namespace

extern "C" int vb_quantize_rows_mx(void* stream, int64_t rows, int32_t K, const float* x, int64_t ldx, uint8_t* q, int64_t ldq,
                                   uint32_t* scales, int64_t scale_rows) {
    return quantize_rows_mx<float>(stream, rows, K, x, ldx, q, ldq, scales, scale_rows);
}

extern "C" int vb_quantize_rows_mx_bf16(void* stream, int64_t rows, int32_t K, const uint16_t* x, int64_t ldx, uint8_t* q,
                                        int64_t ldq, uint32_t* scales, int64_t scale_rows) {
    return quantize_rows_mx<unsigned short>(stream, rows, K, x, ldx, q, ldq, scales, scale_rows);
}

extern "C" int vb_linear_fwd_mx(void* stream, const vb_linear_mx_args* a) {
    if (a == nullptr || a->A == nullptr || a->W == nullptr || a->a_scales == nullptr || a->w_scales == nullptr)
        return VB_E_BADARG;
    // exactly one output form per launch
    if ((a->C != nullptr) + (a->Cq != nullptr) + (a->Cb != nullptr) != 1) return VB_E_BADARG;
    if (a->Cq != nullptr && a->c_scales == nullptr) return VB_E_BADARG;
    if (a->M <= 0 || a->N <= 0 || a->K <= 0) return VB_E_BADARG;
    if (a->act != VB_ACT_NONE && a->act != VB_ACT_GELU) return VB_E_BADARG;
    if (a->K % MX_BK != 0 || a->N % MX_BN != 0 || a->lda % 16 != 0 || a->ldw % 16 != 0 || a->lda < a->K || a->ldw < a->K ||
        !vb_aligned16(a->A) || !vb_aligned16(a->W) || !vb_aligned16(a->a_scales) || !vb_aligned16(a->w_scales))
        return VB_E_ALIGN;
    // a tile's scale words are fetched as whole 1 KiB / 512-byte pieces: the scale planes must cover the padded tile rows
    if (a->a_srows < (a->M + MX_BM - 1) / MX_BM * MX_BM || a->w_srows < a->N || a->a_srows % 4 != 0 || a->w_srows % 4 != 0)
        return VB_E_RANGE;
    if (a->C != nullptr && (a->ldc % 4 != 0 || a->ldc < a->N || !vb_aligned16(a->C))) return VB_E_ALIGN;
    if (a->Cb != nullptr && (a->ldb16 % 8 != 0 || a->ldb16 < a->N || !vb_aligned16(a->Cb))) return VB_E_ALIGN;
    if (a->Cq != nullptr && (a->ldq % 16 != 0 || a->ldq < a->N || !vb_aligned16(a->Cq) || a->c_srows < a->M)) return VB_E_ALIGN;
    if (a->residual != nullptr && (a->ldr % 4 != 0 || a->ldr < a->N || !vb_aligned16(a->residual))) return VB_E_ALIGN;
    if (a->bias != nullptr && !vb_aligned16(a->bias)) return VB_E_ALIGN;
    if (a->residual != nullptr && a->ldr % 2 != 0) return VB_E_ALIGN;
    if (a->residual != nullptr && a->residual_bf16 != nullptr) return VB_E_BADARG;
    if (a->residual_bf16 != nullptr && (a->ldr16 % 8 != 0 || a->ldr16 < a->N || !vb_aligned16(a->residual_bf16)))
        return VB_E_ALIGN;
    if ((long)a->M * a->lda > 0xffffffffL && a->lda * 256 > 0xffffffffL) return VB_E_RANGE;
    MxP p{};
    p.M = (int)a->M; p.N = (int)a->N; p.K = (int)a->K;
    p.A = a->A; p.lda = a->lda; p.sa = a->a_scales; p.sa_rows = a->a_srows;
    p.B = a->W; p.ldb = a->ldw; p.sb = a->w_scales; p.sb_rows = a->w_srows;
    p.bias = a->bias; p.R = a->residual; p.ldr = a->ldr; p.R16 = a->residual_bf16; p.ldr16 = a->ldr16;
    p.C = a->C; p.ldc = a->ldc;
    p.Cq = a->Cq; p.ldq = a->ldq; p.cs = a->c_scales; p.cs_rows = a->c_srows;
    p.Cb = a->Cb; p.ldb16 = a->ldb16;
    p.act = a->act;
    p.tiles_n = p.N / MX_BN;
    p.tiles = ((p.M + MX_BM - 1) / MX_BM) * p.tiles_n;
#ifdef VB_GEMM_LAB
    static const int flags = [] { const char* e = getenv("VB_MX_FLAGS"); return e ? atoi(e) : 0; }();
    p.flags = flags;
#endif
    hipStream_t st = static_cast<hipStream_t>(stream);
    const bool gelu = a->act == VB_ACT_GELU;
    const int res = a->residual != nullptr ? 1 : a->residual_bf16 != nullptr ? 2 : 0;
    // the combinations the model uses (+ the plain ones); anything else is not built
    if (p.Cq != nullptr) {
        if (res == 2) return VB_E_BADARG;
        if (res == 1) return gelu ? VB_E_BADARG : launch_mx<MX_OUT_MX, false, 1>(st, p);
        return gelu ? launch_mx<MX_OUT_MX, true, 0>(st, p) : launch_mx<MX_OUT_MX, false, 0>(st, p);
    }
    if (p.Cb != nullptr) {
        if (gelu) return res == 0 ? launch_mx<MX_OUT_BF16, true, 0>(st, p) : VB_E_BADARG;
        return res == 2 ? launch_mx<MX_OUT_BF16, false, 2>(st, p) : res == 1 ? launch_mx<MX_OUT_BF16, false, 1>(st, p)
                                                                           : launch_mx<MX_OUT_BF16, false, 0>(st, p);
    }
    if (res == 2) return gelu ? VB_E_BADARG : launch_mx<MX_OUT_F32, false, 2>(st, p);
    if (res == 1) return gelu ? launch_mx<MX_OUT_F32, true, 1>(st, p) : launch_mx<MX_OUT_F32, false, 1>(st, p);
    return gelu ? launch_mx<MX_OUT_F32, true, 0>(st, p) : launch_mx<MX_OUT_F32, false, 0>(st, p);
}
