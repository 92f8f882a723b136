// fp32 GEMM emulated on the bf16 matrix cores ("split" mode): every fp32 operand element is split
// exactly into NPL bf16 planes, x = x0 + x1 (+ x2) with x0 = bf16(x), x1 = bf16(x - x0), x2 = bf16(x - x0 - x1)
// (each residual is exact in fp32), and the product a*b is accumulated in fp32 from the partial products
//   NPL = 3:  a0b0 + a0b1 + a1b0 + a1b1 + a0b2 + a2b0      (dropped terms <= 2^-24 |ab|: fp32-class result)
//   NPL = 2:  a0b0 + a0b1 + a1b0                            (dropped terms <= 2^-16 |ab|)
// on v_mfma_f32_32x32x16_bf16 (16x the rate of v_mfma_f32_32x32x2_f32, so 6 products still give 2.7x the
// fp32-MFMA throughput ceiling: 2500 / 6 = 417 TFLOP/s effective). Same contract, tile plan, operand
// layouts (NT / NN / TN), weight segments and epilogues as gemm.hip (shared through gemm_core.h).
//
// Staging: threads 0-127 stage the A tile, 128-255 the B tile: 16 fp32 values each per K step of 16,
// loaded as float4 (k-contiguous operand: one row x 4 k; row-contiguous operand: a 4k x 4rows block that is
// transposed in registers), split with v_cvt_pk_bf16_f32 and written to LDS as 8-byte (4 bf16) pieces.
// LDS: per operand NPL planes of [rows][16 + 8 pad] bf16 (48-byte rows = 3 slots, odd -> the 16-lane
// ds_read_b128 groups are conflict free); double buffered, 72 KiB (NPL 3) / 48 KiB (NPL 2).
// MFMA fragments: lane l reads the 8 consecutive k = 8 (l >> 5) .. +7 of row l & 31 (one ds_read_b128 per
// plane); the accumulator layout equals the fp32 kernel's, so the epilogues are shared.
#include "gemm_core.h"

namespace {

using namespace vbgemm;

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int SROW = 24;                       // bf16 elements per LDS row (16 data + 8 pad)
constexpr int PLANE_ELEMS = 128 * SROW;        // one plane of a 128-row operand tile

template <int NPL> struct SGeo {
    static constexpr int OPER_ELEMS = NPL * PLANE_ELEMS;
    static constexpr int STAGE_ELEMS = 2 * OPER_ELEMS;
    static constexpr int LDS_BYTES = 2 * STAGE_ELEMS * 2;   // 73,728 (NPL 3) / 49,152 (NPL 2)
};

__device__ __forceinline__ unsigned cvt_pk_bf16(float lo, float hi) {
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}

// Split four fp32 values into NPL planes of four bf16 each (one 8-byte LDS piece per plane).
template <int NPL>
__device__ __forceinline__ void split4(const f32x4 v, uint2 (&out)[NPL]) {
    float r0 = v[0], r1 = v[1], r2 = v[2], r3 = v[3];
#pragma unroll
    for (int pl = 0; pl < NPL; ++pl) {
        const unsigned a = cvt_pk_bf16(r0, r1), b = cvt_pk_bf16(r2, r3);
        out[pl] = make_uint2(a, b);
        if (pl + 1 < NPL) {
            r0 -= __uint_as_float(a << 16);
            r1 -= __uint_as_float(a & 0xFFFF0000u);
            r2 -= __uint_as_float(b << 16);
            r3 -= __uint_as_float(b & 0xFFFF0000u);
        }
    }
}

// One (64 TM) x (64 TN) output tile at (m0, n0).
template <int TM, int TN, bool A_KC, bool B_KC, bool VEC, int NPL>
__device__ __forceinline__ void gemm_tile_split(const GemmP& p, unsigned short* __restrict__ smem, const int m0,
                                                const int n0) {
    using G = SGeo<NPL>;
    constexpr int RA = 64 * TM, RB = 64 * TN;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    const bool is_b = tid >= 128;           // staging role: A tile (threads 0-127) or B tile (128-255)
    const int ht = tid & 127;

    const int kt_total = (p.K + BK - 1) / BK;
    const int kt_begin = blockIdx.y * p.ktiles_per_split;
    const int kt_end = min(kt_total, kt_begin + p.ktiles_per_split);
    if (kt_begin >= kt_end) return;

    // ---- staging set-up: every thread stages 4 float4 (16 values) of ITS operand per K step -------------
    const bool my_kc = is_b ? B_KC : A_KC;
    const int my_rows = is_b ? RB : RA;          // rows of my operand tile
    const int my_row0 = is_b ? n0 : m0;
    const int my_nrows = is_b ? p.N : p.M;
    const long my_ld = is_b ? p.ldb : p.lda;
    // k-contiguous: rows (ht >> 2) + 32 it, k offset 4 (ht & 3)
    const float* rowp[4] = {nullptr, nullptr, nullptr, nullptr};
    if (my_kc) {
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int r = (ht >> 2) + 32 * it;
            const int g = my_row0 + r;
            if (r < my_rows && g < my_nrows) {
                if (is_b) {
                    const int sg = g / p.bseg;
                    rowp[it] = p.B[sg] + (long)(g - sg * p.bseg) * my_ld;
                } else {
                    rowp[it] = p.A + (long)g * my_ld;
                }
            }
        }
    }
    // row-contiguous: a 4 (k) x 4 (rows) block: k group ht & 3, row group ht >> 2
    const int kg = ht & 3, rg = ht >> 2;
    const bool rc_active = !my_kc && rg * 4 < my_rows;

    float cs4[4] = {0.f, 0.f, 0.f, 0.f};  // fused bias gradient: column sums of my 4 rows (wgrad A tile)
    const bool want_colsum = !A_KC && !is_b && n0 == 0 && p.colsum[0] != nullptr;

    // Global loads of my 16 values of K step kt (zero beyond the matrix edges).
    auto load_stage = [&](int kt, f32x4 (&stg)[4]) {
        const int k0 = kt * BK;
#pragma unroll
        for (int it = 0; it < 4; ++it) stg[it] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (my_kc) {
            const int k = k0 + (ht & 3) * 4;
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                if (rowp[it] != nullptr) {
                    const float* g = rowp[it] + k;
                    if (VEC) {
                        if (k < p.K) stg[it] = *reinterpret_cast<const f32x4*>(g);
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (k + e < p.K) stg[it][e] = g[e];
                    }
                }
            }
        } else if (rc_active) {
            const float* base = is_b ? p.B[0] : p.A;
            int kk = k0, klim = p.K;
            if (is_b && k0 < p.K) {  // segments stacked along K (dgrad through stacked weights); bseg % 16 == 0
                const int sg = k0 / p.bseg;
                base = p.B[sg];
                kk = k0 - sg * p.bseg;
                klim = min(p.bseg, p.K - sg * p.bseg);
            }
            const int row = my_row0 + rg * 4;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int k = kk + kg * 4 + e;
                if (k < klim) {
                    const float* g = base + (long)k * my_ld + row;
                    if (VEC) {
                        if (row < my_nrows) stg[e] = *reinterpret_cast<const f32x4*>(g);
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (row + r < my_nrows) stg[e][r] = g[r];
                    }
                }
            }
        }
    };

    // Split + LDS write of piece q (0..3) of a staged K step: one float4 = 4 consecutive k of one row.
    auto store_piece = [&](const f32x4 (&stg)[4], int q, int buf, bool count) {
        unsigned short* sop = smem + buf * G::STAGE_ELEMS + (is_b ? G::OPER_ELEMS : 0);
        if (my_kc) {
            const int r = (ht >> 2) + 32 * q;
            if (r < my_rows) {
                uint2 pl[NPL];
                split4<NPL>(stg[q], pl);
#pragma unroll
                for (int z = 0; z < NPL; ++z)
                    *reinterpret_cast<uint2*>(sop + z * PLANE_ELEMS + r * SROW + (ht & 3) * 4) = pl[z];
            }
        } else if (rc_active) {
            const f32x4 col = {stg[0][q], stg[1][q], stg[2][q], stg[3][q]};  // 4 consecutive k of row 4 rg + q
            if (want_colsum && count) cs4[q] += (col[0] + col[1]) + (col[2] + col[3]);
            uint2 pl[NPL];
            split4<NPL>(col, pl);
#pragma unroll
            for (int z = 0; z < NPL; ++z)
                *reinterpret_cast<uint2*>(sop + z * PLANE_ELEMS + (rg * 4 + q) * SROW + kg * 4) = pl[z];
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    f32x16 acc2;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc2[r] = 0.f;

    // Software pipeline, two K steps deep: while the MFMAs of step kt run out of LDS buffer `cur`, the
    // values of step kt+1 (loaded one iteration earlier, so they have landed) are split and written to the
    // other buffer between the MFMA groups, and the global loads of step kt+2 are in flight.
    f32x4 sc[4], sn[4];
    load_stage(kt_begin, sc);
#pragma unroll
    for (int q = 0; q < 4; ++q) store_piece(sc, q, 0, true);
    load_stage(kt_begin + 1, sc);   // zero-filled past the matrix edge; never consumed past kt_end
    __syncthreads();

    for (int kt = kt_begin; kt < kt_end; ++kt) {
        const int cur = (kt - kt_begin) & 1;
        const unsigned short* sA = smem + cur * G::STAGE_ELEMS;
        const unsigned short* sB = sA + G::OPER_ELEMS;
        const bool more = kt + 1 < kt_end;
        load_stage(kt + 2, sn);

        bf16x8 af[TM][NPL], bf[TN][NPL];
#pragma unroll
        for (int t = 0; t < TM; ++t)
#pragma unroll
            for (int q = 0; q < NPL; ++q)
                af[t][q] = *reinterpret_cast<const bf16x8*>(sA + q * PLANE_ELEMS +
                                                            (wm * 32 * TM + t * 32 + l31) * SROW + hi * 8);
#pragma unroll
        for (int t = 0; t < TN; ++t)
#pragma unroll
            for (int q = 0; q < NPL; ++q)
                bf[t][q] = *reinterpret_cast<const bf16x8*>(sB + q * PLANE_ELEMS +
                                                            (wn * 32 * TN + t * 32 + l31) * SROW + hi * 8);
        // Partial products, smallest magnitude first. Product-major order: consecutive MFMAs write
        // different accumulators (a chain on one accumulator would run at the dependent-issue latency);
        // one staging piece (split + LDS write of step kt+1) rides behind each of the first four rounds.
        constexpr int NPROD = NPL == 3 ? 6 : 3;
        constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};   // NPL 3: a2b0 a0b2 a1b1 a1b0 a0b1 a0b0
        constexpr int QA[3] = {1, 0, 0}, QB[3] = {0, 1, 0};                    // NPL 2: a1b0 a0b1 a0b0
#pragma unroll
        for (int pr = 0; pr < NPROD; ++pr) {
            const int pa = NPL == 3 ? PA[pr] : QA[pr], pb = NPL == 3 ? PB[pr] : QB[pr];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    if (TM * TN == 1 && (pr & 1))   // single-tile waves alternate between two accumulators
                        acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][pa], bf[j][pb], acc2, 0, 0, 0);
                    else
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][pa], bf[j][pb], acc[i][j], 0, 0, 0);
                }
            constexpr int ROUNDS = NPROD < 4 ? NPROD : 4;
            if (pr < ROUNDS) {
#pragma unroll
                for (int q = pr * 4 / ROUNDS; q < (pr + 1) * 4 / ROUNDS; ++q) store_piece(sc, q, cur ^ 1, more);
            }
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 4; ++q) sc[q] = sn[q];
    }

    if (TM * TN == 1) acc[0][0] += acc2;

    if (want_colsum && rc_active) {
        // this thread summed 4 of the 16 k of every K step for its 4 rows
        const int cs = m0 / p.cseg;
        const int mloc = m0 - cs * p.cseg;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = rg * 4 + r;
            if (mloc + row < p.cseg && m0 + row < p.M) unsafeAtomicAdd(p.colsum[cs] + mloc + row, cs4[r]);
        }
    }
    tile_epilogue<TM, TN, A_KC, B_KC>(p, acc, m0, n0, false, 0.f);
}

template <bool A_KC, bool B_KC, bool VEC, int NPL>
__global__ __launch_bounds__(256, 2) void gemm_split_kernel(const GemmP p) {
    extern __shared__ __attribute__((aligned(16))) unsigned short smem_s[];
    const int b = blockIdx.x;
    if (b < p.n_big) {
        const int t = xcd_swizzle(b, p.n_big);
        gemm_tile_split<2, 2, A_KC, B_KC, VEC, NPL>(p, smem_s, (t / p.tiles_n) * 128, (t % p.tiles_n) * 128);
    } else {
        const int s = xcd_swizzle(b - p.n_big, p.n_small);
        const int t = p.n_big + (s >> 2);
        const int m0 = (t / p.tiles_n) * 128 + ((s >> 1) & 1) * 64;
        const int n0 = (t % p.tiles_n) * 128 + (s & 1) * 64;
        if (m0 >= p.M || n0 >= p.N) return;
        gemm_tile_split<1, 1, A_KC, B_KC, VEC, NPL>(p, smem_s, m0, n0);
    }
}

template <bool A_KC, bool B_KC, bool VEC, int NPL>
int launch_one(hipStream_t st, const GemmP& p, int splits) {
    auto k = gemm_split_kernel<A_KC, B_KC, VEC, NPL>;
    static bool attr_done = false;  // > 64 KiB of dynamic LDS needs the attribute once per kernel
    if (SGeo<NPL>::LDS_BYTES > 65536 && !attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           SGeo<NPL>::LDS_BYTES);
        if (e != hipSuccess) return (int)e;
        attr_done = true;
    }
    dim3 grid(p.n_big + p.n_small, splits), block(256);
    hipLaunchKernelGGL(k, grid, block, SGeo<NPL>::LDS_BYTES, st, p);
    VB_LAUNCH_CHECK();
    return 0;
}

template <bool A_KC, bool B_KC>
int launch_layout(hipStream_t st, const GemmP& p, bool vec, int splits, int planes) {
    if (planes == 2)
        return vec ? launch_one<A_KC, B_KC, true, 2>(st, p, splits) : launch_one<A_KC, B_KC, false, 2>(st, p, splits);
    return vec ? launch_one<A_KC, B_KC, true, 3>(st, p, splits) : launch_one<A_KC, B_KC, false, 3>(st, p, splits);
}

}  // namespace

int vbgemm::launch_gemm_split(hipStream_t st, const GemmP& p, int layout, bool vec, int splits, int planes) {
    switch (layout) {
        case 0: return launch_layout<true, true>(st, p, vec, splits, planes);
        case 1: return launch_layout<true, false>(st, p, vec, splits, planes);
        default: return launch_layout<false, false>(st, p, vec, splits, planes);
    }
}
