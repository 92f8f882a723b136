// bf16x6 / bf16x3 GEMM kernels: fp32 operands split into bf16 planes on their way into LDS, products on
// v_mfma_f32_32x32x16_bf16. Same tiling, tile plan, segments and epilogues as gemm.hip (shared through
// gemm_core.h); separate translation unit so the two kernel families compile in parallel.
#include "gemm_core.h"

#ifndef VB_NPL
#error "compile with -DVB_NPL=3 (bf16x6), -DVB_NPL=2 (bf16x3) or -DVB_NPL=1 (plain bf16 operands)"
#endif

namespace {

using namespace vbgemm;

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ unsigned cvt_pk_bf16(float lo, float hi) {
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}

// Exact split of 8 fp32 values (a lane's 8 consecutive k of one row) into NPL bf16x8 MFMA fragments:
// x = x0 + x1 (+ x2), x0 = bf16(x), x1 = bf16(x - x0), x2 = bf16(x - x0 - x1); each residual is exact.
template <int NPL>
__device__ __forceinline__ void split8(const f32x4 lo, const f32x4 hi, bf16x8 (&out)[NPL]) {
    float r[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
#pragma unroll
    for (int pl = 0; pl < NPL; ++pl) {
        unsigned w[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            w[q] = cvt_pk_bf16(r[2 * q], r[2 * q + 1]);
            if (pl + 1 < NPL) {
                r[2 * q] -= __uint_as_float(w[q] << 16);
                r[2 * q + 1] -= __uint_as_float(w[q] & 0xFFFF0000u);
            }
        }
        const uint4 packed = make_uint4(w[0], w[1], w[2], w[3]);
        out[pl] = *reinterpret_cast<const bf16x8*>(&packed);
    }
}

// ------------------------------------------------------------------------------------------------
// bf16x6 / bf16x3 tiles ("planes" path, NPL = 3 / 2 planes, the 6 / 3 largest partial products). The fp32 operands are split ONCE per block, on their
// way from the staging registers into LDS, into NPL bf16 planes (x = x0 + x1 (+ x2), exact residuals), so
// the main loop is ds_read_b128 + v_mfma_f32_32x32x16_bf16 only. (Splitting at fragment-read time - every
// element split by two waves - cost 8.8 VALU instructions per MFMA; the matrix pipe hides about 5 issue
// slots per 32-cycle MFMA, PMC: 53 % MFMA busy. Here it is ~3.7.)
// LDS per operand and stage: NPL planes x [128 rows][2 halves] 16-byte slots; slot (row, half) holds the 8
// consecutive k = 8 half .. 8 half + 7 of that row in bf16 = exactly one MFMA fragment. The two halves of a
// row are swapped when bit 2 xor bit 4 of the row index is set: ds_read_b128 is serviced in the lane groups
// {0-3,12-15,20-27}, {4-11,16-19,28-31} (+32) over 64 banks and ds_write_b128 in groups of 8 consecutive
// lanes over 32 banks (MI355X_MICROARCH.md, LDS); with this swap the 16 slots of every read group and the 8
// slots of every write group fall on distinct bank sets.
// ------------------------------------------------------------------------------------------------
constexpr int PL_PLANE = 128 * 32;   // bytes

__device__ __forceinline__ int pl_slot(int row, int half) {
    return (row * 2 + (half ^ (((row >> 4) ^ (row >> 2)) & 1))) * 16;
}

template <int NPL>
__device__ __forceinline__ void split_store8(char* __restrict__ oper, int row, int half, const f32x4 lo, const f32x4 hi) {
    bf16x8 pl[NPL];
    split8<NPL>(lo, hi, pl);
    char* d = oper + pl_slot(row, half);
#pragma unroll
    for (int q = 0; q < NPL; ++q) *reinterpret_cast<bf16x8*>(d + q * PL_PLANE) = pl[q];
}

// 4 consecutive k (k = kq .. kq + 3, kq % 4 == 0) of one row -> 8 bytes per plane
template <int NPL>
__device__ __forceinline__ void split_store4(char* __restrict__ oper, int row, int kq, const f32x4 v) {
    float r[4] = {v[0], v[1], v[2], v[3]};
    char* d = oper + pl_slot(row, kq >> 3) + (kq & 4) * 2;
#pragma unroll
    for (int pl = 0; pl < NPL; ++pl) {
        unsigned w[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            w[q] = cvt_pk_bf16(r[2 * q], r[2 * q + 1]);
            if (pl + 1 < NPL) {
                r[2 * q] -= __uint_as_float(w[q] << 16);
                r[2 * q + 1] -= __uint_as_float(w[q] & 0xFFFF0000u);
            }
        }
        *reinterpret_cast<uint2*>(d + pl * PL_PLANE) = make_uint2(w[0], w[1]);
    }
}

// FULL: the tile lies inside the matrix and K is a multiple of 16 - no bounds handling on row-contiguous operands.
template <int TM, int TN, bool A_KC, bool B_KC, bool VEC, int NPL, bool FULL>
__device__ __forceinline__ void gemm_tile_planes(const GemmP& p, char* __restrict__ smem, const int m0, const int n0) {
    constexpr int RA = 64 * TM, RB = 64 * TN;
    constexpr int NA = RA / 64, NB = RB / 64;
    constexpr int OPER_B = NPL * PL_PLANE, STAGE_B = 2 * OPER_B;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;

    const int kt_total = (p.K + BK - 1) / BK;
    const int kt_begin = blockIdx.y * p.ktiles_per_split;
    const int kt_end = min(kt_total, kt_begin + p.ktiles_per_split);
    if (kt_begin >= kt_end) return;

    // k-contiguous operands: thread owns rows (tid >> 2) + 64 it and k = 4 (tid & 3) .. + 3 (as in gemm_tile)
    const float* arow[NA];
    const float* brow[NB];
    const int kq = (tid & 3) * 4;
#pragma unroll
    for (int it = 0; it < NA; ++it) {
        const int r = (tid >> 2) + 64 * it;
        arow[it] = (A_KC && m0 + r < p.M) ? p.A + (long)(m0 + r) * p.lda : nullptr;
    }
#pragma unroll
    for (int it = 0; it < NB; ++it) {
        const int n = n0 + (tid >> 2) + 64 * it;
        brow[it] = nullptr;
        if (B_KC && n < p.N) {
            const int sg = n / p.bseg;
            brow[it] = p.B[sg] + (long)(n - sg * p.bseg) * p.ldb;
        }
    }
    // row-contiguous operands: thread owns ONE (row, half) unit = 8 consecutive k of row tid % R. `half` is
    // wave-uniform (R >= 64), so the k part of every address is scalar: the 8 loads of a unit use one
    // per-thread row offset and SGPR bases.
    const int ua_row = tid & (RA - 1), ub_row = tid & (RB - 1);
    const int ua_half = __builtin_amdgcn_readfirstlane(tid / RA), ub_half = __builtin_amdgcn_readfirstlane(tid / RB);
    // wave-uniform; compile-time true for 128-row tiles (256 threads = 128 rows x 2 halves): no branch in the K loop
    const bool ua_on = !A_KC && (RA == 128 || ua_half < 2), ub_on = !B_KC && (RB == 128 || ub_half < 2);
    const bool ua_ok = m0 + ua_row < p.M, ub_ok = n0 + ub_row < p.N;
    const unsigned ua_off = (ua_ok ? m0 + ua_row : 0) * 4u, ub_off = (ub_ok ? n0 + ub_row : 0) * 4u;   // bytes

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    f32x4 ra[2], rb[2];   // KC: NA / NB float4 (rows it);  RC: the unit's 8 k values (lo, hi)
    float csum = 0.f;
    const float* const bseg0 = p.B[0];
    const float* const bseg1 = p.B[1];
    const float* const bseg2 = p.B[2];
    const float* const bseg3 = p.B[3];
    int rc_sg = B_KC ? 0 : (kt_begin * BK) / p.bseg, rc_krel = B_KC ? 0 : kt_begin * BK - rc_sg * p.bseg;

    // 8 dword loads, lanes = consecutive rows (coalesced). k is wave-uniform, so each load is
    // `global_load_dword v, v_byte_offset, s[base of row k]` - no per-lane address arithmetic. Addresses are
    // clamped into range and out-of-range values are zeroed at STORE time (a select here would wait for
    // the load before the MFMAs of this K tile).
    auto load_rc = [&](f32x4 (&reg)[2], const float* __restrict__ base, long ld, unsigned offb, int kbase, int K) {
        const char* __restrict__ krow = reinterpret_cast<const char*>(base + (long)(FULL ? kbase : min(kbase, K - 1)) * ld);
        const long step = ld * 4;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            reg[e >> 2][e & 3] = *reinterpret_cast<const float*>(krow + offb);
            if (FULL || kbase + e + 1 < K) krow += step;
        }
    };
    auto mask_rc = [&](f32x4 (&reg)[2], bool ok, int kvalid) {
#pragma unroll
        for (int e = 0; e < 8; ++e)
            if (!(ok && e < kvalid)) reg[e >> 2][e & 3] = 0.f;
    };
    auto load_ab = [&](int kt, f32x4 (&ra)[2], f32x4 (&rb)[2]) {
        const int k0 = kt * BK;
        if (A_KC) {
            if (FULL && VEC) {   // branch-free: the compiler can then count vmcnt exactly (prefetch distance 2)
#pragma unroll
                for (int it = 0; it < NA; ++it) ra[it] = *reinterpret_cast<const f32x4*>(arow[it] + k0 + kq);
            } else {
                load_tile_kc<VEC, NA>(reinterpret_cast<f32x4(&)[NA]>(ra), arow, k0 + kq, p.K);
            }
        } else if (ua_on) {
            load_rc(ra, p.A, p.lda, ua_off, k0 + 8 * ua_half, p.K);
        }
        if (B_KC) {
            if (FULL && VEC) {
#pragma unroll
                for (int it = 0; it < NB; ++it) rb[it] = *reinterpret_cast<const f32x4*>(brow[it] + k0 + kq);
            } else {
                load_tile_kc<VEC, NB>(reinterpret_cast<f32x4(&)[NB]>(rb), brow, k0 + kq, p.K);
            }
        } else if (ub_on) {
            // segments stacked along K (dgrad through stacked weights; bseg is a multiple of BK). Tiles are
            // loaded in increasing order, so the (segment, offset) pair is advanced incrementally - no
            // division and no indexed kernel-argument load (s_load + lgkmcnt(0)) in the K loop.
            const float* bp = rc_sg == 0 ? bseg0 : rc_sg == 1 ? bseg1 : rc_sg == 2 ? bseg2 : bseg3;
            load_rc(rb, bp, p.ldb, ub_off, rc_krel + 8 * ub_half, min(p.bseg, p.K - rc_sg * p.bseg));
            rc_krel += BK;
            const bool wrap = rc_krel >= p.bseg;
            rc_krel = wrap ? 0 : rc_krel;
            rc_sg += wrap ? 1 : 0;
        }
    };
    auto store_ab = [&](char* __restrict__ stage, int kt, f32x4 (&ra)[2], f32x4 (&rb)[2]) {
        char* dA = stage;
        char* dB = stage + OPER_B;
        const int k0 = kt * BK;
        if (A_KC) {
#pragma unroll
            for (int it = 0; it < NA; ++it) split_store4<NPL>(dA, (tid >> 2) + 64 * it, kq, ra[it]);
        } else if (ua_on) {
            if (!FULL) mask_rc(ra, ua_ok, p.K - (k0 + 8 * ua_half));
            split_store8<NPL>(dA, ua_row, ua_half, ra[0], ra[1]);
        }
        if (B_KC) {
#pragma unroll
            for (int it = 0; it < NB; ++it) split_store4<NPL>(dB, (tid >> 2) + 64 * it, kq, rb[it]);
        } else if (ub_on) {
            if (!FULL) mask_rc(rb, ub_ok, p.K - (k0 + 8 * ub_half));   // K = total contraction length
            split_store8<NPL>(dB, ub_row, ub_half, rb[0], rb[1]);
        }
    };

    const bool want_colsum = ua_on && n0 == 0 && p.colsum[0] != nullptr;
    auto add_colsum = [&](const f32x4 (&reg)[2]) {   // after store_ab: out-of-range values are zero by then
        if (want_colsum)
#pragma unroll
            for (int e = 0; e < 8; ++e) csum += reg[e >> 2][e & 3];
    };

    constexpr int NPROD = NPL == 3 ? 6 : (NPL == 2 ? 3 : 1);   // NPL 1: operands rounded to bf16, one product
    constexpr int PA3[6] = {2, 0, 1, 1, 0, 0}, PB3[6] = {0, 2, 1, 0, 1, 0};
    constexpr int PA2[3] = {1, 0, 0}, PB2[3] = {0, 1, 0};

    // One K tile: [global loads of tile kt + 2 -> register set L] [fragments of tile kt from LDS] [MFMAs of
    // tile kt, with the split + LDS store of tile kt + 1 (register set S, loaded one step earlier) woven
    // between them] [barrier]. Prefetch distance 2 so that the split does not have to wait for memory, and
    // the sched_group_barrier pattern makes each wave cover its own VALU work with its own MFMAs
    // (1 MFMA = 32 cycles of matrix pipe = room for ~4 other issues) instead of relying on other waves.
    auto kstep = [&](int kt, f32x4 (&sa)[2], f32x4 (&sb)[2], f32x4 (&la)[2], f32x4 (&lb)[2], bool do_load, bool do_store) {
        const int cur = (kt - kt_begin) & 1;
        const char* sA = smem + cur * STAGE_B;
        const char* sB = sA + OPER_B;
        if (do_load) load_ab(kt + 2, la, lb);
        __builtin_amdgcn_sched_barrier(0);   // keep the prefetch at the top of the step

        bf16x8 ap[TM][NPL], bp[TN][NPL];
#pragma unroll
        for (int t = 0; t < TM; ++t) {
            const char* q = sA + pl_slot(wm * 32 * TM + t * 32 + l31, hi);
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl) ap[t][pl] = *reinterpret_cast<const bf16x8*>(q + pl * PL_PLANE);
        }
#pragma unroll
        for (int t = 0; t < TN; ++t) {
            const char* q = sB + pl_slot(wn * 32 * TN + t * 32 + l31, hi);
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl) bp[t][pl] = *reinterpret_cast<const bf16x8*>(q + pl * PL_PLANE);
        }
        // partial products, smallest magnitude first, product-major (consecutive MFMAs hit different
        // accumulators): NPL 3: a2b0 a0b2 a1b1 a1b0 a0b1 a0b0;  NPL 2: a1b0 a0b1 a0b0
#pragma unroll
        for (int pr = 0; pr < NPROD; ++pr) {
            const int pa = NPL == 3 ? PA3[pr] : (NPL == 2 ? PA2[pr % 3] : 0), pb = NPL == 3 ? PB3[pr] : (NPL == 2 ? PB2[pr % 3] : 0);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[i][pa % NPL], bp[j][pb % NPL],
                                                                       acc[i][j], 0, 0, 0);
        }
        if (do_store) {
            store_ab(smem + (cur ^ 1) * STAGE_B, kt + 1, sa, sb);
            add_colsum(sa);
        }
        if (TM == 2 && TN == 2) {
            // issue order: LDS reads, then {1 MFMA, 4 VALU} x 24 with the LDS writes in the second half
            __builtin_amdgcn_sched_group_barrier(0x100, 4 * NPL, 0);
#pragma unroll
            for (int g = 0; g < NPROD * 4; ++g) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
                if (g >= NPROD * 2) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
            }
        }
        __syncthreads();
    };

    f32x4 ra2[2], rb2[2];
    load_ab(kt_begin, ra, rb);
    store_ab(smem, kt_begin, ra, rb);
    add_colsum(ra);
    if (kt_begin + 1 < kt_end) load_ab(kt_begin + 1, ra, rb);
    __syncthreads();

    int kt = kt_begin;
    for (; kt + 3 < kt_end; kt += 2) {
        kstep(kt, ra, rb, ra2, rb2, true, true);
        kstep(kt + 1, ra2, rb2, ra, rb, true, true);
    }
    for (; kt < kt_end; ++kt) {   // <= 3 tiles left: loads only while a tile kt + 2 exists
        const bool ld = kt + 2 < kt_end, stq = kt + 1 < kt_end;
        if (((kt - kt_begin) & 1) == 0) kstep(kt, ra, rb, ra2, rb2, ld, stq);
        else kstep(kt, ra2, rb2, ra, rb, ld, stq);
    }

    tile_epilogue<TM, TN, A_KC, B_KC>(p, acc, m0, n0, want_colsum, csum, ua_row, ua_half);   // two threads hold a row's sum
}

template <bool A_KC, bool B_KC, bool VEC, int NPL>
__global__ __launch_bounds__(256, 2) void gemm_planes_kernel(const GemmP p) {
    extern __shared__ __attribute__((aligned(16))) char smem_pl[];
    const int b = blockIdx.x;
    if (b < p.n_big) {
        const int t = xcd_swizzle(b, p.n_big);
        const int m0 = (t / p.tiles_n) * 128, n0 = (t % p.tiles_n) * 128;
        if (m0 + 128 <= p.M && n0 + 128 <= p.N && p.K % BK == 0) gemm_tile_planes<2, 2, A_KC, B_KC, VEC, NPL, true>(p, smem_pl, m0, n0);
        else gemm_tile_planes<2, 2, A_KC, B_KC, VEC, NPL, false>(p, smem_pl, m0, n0);
    } else {
        const int s = xcd_swizzle(b - p.n_big, p.n_small);
        const int t = p.n_big + (s >> 2);
        const int m0 = (t / p.tiles_n) * 128 + ((s >> 1) & 1) * 64;
        const int n0 = (t % p.tiles_n) * 128 + (s & 1) * 64;
        if (m0 >= p.M || n0 >= p.N) return;
        if (m0 + 64 <= p.M && n0 + 64 <= p.N && p.K % BK == 0) gemm_tile_planes<1, 1, A_KC, B_KC, VEC, NPL, true>(p, smem_pl, m0, n0);
        else gemm_tile_planes<1, 1, A_KC, B_KC, VEC, NPL, false>(p, smem_pl, m0, n0);
    }
}

template <bool A_KC, bool B_KC>
int launch_planes(hipStream_t st, const GemmP& p, bool vec, int splits) {
    dim3 grid(p.n_big + p.n_small, splits), block(256);
    const int lds = 2 * 2 * VB_NPL * PL_PLANE;
    if (vec) hipLaunchKernelGGL((gemm_planes_kernel<A_KC, B_KC, true, VB_NPL>), grid, block, lds, st, p);
    else hipLaunchKernelGGL((gemm_planes_kernel<A_KC, B_KC, false, VB_NPL>), grid, block, lds, st, p);
    VB_LAUNCH_CHECK();
    return 0;
}

}  // namespace

// one object per plane count (compiled twice, -DVB_NPL=3 and -DVB_NPL=2, so the two builds run in parallel)
#define VB_CAT2(a, b) a##b
#define VB_CAT(a, b) VB_CAT2(a, b)
int vbgemm::VB_CAT(launch_gemm_planes, VB_NPL)(hipStream_t st, const GemmP& p, bool vec, int splits, bool a_kc, bool b_kc) {
    if (a_kc && b_kc) return launch_planes<true, true>(st, p, vec, splits);
    if (a_kc && !b_kc) return launch_planes<true, false>(st, p, vec, splits);
    if (!a_kc && !b_kc) return launch_planes<false, false>(st, p, vec, splits);
    return VB_E_BADARG;
}
