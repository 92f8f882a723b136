// fp32 GEMM family on v_mfma_f32_32x32x2_f32 (exact-fp32 matrix cores, 157.3 TFLOP/s peak on
// gfx950) with fused epilogues. One kernel template covers the three operand layouts the
// encoder needs:
// (opt-in "bf16x6" / "bf16x3": the same tiling with the operands split into bf16 planes on their way into
// LDS and multiplied on v_mfma_f32_32x32x16_bf16 - gemm_tile_planes below):
//   forward  (NT)  C[M,N] = A[M,K] . W[N,K]^T        A k-contiguous,  B k-contiguous
//   dgrad    (NN)  dX[M,K'] = dY[M,N'] . W[N',K']    A k-contiguous,  B j-contiguous
//   wgrad    (TN)  dW[N',K'] = dY[M,N']^T . X[M,K']  A i-contiguous,  B j-contiguous
//
// Block = 256 threads = 4 waves (one per SIMD) in a 2x2 arrangement. Two tile shapes live in the same
// kernel: BIG 128x128 (each wave 2x2 MFMA tiles of 32x32, 64 accumulator VGPRs) and SMALL 64x64 (each
// wave one 32x32 tile). The matrix pipe of a SIMD is the bottleneck resource, so what matters is the
// number of MFMA tiles queued per CU: with T big tiles over 256 CUs the last of ceil(T/256) rounds is
// mostly empty for the model's shapes (432 / 576 / 592 tiles at batch 256). The launch therefore runs
// floor(T/256) full rounds as big tiles and re-cuts the leftover big tiles into 4 small tiles each
// (when that shortens the tail), which the dispatcher spreads over all CUs ("hybrid tail").
//
// K step 16, LDS double buffered: 2 x (A + B) x 128 x 20 floats = 40 KiB -> 3-4 blocks per CU, so
// one block's barrier / prologue / epilogue is covered by the MFMAs of the others. Global -> register
// -> LDS staging: the loads for K tile t+1 are issued before the MFMAs of tile t and written to the
// other LDS buffer after them; one barrier per K tile.
//
// LDS layouts (floats):
//   k-contiguous operand: [rows][20]  (16 + 4 pad; 5 16-byte slots per row, odd): the 16-lane
//       ds_read_b128 groups fall on 16 distinct slots -> conflict free.
//   row-contiguous operand: [16 k][rows + 4]: ds_read_b32, lanes = consecutive rows -> conflict free.
// MFMA operand convention (32x32x2): lane l supplies A[i = l&31][k = l>>5], B[k = l>>5][j = l&31].
// The contraction order inside a K step is permuted (lane half `hi` owns k = 8c + 4hi + e) so that
// a k-contiguous operand is fetched with one ds_read_b128 per four MFMAs; A and B use the same
// permutation, which only reorders the fp32 summation.
//
// Workgroup -> tile map is XCD aware: block b runs on XCD b % 8, so XCD x gets a contiguous run
// of logical tiles (N fastest) and the blocks sharing an A panel share one L2.
#include <string.h>

#include "gemm_core.h"

namespace {

using namespace vbgemm;

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

// row-contiguous operand (global [k][ld], rows contiguous): R / 4 float4 per k row.
template <bool VEC, int R>
__device__ __forceinline__ void load_tile_rc(f32x4 (&reg)[R / 64], const float* __restrict__ base, long ld,
                                             int row0, int nrows, int k0, int K, int tid) {
#pragma unroll
    for (int it = 0; it < R / 64; ++it) {
        const int f = tid + 256 * it;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        const int k = k0 + f / (R / 4), row = row0 + (f % (R / 4)) * 4;
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

template <bool KC, int R>
__device__ __forceinline__ void store_tile(float* __restrict__ s, const f32x4 (&reg)[R / 64], int tid) {
#pragma unroll
    for (int it = 0; it < R / 64; ++it) {
        const int f = tid + 256 * it;
        const int off = KC ? (f >> 2) * KC_LD + (f & 3) * 4 : (f / (R / 4)) * (R + 4) + (f % (R / 4)) * 4;
        *reinterpret_cast<f32x4*>(s + off) = reg[it];
    }
}

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

// One (64 TM) x (64 TN) output tile at (m0, n0): exact fp32 products on v_mfma_f32_32x32x2_f32.
// FULL: the tile lies inside the matrix, K % 16 == 0 and 16-byte loads are legal - branch-free staging loads
// (one basic block per K tile: exact vmcnt waits, free scheduling).
template <int TM, int TN, bool A_KC, bool B_KC, bool VEC, bool FULL>
__device__ __forceinline__ void gemm_tile(const GemmP& p, float* __restrict__ smem, const int m0, const int n0) {
    constexpr int RA = 64 * TM, RB = 64 * TN;   // operand tile rows
    constexpr int NA = RA / 64, NB = RB / 64;   // float4 per thread per operand tile
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;

    const int kt_total = (p.K + BK - 1) / BK;
    const int kt_begin = blockIdx.y * p.ktiles_per_split;
    const int kt_end = min(kt_total, kt_begin + p.ktiles_per_split);
    if (kt_begin >= kt_end) return;

    // Row base pointers of the k-contiguous operands. B rows (= output columns) are mapped to their
    // weight segment here: the segments are stacked along N (q | k | v projections in one launch).
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

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    f32x4 ra[NA], rb[NB];
    float csum = 0.f;

    auto load_ab = [&](int kt) {
        const int k0 = kt * BK;
        if (FULL) {
            if (A_KC) {
#pragma unroll
                for (int it = 0; it < NA; ++it) ra[it] = *reinterpret_cast<const f32x4*>(arow[it] + k0 + kq);
            } else {
#pragma unroll
                for (int it = 0; it < NA; ++it) {
                    const int f = tid + 256 * it;
                    ra[it] = *reinterpret_cast<const f32x4*>(p.A + (long)(k0 + f / (RA / 4)) * p.lda + m0 + (f % (RA / 4)) * 4);
                }
            }
            if (B_KC) {
#pragma unroll
                for (int it = 0; it < NB; ++it) rb[it] = *reinterpret_cast<const f32x4*>(brow[it] + k0 + kq);
            } else {
                const int sg = k0 / p.bseg;
                const float* __restrict__ bb = p.B[sg] + (long)(k0 - sg * p.bseg) * p.ldb + n0;
#pragma unroll
                for (int it = 0; it < NB; ++it) {
                    const int f = tid + 256 * it;
                    rb[it] = *reinterpret_cast<const f32x4*>(bb + (long)(f / (RB / 4)) * p.ldb + (f % (RB / 4)) * 4);
                }
            }
            return;
        }
        if (A_KC) load_tile_kc<VEC, NA>(ra, arow, k0 + kq, p.K);
        else load_tile_rc<VEC, RA>(ra, p.A, p.lda, m0, p.M, k0, p.K, tid);
        if (B_KC) {
            load_tile_kc<VEC, NB>(rb, brow, k0 + kq, p.K);
        } else {
            // segments stacked along K (dgrad through stacked weights); bseg is a multiple of BK
            const int sg = k0 / p.bseg;
            load_tile_rc<VEC, RB>(rb, p.B[sg], p.ldb, n0, p.N, k0 - sg * p.bseg, min(p.bseg, p.K - sg * p.bseg), tid);
        }
    };

    load_ab(kt_begin);
    store_tile<A_KC, RA>(smem, ra, tid);
    store_tile<B_KC, RB>(smem + OPER_SZ, rb, tid);
    __syncthreads();

    const bool want_colsum = !A_KC && n0 == 0 && tid < RA && p.colsum[0] != nullptr;

    for (int kt = kt_begin; kt < kt_end; ++kt) {
        const int cur = (kt - kt_begin) & 1;
        const float* sA = smem + cur * STAGE_SZ;
        const float* sB = sA + OPER_SZ;
        const bool more = kt + 1 < kt_end;
        if (more) load_ab(kt + 1);

        if (p.flags & 1) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kc = 0; kc < BK / 8; ++kc) {
            f32x4 af[TM], bf[TN];
#pragma unroll
            for (int t = 0; t < TM; ++t) {
                if (A_KC) {
                    af[t] = *reinterpret_cast<const f32x4*>(
                        sA + (wm * 32 * TM + t * 32 + l31) * KC_LD + kc * 8 + hi * 4);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        af[t][e] = sA[(kc * 8 + hi * 4 + e) * (RA + 4) + wm * 32 * TM + t * 32 + l31];
                }
            }
#pragma unroll
            for (int t = 0; t < TN; ++t) {
                if (B_KC) {
                    bf[t] = *reinterpret_cast<const f32x4*>(
                        sB + (wn * 32 * TN + t * 32 + l31) * KC_LD + kc * 8 + hi * 4);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        bf[t][e] = sB[(kc * 8 + hi * 4 + e) * (RB + 4) + wn * 32 * TN + t * 32 + l31];
                }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][e], bf[j][e],
                                                                        acc[i][j], 0, 0, 0);
        }
        if (p.flags & 1) __builtin_amdgcn_s_setprio(0);

        if (want_colsum) {
            // bias gradient fused into wgrad: A = dY^T, so the sum over this K tile of row i = tid
#pragma unroll
            for (int kk = 0; kk < BK; ++kk) csum += sA[kk * (RA + 4) + tid];
        }

        if (more) {
            float* dA = smem + (cur ^ 1) * STAGE_SZ;
            store_tile<A_KC, RA>(dA, ra, tid);
            store_tile<B_KC, RB>(dA + OPER_SZ, rb, tid);
        }
        __syncthreads();
    }

    tile_epilogue<TM, TN, A_KC, B_KC>(p, acc, m0, n0, want_colsum, csum, tid);
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

    constexpr int NPROD = NPL == 3 ? 6 : 3;
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
            const int pa = NPL == 3 ? PA3[pr] : PA2[pr % 3], pb = NPL == 3 ? PB3[pr] : PB2[pr % 3];
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

    tile_epilogue<TM, TN, A_KC, B_KC>(p, acc, m0, n0, want_colsum, csum, ua_row);
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

template <bool A_KC, bool B_KC, bool VEC>
__global__ __launch_bounds__(256, 2) void gemm_f32_kernel(const GemmP p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int b = blockIdx.x;
    if (b < p.n_big) {
        const int t = xcd_swizzle(b, p.n_big);
        const int m0 = (t / p.tiles_n) * 128, n0 = (t % p.tiles_n) * 128;
        // measured: the branch-free variant gains 10-18 % on wgrad (TN) and 1-4 % on dgrad (NN) but loses
        // 2-14 % on the forward (NT) layout (also with the prefetch pinned to the top of the step), whose
        // predicated loads schedule better as they are
        if (!(A_KC && B_KC) && VEC && m0 + 128 <= p.M && n0 + 128 <= p.N && p.K % BK == 0)
            gemm_tile<2, 2, A_KC, B_KC, VEC, true>(p, smem, m0, n0);
        else gemm_tile<2, 2, A_KC, B_KC, VEC, false>(p, smem, m0, n0);
    } else {
        // leftover big tiles, re-cut into four 64x64 tiles each
        const int s = xcd_swizzle(b - p.n_big, p.n_small);
        const int t = p.n_big + (s >> 2);
        const int m0 = (t / p.tiles_n) * 128 + ((s >> 1) & 1) * 64;
        const int n0 = (t % p.tiles_n) * 128 + (s & 1) * 64;
        if (m0 >= p.M || n0 >= p.N) return;
        gemm_tile<1, 1, A_KC, B_KC, VEC, false>(p, smem, m0, n0);
    }
}

// GEMM arithmetic mode: 0 = exact fp32 MFMA, 3 = bf16x6, 2 = bf16x3 (number of bf16 operand planes).
int g_gemm_mode = -1;

int gemm_mode() {
    if (g_gemm_mode < 0) {
        const char* e = getenv("VB_GEMM_MODE");
        g_gemm_mode = 0;
        if (e != nullptr && !strcmp(e, "bf16x6")) g_gemm_mode = 3;
        if (e != nullptr && !strcmp(e, "bf16x3")) g_gemm_mode = 2;
    }
    return g_gemm_mode;
}

template <bool A_KC, bool B_KC>
int launch_gemm(hipStream_t st, GemmP p, bool vec, int splits) {
    static const int flags = [] { const char* e = getenv("VB_GEMM_FLAGS"); return e ? atoi(e) : 0; }();
    p.flags = flags;
    plan_tiles(p, splits, gemm_mode() != 0);
    // VB_GEMM_MODE: "f32" (default) = exact fp32 MFMA; "bf16x6" / "bf16x3" = fp32 emulated on the bf16
    // matrix cores with 3 / 2 operand planes (gemm_split.hip)
    const int planes = gemm_mode();
    dim3 grid(p.n_big + p.n_small, splits), block(256);
    if (planes != 0) {
        // operands split into bf16 planes on their way into LDS (gemm_tile_planes)
        const int lds = 2 * 2 * planes * PL_PLANE;
        if (planes == 3) {
            if (vec) hipLaunchKernelGGL((gemm_planes_kernel<A_KC, B_KC, true, 3>), grid, block, lds, st, p);
            else hipLaunchKernelGGL((gemm_planes_kernel<A_KC, B_KC, false, 3>), grid, block, lds, st, p);
        } else {
            if (vec) hipLaunchKernelGGL((gemm_planes_kernel<A_KC, B_KC, true, 2>), grid, block, lds, st, p);
            else hipLaunchKernelGGL((gemm_planes_kernel<A_KC, B_KC, false, 2>), grid, block, lds, st, p);
        }
    } else {
        if (vec) hipLaunchKernelGGL((gemm_f32_kernel<A_KC, B_KC, true>), grid, block, GEMM_LDS_BYTES, st, p);
        else hipLaunchKernelGGL((gemm_f32_kernel<A_KC, B_KC, false>), grid, block, GEMM_LDS_BYTES, st, p);
    }
    VB_LAUNCH_CHECK();
    return 0;
}

}  // namespace

extern "C" int vb_set_gemm_mode(int planes) {
    const int prev = gemm_mode();
    if (planes == 0 || planes == 2 || planes == 3) g_gemm_mode = planes;
    return prev;
}

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
    p.C[0] = a->C; p.ldc = a->ldc; p.cseg = (p.M + 127) / 128 * 128;
    p.R = a->residual; p.ldr = a->ldr;
    p.P = a->preact; p.ldp = a->ldp;
    p.act = a->act; p.accumulate = 0;
    if (!(a->dropout_p >= 0.f && a->dropout_p < 1.f)) return VB_E_BADARG;
    if (a->dropout_p > 0.f && a->ldc != p.N) return VB_E_ALIGN;
    p.drop_p = a->dropout_p; p.drop_scale = 1.0f / (1.0f - a->dropout_p); p.seed = a->seed;
    if (a->dropout_p > 0.f)
        p.epi = (a->act == VB_ACT_NONE && a->preact == nullptr && a->residual != nullptr) ? EPI_RES_DROP : EPI_GENERIC;
    else if (a->act == VB_ACT_NONE && a->preact == nullptr) p.epi = a->residual != nullptr ? EPI_RES : EPI_STORE;
    else if (a->act == VB_ACT_GELU && a->residual == nullptr) p.epi = a->preact != nullptr ? EPI_PRE_GELU : EPI_GELU;
    else p.epi = EPI_GENERIC;
    p.ktiles_per_split = (p.K + BK - 1) / BK;
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
    const bool fused = a->nseg == 1 || (a->seg_n % BK) == 0;
    if (!fused && a->residual != nullptr) return VB_E_SEGMENT;
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
        p.C[0] = a->dX; p.ldc = a->ldx; p.cseg = (p.M + 127) / 128 * 128;
        p.act = VB_ACT_NONE;
        p.accumulate = (a->accumulate || l > 0) ? 1 : 0;
        p.R = a->residual; p.ldr = a->ldr;
        if (p.R != nullptr) p.epi = p.accumulate ? EPI_GENERIC : EPI_RES;
        else p.epi = p.accumulate ? EPI_ACCUM : EPI_STORE;
        p.ktiles_per_split = (p.K + BK - 1) / BK;
        if (int e = launch_gemm<true, false>(st, p, vec, 1)) return e;
    }
    return 0;
}

// dW_s[seg_n,K] (+)= dY[:, s]^T . X[M,K] and dbias_s[seg_n] (+)= column sums of dY[:, s] - nn.Linear
// backward w.r.t. weight and bias of the stacked segments. The contraction runs over the M rows: split
// over gridDim.y with fp32 atomics. Segments whose size is a multiple of 128 share one launch.
extern "C" int vb_linear_bwd_weight(void* stream, const vb_linear_bwd_weight_args* a) {
    if (a == nullptr || a->dY == nullptr || a->X == nullptr) return VB_E_BADARG;
    if (a->M <= 0 || a->K <= 0 || a->seg_n <= 0) return VB_E_BADARG;
    if (a->nseg < 1 || a->nseg > VB_MAX_SEGMENTS) return VB_E_SEGMENT;
    hipStream_t st = static_cast<hipStream_t>(stream);
    for (int s = 0; s < a->nseg; ++s) {
        if (a->dW[s] == nullptr) return VB_E_SEGMENT;
        if (!a->accumulate) {
            hipError_t e = hipMemset2DAsync(a->dW[s], a->ldw * sizeof(float), 0, (size_t)a->K * sizeof(float),
                                            a->seg_n, st);
            if (e != hipSuccess) return (int)e;
            if (a->dbias[s] != nullptr) {
                e = hipMemsetAsync(a->dbias[s], 0, (size_t)a->seg_n * sizeof(float), st);
                if (e != hipSuccess) return (int)e;
            }
        }
    }
    bool same_bias = true;  // the fused launch needs bias gradients for all segments or for none
    for (int s = 1; s < a->nseg; ++s) same_bias = same_bias && ((a->dbias[s] != nullptr) == (a->dbias[0] != nullptr));
    const bool fused = a->nseg == 1 || ((a->seg_n % 128) == 0 && same_bias);
    const int launches = fused ? 1 : a->nseg;
    const int segs = fused ? a->nseg : 1;
    for (int l = 0; l < launches; ++l) {
        GemmP p{};
        p.M = segs * a->seg_n; p.N = a->K; p.K = a->M;
        p.A = a->dY + (long)l * a->seg_n; p.lda = a->ldy;
        p.B[0] = a->X; p.ldb = a->ldx; p.bseg = a->M;
        p.ldc = a->ldw;
        p.cseg = segs == 1 ? (a->seg_n + 127) / 128 * 128 : a->seg_n;
        for (int s = 0; s < segs; ++s) {
            p.C[s] = a->dW[l + s];
            p.colsum[s] = a->dbias[l + s];
        }
        p.act = VB_ACT_NONE; p.accumulate = 1;
        const int tiles = ((p.M + 127) / 128) * ((p.N + 127) / 128);
        const int kt_total = (p.K + BK - 1) / BK;
        // Split count: tiles x splits workgroups should fill r whole "one block per CU" rounds of the 256
        // CUs (all co-resident, so r = blocks per CU) WITHOUT spilling into a partial extra round.
        // Measured: r = 4 beats fewer, longer blocks (one block per CU leaves the matrix pipe idle during
        // every barrier / epilogue); take the largest r <= 4 that fills >= 93 % of its slots.
        int splits = 1;
        {
            double best = -1.0;
            static const int rmax = [] { const char* e = getenv("VB_WGRAD_RMAX"); return e ? atoi(e) : 4; }();
            for (int r = rmax; r >= 2; --r) {
                int s = (256 * r) / tiles;
                if (s < 1) s = 1;
                if (s > kt_total / 4) s = kt_total / 4 > 0 ? kt_total / 4 : 1;  // >= 4 K tiles per block
                const int blocks = tiles * s;
                const double fill = (double)blocks / (256.0 * ((blocks + 255) / 256));
                if (fill > best + 1e-9) { best = fill; splits = s; }
                if (fill >= 0.93) break;
            }
        }
        p.ktiles_per_split = (kt_total + splits - 1) / splits;
        splits = (kt_total + p.ktiles_per_split - 1) / p.ktiles_per_split;
        const bool vec = (a->seg_n % 4 == 0) && (a->K % 4 == 0) && (a->ldy % 4 == 0) && (a->ldx % 4 == 0) &&
                         vb_aligned16(p.A) && vb_aligned16(a->X);
        p.epi = splits > 1 ? EPI_ATOMIC : EPI_ACCUM;
        if (int e = launch_gemm<false, false>(st, p, vec, splits)) return e;
    }
    return 0;
}
