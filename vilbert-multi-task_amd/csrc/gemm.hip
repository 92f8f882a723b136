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

#include <mutex>

#include "gemm_v2.h"

namespace {

using namespace vbgemm;

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

unsigned long long* g_dbg = nullptr;   // lab only: where block 128 of the next v2 launches stores its cycle span

// GEMM arithmetic mode: 0 = exact fp32 MFMA, 3 = bf16x6, 2 = bf16x3, 1 = plain bf16 (number of bf16 operand planes).
int g_gemm_mode = -1;

int gemm_mode() {
    if (g_gemm_mode < 0) {
        const char* e = getenv("VB_GEMM_MODE");
        g_gemm_mode = 0;
        if (e != nullptr && !strcmp(e, "bf16x6")) g_gemm_mode = 3;
        if (e != nullptr && !strcmp(e, "bf16x3")) g_gemm_mode = 2;
        if (e != nullptr && !strcmp(e, "bf16")) g_gemm_mode = 1;
    }
    return g_gemm_mode;
}

// Tile selection of the second-generation kernel: 0 = cost model, 10 TM + TN = force a menu entry (22 | 33 | 34 | 43 |
// 44; launches it cannot serve fall back to the round-1 kernel), -1 = round-1 kernel only. Initial value from the
// environment (VB_GEMM_V2=0 -> -1, VB_GEMM_TILE=<code>).
int g_gemm_tile = -2;

int gemm_tile_code() {
    if (g_gemm_tile == -2) {
        const char* v2 = getenv("VB_GEMM_V2");
        const char* t = getenv("VB_GEMM_TILE");
        g_gemm_tile = (v2 != nullptr && atoi(v2) == 0) ? -1 : (t != nullptr ? atoi(t) : 0);
    }
    return g_gemm_tile;
}

// ---- second-generation kernel (gemm_v2.h): eligibility + tile / split plan --------------------------------------
// Cost model: a CU retires "16 x 16 tile K-steps" at a fixed rate once its matrix pipes are saturated, the blocks of
// a launch are dealt round-robin, so the launch takes ceil(blocks / 256) blocks of TM TN (K steps + overhead) tile
// steps on the busiest CU; eff = measured relative main-loop efficiency of the tile shape (tools/gemm_lab).
struct V2Plan { int tm1, tm2, tn, big_rows, small_rows, tiles_n, splits, kt_per_split; double cost; };

bool aligned_ld(const void* ptr, long ld) { return ptr == nullptr || (vb_aligned16(ptr) && ld % 4 == 0); }

// Modelled duration (arbitrary unit: one 16 x 16 tile K step on a saturated CU) of a launch of n1 tiles of area a1 (in
// 16 x 16 units) followed by n2 tiles of area a2, every block running `steps` K steps, `occ` blocks resident per CU.
// Blocks are dealt to the 256 CUs round-robin; a CU's matrix pipes are shared by its resident blocks and lose
// efficiency when fewer than 3 blocks cover each other's barriers / prologues / epilogues (occ_eff, measured).
double v2_launch_cost(long n1, int a1, long n2, int a2, double steps, int occ) {
    static const double occ_eff[5] = {1.0, 0.70, 0.90, 0.97, 1.0};
    double worst = 0.0;
    const long q1 = n1 / 256, r1 = n1 % 256, q2 = n2 / 256, r2 = n2 % 256;
    // the CU classes of a round-robin deal: (extra big tile?, extra small tile?)
    for (int cls = 0; cls < 4; ++cls) {
        const bool x1 = cls & 1, x2 = cls & 2;
        // CUs [0, r1) hold an extra big tile; the small tiles continue the deal at CU r1: CUs [r1, r1 + r2) mod 256
        long cnt;   // number of CUs in this class
        const long lo2 = r1, hi2 = r1 + r2;   // extra-small range, may wrap
        auto in2 = [&](long c) { return hi2 <= 256 ? (c >= lo2 && c < hi2) : (c >= lo2 || c < hi2 - 256); };
        cnt = 0;
        // count analytically would be fiddly; 256 iterations only when the class is otherwise plausible
        for (long c = 0; c < 256; ++c) cnt += ((c < r1) == x1) && (in2(c) == x2);
        if (cnt == 0) continue;
        const long b1 = q1 + (x1 ? 1 : 0), b2 = q2 + (x2 ? 1 : 0);
        long left1 = b1, left2 = b2;
        double t = 0.0;
        while (left1 + left2 > 0) {   // resident batches of up to occ blocks (big tiles first)
            const long take = left1 + left2 < occ ? left1 + left2 : occ;
            const long t1 = left1 < take ? left1 : take, t2 = take - t1;
            t += (double)(t1 * a1 + t2 * a2) * steps / occ_eff[take];
            left1 -= t1;
            left2 -= t2;
        }
        if (t > worst) worst = t;
    }
    return worst;
}

template <bool A_KC, bool B_KC>
bool plan_v2(const GemmP& p, bool vec, int splits, V2Plan& best) {
    const int code = gemm_tile_code();
    const bool enabled = code >= 0;
    // forced tile: 10 TM + TN (single height) or 100 TM1 + 10 TM2 + TN (mixed heights)
    const int forced_tm1 = code >= 100 ? code / 100 : code / 10, forced_tm2 = code >= 100 ? (code / 10) % 10 : code / 10;
    const int forced_tn = code % 10;
    if (!enabled || !vec || p.K % V2_BK != 0 || p.N % 4 != 0) return false;
    if (!A_KC && p.M % 4 != 0 && p.lda < (p.M + 3) / 4 * 4) return false;
    if (!B_KC && p.bseg % V2_BK != 0) return false;   // a K tile must not straddle two stacked weight segments
    for (int s = 0; s < VB_MAX_SEGMENTS; ++s)
        if (!aligned_ld(p.C[s], p.ldc) || !aligned_ld(p.bias[s], 4)) return false;
    if (!aligned_ld(p.R, p.ldr) || !aligned_ld(p.D, p.ldd) || !aligned_ld(p.mul, p.ldmul)) return false;
    if (B_KC && p.bseg % 4 != 0) return false;
    constexpr bool FWD = A_KC && B_KC, DGRAD = A_KC && !B_KC;
    const int e = p.epi;
    const bool epi_ok = e == EPI_STORE || (FWD && (e == EPI_GELU || e == EPI_DGELU || e == EPI_RES_DROP)) ||
                        ((FWD || DGRAD) && e == EPI_RES) || (DGRAD && (e == EPI_MUL || e == EPI_ACCUM)) ||
                        (!A_KC && (e == EPI_ATOMIC || e == EPI_ACCUM || splits != 1)) ||
                        (DGRAD && splits < 0 && e == EPI_ACCUM);   // split-K dgrad of a small output (vb_linear_bwd_input)
    if (!epi_ok) return false;
    const bool multi_seg = p.C[1] != nullptr;
    // plans are cached per problem shape (a training step launches the same ~30 shapes thousands of times)
    struct Key { int layout, M, N, K, cseg, splits, code; };
    struct Entry { Key k; bool ok; V2Plan pl; };
    static thread_local Entry cache[64];
    static thread_local int cache_n = 0;
    const Key key{(A_KC ? 2 : 0) + (B_KC ? 1 : 0), p.M, p.N, p.K, multi_seg ? p.cseg : 0, splits, code};
    for (int i = 0; i < cache_n; ++i)
        if (!memcmp(&cache[i].k, &key, sizeof(Key))) { best = cache[i].pl; return cache[i].ok; }

    // {tm1, tm2, tn}: single-height tiles and the mixed-height pairs compiled in gemm_v2.hip
    static const int menu[9][3] = {{4, 4, 4}, {3, 3, 4}, {4, 4, 3}, {3, 3, 3}, {2, 2, 2}, {4, 3, 4}, {4, 3, 3}, {3, 2, 4}, {3, 2, 3}};
    // relative main-loop efficiency of a tile shape (tools/gemm_lab, round 2): bigger tiles move fewer bytes per FLOP
    auto eff = [](int tm, int tn) { return tm * tn >= 16 ? 1.0 : tm * tn >= 12 ? 0.98 : tm * tn >= 9 ? 0.93 : 0.80; };
    const int kt_total = p.K / V2_BK;
    double best_cost = 1e300;
    for (int c = 0; c < 9; ++c) {
        const int tm1 = menu[c][0], tm2 = menu[c][1], tn = menu[c][2];
        if (code > 0 && (tm1 != forced_tm1 || tm2 != forced_tm2 || tn != forced_tn)) continue;
        if (tm1 != tm2 && splits != 1) continue;   // mixed heights: forward / dgrad only (wgrad tiles a weight matrix)
        if (multi_seg && (tm1 != tm2 || p.cseg % (32 * tm1) != 0)) continue;   // tiles must not straddle two C row segments
        const int bm1 = 32 * tm1, bm2 = 32 * tm2;
        const int tiles_n = (p.N + 32 * tn - 1) / (32 * tn);
        const int occ = (tm1 * tn <= 9 && FWD) ? 4 : 3;
        const int max_big = tm1 == tm2 ? 0 : p.M / bm1;
        for (int nb = 0; nb <= max_big; ++nb) {
            // nb row tiles of the taller kind (mixed launches only), the rest of the rows in bm2-row tiles
            const int rest = p.M - nb * bm1;
            const int ns = tm1 == tm2 ? (p.M + bm2 - 1) / bm2 : (rest + bm2 - 1) / bm2;
            if (tm1 != tm2 && (nb == 0 || ns == 0)) continue;
            const long n1 = (long)nb * tiles_n, n2 = (long)ns * tiles_n;
            const int smax = splits < 0 ? (kt_total / 4 > 0 ? (kt_total / 4 < 96 ? kt_total / 4 : 96) : 1) : 1;
            for (int sp = 1; sp <= smax; ++sp) {
                const int per = (kt_total + sp - 1) / sp;
                if ((kt_total + per - 1) / per != sp) continue;
                const double steps = per + (sp > 1 ? 3.5 : 2.0);
                const double cost = v2_launch_cost(n1 * sp, tm1 * tn, n2 * sp, tm2 * tn, steps, occ) / eff(tm2, tn);
                if (cost < best_cost - 1e-9) {
                    best_cost = cost;
                    best = {tm1, tm2, tn, nb, ns, tiles_n, sp, per, cost};
                }
            }
        }
    }
    const bool ok = best_cost < 1e299;
    if (cache_n < 64) cache[cache_n++] = Entry{key, ok, best};
    return ok;
}

// Persistent one-block-per-CU kernel (gemm_v4.h): -> tile width code (3 = 96, 4 = 128 columns) or 0 = not used. Called
// after plan_v2 accepted the launch (alignment, epilogue). Mode (vb_set_gemm_v4 / VB_GEMM_V4): 0 = never, 1 = wherever
// its tiles fill whole rounds of the 256 CUs (default), 2 = every eligible launch (tests, lab).
// Measured in one process on the product library (tools/gemm_lab_prod LAB_V4_AB=1, profiles/r03_gemm_lab_v4_ab*.txt):
// +2 ... +13 % on every forward / dgrad shape of the model at M = 9216 and 18432 (137-147 TF against 120-136 for the
// 4-wave blocks on the same box), bert_large shapes included.
int g_gemm_v4 = -1;

int gemm_v4_mode() {
    if (g_gemm_v4 < 0) {
        const char* e = getenv("VB_GEMM_V4");
        g_gemm_v4 = e != nullptr ? atoi(e) : 1;
        if (g_gemm_v4 < 0 || g_gemm_v4 > 2) g_gemm_v4 = 1;
    }
    return g_gemm_v4;
}

// Deterministic split-K (vb_set_deterministic): the splits of a launch store their partial products to a workspace
// registered by the caller and splitk_reduce_kernel adds them to C in split order - bit-identical results from run to
// run, where the default (fp32 atomics from all splits into C) depends on the order the blocks happen to finish in.
int g_det = -1;
// One workspace PER DEVICE (a process may drive several GPUs: nn.DataParallel replicas, the reference's non-distributed
// multi-GPU path train_concap.py:513-515): a launch only ever uses the workspace registered for the device it runs on, and
// falls back to the fp32-atomics split-K when there is none. Each workspace is cut into DET_SLICES equal slices; every
// (device, stream) that issues split launches gets its own (first come, first served, for the lifetime of that
// registration), so the text / image / weight-gradient side streams keep overlapping. A stream that comes after the
// slices are taken, or a launch whose partials do not fit a slice, runs with atomics (correct, just not bit-reproducible)
// and is counted (vb_deterministic_fallbacks) - it is never an error.
constexpr int DET_SLICES = 8;
constexpr int DET_MAX_DEV = 16;
struct DetDevice {
    float* ws = nullptr;
    size_t bytes = 0;
    hipStream_t streams[DET_SLICES];
    int nstreams = 0;
};
DetDevice g_det_dev[DET_MAX_DEV];
long g_det_fallbacks = 0;
std::mutex g_det_mutex;     // autograd runs backward nodes on its own threads

bool deterministic() {
    if (g_det < 0) g_det = 0;
    return g_det != 0;
}

// workspace slice of (current device, stream): base pointer + slice size, or nullptr when the device has no workspace or
// its slices are taken
float* det_slice_of(hipStream_t st, size_t* slice_bytes) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= DET_MAX_DEV) return nullptr;
    std::lock_guard<std::mutex> lock(g_det_mutex);
    DetDevice& d = g_det_dev[dev];
    if (d.ws == nullptr) return nullptr;
    const size_t slice = d.bytes / DET_SLICES / 16 * 16;
    *slice_bytes = slice;
    int k = -1;
    for (int i = 0; i < d.nstreams; ++i)
        if (d.streams[i] == st) { k = i; break; }
    if (k < 0) {
        if (d.nstreams == DET_SLICES) return nullptr;
        d.streams[d.nstreams] = st;
        k = d.nstreams++;
    }
    return d.ws + (size_t)k * (slice / sizeof(float));
}

void det_count_fallback() {
    std::lock_guard<std::mutex> lock(g_det_mutex);
    ++g_det_fallbacks;
}

__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ ws, const float* __restrict__ cs_ws,
                                                            int splits, int cs_slices, long stride, int M, int N, GemmP p) {
    // C[r][c] += sum over the splits (in order) of ws[s][r][c]; bias gradient: colsum[r] += sum of cs_ws[s][r]
    const long n4 = (long)M * (N / 4);
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n4) {
        const int r = (int)(i / (N / 4)), c = (int)(i % (N / 4)) * 4;
        f32x4 acc = *reinterpret_cast<const f32x4*>(ws + (long)r * N + c);
        for (int s = 1; s < splits; ++s) acc += *reinterpret_cast<const f32x4*>(ws + s * stride + (long)r * N + c);
        const int sg = r / p.cseg;
        float* dst = p.C[sg] + (long)(r - sg * p.cseg) * p.ldc + c;
        *reinterpret_cast<f32x4*>(dst) = *reinterpret_cast<const f32x4*>(dst) + acc;
    } else if (cs_ws != nullptr && i - n4 < M) {
        const int r = (int)(i - n4);
        float acc = cs_ws[r];
        for (int s = 1; s < cs_slices; ++s) acc += cs_ws[(long)s * M + r];
        const int sg = r / p.cseg;
        p.colsum[sg][r - sg * p.cseg] += acc;
    }
}

// Zero fill of a row-strided fp32 matrix AS A KERNEL (round 6). hipMemset2DAsync / hipMemsetAsync become MEMSET NODES when the
// step is captured into a HIP graph, and a memset node in front of the kernels that accumulate into the buffer was found not to
// take effect at replay on this runtime (tools/memset_node_repro.py; the fp8 chain-graph issue of round 5: the split-K input
// gradient of the MLM decoder then added its partial sums to whatever the block held from the previous replay - e4m3 codes read
// as fp32 are ~1e38 - and AdamW's second moments overflowed). A kernel node is ordered like every other kernel of the chain.
__global__ __launch_bounds__(256) void zero_rows_kernel(float* __restrict__ c, long ldc, int rows, int cols) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    const int c4 = (cols + 3) / 4;
    if (i >= (long)rows * c4) return;
    const int r = (int)(i / c4), q = (int)(i % c4) * 4;
    float* __restrict__ d = c + (long)r * ldc + q;
    if (q + 4 <= cols && ((reinterpret_cast<uintptr_t>(d) & 15u) == 0)) {
        *reinterpret_cast<f32x4*>(d) = f32x4{0.f, 0.f, 0.f, 0.f};
    } else {
        for (int e = 0; e < 4 && q + e < cols; ++e) d[e] = 0.f;
    }
}
int zero_rows(hipStream_t st, float* c, long ldc, int rows, int cols) {
    if (rows <= 0 || cols <= 0) return 0;
    const long work = (long)rows * ((cols + 3) / 4);
    hipLaunchKernelGGL(zero_rows_kernel, dim3((unsigned)((work + 255) / 256)), dim3(256), 0, st, c, ldc, rows, cols);
    VB_LAUNCH_CHECK();
    return 0;
}

// Points the launch at the workspace (-> the kernels store partials instead of adding atomically). false = no slice for
// this (device, stream) or the slice is too small for `splits` partial copies of C: the caller launches with atomics.
bool det_prepare(hipStream_t st, GemmP& p, int splits) {
    const int cs_parts = p.det_cs_parts < 1 ? 1 : p.det_cs_parts;
    const size_t need = ((size_t)splits * p.M * p.N + (size_t)splits * cs_parts * p.M) * sizeof(float);
    size_t slice = 0;
    float* base = det_slice_of(st, &slice);
    if (base == nullptr || need > slice) { det_count_fallback(); return false; }
    p.det_cs_parts = cs_parts;
    p.det_ws = base;
    p.det_stride = (long)p.M * p.N;
    p.det_cs = base + (size_t)splits * p.M * p.N;
    return true;
}

int det_finish(hipStream_t st, const GemmP& p, int splits) {
    const bool cs = p.colsum[0] != nullptr;
    const long work = (long)p.M * (p.N / 4) + (cs ? p.M : 0);
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((work + 255) / 256)), dim3(256), 0, st, p.det_ws, cs ? p.det_cs : nullptr,
                       splits, splits * p.det_cs_parts, p.det_stride, p.M, p.N, p);
    VB_LAUNCH_CHECK();
    return 0;
}

// ---- skinny weight gradient: dW[seg_n, K] += dY^T X for K <= 8 input features (the 5-wide region-location projection
// of the image embeddings, reference vilbert.py:385-386). As a GEMM this is a 1024 x 5 output reduced over ~9.5 k rows:
// the tiled kernels waste 27 of 32 columns and, without atomics, cannot split the reduction (482 us per step in
// profiles/r03_bench_train_b256_gemm_breakdown.txt). Here it is what it is - one streaming pass over dY: a block owns
// 256 columns x one slab of rows, a wave every fourth row of the slab (the X row is wave-uniform: scalar loads), a lane 4
// columns; the four waves are summed through LDS and the slabs through the deterministic workspace in slab order
// (or with fp32 atomics when no workspace is registered). Bias gradient = the same sum with x = 1.
constexpr int SK_KMAX = 8, SK_SLABS = 128;

__global__ __launch_bounds__(256) void wgrad_skinny_kernel(int M, int seg_n, int K, const float* __restrict__ dY, long ldy,
                                                           const float* __restrict__ X, long ldx, float* __restrict__ dW,
                                                           long ldw, float* __restrict__ dbias, float* __restrict__ ws,
                                                           int rows_per_slab) {
    __shared__ float red[3][64][4 * (SK_KMAX + 1)];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n0 = blockIdx.x * 256 + 4 * lane;
    const bool live = n0 < seg_n;                      // (seg_n % 4 == 0: a lane's 4 columns are all in or all out)
    const int slab = blockIdx.y;
    const int r0 = slab * rows_per_slab, r1 = min(M, r0 + rows_per_slab);
    f32x4 acc[SK_KMAX + 1];
#pragma unroll
    for (int k = 0; k <= SK_KMAX; ++k) acc[k] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 2
    for (int r = r0 + wave; r < r1; r += 4) {
        const f32x4 dy = live ? *reinterpret_cast<const f32x4*>(dY + (long)r * ldy + n0) : f32x4{0.f, 0.f, 0.f, 0.f};
        const float* __restrict__ xr = X + (long)r * ldx;
#pragma unroll
        for (int k = 0; k < SK_KMAX; ++k)
            if (k < K) acc[k] += dy * xr[k];
        acc[SK_KMAX] += dy;
    }
    if (wave > 0) {
#pragma unroll
        for (int k = 0; k <= SK_KMAX; ++k)
#pragma unroll
            for (int e = 0; e < 4; ++e) red[wave - 1][lane][4 * k + e] = acc[k][e];
    }
    __syncthreads();
    if (wave != 0 || !live) return;
#pragma unroll
    for (int k = 0; k <= SK_KMAX; ++k)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[k][e] += (red[0][lane][4 * k + e] + red[1][lane][4 * k + e]) + red[2][lane][4 * k + e];
    if (ws != nullptr) {
        // partials [slab][j = 0 .. K][seg_n], j = K: bias
        float* __restrict__ w = ws + (long)slab * (K + 1) * seg_n + n0;
#pragma unroll
        for (int k = 0; k < SK_KMAX; ++k)
            if (k < K) *reinterpret_cast<f32x4*>(w + (long)k * seg_n) = acc[k];
        *reinterpret_cast<f32x4*>(w + (long)K * seg_n) = acc[SK_KMAX];
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
#pragma unroll
            for (int k = 0; k < SK_KMAX; ++k)
                if (k < K) atomicAdd(dW + (long)(n0 + e) * ldw + k, acc[k][e]);
            if (dbias != nullptr) atomicAdd(dbias + n0 + e, acc[SK_KMAX][e]);
        }
    }
}

__global__ __launch_bounds__(256) void wgrad_skinny_reduce_kernel(int slabs, int seg_n, int K, const float* __restrict__ ws,
                                                                  float* __restrict__ dW, long ldw, float* __restrict__ dbias) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)(K + 1) * seg_n) return;
    const int j = (int)(i / seg_n), n = (int)(i % seg_n);
    float acc = 0.f;
    for (int sl = 0; sl < slabs; ++sl) acc += ws[((long)sl * (K + 1) + j) * seg_n + n];
    if (j < K) dW[(long)n * ldw + j] += acc;
    else if (dbias != nullptr) dbias[n] += acc;
}

// -> 0 launched, > 0 error, -1 not eligible (the caller takes the GEMM path)
int launch_wgrad_skinny(hipStream_t st, const vb_linear_bwd_weight_args* a) {
    if (a->nseg != 1 || a->K > SK_KMAX || a->M < 256 || a->seg_n % 4 != 0 || a->ldy % 4 != 0 || !vb_aligned16(a->dY)) return -1;
    float* ws = nullptr;
    const int slabs = SK_SLABS;
    if (deterministic()) {
        const size_t need = (size_t)slabs * (a->K + 1) * a->seg_n * sizeof(float);
        size_t slice = 0;
        ws = det_slice_of(st, &slice);
        if (ws != nullptr && need > slice) ws = nullptr;
        if (ws == nullptr) det_count_fallback();       // slabs added with fp32 atomics instead
    }
    const int rows_per_slab = (int)((a->M + slabs - 1) / slabs);
    hipLaunchKernelGGL(wgrad_skinny_kernel, dim3((unsigned)((a->seg_n + 255) / 256), (unsigned)slabs), dim3(256), 0, st, (int)a->M,
                       (int)a->seg_n, (int)a->K, a->dY, (long)a->ldy, a->X, (long)a->ldx, a->dW[0], (long)a->ldw, a->dbias[0], ws,
                       rows_per_slab);
    VB_LAUNCH_CHECK();
    if (ws != nullptr) {
        const long work = (long)(a->K + 1) * a->seg_n;
        hipLaunchKernelGGL(wgrad_skinny_reduce_kernel, dim3((unsigned)((work + 255) / 256)), dim3(256), 0, st, slabs, (int)a->seg_n,
                           (int)a->K, ws, a->dW[0], (long)a->ldw, a->dbias[0]);
        VB_LAUNCH_CHECK();
    }
    return 0;
}

// -> configuration code WM * 1000 + TM * 100 + TM2 * 10 + TN of the persistent kernel (gemm_v4.h, dispatch_v4 in
// gemm_v2.hip), 0 = not used. Round 4: a tile menu instead of the two 288-row shapes -
//   * the round-3 shapes 288 x 128 / 288 x 96 (12 MFMA waves);
//   * MIXED 320 | 256-row tiles on 8 MFMA waves when M = 320 a + 256 (32 - a): the 37-region image stream at batch 256
//     (M = 9472 = 20 x 320 + 12 x 256) becomes exactly 32 row tiles x N / 128 column tiles - one tile per CU and round;
//   * small-M shapes (per-GPU batch 64: M = 2304 / 2368 rows - 64 tiles of 288 rows would leave 192 CUs idle): 192 x 128,
//     96 x 128, 96 x 96 (12 waves), 128 x 64, 64 x 128 (8 waves).
// Choice by a TIME model fitted to in-process A/B runs of every configuration on the model's shapes
// (tools/lab_v4_menu.sh, profiles/r04_gemm_lab_v4_menu_*.txt): a persistent launch costs
//     9.4 us  +  rounds x K steps x (ideal matrix-pipe time of one tile K step) / 0.94  +  (rounds - 1) x 12 us
// (launch + prologue + epilogue are ~9.4 us whatever the tile; every configuration's K step runs at ~0.94 of the pipe;
// an output-tile boundary inside a launch is a store burst), with the mixed launch timed by its 320-row
// tiles; the 4-wave alternative costs 0.93 x (plan_v2's modelled cost, in 32 x 32-tile K steps of 53.4 ns). The persistent
// kernel is taken when its modelled time is lower (mode 1), always when eligible (mode 2).
// VB_GEMM_V4_CFG=<code> / vblab_set_gemm_v4_cfg force one configuration wherever the shape allows it (laboratory, tests).
struct V4Opt { int wm, tm, tn; };
int g_v4_force_cfg = -1;
int g_v4_last_cfg = 0;
int plan_v4(const GemmP& p, bool b_kc, double v2_cost) {
    const int mode = gemm_v4_mode();
    if (mode == 0) return 0;
    if (p.K % 32 != 0 || p.C[1] != nullptr || p.epi == EPI_ATOMIC || p.epi == EPI_GENERIC || p.epi == EPI_PRE_GELU) return 0;
    static const int force_tn = [] { const char* e = getenv("VB_GEMM_V4_TN"); return e ? atoi(e) : 0; }();
    if (g_v4_force_cfg < 0) { const char* e = getenv("VB_GEMM_V4_CFG"); g_v4_force_cfg = e ? atoi(e) : 0; }
    const int force_cfg = g_v4_force_cfg;
    static const double margin = [] { const char* e = getenv("VB_GEMM_V4_MARGIN"); return e ? atof(e) : 0.98; }();
    auto cols_ok = [&](int tn) {
        if (p.N % (32 * tn) != 0) return false;
        return !(b_kc && p.B[1] != nullptr && p.bseg % (32 * tn) != 0);   // a tile must not straddle two stacked weights
    };
    constexpr double CU_FLOPS = 157.3e12 / 256.0, T_FIXED = 9.4e-6, T_BOUNDARY = 12e-6, STEP_EFF = 0.94;
    const double nk = p.K / 16;
    auto model = [&](int bm, int bn, long tiles) {
        const double rounds = (double)((tiles + 255) / 256);
        return T_FIXED + rounds * nk * (2.0 * bm * bn * 16.0 / CU_FLOPS) / STEP_EFF + (rounds - 1.0) * T_BOUNDARY;
    };
    static const V4Opt menu[] = {{6, 3, 4}, {6, 3, 3}, {6, 2, 4}, {6, 1, 4}, {6, 1, 3}, {4, 2, 2}, {4, 1, 4}};
    // VB_GEMM_V4_MENU=0: the round-3 planner (288-row shapes only, >= 0.90 fill) for A/B runs
    static const bool menu_on = [] { const char* e = getenv("VB_GEMM_V4_MENU"); return e == nullptr || atoi(e) != 0; }();
    double best = 1e30;
    int best_cfg = 0;
    bool best_fills = false;
    for (const V4Opt& o : menu) {
        if (!cols_ok(o.tn) || (force_tn != 0 && force_tn != o.tn)) continue;
        const int cfg = o.wm * 1000 + o.tm * 100 + o.tn;
        if (force_cfg != 0 && force_cfg != cfg) continue;
        if (!menu_on && o.tm != 3) continue;
        const int bm = 16 * o.tm * o.wm;
        const long rows = (p.M + bm - 1) / bm, tiles = rows * (p.N / (32 * o.tn));
        const double t = model(bm, 32 * o.tn, tiles);
        if (t < best - 1e-12) {
            best = t;
            best_cfg = cfg;
            // the round-3 rule: a 288-row shape whose launched tile slots are >= 90 % useful
            best_fills = o.tm == 3 && (double)tiles / (double)((tiles + 255) / 256 * 256) * ((double)p.M / (rows * 288.0)) >= 0.90;
        }
    }
    bool best_mixed = false;
    // mixed 320 | 256-row tiles (8 MFMA waves): M = 320 a + 256 (32 - a), 0 < a < 32
    for (int tn = 4; tn >= 3 && menu_on; --tn) {
        const int cfg = 4540 + tn;
        if (!cols_ok(tn) || (p.N / (32 * tn)) % 8 != 0 || (force_cfg != 0 && force_cfg != cfg) || (force_tn != 0 && force_tn != tn)) continue;
        const int rest = p.M - 32 * 256;
        if (rest <= 0 || rest % 64 != 0 || rest / 64 >= 32) continue;
        const double t = model(320, 32 * tn, 32L * (p.N / (32 * tn)));
        if (t < best - 1e-12) { best = t; best_cfg = cfg; best_mixed = true; best_fills = false; }
    }
    if (best_cfg == 0) return 0;
    if (mode == 2 || force_cfg != 0) return best_cfg;
    // mode 1. Measured (in-process A/B on every shape of the model, profiles/r03_gemm_lab_v4_ab_*.txt,
    // r04_gemm_lab_v4_menu_*.txt): the 288-row shapes that fill the chip and the mixed launch beat the 4-wave blocks on
    // every forward / dgrad shape but one (+4 ... +14 %; 9472 x 1024 x 3 x 1024 forward: -1 %). For the small-M menu the two models are compared; plan_v2's cost is in
    // 32 x 32-tile K steps (53.4 ns on a saturated CU) and tracks the measured time (x 0.93) while a CU holds at most two
    // 4-wave blocks - beyond that (large M, where the menu has nothing to offer anyway) it is not calibrated: 4-wave.
    if (best_fills || best_mixed) return best_cfg;
    // The small-M menu wins most isolated A/Bs (profiles/r04_gemm_lab_v4_menu_M2304.txt / _M2368.txt, planner's choice against the
    // 4-wave blocks: +8 ... +20 % on 15 of 20 forward / dgrad launches, -1 ... -13 % on 5) and is a WASH inside the batch-64 training
    // step (profiles/r04_bench_b64_menu_ab.txt: 1,986 -> 2,022 samples/s eager, 1,716 -> 1,773 single-stream, 1,887 -> 1,932 as
    // one HIP graph on one box; 1,954 -> 1,866 on an earlier one): there the text / image / weight-gradient streams keep
    // several kernels in flight, the 4-wave blocks of different kernels co-reside on a CU and cover each other's bubbles,
    // while a persistent block owns its CU - so it is opt-in (VB_GEMM_V4_SMALLM=1: single-stream inference, laboratory).
    static const bool small_m = [] { const char* e = getenv("VB_GEMM_V4_SMALLM"); return e != nullptr && atoi(e) != 0; }();
    if (!menu_on || !small_m) return 0;
    const long v2_tiles = (long)((p.M + 95) / 96) * ((p.N + 95) / 96);     // upper bound of plan_v2's block count
    if (v2_tiles > 3 * 256) return 0;
    const double t_v2 = 0.93 * 53.4e-9 * v2_cost;
    return best < margin * t_v2 ? best_cfg : 0;
}

// Persistent weight-gradient kernel (gemm_v4w.h): 384 x 96 tiles of dW times K splits as equal work units. Fills the
// launch fields and returns true when the shape divides (text-stream weights: 768 / 2304 / 3072 rows, 768 / 3072
// columns) and the units fill the chip.
int plan_v4w(GemmP& p) {
    const int mode = gemm_v4_mode();
    if (mode == 0 || p.K % 32 != 0) return -1;
    // {tile rows, tile columns, relative main-loop efficiency}: 12 MFMA waves for the 384-row tiles, 8 for the 256-row ones
    static const struct { int bm, bn; double eff; } cfgs[3] = {{384, 96, 1.0}, {256, 128, 0.95}, {256, 96, 0.93}};
    const int kt = p.K / V2_BK;
    double best = 1e300;
    int best_s = 0, best_c = -1;
    for (int c = 0; c < 3; ++c) {
        const int BM = cfgs[c].bm, BN = cfgs[c].bn;
        if (p.M % BM != 0 || p.N % BN != 0 || p.cseg % BM != 0) continue;
        const long tiles = (long)(p.M / BM) * (p.N / BN);
        for (int s = 1; s <= 64; ++s) {
            if (kt % s != 0) continue;
            const int nk = kt / s;
            if (nk % 2 != 0 || nk < 8) continue;
            const long units = tiles * s, rounds = (units + 255) / 256;
            const double eff = (double)units / (double)(rounds * 256);
            // measured (tools/gemm_lab_prod LAB_V4_AB=1, profiles/r03_gemm_lab_v4w_ab.txt): +8.5 % with 144 K steps per unit
            // (W[3072, 768], W[768, 3072] at 9216 rows), -5 % with 36 (W[768, 768] needs 16 splits to fill the chip and
            // every unit ends in a 147 KB burst of stores that all 256 blocks issue at the same instant)
            if (mode != 2 && (eff < 0.85 || nk < 64)) continue;
            // ~8 K steps of epilogue per unit; cost in units of one 16 x 16 tile K step per CU
            const double cost = (double)rounds * (nk + 8.0) * (BM / 16) * (BN / 16) / cfgs[c].eff;
            if (cost < best - 1e-9) { best = cost; best_s = s; best_c = c; }
        }
    }
    if (best_c < 0) return -1;
    const int BM = cfgs[best_c].bm, BN = cfgs[best_c].bn;
    const long tiles = (long)(p.M / BM) * (p.N / BN);
    p.tiles_n = p.N / BN;
    p.n_small = (int)tiles;
    p.n_big = (int)(tiles * best_s);
    p.ktiles_per_split = kt / best_s;
    p.epi = best_s > 1 ? EPI_ATOMIC : EPI_ACCUM;
    return best_c;
}

// splits: 1 = no split-K; < 0 = split-K launch (wgrad), choose the count; legacy_splits = count for the round-1 kernel
template <bool A_KC, bool B_KC>
int launch_gemm(hipStream_t st, GemmP p, bool vec, int splits, int legacy_splits = 1, int vec_v2 = -1) {
    static const int flags = [] { const char* e = getenv("VB_GEMM_FLAGS"); return e ? atoi(e) : 0; }();
    p.flags = flags;
    // VB_GEMM_MODE: "f32" (default) = exact fp32 MFMA; "bf16x6" / "bf16x3" = fp32 emulated on the bf16
    // matrix cores with 3 / 2 operand planes (gemm_planes.hip)
    const int planes = gemm_mode();
    V2Plan pl;
    p.dbg = g_dbg;
    // vec_v2: 16-byte loads legal for the second-generation kernel (it tolerates a row-contiguous A whose row count is
    // not a multiple of 4 when the leading dimension leaves room for the last float4); default = same as `vec`
    if (planes == 0 && plan_v2<A_KC, B_KC>(p, vec_v2 < 0 ? vec : vec_v2 != 0, splits, pl)) {
        const int c4w = (!A_KC && !B_KC && splits < 0) ? plan_v4w(p) : -1;
        if (c4w >= 0) {
            const int s4 = p.n_big / p.n_small;
            const bool det = s4 > 1 && deterministic() && det_prepare(st, p, s4);
            if (int e = launch_gemm_v4_tn(st, p, c4w)) return e;
            return det ? det_finish(st, p, s4) : 0;
        }
        // persistent 288-row tiles (gemm_v4.h) where they fill the chip in whole rounds
        if (A_KC && splits == 1) {
            const int cfg4 = plan_v4(p, B_KC, pl.cost);
            g_v4_last_cfg = cfg4;
            if (cfg4 != 0) return B_KC ? launch_gemm_v4_nt(st, p, cfg4) : launch_gemm_v4_nn(st, p, cfg4);
        }
        p.tiles_n = pl.tiles_n;
        p.ktiles_per_split = pl.kt_per_split;
        p.n_big = pl.big_rows * pl.tiles_n;
        p.m_split = pl.big_rows * 32 * pl.tm1;
        if (splits < 0) p.epi = pl.splits > 1 ? EPI_ATOMIC : EPI_ACCUM;
        const int tiles = (pl.big_rows + pl.small_rows) * pl.tiles_n;
        const bool det = splits < 0 && pl.splits > 1 && deterministic() && det_prepare(st, p, pl.splits);
        int e = 0;
        if (A_KC && B_KC) e = launch_gemm_v2_nt(st, p, pl.tm1, pl.tm2, pl.tn, tiles, pl.splits);
        else if (A_KC) e = launch_gemm_v2_nn(st, p, pl.tm1, pl.tm2, pl.tn, tiles, pl.splits);
        else e = launch_gemm_v2_tn(st, p, pl.tm1, pl.tm2, pl.tn, tiles, pl.splits);
        if (e) return e;
        return det ? det_finish(st, p, pl.splits) : 0;
    }
    // round-1 kernel: any alignment, ragged K, every epilogue. It stores the pre-activation where the activation
    // derivative is wanted and leaves the multiplier to a post-pass (both fused only in the kernel above; keeping
    // erf + exp out of this kernel's generic epilogue keeps it free of register spills).
    float* const want_d = p.D;
    const float* const want_mul = p.mul;
    if (want_d != nullptr) {
        if (p.P != nullptr) return VB_E_BADARG;
        p.P = p.D; p.ldp = p.ldd; p.D = nullptr;
        if (p.epi == EPI_DGELU) p.epi = EPI_PRE_GELU;
    }
    if (want_mul != nullptr) {
        if (p.accumulate || splits != 1) return VB_E_BADARG;
        p.mul = nullptr;
        if (p.epi == EPI_MUL) p.epi = EPI_STORE;
    }
    bool det_legacy = false;
    if (splits < 0) {
        // weight gradient on the round-1 / bf16-planes kernels: split-K with fp32 atomics, or - deterministic mode -
        // through the workspace (N % 4 == 0: the reduce kernel works in float4; no room in the slice: unsplit)
        splits = legacy_splits;
        const int kt_total = (p.K + BK - 1) / BK;
        p.ktiles_per_split = (kt_total + splits - 1) / splits;
        splits = (kt_total + p.ktiles_per_split - 1) / p.ktiles_per_split;
        if (deterministic() && splits > 1) {
            p.det_cs_parts = planes != 0 ? 2 : 1;
            det_legacy = p.N % 4 == 0 && det_prepare(st, p, splits);
            if (!det_legacy) { splits = 1; p.ktiles_per_split = kt_total; }
        }
        p.epi = splits > 1 ? EPI_ATOMIC : EPI_ACCUM;
    }
    plan_tiles(p, splits, planes != 0);
    dim3 grid(p.n_big + p.n_small, splits), block(256);
    if (planes != 0) {
        // operands split into bf16 planes on their way into LDS (gemm_planes.hip)
        if (int e = planes == 3 ? launch_gemm_planes3(st, p, vec, splits, A_KC, B_KC)
                    : planes == 2 ? launch_gemm_planes2(st, p, vec, splits, A_KC, B_KC)
                                  : launch_gemm_planes1(st, p, vec, splits, A_KC, B_KC))
            return e;
    } else {
        if (vec) hipLaunchKernelGGL((gemm_f32_kernel<A_KC, B_KC, true>), grid, block, GEMM_LDS_BYTES, st, p);
        else hipLaunchKernelGGL((gemm_f32_kernel<A_KC, B_KC, false>), grid, block, GEMM_LDS_BYTES, st, p);
    }
    VB_LAUNCH_CHECK();
    if (det_legacy) return det_finish(st, p, splits);
    if (want_d != nullptr)
        if (int e = launch_act_grad_inplace(st, p.M, p.N, want_d, p.ldp, p.act)) return e;
    if (want_mul != nullptr)
        if (int e = launch_mul_inplace(st, p.M, p.N, p.C[0], p.ldc, want_mul, p.ldmul)) return e;
    return 0;
}

}  // namespace

// Laboratory hook (tools/gemm_lab.cpp, not part of the product ABI): device buffer of 2 x uint64 receiving
// {shader cycles of the K loop of block 128, its K steps} of every following second-generation GEMM launch.
extern "C" void vblab_gemm_cycles(unsigned long long* dev_buf) { g_dbg = dev_buf; }
// Laboratory / test hook (not part of the product ABI): force one persistent-kernel configuration (plan_v4 code, 0 = the
// planner's choice) wherever the shape allows it. Returns the previous value.
// configuration code of the most recent forward / dgrad launch that reached the planner (0 = it ran on the 4-wave blocks)
extern "C" int vblab_last_gemm_v4_cfg(void) { return g_v4_last_cfg; }
extern "C" int vblab_set_gemm_v4_cfg(int cfg) {
    const int prev = g_v4_force_cfg < 0 ? 0 : g_v4_force_cfg;
    g_v4_force_cfg = cfg;
    return prev;
}

extern "C" int vb_set_gemm_tile(int code) {
    const int prev = gemm_tile_code();
    if (code == -1 || code == 0 || code == 22 || code == 33 || code == 34 || code == 43 || code == 44 || code == 434 ||
        code == 433 || code == 324 || code == 323)
        g_gemm_tile = code;
    return prev;
}

extern "C" int vb_set_deterministic(int on, void* workspace, int64_t workspace_bytes) {
    const int prev = g_det > 0 ? 1 : 0;
    if (on != 0 && on != 1) return prev;
    if (on && (workspace == nullptr || workspace_bytes <= 0 || !vb_aligned16(workspace))) return VB_E_BADARG;
    int dev = -1;
    if (on) {
        // the workspace serves the device it lives on, whichever device is current now
        hipPointerAttribute_t attr;
        if (hipPointerGetAttributes(&attr, workspace) != hipSuccess) { (void)hipGetLastError(); return VB_E_BADARG; }
        dev = attr.device;
        if (dev < 0 || dev >= DET_MAX_DEV) return VB_E_BADARG;
    }
    std::lock_guard<std::mutex> lock(g_det_mutex);
    if (!on) {
        for (DetDevice& d : g_det_dev) d = DetDevice();
        g_det = 0;
        return prev;
    }
    DetDevice& d = g_det_dev[dev];
    // re-registering the same buffer keeps the stream -> slice assignment (captured graphs have it baked in)
    if (d.ws != static_cast<float*>(workspace) || d.bytes != (size_t)workspace_bytes) {
        d = DetDevice();
        d.ws = static_cast<float*>(workspace);
        d.bytes = (size_t)workspace_bytes;
    }
    g_det = 1;
    g_det_fallbacks = 0;
    return prev;
}

namespace vbgemm {
bool det_on() { return deterministic(); }
float* det_slice(hipStream_t st, size_t* slice_bytes) { return det_slice_of(st, slice_bytes); }
void det_fallback() { det_count_fallback(); }
}  // namespace vbgemm

extern "C" int64_t vb_deterministic_fallbacks(void) {
    std::lock_guard<std::mutex> lock(g_det_mutex);
    return (int64_t)g_det_fallbacks;
}

extern "C" int vb_set_gemm_v4(int mode) {
    const int prev = gemm_v4_mode();
    if (mode >= 0 && mode <= 2) g_gemm_v4 = mode;
    return prev;
}

extern "C" int vb_set_gemm_mode(int planes) {
    const int prev = gemm_mode();
    if (planes >= 0 && planes <= 3) g_gemm_mode = planes;
    return prev;
}

extern "C" int vb_linear_fwd(void* stream, const vb_linear_args* a) {
    if (a == nullptr || a->A == nullptr || a->C == nullptr) return VB_E_BADARG;
    if (a->M <= 0 || a->K <= 0 || a->seg_n <= 0) return VB_E_BADARG;
    if (a->nseg < 1 || a->nseg > VB_MAX_SEGMENTS) return VB_E_SEGMENT;
    if (a->act < VB_ACT_NONE || a->act > VB_ACT_SWISH) return VB_E_BADARG;
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
    p.D = a->act_grad; p.ldd = a->ldg;
    p.act = a->act; p.accumulate = 0;
    if (!(a->dropout_p >= 0.f && a->dropout_p < 1.f)) return VB_E_BADARG;
    if (a->dropout_p > 0.f && a->ldc != p.N) return VB_E_ALIGN;
    p.drop_p = a->dropout_p; p.drop_scale = 1.0f / (1.0f - a->dropout_p); p.seed = a->seed;
    p.epoch = a->dropout_p > 0.f ? vb_seed_epoch() : nullptr;
    if (a->dropout_p > 0.f)
        p.epi = (a->act == VB_ACT_NONE && a->preact == nullptr && a->residual != nullptr) ? EPI_RES_DROP : EPI_GENERIC;
    else if (a->act == VB_ACT_NONE && a->preact == nullptr) p.epi = a->residual != nullptr ? EPI_RES : EPI_STORE;
    else if (a->act == VB_ACT_GELU && a->residual == nullptr) p.epi = a->preact != nullptr ? EPI_PRE_GELU : EPI_GELU;
    else p.epi = EPI_GENERIC;
    if (a->act_grad != nullptr)
        p.epi = (p.epi == EPI_GELU && a->dropout_p == 0.f) ? EPI_DGELU : EPI_GENERIC;
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
        if (a->mul != nullptr) {
            if (!fused) return VB_E_SEGMENT;
            p.mul = a->mul; p.ldmul = a->ldm;
            p.epi = (p.R == nullptr && !p.accumulate) ? EPI_MUL : EPI_GENERIC;
        }
        p.ktiles_per_split = (p.K + BK - 1) / BK;
        // A contraction length that is not a multiple of 16 (the MLM decoder: 30522 out-features) would send the whole
        // GEMM to the round-1 kernel with scalar loads: run the aligned bulk on the second-generation kernel and
        // add the <= 15 leftover k with a second, tiny launch.
        // A small output with a long contraction (the MLM decoder: dX [1628 x 768] over 30522 out-features = 136 tiles
        // for 256 CUs, 66 TF) is cut along K as the wgrad launches are: the planner picks the split count and the splits
        // add into dX with atomics (dX is zero-filled first unless it already holds a contribution).
        // (round 6: in the bf16-operand modes too - the bf16 training mode left this one launch unsplit on 96 blocks:
        // 0.90 ms of its 27 ms step at B = 256, profiles/r06_bf16_ragged_wgrad_ab.txt; there the count is chosen here:
        // about three blocks per CU, at least 1024 contraction elements per split)
        const bool split_k = p.mul == nullptr && p.R == nullptr && (long)p.M * p.N <= 2048L * 1024 && p.K >= 4096 &&
                             (p.epi == EPI_STORE || p.epi == EPI_ACCUM);
        auto launch_main = [&](GemmP q, bool vq) -> int {
            if (!split_k) return launch_gemm<true, false>(st, q, vq, 1);
            int planes_splits = 1;
            if (gemm_mode() != 0) {
                const long tiles = (long)((q.M + 127) / 128) * ((q.N + 127) / 128);
                long sp = (768 + tiles - 1) / tiles;
                if (sp > q.K / 1024) sp = q.K / 1024;
                planes_splits = (int)(sp < 1 ? 1 : (sp > 32 ? 32 : sp));
                if (planes_splits == 1) return launch_gemm<true, false>(st, q, vq, 1);
            }
            if (q.epi == EPI_STORE) {
                if (int e = zero_rows(st, q.C[0], q.ldc, q.M, q.N)) return e;       // (a kernel, not a memset node: see zero_rows)
            }
            q.epi = EPI_ACCUM;
            q.accumulate = 1;
            return launch_gemm<true, false>(st, q, vq, -1, planes_splits);
        };
        const int k_main = p.K / V2_BK * V2_BK;
        if (fused && a->nseg == 1 && k_main >= 256 && k_main != p.K && p.R == nullptr && p.mul == nullptr &&
            a->K % 4 == 0 && a->ldy % 4 == 0 && a->ldw % 4 == 0 && vb_aligned16(p.A) && vb_aligned16(p.B[0])) {
            GemmP m = p, t = p;
            m.K = k_main; m.bseg = k_main; m.ktiles_per_split = k_main / BK;
            if (int e = launch_main(m, true)) return e;
            t.K = p.K - k_main; t.bseg = t.K; t.A = p.A + k_main; t.B[0] = p.B[0] + (long)k_main * p.ldb;
            t.accumulate = 1; t.epi = EPI_ACCUM; t.ktiles_per_split = 1;
            if (int e = launch_gemm<true, false>(st, t, false, 1)) return e;
            continue;
        }
        if (int e = launch_main(p, vec)) return e;
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
            if (int e = zero_rows(st, a->dW[s], a->ldw, a->seg_n, a->K)) return e;
            if (a->dbias[s] != nullptr)
                if (int e = zero_rows(st, a->dbias[s], a->seg_n, 1, a->seg_n)) return e;
        }
    }
    {
        const int e = launch_wgrad_skinny(st, a);      // in_features <= 8: one streaming pass instead of a 5-column GEMM
        if (e >= 0) return e;
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
        const bool vec = (a->seg_n % 4 == 0) && (a->K % 4 == 0) && (a->ldy % 4 == 0) && (a->ldx % 4 == 0) &&
                         vb_aligned16(p.A) && vb_aligned16(a->X);
        p.epi = EPI_ATOMIC;
        // second-generation kernel: 16-byte loads along the out-feature dimension are legal when the row stride of dY
        // leaves room for the last float4 (a [rows, 30522] gradient stored with leading dimension 30524)
        const bool vec2 = (a->K % 4 == 0) && (a->ldy % 4 == 0) && (a->ldx % 4 == 0) && vb_aligned16(p.A) && vb_aligned16(a->X) &&
                          (a->seg_n % 4 == 0 || (segs == 1 && a->ldy >= (a->seg_n + 3) / 4 * 4));
        const int k_main = p.K / V2_BK * V2_BK;
        if (vec2 && k_main >= 256 && k_main != p.K) {
            // contraction (row count) not a multiple of 16: aligned bulk + a tiny launch for the <= 15 leftover rows
            GemmP m = p, t = p;
            m.K = k_main; m.bseg = k_main;
            if (int e = launch_gemm<false, false>(st, m, vec, -1, splits, 1)) return e;
            t.K = p.K - k_main; t.bseg = t.K; t.A = p.A + (long)k_main * p.lda; t.B[0] = p.B[0] + (long)k_main * p.ldb;
            if (int e = launch_gemm<false, false>(st, t, false, -1, 1, 0)) return e;
            continue;
        }
        if (int e = launch_gemm<false, false>(st, p, vec, -1, splits, vec2 ? 1 : 0)) return e;
    }
    return 0;
}
