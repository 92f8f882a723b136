// Fused short-sequence attention:  O = dropout(softmax(Q K^T * scale + mask)) V  per (sample, head),
// forward and backward.
//
// ViLBERT sequences are short (36 tokens / 36 regions pre-training, <= 306 in the task table), so
// this is not a flash-attention problem: one 64-lane wave owns a 16-row tile of one (sample, head)
// and keeps the whole score row block in registers - no S x S round trip through HBM. Two kernel families share the arithmetic: the generic one below (operands
// straight from L2, any length <= 320 keys, batch broadcast) and an LDS-staged one for sequences of at
// most 48 rows (one block per (sample, head), K / V or Q / dO staged once) further down. Neither makes
// head split / merge copies (Q, K, V are read straight out of the fused [q | k | v] projection, O is
// written token-major, the backward writes dQ/dK/dV straight into the fused gradient buffer).
//
// MFMA: v_mfma_f32_16x16x4_f32 (exact fp32). Lane l = (c = l & 15, g = l >> 4) supplies
// A[i = c][k = g] and B[k = g][j = c]; D[row = 4g + r][col = c], r = 0..3. Contractions over
// head_dim are permuted so that lane group g owns dims 16s + 4g + e (one 16-byte load per 4 MFMAs).
//
// forward / backward pass 1 (one wave per 16 queries):
//  * scores are computed TRANSPOSED, S^T = K Q^T (A = key rows, B = query rows): lane (c, g) holds
//    S[q = c][key = 16 kt + 4g + r] - a fixed query per lane, so softmax row reductions are
//    in-register plus two cross-lane steps (xor 16, xor 32);
//  * that is exactly the A-operand layout of P V (A[i = q][k = key], lane group g owning keys
//    16 kt + 4g + r), so P never leaves its registers; B = V[key][d] is read with plain dword loads
//    (16 lanes = 64 contiguous bytes). Pass 1 of the backward reuses the structure for
//    dP^T = V dO^T, D = rowsum(P dP), dS = P (dP - D) scale and dQ = dS K.
// backward pass 2 (one wave per 16 keys, looping over query tiles): S = Q K^T with A = query rows,
//    B = key rows leaves lane (c, g) with S[q = 16 qt + 4g + r][key = c]; P^T and dS^T are then the
//    A operands of dV = P^T dO and dK = dS^T Q. Softmax statistics come from the saved
//    log-sum-exp and the D vector written by pass 1. No atomics anywhere.
// Dropout masks are regenerated from (seed, element index) - see rng.h.
#include "common.h"
#include "rng.h"

namespace {

// The file is compiled twice (csrc/Makefile): as is - fp32 tensors, vb_attention_fwd / vb_attention_bwd - and with
// -DVB_ATTN_BF16 - the bf16 training path: Q, K, V, dO read and O, dQ, dK, dV written as bf16 (half the HBM bytes of
// kernels that are HBM-bound at these sequence lengths), vb_attention_fwd_bf16 / vb_attention_bwd_bf16. The arithmetic
// is the same in both: operands are widened to fp32 on their way into registers / LDS, fp32 MFMA, fp32 softmax
// statistics (mask, lse, D vector and the optional probabilities stay fp32 tensors).
#ifdef VB_ATTN_BF16
typedef unsigned short io_t;
__device__ __forceinline__ f32x4 ld4(const io_t* p) {
    const uint2 w = *reinterpret_cast<const uint2*>(p);
    return f32x4{__uint_as_float(w.x << 16), __uint_as_float(w.x & 0xffff0000u), __uint_as_float(w.y << 16),
                 __uint_as_float(w.y & 0xffff0000u)};
}
__device__ __forceinline__ float ld1(const io_t* p) { return __uint_as_float((unsigned)*p << 16); }
__device__ __forceinline__ void st1(io_t* p, float v) {
    const unsigned u = __float_as_uint(v);
    *p = (io_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}
// one output row piece: element a[dt][r] belongs to column 16 dt + c of the row; `op` already points at column c. Lanes c and
// c ^ 1 trade one value per pair of 16-column blocks, so that every lane stores TWO adjacent bf16 columns (4 bytes) per pair
// instead of one 2-byte element per block: half the store instructions (these kernels' tails are store-issue-bound).
template <int DS>
__device__ __forceinline__ void store_cols(io_t* op, int c, const f32x4 (&a)[DS], int r) {
    auto bf = [](float v) -> unsigned { const unsigned u = __float_as_uint(v); return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16; };
#pragma unroll
    for (int k = 0; k < DS / 2; ++k) {
        const float even = a[2 * k][r], odd = a[2 * k + 1][r];
        const float recv = __shfl_xor((c & 1) ? even : odd, 1, 64);
        if (c & 1) *reinterpret_cast<unsigned*>(op + 32 * k + 15) = bf(recv) | (bf(odd) << 16);     // columns 32 k + 16 + c - 1, + c
        else *reinterpret_cast<unsigned*>(op + 32 * k) = bf(even) | (bf(recv) << 16);               // columns 32 k + c, + c + 1
    }
}
#else
typedef float io_t;
__device__ __forceinline__ f32x4 ld4(const io_t* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ float ld1(const io_t* p) { return *p; }
__device__ __forceinline__ void st1(io_t* p, float v) { *p = v; }
template <int DS>
__device__ __forceinline__ void store_cols(io_t* op, int c, const f32x4 (&a)[DS], int r) {
#pragma unroll
    for (int dt = 0; dt < DS; ++dt) op[16 * dt] = a[dt][r];
}
#endif

struct AttnP {
    int batch, heads, n_q, n_k, n_qt, n_kt;
    long q_bstride, kv_bstride, m_bstride;  // rows (or mask floats) per sample; 0 = broadcast
    const io_t* Q; long ldq;
    const io_t* K; long ldk;
    const io_t* V; long ldv;
    const float* mask;
    io_t* O; long ldo;
    float* probs;
    float* lse;       // [batch, heads, n_q]   forward: out (optional), backward: in
    float scale;
    float drop_p, drop_scale;
    uint64_t seed;
    const uint64_t* epoch;   // device step counter mixed into the seed (vb_set_seed_epoch), may be null
    long total;       // wave items
    // backward only
    const io_t* dO; long lddo;
    io_t* dQ; long lddq;
    io_t* dK; long lddk;
    io_t* dV; long lddv;
    float* dvec;      // [batch, heads, n_q]  D = rowsum(P dP): pass 1 out, pass 2 in
    int dvec_mode;    // VB_DVEC_*: key chunks of a longer sequence (pass 1: accumulate D only / take D as given)
};

template <int DS>
__device__ __forceinline__ void load_frag(f32x4 (&f)[DS], const io_t* p) {
#pragma unroll
    for (int s = 0; s < DS; ++s) f[s] = ld4(p + 16 * s);
}

// The two contraction patterns of these kernels. fp32 build: v_mfma_f32_16x16x4_f32, four instructions per 16 contraction
// values (lane group g supplies index 4 g + e in step e). bf16 build (VB_ATTN_BF16): ONE v_mfma_f32_16x16x16_bf16 per 16 values -
// its operand layout is exactly "lane group g holds the contraction indices 4 g .. 4 g + 3", i.e. the four fp32 operands of
// the four steps packed into one register pair - a quarter of the matrix instructions, each at the bf16 rate. Q, K, V and dO
// ARE bf16 values there (widened on their way in), so their products are exact as before; probabilities and dS are rounded
// to bf16 (2^-9 relative) on their way into the second contraction, accumulation stays fp32.
#ifdef VB_ATTN_BF16
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ s16x4 pk4(const f32x4 v) {
    return __builtin_bit_cast(s16x4, __builtin_convertvector(v, bf16x4_t));     // v_cvt_pk_bf16_f32: round to nearest even
}
#endif

template <int DS>
__device__ __forceinline__ f32x4 dot_tile(const f32x4 (&a)[DS], const f32x4 (&b)[DS]) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < DS; ++s) {
#ifdef VB_ATTN_BF16
        acc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(pk4(a[s]), pk4(b[s]), acc, 0, 0, 0);
#else
#pragma unroll
        for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s][e], b[s][e], acc, 0, 0, 0);
#endif
    }
    return acc;
}

// acc[dt] += sum_r a[r] b[dt][r]: the rank-4 update of the second contractions (a[r] = this lane's probability / dS value of
// contraction index 4 g + r, b[dt][r] = the operand row of that index at column 16 dt + c)
template <int DS>
__device__ __forceinline__ void mma_rank4(f32x4 (&acc)[DS], const f32x4 a, const f32x4 (&b)[DS]) {
#ifdef VB_ATTN_BF16
    const s16x4 pa = pk4(a);
#pragma unroll
    for (int dt = 0; dt < DS; ++dt) acc[dt] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(pa, pk4(b[dt]), acc[dt], 0, 0, 0);
#else
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int dt = 0; dt < DS; ++dt) acc[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[r], b[dt][r], acc[dt], 0, 0, 0);
#endif
}

__device__ __forceinline__ float group_max(float v) {
    v = fmaxf(v, __shfl_xor(v, 16, 64));
    return fmaxf(v, __shfl_xor(v, 32, 64));
}

__device__ __forceinline__ float group_sum(float v) {
    v += __shfl_xor(v, 16, 64);
    return v + __shfl_xor(v, 32, 64);
}

// BWD = false: forward.  BWD = true: backward pass 1 (dQ and the D vector).
template <int D, int NT, bool BWD>
__global__ __launch_bounds__(256) void attn_q_kernel(const AttnP p_in) {
    AttnP p = p_in;
    p.seed = vb_seed_with_epoch(p_in.seed, p_in.epoch);
    const int lane = threadIdx.x & 63;
    const long item = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (item >= p.total) return;
    const int qt = (int)(item % p.n_qt);
    const long bh = item / p.n_qt;
    const int h = (int)(bh % p.heads);
    const int b = (int)(bh / p.heads);
    const int c = lane & 15, g = lane >> 4;
    constexpr int DS = D / 16;
    const int nkt = p.n_kt;
    const bool drop = p.drop_p > 0.f;

    // Row fragments of this lane's query (B operands): Q[q = 16 qt + c][16 s + 4 g + e], same for dO.
    const int q_row = min(qt * 16 + c, p.n_q - 1);
    f32x4 qf[DS];
    load_frag<DS>(qf, p.Q + ((long)b * p.q_bstride + q_row) * p.ldq + h * D + 4 * g);

    const io_t* kbase = p.K + (long)b * p.kv_bstride * p.ldk + h * D;
    const io_t* vbase = p.V + (long)b * p.kv_bstride * p.ldv + h * D;
    const float* mrow = p.mask != nullptr ? p.mask + (long)b * p.m_bstride : nullptr;
    const long prow = (bh * p.n_q + q_row) * p.n_k;  // element index of P[b, h, q, 0]

    f32x4 st[NT];
    float mx = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < NT; ++kt) {
        st[kt] = f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        if (kt < nkt) {
            const int k_row = min(kt * 16 + c, p.n_k - 1);
            f32x4 kf[DS];
            load_frag<DS>(kf, kbase + (long)k_row * p.ldk + 4 * g);
            f32x4 acc = dot_tile<DS>(kf, qf);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int key = kt * 16 + 4 * g + r;
                float v = -INFINITY;
                if (key < p.n_k) {
                    // vilbert.py:435-439: scores / sqrt(d) + mask. Two separately rounded steps like
                    // the reference (no fma contraction): with the -10000 mask one fp32 ulp of the
                    // sum is 1e-3, so the rounding order shows.
                    v = __fmul_rn(acc[r], p.scale);
                    if (mrow != nullptr) v = __fadd_rn(v, mrow[key]);
                }
                acc[r] = v;
                mx = fmaxf(mx, v);
            }
            st[kt] = acc;
        }
    }

    if (!BWD) {
        mx = group_max(mx);
        float sum = 0.f;
#pragma unroll
        for (int kt = 0; kt < NT; ++kt)
            if (kt < nkt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float e = expf(st[kt][r] - mx);
                    st[kt][r] = e;
                    sum += e;
                }
        sum = group_sum(sum);
        const float inv = 1.0f / sum;
        if (p.lse != nullptr && g == 0 && qt * 16 + c < p.n_q) p.lse[bh * p.n_q + qt * 16 + c] = mx + logf(sum);
#pragma unroll
        for (int kt = 0; kt < NT; ++kt)
            if (kt < nkt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float pv = st[kt][r] * inv;
                    if (drop) {
                        const int key = kt * 16 + 4 * g + r;
                        pv = vb_keep(p.seed, (uint64_t)(prow + key), p.drop_p) ? pv * p.drop_scale : 0.f;
                    }
                    st[kt][r] = pv;
                }
        if (p.probs != nullptr && qt * 16 + c < p.n_q) {
            float* pr = p.probs + prow;
#pragma unroll
            for (int kt = 0; kt < NT; ++kt)
                if (kt < nkt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int key = kt * 16 + 4 * g + r;
                        if (key < p.n_k) pr[key] = st[kt][r];
                    }
        }
    } else {
        // P from the saved log-sum-exp, dP^T = V dO^T, D = rowsum(P dP), dS = P (dP - D) scale
        const float lse = p.lse[bh * p.n_q + q_row];
        f32x4 dof[DS];
        load_frag<DS>(dof, p.dO + ((long)b * p.n_q + q_row) * p.lddo + h * D + 4 * g);
        f32x4 dp[NT];
        float dsum = 0.f;
#pragma unroll
        for (int kt = 0; kt < NT; ++kt) {
            dp[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (kt < nkt) {
                const int k_row = min(kt * 16 + c, p.n_k - 1);
                f32x4 vf[DS];
                load_frag<DS>(vf, vbase + (long)k_row * p.ldv + 4 * g);
                f32x4 acc = dot_tile<DS>(vf, dof);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int key = min(kt * 16 + 4 * g + r, p.n_k - 1);
                    const float pv = expf(st[kt][r] - lse);  // exp(-inf) = 0 past n_k
                    float d = acc[r];
                    if (drop) d = vb_keep(p.seed, (uint64_t)(prow + key), p.drop_p) ? d * p.drop_scale : 0.f;
                    st[kt][r] = pv;
                    acc[r] = d;
                    dsum += pv * d;
                }
                dp[kt] = acc;
            }
        }
        dsum = group_sum(dsum);
        if (p.dvec_mode == VB_DVEC_ACCUMULATE) {
            // one chunk of a longer key sequence: its share of D (launches of one stream: plain read-modify-write), no dQ yet
            if (g == 0 && qt * 16 + c < p.n_q) p.dvec[bh * p.n_q + qt * 16 + c] += dsum;
            return;
        }
        if (p.dvec_mode == VB_DVEC_GIVEN) dsum = p.dvec[bh * p.n_q + q_row];
        else if (g == 0 && qt * 16 + c < p.n_q) p.dvec[bh * p.n_q + qt * 16 + c] = dsum;
#pragma unroll
        for (int kt = 0; kt < NT; ++kt)
            if (kt < nkt)
#pragma unroll
                for (int r = 0; r < 4; ++r) st[kt][r] = st[kt][r] * (dp[kt][r] - dsum) * p.scale;
    }

    // forward: O[q][d] = sum_key P[q][key] V[key][d];  backward: dQ[q][d] = sum_key dS[q][key] K[key][d]
    const io_t* rbase = (BWD ? kbase : vbase) + c;
    const long ldr = BWD ? p.ldk : p.ldv;
    f32x4 oacc[DS];
#pragma unroll
    for (int dt = 0; dt < DS; ++dt) oacc[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kt = 0; kt < NT; ++kt) {
        if (kt < nkt) {
            f32x4 av, bv[DS];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int key = min(kt * 16 + 4 * g + r, p.n_k - 1);  // the A operand is 0 past n_k
                const io_t* rp = rbase + (long)key * ldr;
                av[r] = st[kt][r];
#pragma unroll
                for (int dt = 0; dt < DS; ++dt) bv[dt][r] = ld1(rp + 16 * dt);
            }
            mma_rank4<DS>(oacc, av, bv);
        }
    }

    // D[row = q = 4g + r][col = d = 16 dt + c]
    io_t* obase = BWD ? p.dQ : p.O;
    const long ldo = BWD ? p.lddq : p.ldo;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int q = qt * 16 + 4 * g + r;
        if (q < p.n_q) {
            io_t* op = obase + ((long)b * p.n_q + q) * ldo + h * D + c;
            store_cols<DS>(op, c, oacc, r);
        }
    }
}

// Backward pass 2: one wave per (sample, head, 16-key tile); dK and dV.
template <int D>
__global__ __launch_bounds__(256) void attn_bwd_kv_kernel(const AttnP p_in) {
    AttnP p = p_in;
    p.seed = vb_seed_with_epoch(p_in.seed, p_in.epoch);
    const int lane = threadIdx.x & 63;
    const long item = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (item >= p.total) return;
    const int kt = (int)(item % p.n_kt);
    const long bh = item / p.n_kt;
    const int h = (int)(bh % p.heads);
    const int b = (int)(bh / p.heads);
    const int c = lane & 15, g = lane >> 4;
    constexpr int DS = D / 16;
    const bool drop = p.drop_p > 0.f;

    const int key = kt * 16 + c;
    const int k_row = min(key, p.n_k - 1);
    const bool key_ok = key < p.n_k;
    f32x4 kf[DS], vf[DS];
    load_frag<DS>(kf, p.K + ((long)b * p.n_k + k_row) * p.ldk + h * D + 4 * g);
    load_frag<DS>(vf, p.V + ((long)b * p.n_k + k_row) * p.ldv + h * D + 4 * g);
    const float madd = p.mask != nullptr ? p.mask[(long)b * p.n_k + k_row] : 0.f;

    const io_t* qbase = p.Q + (long)b * p.n_q * p.ldq + h * D;
    const io_t* dobase = p.dO + (long)b * p.n_q * p.lddo + h * D;
    const float* lse = p.lse + bh * p.n_q;
    const float* dvec = p.dvec + bh * p.n_q;

    f32x4 dk[DS], dv[DS];
#pragma unroll
    for (int dt = 0; dt < DS; ++dt) {
        dk[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
        dv[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

    for (int qt = 0; qt < p.n_qt; ++qt) {
        const int q_row = min(qt * 16 + c, p.n_q - 1);
        f32x4 qf[DS], dof[DS];
        load_frag<DS>(qf, qbase + (long)q_row * p.ldq + 4 * g);
        load_frag<DS>(dof, dobase + (long)q_row * p.lddo + 4 * g);
        const f32x4 s = dot_tile<DS>(qf, kf);    // S[q = 16 qt + 4g + r][key = c]
        const f32x4 dpr = dot_tile<DS>(dof, vf);  // dO V^T, same layout
        f32x4 pd, dsv;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int q = qt * 16 + 4 * g + r;
            float dsr = 0.f, pdr = 0.f;
            if (q < p.n_q && key_ok) {
                const float sv = __fadd_rn(__fmul_rn(s[r], p.scale), madd);
                const float pv = expf(sv - lse[q]);
                float d = dpr[r];
                pdr = pv;
                if (drop) {
                    const bool keep = vb_keep(p.seed, (uint64_t)((bh * p.n_q + q) * p.n_k + key), p.drop_p);
                    d = keep ? d * p.drop_scale : 0.f;
                    pdr = keep ? pv * p.drop_scale : 0.f;
                }
                dsr = pv * (d - dvec[q]) * p.scale;
            }
            pd[r] = pdr;
            dsv[r] = dsr;
        }
        // dV[key][d] += sum_q Pdrop[q][key] dO[q][d];  dK[key][d] += sum_q dS[q][key] Q[q][d]
        f32x4 dov[DS], qv[DS];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int q = min(qt * 16 + 4 * g + r, p.n_q - 1);  // A operands are 0 past n_q
            const io_t* dop = dobase + (long)q * p.lddo + c;
            const io_t* qp = qbase + (long)q * p.ldq + c;
#pragma unroll
            for (int dt = 0; dt < DS; ++dt) {
                dov[dt][r] = ld1(dop + 16 * dt);
                qv[dt][r] = ld1(qp + 16 * dt);
            }
        }
        mma_rank4<DS>(dv, pd, dov);
        mma_rank4<DS>(dk, dsv, qv);
    }

    // D[row = key = 4g + r][col = d = 16 dt + c]
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int kk = kt * 16 + 4 * g + r;
        if (kk < p.n_k) {
            io_t* kp = p.dK + ((long)b * p.n_k + kk) * p.lddk + h * D + c;
            io_t* vp = p.dV + ((long)b * p.n_k + kk) * p.lddv + h * D + c;
            store_cols<DS>(kp, c, dk, r);
            store_cols<DS>(vp, c, dv, r);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Short-sequence fast path (n_q, n_k <= 48: the pre-training / benchmark shape of 36-37 tokens and
// regions). One 4-wave block owns one (sample, head): K and V (forward, pass 1) or Q and dO (pass 2) are
// staged ONCE into LDS with coalesced 16-byte loads instead of being re-fetched from L2 by every 16-row
// tile, and the per-MFMA operand reads become ds_read_b128 / ds_read_b32 (rows padded by 4 floats: both
// access patterns are conflict free). Same arithmetic, same register layout as the generic kernels
// above, which were bound by global-load latency (12 dependent load -> MFMA rounds per wave).
// ------------------------------------------------------------------------------------------------
constexpr int LDS_MAX_ROWS = 48;

// Two operand blocks at once, every global load of the thread issued before the first LDS store (as one loop per
// block the compiler emitted load -> s_waitcnt vmcnt(0) -> ds_write per 16 bytes: 12 dependent trips to HBM per block
// at head_dim 128 - profiles/r03_attn_bench.txt).
template <int D>
__device__ __forceinline__ void stage_rows2(float* __restrict__ s0, const io_t* __restrict__ g0, long ld0,
                                            float* __restrict__ s1, const io_t* __restrict__ g1, long ld1, int n_rows) {
    // [n_rows][D] global (row stride ld) -> [LDS_MAX_ROWS][D + 4] LDS, rows >= n_rows zero-filled
    constexpr int V4 = D / 4, IT = (LDS_MAX_ROWS * V4 + 255) / 256;
    f32x4 r0[IT], r1[IT];
    // (unconditional loads from a clamped row: a select on the loaded value would put a wait behind every load)
#pragma unroll
    for (int i = 0; i < IT; ++i) {
        const int f = threadIdx.x + 256 * i;
        const int r = min(f / V4, n_rows - 1), c4 = f % V4;
        r0[i] = ld4(g0 + (long)r * ld0 + c4 * 4);
        r1[i] = ld4(g1 + (long)r * ld1 + c4 * 4);
    }
#pragma unroll
    for (int i = 0; i < IT; ++i) {
        const int f = threadIdx.x + 256 * i;
        const int r = f / V4, c4 = f % V4;
        if (f < LDS_MAX_ROWS * V4) {
            const bool real = r < n_rows;
            *reinterpret_cast<f32x4*>(s0 + r * (D + 4) + c4 * 4) = real ? r0[i] : f32x4{0.f, 0.f, 0.f, 0.f};
            *reinterpret_cast<f32x4*>(s1 + r * (D + 4) + c4 * 4) = real ? r1[i] : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
}

template <int DS>
__device__ __forceinline__ void load_frag_lds(f32x4 (&f)[DS], const float* s) {
#pragma unroll
    for (int q = 0; q < DS; ++q) f[q] = *reinterpret_cast<const f32x4*>(s + 16 * q);
}

template <int D, bool BWD>
__global__ __launch_bounds__(256, 3) void attn_q_lds_kernel(const AttnP p_in) {   // 3 blocks per CU fit the LDS at head_dim 128
    AttnP p = p_in;
    p.seed = vb_seed_with_epoch(p_in.seed, p_in.epoch);
    extern __shared__ __attribute__((aligned(16))) float smem_att[];
    constexpr int LD = D + 4, DS = D / 16, NT = LDS_MAX_ROWS / 16;
    float* sK = smem_att;
    float* sV = smem_att + LDS_MAX_ROWS * LD;
    const long bh = blockIdx.x;
    const int h = (int)(bh % p.heads), b = (int)(bh / p.heads);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 15, g = lane >> 4;
    const int nkt = p.n_kt;
    const bool drop = p.drop_p > 0.f;

    // the wave's first query fragment travels together with the K / V blocks (one trip to HBM instead of two)
    f32x4 qf_first[DS];
    load_frag<DS>(qf_first, p.Q + ((long)b * p.n_q + min(wave * 16 + c, p.n_q - 1)) * p.ldq + h * D + 4 * g);
    stage_rows2<D>(sK, p.K + (long)b * p.n_k * p.ldk + h * D, p.ldk, sV, p.V + (long)b * p.n_k * p.ldv + h * D, p.ldv, p.n_k);
    __syncthreads();
    const float* mrow = p.mask != nullptr ? p.mask + (long)b * p.n_k : nullptr;

    for (int qt = wave; qt < p.n_qt; qt += 4) {
        const int q_row = min(qt * 16 + c, p.n_q - 1);
        f32x4 qf[DS];
        if (qt == wave) {
#pragma unroll
            for (int i = 0; i < DS; ++i) qf[i] = qf_first[i];
        } else {
            load_frag<DS>(qf, p.Q + ((long)b * p.n_q + q_row) * p.ldq + h * D + 4 * g);
        }
        const long prow = (bh * p.n_q + q_row) * p.n_k;

        f32x4 st[NT];
        float mx = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < NT; ++kt) {
            st[kt] = f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
            if (kt < nkt) {
                f32x4 kf[DS];
                load_frag_lds<DS>(kf, sK + (kt * 16 + c) * LD + 4 * g);
                f32x4 acc = dot_tile<DS>(kf, qf);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int key = kt * 16 + 4 * g + r;
                    float v = -INFINITY;
                    if (key < p.n_k) {
                        v = __fmul_rn(acc[r], p.scale);
                        if (mrow != nullptr) v = __fadd_rn(v, mrow[key]);
                    }
                    acc[r] = v;
                    mx = fmaxf(mx, v);
                }
                st[kt] = acc;
            }
        }

        if (!BWD) {
            mx = group_max(mx);
            float sum = 0.f;
#pragma unroll
            for (int kt = 0; kt < NT; ++kt)
                if (kt < nkt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float e = expf(st[kt][r] - mx);
                        st[kt][r] = e;
                        sum += e;
                    }
            sum = group_sum(sum);
            const float inv = 1.0f / sum;
            if (p.lse != nullptr && g == 0 && qt * 16 + c < p.n_q) p.lse[bh * p.n_q + qt * 16 + c] = mx + logf(sum);
#pragma unroll
            for (int kt = 0; kt < NT; ++kt)
                if (kt < nkt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float pv = st[kt][r] * inv;
                        if (drop) {
                            const int key = kt * 16 + 4 * g + r;
                            pv = vb_keep(p.seed, (uint64_t)(prow + key), p.drop_p) ? pv * p.drop_scale : 0.f;
                        }
                        st[kt][r] = pv;
                    }
            if (p.probs != nullptr && qt * 16 + c < p.n_q) {
                float* pr = p.probs + prow;
#pragma unroll
                for (int kt = 0; kt < NT; ++kt)
                    if (kt < nkt)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int key = kt * 16 + 4 * g + r;
                            if (key < p.n_k) pr[key] = st[kt][r];
                        }
            }
        } else {
            const float lse = p.lse[bh * p.n_q + q_row];
            f32x4 dof[DS];
            load_frag<DS>(dof, p.dO + ((long)b * p.n_q + q_row) * p.lddo + h * D + 4 * g);
            f32x4 dp[NT];
            float dsum = 0.f;
#pragma unroll
            for (int kt = 0; kt < NT; ++kt) {
                dp[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (kt < nkt) {
                    f32x4 vf[DS];
                    load_frag_lds<DS>(vf, sV + (kt * 16 + c) * LD + 4 * g);
                    f32x4 acc = dot_tile<DS>(vf, dof);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int key = min(kt * 16 + 4 * g + r, p.n_k - 1);
                        const float pv = expf(st[kt][r] - lse);
                        float d = acc[r];
                        if (drop) d = vb_keep(p.seed, (uint64_t)(prow + key), p.drop_p) ? d * p.drop_scale : 0.f;
                        st[kt][r] = pv;
                        acc[r] = d;
                        dsum += pv * d;
                    }
                    dp[kt] = acc;
                }
            }
            dsum = group_sum(dsum);
            if (g == 0 && qt * 16 + c < p.n_q) p.dvec[bh * p.n_q + qt * 16 + c] = dsum;
#pragma unroll
            for (int kt = 0; kt < NT; ++kt)
                if (kt < nkt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) st[kt][r] = st[kt][r] * (dp[kt][r] - dsum) * p.scale;
        }

        // forward: O = P V;  backward: dQ = dS K   (B operand rows from LDS, zero rows past n_k)
        const float* rb = (BWD ? sK : sV) + c;
        f32x4 oacc[DS];
#pragma unroll
        for (int dt = 0; dt < DS; ++dt) oacc[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kt = 0; kt < NT; ++kt) {
            if (kt < nkt) {
                f32x4 av, bv[DS];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float* rp = rb + (kt * 16 + 4 * g + r) * LD;
                    av[r] = st[kt][r];
#pragma unroll
                    for (int dt = 0; dt < DS; ++dt) bv[dt][r] = rp[16 * dt];
                }
                mma_rank4<DS>(oacc, av, bv);
            }
        }
        io_t* obase = BWD ? p.dQ : p.O;
        const long ldo = BWD ? p.lddq : p.ldo;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int q = qt * 16 + 4 * g + r;
            if (q < p.n_q) {
                io_t* op = obase + ((long)b * p.n_q + q) * ldo + h * D + c;
                store_cols<DS>(op, c, oacc, r);
            }
        }
    }
}

// Backward pass 2, short sequences: Q and dO of the (sample, head) staged in LDS, one wave per 16 keys.
template <int D>
__global__ __launch_bounds__(256) void attn_bwd_kv_lds_kernel(const AttnP p_in) {
    AttnP p = p_in;
    p.seed = vb_seed_with_epoch(p_in.seed, p_in.epoch);
    extern __shared__ __attribute__((aligned(16))) float smem_att[];
    constexpr int LD = D + 4, DS = D / 16;
    float* sQ = smem_att;
    float* sO = smem_att + LDS_MAX_ROWS * LD;
    const long bh = blockIdx.x;
    const int h = (int)(bh % p.heads), b = (int)(bh / p.heads);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 15, g = lane >> 4;
    const bool drop = p.drop_p > 0.f;

    stage_rows2<D>(sQ, p.Q + (long)b * p.n_q * p.ldq + h * D, p.ldq, sO, p.dO + (long)b * p.n_q * p.lddo + h * D, p.lddo, p.n_q);
    __syncthreads();
    const float* lse = p.lse + bh * p.n_q;
    const float* dvec = p.dvec + bh * p.n_q;

    for (int kt = wave; kt < p.n_kt; kt += 4) {
        const int key = kt * 16 + c;
        const int k_row = min(key, p.n_k - 1);
        const bool key_ok = key < p.n_k;
        f32x4 kf[DS], vf[DS];
        load_frag<DS>(kf, p.K + ((long)b * p.n_k + k_row) * p.ldk + h * D + 4 * g);
        load_frag<DS>(vf, p.V + ((long)b * p.n_k + k_row) * p.ldv + h * D + 4 * g);
        const float madd = p.mask != nullptr ? p.mask[(long)b * p.n_k + k_row] : 0.f;
        f32x4 dk[DS], dv[DS];
#pragma unroll
        for (int dt = 0; dt < DS; ++dt) {
            dk[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
            dv[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        for (int qt = 0; qt < p.n_qt; ++qt) {
            f32x4 qf[DS], dof[DS];
            load_frag_lds<DS>(qf, sQ + (qt * 16 + c) * LD + 4 * g);     // rows past n_q are zero
            load_frag_lds<DS>(dof, sO + (qt * 16 + c) * LD + 4 * g);
            const f32x4 s = dot_tile<DS>(qf, kf);
            const f32x4 dpr = dot_tile<DS>(dof, vf);
            f32x4 pd, dsv;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int q = qt * 16 + 4 * g + r;
                float dsr = 0.f, pdr = 0.f;
                if (q < p.n_q && key_ok) {
                    const float sv = __fadd_rn(__fmul_rn(s[r], p.scale), madd);
                    const float pv = expf(sv - lse[q]);
                    float d = dpr[r];
                    pdr = pv;
                    if (drop) {
                        const bool keep = vb_keep(p.seed, (uint64_t)((bh * p.n_q + q) * p.n_k + key), p.drop_p);
                        d = keep ? d * p.drop_scale : 0.f;
                        pdr = keep ? pv * p.drop_scale : 0.f;
                    }
                    dsr = pv * (d - dvec[q]) * p.scale;
                }
                pd[r] = pdr;
                dsv[r] = dsr;
            }
            f32x4 dov[DS], qv[DS];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float* dop = sO + (qt * 16 + 4 * g + r) * LD + c;
                const float* qp = sQ + (qt * 16 + 4 * g + r) * LD + c;
#pragma unroll
                for (int dt = 0; dt < DS; ++dt) {
                    dov[dt][r] = dop[16 * dt];
                    qv[dt][r] = qp[16 * dt];
                }
            }
            mma_rank4<DS>(dv, pd, dov);
            mma_rank4<DS>(dk, dsv, qv);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int kk = kt * 16 + 4 * g + r;
            if (kk < p.n_k) {
                io_t* kp = p.dK + ((long)b * p.n_k + kk) * p.lddk + h * D + c;
                io_t* vp = p.dV + ((long)b * p.n_k + kk) * p.lddv + h * D + c;
                store_cols<DS>(kp, c, dk, r);
                store_cols<DS>(vp, c, dv, r);
            }
        }
    }
}

// Backward in ONE launch, short sequences (round 2): K, V, Q and dO of the (sample, head) are staged once - exactly
// n rows each, unpadded, [n][D + 4] - so the block moves 4 reads + 3 writes of H floats per token row instead of the 8
// + 3 of the two-pass version. Phase 1 (one wave per 16 queries): dP, D = rowsum(P dP) (kept in LDS), dS, dQ = dS K.
// Phase 2 (one wave per 16 keys): dV = P^T dO, dK = dS^T Q. Rows past the end of a tile are read CLAMPED to the last
// real row (no zero padding, which would cost a third block of LDS at head_dim 128): every such value only ever
// meets a factor that is exactly 0 (masked score / probability), and the clamped data is finite model data.
template <int D>
__global__ __launch_bounds__(256) void attn_bwd_fused_lds_kernel(const AttnP p_in, const int ps_off) {
    AttnP p = p_in;
    p.seed = vb_seed_with_epoch(p_in.seed, p_in.epoch);
    extern __shared__ __attribute__((aligned(16))) float smem_att[];
    constexpr int LD = D + 4, DS = D / 16, NT = LDS_MAX_ROWS / 16, V4 = D / 4;
    float* sK = smem_att;
    float* sV = sK + p.n_k * LD;
    float* sQ = sV + p.n_k * LD;
    float* sO = sQ + p.n_q * LD;
    float* sD = sO + p.n_q * LD;          // [n_q] D vector (unused since phase 2 reads dS; kept for the lse-free layout)
    // phase-1 results handed to phase 2 through LDS: Pd[q][key] (probabilities incl. the dropout mask / scale) and
    // dS[q][key], row stride LDP. They live at float offset ps_off: over the V block when they fit there (V is dead
    // after phase 1 - this keeps head_dim 128 at two blocks per CU), else behind the D vector.
    const int LDP = (p.n_k + 3) / 4 * 4 + 4;
    float* sP = smem_att + ps_off;
    float* sS = sP + p.n_q * LDP;
    const long bh = blockIdx.x;
    const int h = (int)(bh % p.heads), b = (int)(bh / p.heads);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 15, g = lane >> 4;
    const int nkt = p.n_kt;
    const bool drop = p.drop_p > 0.f;

    {   // stage the four operand blocks (coalesced 16-byte loads, rows of D floats). ALL loads of a thread are issued
        // before the first LDS store: with runtime loop bounds the compiler kept one load -> wait -> store round per
        // iteration, i.e. ~10 dependent trips to HBM per block with only 8 waves per CU to hide them (round 3).
        const io_t* gk = p.K + (long)b * p.n_k * p.ldk + h * D;
        const io_t* gv = p.V + (long)b * p.n_k * p.ldv + h * D;
        const io_t* gq = p.Q + (long)b * p.n_q * p.ldq + h * D;
        const io_t* go = p.dO + (long)b * p.n_q * p.lddo + h * D;
        constexpr int IT = (LDS_MAX_ROWS * V4 + 255) / 256;
        f32x4 rk[IT], rv[IT], rq[IT], ro[IT];
#pragma unroll
        for (int i = 0; i < IT; ++i) {
            const int f = threadIdx.x + 256 * i;
            const int r = f / V4, c4 = (f % V4) * 4;
            if (r < p.n_k) {
                rk[i] = ld4(gk + (long)r * p.ldk + c4);
                rv[i] = ld4(gv + (long)r * p.ldv + c4);
            }
            if (r < p.n_q) {
                rq[i] = ld4(gq + (long)r * p.ldq + c4);
                ro[i] = ld4(go + (long)r * p.lddo + c4);
            }
        }
#pragma unroll
        for (int i = 0; i < IT; ++i) {
            const int f = threadIdx.x + 256 * i;
            const int r = f / V4, c4 = (f % V4) * 4;
            if (r < p.n_k) {
                *reinterpret_cast<f32x4*>(sK + r * LD + c4) = rk[i];
                *reinterpret_cast<f32x4*>(sV + r * LD + c4) = rv[i];
            }
            if (r < p.n_q) {
                *reinterpret_cast<f32x4*>(sQ + r * LD + c4) = rq[i];
                *reinterpret_cast<f32x4*>(sO + r * LD + c4) = ro[i];
            }
        }
    }
    __syncthreads();
    const float* mrow = p.mask != nullptr ? p.mask + (long)b * p.n_k : nullptr;
    const float* lse_g = p.lse + bh * p.n_q;

    // ---- phase 1: dQ, Pd and dS; one wave per 16 queries (n_q <= 48: at most one tile per wave) --------------------
    const int qt = wave;
    const bool has_q = qt < p.n_qt;
    f32x4 pdm[NT], dsm[NT];      // Pd and dS of this wave's 16 queries: lane (c = query, g) holds keys 16 kt + 4 g + r
    if (has_q) {
        const int q_row = min(qt * 16 + c, p.n_q - 1);
        f32x4 qf[DS], dof[DS];
        load_frag_lds<DS>(qf, sQ + q_row * LD + 4 * g);
        load_frag_lds<DS>(dof, sO + q_row * LD + 4 * g);
        const long prow = (bh * p.n_q + q_row) * p.n_k;
        const float lse = lse_g[q_row];
        f32x4 dp[NT];
        float dsum = 0.f;
#pragma unroll
        for (int kt = 0; kt < NT; ++kt) {
            pdm[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
            dsm[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
            dp[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (kt < nkt) {
                const int krow = min(kt * 16 + c, p.n_k - 1);
                f32x4 kf[DS], vf[DS];
                load_frag_lds<DS>(kf, sK + krow * LD + 4 * g);
                load_frag_lds<DS>(vf, sV + krow * LD + 4 * g);
                const f32x4 sacc = dot_tile<DS>(kf, qf);
                f32x4 dacc = dot_tile<DS>(vf, dof);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int key = kt * 16 + 4 * g + r;
                    float pv = 0.f, d = 0.f, pd = 0.f;
                    if (key < p.n_k) {
                        float sv = __fmul_rn(sacc[r], p.scale);
                        if (mrow != nullptr) sv = __fadd_rn(sv, mrow[key]);
                        pv = expf(sv - lse);
                        d = dacc[r];
                        pd = pv;
                        if (drop) {
                            const bool keep = vb_keep(p.seed, (uint64_t)(prow + key), p.drop_p);
                            d = keep ? d * p.drop_scale : 0.f;
                            pd = keep ? pv * p.drop_scale : 0.f;
                        }
                    }
                    dsm[kt][r] = pv;
                    pdm[kt][r] = pd;
                    dacc[r] = d;
                    dsum += pv * d;
                }
                dp[kt] = dacc;
            }
        }
        dsum = group_sum(dsum);
#pragma unroll
        for (int kt = 0; kt < NT; ++kt)
            if (kt < nkt)
#pragma unroll
                for (int r = 0; r < 4; ++r) dsm[kt][r] = dsm[kt][r] * (dp[kt][r] - dsum) * p.scale;
        // dQ = dS K  (B operand rows from LDS, clamped past n_k where dS is exactly 0)
        f32x4 oacc[DS];
#pragma unroll
        for (int dt = 0; dt < DS; ++dt) oacc[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kt = 0; kt < NT; ++kt) {
            if (kt < nkt) {
                f32x4 av, bv[DS];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float* rp = sK + min(kt * 16 + 4 * g + r, p.n_k - 1) * LD + c;
                    av[r] = dsm[kt][r];
#pragma unroll
                    for (int dt = 0; dt < DS; ++dt) bv[dt][r] = rp[16 * dt];
                }
                mma_rank4<DS>(oacc, av, bv);
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int q = qt * 16 + 4 * g + r;
            if (q < p.n_q) {
                io_t* op = p.dQ + ((long)b * p.n_q + q) * p.lddq + h * D + c;
                store_cols<DS>(op, c, oacc, r);
            }
        }
    }
    __syncthreads();            // every wave is done reading K and V: their LDS may be overwritten
    if (has_q && qt * 16 + c < p.n_q) {
        float* pr = sP + (qt * 16 + c) * LDP + 4 * g;
        float* sr = sS + (qt * 16 + c) * LDP + 4 * g;
#pragma unroll
        for (int kt = 0; kt < NT; ++kt)
            if (kt < nkt && kt * 16 + 4 * g < LDP) {
                *reinterpret_cast<f32x4*>(pr + kt * 16) = pdm[kt];
                *reinterpret_cast<f32x4*>(sr + kt * 16) = dsm[kt];
            }
    }
    __syncthreads();

    // ---- phase 2: dV = Pd^T dO, dK = dS^T Q; one wave per 16 keys ---------------------------------------------------
    for (int kt = wave; kt < p.n_kt; kt += 4) {
        const int key = min(kt * 16 + c, p.n_k - 1);      // columns past n_k: clamped, their results are not stored
        f32x4 dk[DS], dv[DS];
#pragma unroll
        for (int dt = 0; dt < DS; ++dt) {
            dk[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
            dv[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        for (int q4 = 0; q4 < p.n_q; q4 += 16) {
            f32x4 pd, dsv, dov[DS], qv[DS];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int q = q4 + 4 * g + r;
                const int qr = min(q, p.n_q - 1);
                // A operands: Pd^T / dS^T element (key = c, q); queries past n_q contribute exactly 0
                pd[r] = q < p.n_q ? sP[qr * LDP + key] : 0.f;
                dsv[r] = q < p.n_q ? sS[qr * LDP + key] : 0.f;
                const float* dop = sO + qr * LD + c;
                const float* qp = sQ + qr * LD + c;
#pragma unroll
                for (int dt = 0; dt < DS; ++dt) {
                    dov[dt][r] = dop[16 * dt];
                    qv[dt][r] = qp[16 * dt];
                }
            }
            mma_rank4<DS>(dv, pd, dov);
            mma_rank4<DS>(dk, dsv, qv);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int kk = kt * 16 + 4 * g + r;
            if (kk < p.n_k) {
                io_t* kp = p.dK + ((long)b * p.n_k + kk) * p.lddk + h * D + c;
                io_t* vp = p.dV + ((long)b * p.n_k + kk) * p.lddv + h * D + c;
                store_cols<DS>(kp, c, dk, r);
                store_cols<DS>(vp, c, dv, r);
            }
        }
    }
}

template <int D>
int launch_bwd_fused_lds(hipStream_t st, const AttnP& p) {
    const int ldp = (p.n_k + 3) / 4 * 4 + 4;
    const int base = (2 * p.n_k + 2 * p.n_q) * (D + 4) + LDS_MAX_ROWS;       // K | V | Q | dO | D vector (floats)
    const bool over_v = 2 * p.n_q * ldp <= p.n_k * (D + 4);                   // Pd | dS fit over the dead V block
    const int ps_off = over_v ? p.n_k * (D + 4) : base;
    const int bytes = (over_v ? base : base + 2 * p.n_q * ldp) * 4;
    // up to 4 x 48 x 132 floats = 101 KiB at head_dim 128: above the 64 KiB a kernel gets by default
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_fused_lds_kernel<D>),
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (attr != hipSuccess) return (int)attr;
    hipLaunchKernelGGL((attn_bwd_fused_lds_kernel<D>), dim3((unsigned)(p.batch * p.heads)), dim3(256), bytes, st, p, ps_off);
    VB_LAUNCH_CHECK();
    return 0;
}

inline bool use_lds_path(const AttnP& p) {
    static const int on = [] { const char* e = getenv("VB_ATTN_LDS"); return e ? atoi(e) : 1; }();
    return on && p.n_q <= LDS_MAX_ROWS && p.n_k <= LDS_MAX_ROWS && p.q_bstride == p.n_q && p.kv_bstride == p.n_k;
}

template <int D, bool BWD>
int launch_q_lds(hipStream_t st, const AttnP& p) {
    constexpr int bytes = 2 * LDS_MAX_ROWS * (D + 4) * 4;
    hipLaunchKernelGGL((attn_q_lds_kernel<D, BWD>), dim3((unsigned)(p.batch * p.heads)), dim3(256), bytes, st, p);
    VB_LAUNCH_CHECK();
    return 0;
}

template <int D>
int launch_kv_lds(hipStream_t st, const AttnP& p) {
    constexpr int bytes = 2 * LDS_MAX_ROWS * (D + 4) * 4;
    hipLaunchKernelGGL((attn_bwd_kv_lds_kernel<D>), dim3((unsigned)(p.batch * p.heads)), dim3(256), bytes, st, p);
    VB_LAUNCH_CHECK();
    return 0;
}

template <int D, bool BWD>
int launch_q(hipStream_t st, const AttnP& p) {
    if (use_lds_path(p) && p.dvec_mode == VB_DVEC_COMPUTE) return launch_q_lds<D, BWD>(st, p);   // (key chunks: generic kernel)
    dim3 block(256), grid((unsigned)((p.total + 3) / 4));
    if (p.n_kt <= 3) hipLaunchKernelGGL((attn_q_kernel<D, 3, BWD>), grid, block, 0, st, p);
    else if (p.n_kt <= 8) hipLaunchKernelGGL((attn_q_kernel<D, 8, BWD>), grid, block, 0, st, p);
    else if (p.n_kt <= 20) hipLaunchKernelGGL((attn_q_kernel<D, 20, BWD>), grid, block, 0, st, p);
    else return VB_E_RANGE;
    VB_LAUNCH_CHECK();
    return 0;
}

template <int D>
int launch_kv(hipStream_t st, const AttnP& p) {
    if (use_lds_path(p)) return launch_kv_lds<D>(st, p);
    dim3 block(256), grid((unsigned)((p.total + 3) / 4));
    hipLaunchKernelGGL((attn_bwd_kv_kernel<D>), grid, block, 0, st, p);
    VB_LAUNCH_CHECK();
    return 0;
}

#ifdef VB_ATTN_BF16
typedef vb_attention_bf16_args attn_args_t;
typedef vb_attention_bf16_grads attn_grads_t;
#define VB_ATTN_FWD vb_attention_fwd_bf16
#define VB_ATTN_BWD vb_attention_bwd_bf16
constexpr unsigned IO_ALIGN = 7u;       // 4 bf16 per load
#else
typedef vb_attention_args attn_args_t;
typedef vb_attention_grads attn_grads_t;
#define VB_ATTN_FWD vb_attention_fwd
#define VB_ATTN_BWD vb_attention_bwd
constexpr unsigned IO_ALIGN = 15u;
#endif
inline bool io_aligned(const void* q) { return (reinterpret_cast<uintptr_t>(q) & IO_ALIGN) == 0; }

int fill_common(AttnP& p, const attn_args_t* a) {
    if (a == nullptr || a->Q == nullptr || a->K == nullptr || a->V == nullptr) return VB_E_BADARG;
    if (a->batch <= 0 || a->heads <= 0 || a->n_q <= 0 || a->n_k <= 0) return VB_E_BADARG;
    if (a->n_k > VB_MAX_KEYS) return VB_E_RANGE;
    if ((a->q_batch != a->batch && a->q_batch != 1) || (a->kv_batch != a->batch && a->kv_batch != 1))
        return VB_E_BADARG;
    if (!(a->dropout_p >= 0.f && a->dropout_p < 1.f)) return VB_E_BADARG;
    if ((a->ldq | a->ldk | a->ldv) % 4 != 0 || !io_aligned(a->Q) || !io_aligned(a->K) || !io_aligned(a->V))
        return VB_E_ALIGN;
    p.batch = a->batch; p.heads = a->heads; p.n_q = a->n_q; p.n_k = a->n_k;
    p.n_qt = (a->n_q + 15) / 16;
    p.n_kt = (a->n_k + 15) / 16;
    p.q_bstride = a->q_batch == 1 && a->batch > 1 ? 0 : a->n_q;
    p.kv_bstride = a->kv_batch == 1 && a->batch > 1 ? 0 : a->n_k;
    p.m_bstride = p.kv_bstride;
    p.Q = a->Q; p.ldq = a->ldq; p.K = a->K; p.ldk = a->ldk; p.V = a->V; p.ldv = a->ldv;
    p.mask = a->mask_add; p.scale = a->scale; p.lse = a->lse;
    p.drop_p = a->dropout_p; p.drop_scale = 1.0f / (1.0f - a->dropout_p); p.seed = a->seed;
    p.epoch = a->dropout_p > 0.f ? vb_seed_epoch() : nullptr;
    return 0;
}

}  // namespace

extern "C" int VB_ATTN_FWD(void* stream, const attn_args_t* a) {
    AttnP p{};
    if (int e = fill_common(p, a)) return e;
    if (a->O == nullptr) return VB_E_BADARG;
    p.O = a->O; p.ldo = a->ldo; p.probs = a->probs;
    p.total = (long)a->batch * a->heads * p.n_qt;
    hipStream_t st = static_cast<hipStream_t>(stream);
    switch (a->head_dim) {
        case 32: return launch_q<32, false>(st, p);
        case 64: return launch_q<64, false>(st, p);
        case 128: return launch_q<128, false>(st, p);
        default: return VB_E_RANGE;
    }
}

extern "C" int VB_ATTN_BWD(void* stream, const attn_args_t* a, const attn_grads_t* gr) {
    AttnP p{};
    if (int e = fill_common(p, a)) return e;
    if (gr == nullptr || gr->dO == nullptr || gr->dQ == nullptr || gr->dK == nullptr || gr->dV == nullptr ||
        gr->dvec == nullptr || a->lse == nullptr)
        return VB_E_BADARG;
    if (a->q_batch != a->batch || a->kv_batch != a->batch) return VB_E_BADARG;  // no broadcast in training
    if (gr->lddo % 4 != 0 || !io_aligned(gr->dO)) return VB_E_ALIGN;
    p.dO = gr->dO; p.lddo = gr->lddo;
    p.dQ = gr->dQ; p.lddq = gr->lddq; p.dK = gr->dK; p.lddk = gr->lddk; p.dV = gr->dV; p.lddv = gr->lddv;
    p.dvec = gr->dvec;
    p.dvec_mode = gr->dvec_mode;
    if (p.dvec_mode < VB_DVEC_COMPUTE || p.dvec_mode > VB_DVEC_GIVEN) return VB_E_BADARG;
    hipStream_t st = static_cast<hipStream_t>(stream);
    int e = 0;
    static const int fused = [] { const char* ev = getenv("VB_ATTN_FUSED_BWD"); return ev ? atoi(ev) : 1; }();
    if (p.dvec_mode == VB_DVEC_COMPUTE && fused && use_lds_path(p) && (gr->lddq | gr->lddk | gr->lddv) % 4 == 0) {
        // short sequences: the whole backward in one launch (K, V, Q, dO staged once)
        switch (a->head_dim) {
            case 32: return launch_bwd_fused_lds<32>(st, p);
            case 64: return launch_bwd_fused_lds<64>(st, p);
            case 128: return launch_bwd_fused_lds<128>(st, p);
            default: return VB_E_RANGE;
        }
    }
    p.total = (long)a->batch * a->heads * p.n_qt;
    switch (a->head_dim) {
        case 32: e = launch_q<32, true>(st, p); break;
        case 64: e = launch_q<64, true>(st, p); break;
        case 128: e = launch_q<128, true>(st, p); break;
        default: return VB_E_RANGE;
    }
    if (e) return e;
    if (p.dvec_mode == VB_DVEC_ACCUMULATE) return 0;
    p.total = (long)a->batch * a->heads * p.n_kt;
    switch (a->head_dim) {
        case 32: return launch_kv<32>(st, p);
        case 64: return launch_kv<64>(st, p);
        default: return launch_kv<128>(st, p);
    }
}
