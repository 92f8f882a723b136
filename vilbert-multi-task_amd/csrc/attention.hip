// Fused short-sequence attention, forward:  O = softmax(Q K^T * scale + mask) V  per (sample, head).
//
// ViLBERT sequences are short (36 tokens / 36 regions pre-training, <= 306 in the task table), so
// this is not a flash-attention problem: one 64-lane wave owns a 16-query tile of one
// (sample, head) and keeps the whole score row block in registers - no LDS, no S x S round trip
// through HBM, no head split / merge copies (Q, K, V are read straight out of the fused
// [q | k | v] projection, O is written token-major).
//
// MFMA: v_mfma_f32_16x16x4_f32 (exact fp32). Lane l = (c = l & 15, g = l >> 4) supplies
// A[i = c][k = g] and B[k = g][j = c]; D[row = 4g + r][col = c], r = 0..3.
//  * scores are computed TRANSPOSED, S^T = K Q^T (A = key rows, B = query rows, contraction over
//    head_dim with lane group g owning dims 16s + 4g + e): lane (c, g) ends up holding
//    S[q = c][key = 16 kt + 4g + r] - a fixed query per lane, so the softmax row reduction is
//    in-register plus two cross-lane steps (xor 16, xor 32);
//  * that is exactly the A-operand layout of P V (A[i = q = c][k = key], lane group g owning keys
//    16 kt + 4g + r), so P never leaves its registers; B = V[key][d] is read with plain dword
//    loads (16 lanes = 64 contiguous bytes).
#include "common.h"

namespace {

struct AttnP {
    int batch, heads, n_q, n_k, n_qt;
    long q_bstride, kv_bstride, m_bstride;  // rows (or mask floats) per sample; 0 = broadcast
    const float* Q; long ldq;
    const float* K; long ldk;
    const float* V; long ldv;
    const float* mask;
    float* O; long ldo;
    float* probs;
    float scale;
    long total;  // batch * heads * n_qt wave items
};

template <int D, int NT>
__global__ __launch_bounds__(256) void attn_fwd_kernel(const AttnP p) {
    const int lane = threadIdx.x & 63;
    const long item = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (item >= p.total) return;
    const int qt = (int)(item % p.n_qt);
    const long bh = item / p.n_qt;
    const int h = (int)(bh % p.heads);
    const int b = (int)(bh / p.heads);
    const int c = lane & 15, g = lane >> 4;
    constexpr int DS = D / 16;
    const int nkt = (p.n_k + 15) >> 4;

    // Query fragment (B operand of S^T = K Q^T): Q[q = 16 qt + c][16 s + 4 g + e].
    const int q_row = min(qt * 16 + c, p.n_q - 1);
    const float* qp = p.Q + ((long)b * p.q_bstride + q_row) * p.ldq + h * D + 4 * g;
    f32x4 qf[DS];
#pragma unroll
    for (int s = 0; s < DS; ++s) qf[s] = *reinterpret_cast<const f32x4*>(qp + 16 * s);

    const float* kbase = p.K + (long)b * p.kv_bstride * p.ldk + h * D + 4 * g;
    const float* mrow = p.mask != nullptr ? p.mask + (long)b * p.m_bstride : nullptr;

    f32x4 st[NT];
    float mx = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < NT; ++kt) {
        if (kt < nkt) {
            const int k_row = min(kt * 16 + c, p.n_k - 1);
            const float* kp = kbase + (long)k_row * p.ldk;
            f32x4 kf[DS];
#pragma unroll
            for (int s = 0; s < DS; ++s) kf[s] = *reinterpret_cast<const f32x4*>(kp + 16 * s);
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < DS; ++s)
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[s][e], qf[s][e], acc, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int key = kt * 16 + 4 * g + r;
                float v = -INFINITY;
                if (key < p.n_k) {
                    // vilbert.py:435-439: scores / sqrt(d) + mask
                    v = acc[r] * p.scale;
                    if (mrow != nullptr) v += mrow[key];
                }
                acc[r] = v;
                mx = fmaxf(mx, v);
            }
            st[kt] = acc;
        } else {
            st[kt] = f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        }
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));

    float sum = 0.f;
#pragma unroll
    for (int kt = 0; kt < NT; ++kt) {
        if (kt < nkt) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float e = expf(st[kt][r] - mx);
                st[kt][r] = e;
                sum += e;
            }
        }
    }
    sum += __shfl_xor(sum, 16, 64);
    sum += __shfl_xor(sum, 32, 64);
    const float inv = 1.0f / sum;
#pragma unroll
    for (int kt = 0; kt < NT; ++kt)
        if (kt < nkt) st[kt] *= inv;

    if (p.probs != nullptr && qt * 16 + c < p.n_q) {
        float* pr = p.probs + (((long)b * p.heads + h) * p.n_q + qt * 16 + c) * p.n_k;
#pragma unroll
        for (int kt = 0; kt < NT; ++kt)
            if (kt < nkt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int key = kt * 16 + 4 * g + r;
                    if (key < p.n_k) pr[key] = st[kt][r];
                }
    }

    // O[q][d] = sum_key P[q][key] V[key][d]; one 16-wide d tile per accumulator.
    const float* vbase = p.V + (long)b * p.kv_bstride * p.ldv + h * D + c;
    f32x4 oacc[DS];
#pragma unroll
    for (int dt = 0; dt < DS; ++dt) oacc[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kt = 0; kt < NT; ++kt) {
        if (kt < nkt) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int key = min(kt * 16 + 4 * g + r, p.n_k - 1);  // P is 0 past n_k
                const float* vp = vbase + (long)key * p.ldv;
                float vv[DS];
#pragma unroll
                for (int dt = 0; dt < DS; ++dt) vv[dt] = vp[16 * dt];
#pragma unroll
                for (int dt = 0; dt < DS; ++dt)
                    oacc[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(st[kt][r], vv[dt], oacc[dt], 0, 0, 0);
            }
        }
    }

    // D[row = q = 4g + r][col = d = 16 dt + c]
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int q = qt * 16 + 4 * g + r;
        if (q < p.n_q) {
            float* op = p.O + ((long)b * p.n_q + q) * p.ldo + h * D + c;
#pragma unroll
            for (int dt = 0; dt < DS; ++dt) op[16 * dt] = oacc[dt][r];
        }
    }
}

template <int D>
int launch_attn(hipStream_t st, const AttnP& p) {
    const int nkt = (p.n_k + 15) / 16;
    dim3 block(256), grid((unsigned)((p.total + 3) / 4));
    if (nkt <= 3) hipLaunchKernelGGL((attn_fwd_kernel<D, 3>), grid, block, 0, st, p);
    else if (nkt <= 8) hipLaunchKernelGGL((attn_fwd_kernel<D, 8>), grid, block, 0, st, p);
    else if (nkt <= 20) hipLaunchKernelGGL((attn_fwd_kernel<D, 20>), grid, block, 0, st, p);
    else return VB_E_RANGE;
    VB_LAUNCH_CHECK();
    return 0;
}

}  // namespace

extern "C" int vb_attention_fwd(void* stream, const vb_attention_args* a) {
    if (a == nullptr || a->Q == nullptr || a->K == nullptr || a->V == nullptr || a->O == nullptr)
        return VB_E_BADARG;
    if (a->batch <= 0 || a->heads <= 0 || a->n_q <= 0 || a->n_k <= 0) return VB_E_BADARG;
    if (a->n_k > VB_MAX_KEYS) return VB_E_RANGE;
    if ((a->q_batch != a->batch && a->q_batch != 1) || (a->kv_batch != a->batch && a->kv_batch != 1))
        return VB_E_BADARG;
    if ((a->ldq | a->ldk) % 4 != 0 || !vb_aligned16(a->Q) || !vb_aligned16(a->K)) return VB_E_ALIGN;
    AttnP p{};
    p.batch = a->batch; p.heads = a->heads; p.n_q = a->n_q; p.n_k = a->n_k;
    p.n_qt = (a->n_q + 15) / 16;
    p.q_bstride = a->q_batch == 1 && a->batch > 1 ? 0 : a->n_q;
    p.kv_bstride = a->kv_batch == 1 && a->batch > 1 ? 0 : a->n_k;
    p.m_bstride = p.kv_bstride;
    p.Q = a->Q; p.ldq = a->ldq; p.K = a->K; p.ldk = a->ldk; p.V = a->V; p.ldv = a->ldv;
    p.mask = a->mask_add; p.O = a->O; p.ldo = a->ldo; p.probs = a->probs; p.scale = a->scale;
    p.total = (long)a->batch * a->heads * p.n_qt;
    hipStream_t st = static_cast<hipStream_t>(stream);
    switch (a->head_dim) {
        case 32: return launch_attn<32>(st, p);
        case 64: return launch_attn<64>(st, p);
        case 128: return launch_attn<128>(st, p);
        default: return VB_E_RANGE;
    }
}
