// Whole-layer launchers (round 6; include/vilbert_hip.h "Whole-layer entry points"): vb_layer_fwd / vb_layer_bwd enqueue the
// kernel sequence of a BertLayer / BertImageLayer (reference vilbert.py:527-533, 688-694) or of a BertConnectionLayer
// (:871-900) with ONE call across the C ABI. Host code only: every launch below is one of the per-op entry points of this
// library (same kernels, same arithmetic, same order as the Python autograd nodes of vilbert/autograd_ops.py) - what
// disappears is the Python between them (argument structs, tensor allocation, autograd bookkeeping per op): ~55 ops per
// connection layer and pass became one.
#include "common.h"
#include <mutex>
#include <unordered_map>

namespace {

constexpr bool kF32 = false, kB16 = true;

#define VB_TRY(expr)               \
    do {                           \
        const int _e = (expr);     \
        if (_e != 0) return _e;    \
    } while (0)

template <bool B16>
struct Elem { static constexpr size_t size = B16 ? 2 : 4; };

// byte offset of column c in a row-major matrix of this element type
template <bool B16>
inline const void* col(const void* p, long c) { return static_cast<const char*>(p) + c * (long)Elem<B16>::size; }
template <bool B16>
inline void* col(void* p, long c) { return static_cast<char*>(p) + c * (long)Elem<B16>::size; }

bool has_bias(const vb_layer_linear& L) { return L.bias[0] != nullptr; }
bool wants_wgrad(const vb_layer_linear& L) { return L.dw[0] != nullptr; }

// ---- y = act(A . W^T + b) [dropout] [+ residual]; act_grad optional ---------------------------------------------------
template <bool B16>
int linear(void* st, long M, const vb_layer_linear& L, const void* A, void* C, int act, const void* residual, float p,
           uint64_t seed, void* act_grad) {
    const long N = (long)L.nseg * L.seg_n;
    if (B16) {
        vb_linear_bf16_args a{};
        a.A = static_cast<const uint16_t*>(A); a.lda = L.K;
        a.W = L.w16; a.ldw = L.K;
        if (has_bias(L)) {
            a.bias_segments = L.nseg;
            for (int s = 0; s < L.nseg; ++s) a.bias[s] = L.bias[s];
        }
        a.C = static_cast<uint16_t*>(C); a.ldc = N;
        a.residual = static_cast<const uint16_t*>(residual); a.ldr = N;
        a.act_grad = static_cast<uint16_t*>(act_grad); a.ldg = N;
        a.M = M; a.N = N; a.K = L.K;
        a.act = act; a.dropout_p = p; a.seed = seed;
        return vb_linear_bf16(st, &a);
    }
    vb_linear_args a{};
    a.M = (int32_t)M; a.K = L.K; a.nseg = L.nseg; a.seg_n = L.seg_n;
    a.A = static_cast<const float*>(A); a.lda = L.K;
    for (int s = 0; s < L.nseg; ++s) { a.W[s] = L.w[s]; a.bias[s] = L.bias[s]; }
    a.ldw = L.K;
    a.C = static_cast<float*>(C); a.ldc = N;
    a.residual = static_cast<const float*>(residual); a.ldr = N;
    a.act_grad = static_cast<float*>(act_grad); a.ldg = N;
    a.act = act; a.dropout_p = p; a.seed = seed;
    return vb_linear_fwd(st, &a);
}

// ---- dX = (dY . W + residual) * mul -------------------------------------------------------------------------------
template <bool B16>
int dgrad(void* st, long M, const vb_layer_linear& L, const void* dY, void* dX, const void* residual, const void* mul) {
    const long N = (long)L.nseg * L.seg_n;
    if (B16) {
        vb_linear_bf16_args a{};
        a.A = static_cast<const uint16_t*>(dY); a.lda = N;
        a.W = L.wt16; a.ldw = N;
        a.C = static_cast<uint16_t*>(dX); a.ldc = L.K;
        a.residual = static_cast<const uint16_t*>(residual); a.ldr = L.K;
        a.mul = static_cast<const uint16_t*>(mul); a.ldm = L.K;
        a.M = M; a.N = L.K; a.K = N;
        return vb_linear_bf16(st, &a);
    }
    vb_linear_bwd_input_args a{};
    a.M = (int32_t)M; a.K = L.K; a.nseg = L.nseg; a.seg_n = L.seg_n;
    a.dY = static_cast<const float*>(dY); a.ldy = N;
    for (int s = 0; s < L.nseg; ++s) a.W[s] = L.w[s];
    a.ldw = L.K;
    a.dX = static_cast<float*>(dX); a.ldx = L.K;
    a.residual = static_cast<const float*>(residual); a.ldr = L.K;
    a.mul = static_cast<const float*>(mul); a.ldm = L.K;
    return vb_linear_bwd_input(st, &a);
}

// ---- dW_s += dY[:, s]^T . X, db_s += colsum(dY[:, s]) ----------------------------------------------------------------
template <bool B16>
int wgrad(void* st, long M, const vb_layer_linear& L, const void* dY, const void* X) {
    if (!wants_wgrad(L)) return 0;
    const long N = (long)L.nseg * L.seg_n;
    if (B16) {
        vb_wgrad_bf16_args a{};
        a.dY = static_cast<const uint16_t*>(dY); a.ldy = N;
        a.X = static_cast<const uint16_t*>(X); a.ldx = L.K;
        for (int s = 0; s < L.nseg; ++s) { a.dW[s] = L.dw[s]; a.dbias[s] = L.dbias[s]; }
        a.ldw = L.K; a.M = M; a.K = L.K; a.nseg = L.nseg; a.seg_n = L.seg_n;
        return vb_wgrad_bf16(st, &a);
    }
    vb_linear_bwd_weight_args a{};
    a.M = (int32_t)M; a.K = L.K; a.nseg = L.nseg; a.seg_n = L.seg_n;
    a.dY = static_cast<const float*>(dY); a.ldy = N;
    a.X = static_cast<const float*>(X); a.ldx = L.K;
    for (int s = 0; s < L.nseg; ++s) { a.dW[s] = L.dw[s]; a.dbias[s] = L.dbias[s]; }
    a.ldw = L.K; a.accumulate = 1;
    return vb_linear_bwd_weight(st, &a);
}

template <bool B16>
int ln_fwd(void* st, long rows, int cols, const void* x, const vb_layer_norm& n, float eps, void* y, float* mean, float* rstd) {
    if (B16)
        return vb_layernorm_fwd_bf16(st, rows, cols, static_cast<const uint16_t*>(x), n.gamma, n.beta, eps,
                                     static_cast<uint16_t*>(y), mean, rstd);
    return vb_layernorm_fwd(st, rows, cols, static_cast<const float*>(x), nullptr, n.gamma, n.beta, eps, static_cast<float*>(y),
                            mean, rstd);
}

template <bool B16>
int ln_bwd(void* st, long rows, int cols, const void* dy, const void* x, const float* mean, const float* rstd,
           const vb_layer_norm& n, void* dx, float* ws, void* dx_drop, float p, uint64_t seed) {
    const bool twin = dx_drop != nullptr && p > 0.f;
    if (B16)
        return vb_layernorm_bwd_bf16(st, rows, cols, static_cast<const uint16_t*>(dy), static_cast<const uint16_t*>(x), mean, rstd,
                                     n.gamma, static_cast<uint16_t*>(dx), n.dgamma, n.dbeta, ws,
                                     twin ? static_cast<uint16_t*>(dx_drop) : nullptr, twin ? p : 0.f, twin ? seed : 0);
    if (twin)
        return vb_layernorm_bwd_drop(st, rows, cols, static_cast<const float*>(dy), static_cast<const float*>(x), mean, rstd, n.gamma,
                                     static_cast<float*>(dx), n.dgamma, n.dbeta, ws, static_cast<float*>(dx_drop), p, seed);
    return vb_layernorm_bwd(st, rows, cols, static_cast<const float*>(dy), static_cast<const float*>(x), mean, rstd, n.gamma,
                            static_cast<float*>(dx), n.dgamma, n.dbeta, ws);
}

struct AttnCall {
    int batch, heads, d, n_q, n_k;
    const void *q, *k, *v;       // column slices of fused projections, row stride ld (elements)
    long ldq, ldk;
    const float* mask;
    void* out; long ldo;
    float* lse;
    float p; uint64_t seed;
};

template <bool B16>
int attn_fwd(void* st, const AttnCall& c) {
    const float scale = (float)(1.0 / sqrt((double)c.d));       // (the double -> float rounding the Python launchers pass)
    if (B16) {
        vb_attention_bf16_args a{};
        a.batch = c.batch; a.heads = c.heads; a.head_dim = c.d; a.n_q = c.n_q; a.n_k = c.n_k;
        a.q_batch = c.batch; a.kv_batch = c.batch;
        a.Q = static_cast<const uint16_t*>(c.q); a.ldq = c.ldq;
        a.K = static_cast<const uint16_t*>(c.k); a.ldk = c.ldk;
        a.V = static_cast<const uint16_t*>(c.v); a.ldv = c.ldk;
        a.mask_add = c.mask; a.O = static_cast<uint16_t*>(c.out); a.ldo = c.ldo; a.lse = c.lse;
        a.scale = scale; a.dropout_p = c.p; a.seed = c.seed;
        return vb_attention_fwd_bf16(st, &a);
    }
    vb_attention_args a{};
    a.batch = c.batch; a.heads = c.heads; a.head_dim = c.d; a.n_q = c.n_q; a.n_k = c.n_k;
    a.q_batch = c.batch; a.kv_batch = c.batch;
    a.Q = static_cast<const float*>(c.q); a.ldq = c.ldq;
    a.K = static_cast<const float*>(c.k); a.ldk = c.ldk;
    a.V = static_cast<const float*>(c.v); a.ldv = c.ldk;
    a.mask_add = c.mask; a.O = static_cast<float*>(c.out); a.ldo = c.ldo; a.lse = c.lse;
    a.scale = scale; a.dropout_p = c.p; a.seed = c.seed;
    return vb_attention_fwd(st, &a);
}

template <bool B16>
int attn_bwd(void* st, const AttnCall& c, const void* d_out, void* dq, void* dk, void* dv, long lddq, long lddk, float* dvec) {
    const float scale = (float)(1.0 / sqrt((double)c.d));       // (the double -> float rounding the Python launchers pass)
    if (B16) {
        vb_attention_bf16_args a{};
        a.batch = c.batch; a.heads = c.heads; a.head_dim = c.d; a.n_q = c.n_q; a.n_k = c.n_k;
        a.q_batch = c.batch; a.kv_batch = c.batch;
        a.Q = static_cast<const uint16_t*>(c.q); a.ldq = c.ldq;
        a.K = static_cast<const uint16_t*>(c.k); a.ldk = c.ldk;
        a.V = static_cast<const uint16_t*>(c.v); a.ldv = c.ldk;
        a.mask_add = c.mask; a.lse = c.lse; a.scale = scale; a.dropout_p = c.p; a.seed = c.seed;
        vb_attention_bf16_grads g{};
        g.dO = static_cast<const uint16_t*>(d_out); g.lddo = c.ldo;
        g.dQ = static_cast<uint16_t*>(dq); g.lddq = lddq;
        g.dK = static_cast<uint16_t*>(dk); g.lddk = lddk;
        g.dV = static_cast<uint16_t*>(dv); g.lddv = lddk;
        g.dvec = dvec;
        return vb_attention_bwd_bf16(st, &a, &g);
    }
    vb_attention_args a{};
    a.batch = c.batch; a.heads = c.heads; a.head_dim = c.d; a.n_q = c.n_q; a.n_k = c.n_k;
    a.q_batch = c.batch; a.kv_batch = c.batch;
    a.Q = static_cast<const float*>(c.q); a.ldq = c.ldq;
    a.K = static_cast<const float*>(c.k); a.ldk = c.ldk;
    a.V = static_cast<const float*>(c.v); a.ldv = c.ldk;
    a.mask_add = c.mask; a.lse = c.lse; a.scale = scale; a.dropout_p = c.p; a.seed = c.seed;
    vb_attention_grads g{};
    g.dO = static_cast<const float*>(d_out); g.lddo = c.ldo;
    g.dQ = static_cast<float*>(dq); g.lddq = lddq;
    g.dK = static_cast<float*>(dk); g.lddk = lddk;
    g.dV = static_cast<float*>(dv); g.lddv = lddk;
    g.dvec = dvec;
    return vb_attention_bwd(st, &a, &g);
}

// ---- weight gradients on a side stream: one (re-recorded) event per launch stream -------------------------------------
std::mutex g_ev_mutex;
std::unordered_map<void*, hipEvent_t> g_fork_events;

int fork_to(void* main_st, void* side_st) {
    hipEvent_t ev;
    {
        std::lock_guard<std::mutex> lock(g_ev_mutex);
        auto it = g_fork_events.find(main_st);
        if (it == g_fork_events.end()) {
            const hipError_t e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
            if (e != hipSuccess) return (int)e;
            g_fork_events.emplace(main_st, ev);
        } else {
            ev = it->second;
        }
    }
    hipError_t e = hipEventRecord(ev, static_cast<hipStream_t>(main_st));
    if (e != hipSuccess) return (int)e;
    e = hipStreamWaitEvent(static_cast<hipStream_t>(side_st), ev, 0);
    return e == hipSuccess ? 0 : (int)e;
}

// The weight gradients of one vb_layer_bwd call are COLLECTED while the launch stream runs the call's critical path
// (LayerNorm / input-gradient / attention backward) and launched together at its end: behind ONE event on the side stream
// when there is one. A cross-stream rendezvous costs the PRODUCER stream 13 - 19 us on this runtime whatever the event
// flags (tools/event_cost.hip, profiles/r06_event_cost.txt: the marker drains the stream's pipeline), so one fork per weight
// gradient - 121 per training step - was ~1.9 ms of bubbles on the backward stream; one per call is 36. Nothing a deferred
// launch reads is overwritten later in the same call (every gradient buffer of a call is written once), and the kernels,
// their inputs and therefore the results are the same.
struct WgradJob { long M; const vb_layer_linear* L; const void* dY; const void* X; };
struct WgradQueue {
    WgradJob job[8];
    int n = 0;
    void push(long M, const vb_layer_linear& L, const void* dY, const void* X) {
        if (wants_wgrad(L) && n < 8) job[n++] = WgradJob{M, &L, dY, X};
    }
};

template <bool B16>
int wgrad_flush(void* st, void* side, const WgradQueue& q) {
    if (q.n == 0) return 0;
    void* target = st;
    if (side != nullptr && side != st) {
        VB_TRY(fork_to(st, side));
        target = side;
    }
    for (int i = 0; i < q.n; ++i) VB_TRY(wgrad<B16>(target, q.job[i].M, *q.job[i].L, q.job[i].dY, q.job[i].X));
    return 0;
}

AttnCall self_call(const vb_attn_block& b, bool b16) {
    const long Hb = (long)b.heads * b.head_dim, es = b16 ? 2 : 4;
    AttnCall c{};
    c.batch = b.batch; c.heads = b.heads; c.d = b.head_dim; c.n_q = b.n1; c.n_k = b.n1;
    c.q = b.qkv1_out;
    c.k = static_cast<const char*>(b.qkv1_out) + Hb * es;
    c.v = static_cast<const char*>(b.qkv1_out) + 2 * Hb * es;
    c.ldq = c.ldk = 3 * Hb;
    c.mask = b.mask1; c.out = b.ctx1; c.ldo = Hb; c.lse = b.lse1; c.p = b.p1; c.seed = b.seed1;
    return c;
}
// ctx1 = attn(q2; k1, v1 | mask1)  [text queries over image keys]; ctx2 = attn(q1; k2, v2 | mask2)
AttnCall cross_call(const vb_attn_block& b, bool b16, int which) {
    const long Hb = (long)b.heads * b.head_dim, es = b16 ? 2 : 4;
    const void* qsrc = which == 1 ? b.qkv2_out : b.qkv1_out;
    const void* ksrc = which == 1 ? b.qkv1_out : b.qkv2_out;
    AttnCall c{};
    c.batch = b.batch; c.heads = b.heads; c.d = b.head_dim;
    c.n_q = which == 1 ? b.n2 : b.n1;
    c.n_k = which == 1 ? b.n1 : b.n2;
    c.q = qsrc;
    c.k = static_cast<const char*>(ksrc) + Hb * es;
    c.v = static_cast<const char*>(ksrc) + 2 * Hb * es;
    c.ldq = c.ldk = 3 * Hb;
    c.mask = which == 1 ? b.mask1 : b.mask2;
    c.out = which == 1 ? b.ctx1 : b.ctx2;
    c.ldo = Hb;
    c.lse = which == 1 ? b.lse1 : b.lse2;
    c.p = which == 1 ? b.p1 : b.p2;
    c.seed = which == 1 ? b.seed1 : b.seed2;
    return c;
}

template <bool B16>
int attn_block_fwd(void* st, const vb_attn_block& b) {
    const long M1 = (long)b.batch * b.n1;
    VB_TRY(linear<B16>(st, M1, b.qkv1, b.x1, b.qkv1_out, VB_ACT_NONE, nullptr, 0.f, 0, nullptr));
    if (b.n2 == 0) return attn_fwd<B16>(st, self_call(b, B16));
    const long M2 = (long)b.batch * b.n2;
    VB_TRY(linear<B16>(st, M2, b.qkv2, b.x2, b.qkv2_out, VB_ACT_NONE, nullptr, 0.f, 0, nullptr));
    VB_TRY(attn_fwd<B16>(st, cross_call(b, B16, 1)));
    return attn_fwd<B16>(st, cross_call(b, B16, 2));
}

template <bool B16>
int ffn_block_fwd(void* st, const vb_ffn_block& f, bool training) {
    if (f.M == 0) return 0;
    VB_TRY(linear<B16>(st, f.M, f.o, f.ctx, f.sum1, VB_ACT_NONE, f.x, f.p_o, f.seed_o, nullptr));
    VB_TRY(ln_fwd<B16>(st, f.M, f.H, f.sum1, f.ln1, f.eps, f.a1, training ? f.mean1 : nullptr, training ? f.rstd1 : nullptr));
    VB_TRY(linear<B16>(st, f.M, f.f1, f.a1, f.h, VB_ACT_GELU, nullptr, 0.f, 0, training ? f.dact : nullptr));
    VB_TRY(linear<B16>(st, f.M, f.f2, f.h, f.sum2, VB_ACT_NONE, f.a1, f.p_f, f.seed_f, nullptr));
    return ln_fwd<B16>(st, f.M, f.H, f.sum2, f.ln2, f.eps, f.y, training ? f.mean2 : nullptr, training ? f.rstd2 : nullptr);
}

template <bool B16>
int ffn_block_bwd(void* st, WgradQueue& wq, const vb_ffn_block& f) {
    if (f.M == 0) return 0;
    // y = LN(sum2)
    VB_TRY(ln_bwd<B16>(st, f.M, f.H, f.dy, f.sum2, f.mean2, f.rstd2, f.ln2, f.d_sum2, f.ln_ws, f.d_sum2_drop, f.p_f, f.seed_f));
    const void* dyd = f.p_f > 0.f ? f.d_sum2_drop : f.d_sum2;
    // sum2 = dropout(h W2^T + b2) + a1;  h = gelu(pre): d_pre = (dyd W2) * gelu'(pre)
    VB_TRY(dgrad<B16>(st, f.M, f.f2, dyd, f.d_pre, nullptr, f.dact));
    wq.push(f.M, f.f2, dyd, f.h);
    // d_a1 = d_pre W1 + d_sum2 (skip connection)
    VB_TRY(dgrad<B16>(st, f.M, f.f1, f.d_pre, f.d_a1, f.d_sum2, nullptr));
    wq.push(f.M, f.f1, f.d_pre, f.a1);
    // a1 = LN(sum1)
    VB_TRY(ln_bwd<B16>(st, f.M, f.H, f.d_a1, f.sum1, f.mean1, f.rstd1, f.ln1, f.d_sum1, f.ln_ws, f.d_sum1_drop, f.p_o, f.seed_o));
    const void* dod = f.p_o > 0.f ? f.d_sum1_drop : f.d_sum1;
    // sum1 = dropout(ctx Wo^T + bo) + x
    VB_TRY(dgrad<B16>(st, f.M, f.o, dod, f.d_ctx, nullptr, nullptr));
    wq.push(f.M, f.o, dod, f.ctx);
    return 0;
}

template <bool B16>
int attn_block_bwd(void* st, WgradQueue& wq, const vb_attn_block& b) {
    const long Hb = (long)b.heads * b.head_dim, M1 = (long)b.batch * b.n1;
    if (b.n2 == 0) {
        VB_TRY(attn_bwd<B16>(st, self_call(b, B16), b.d_ctx1, b.dqkv1, col<B16>(b.dqkv1, Hb), col<B16>(b.dqkv1, 2 * Hb), 3 * Hb,
                             3 * Hb, b.dvec));
    } else {
        // each slice of dqkv1 / dqkv2 is written exactly once: dq2, dk1, dv1 by direction 1; dq1, dk2, dv2 by direction 2
        VB_TRY(attn_bwd<B16>(st, cross_call(b, B16, 1), b.d_ctx1, b.dqkv2, col<B16>(b.dqkv1, Hb), col<B16>(b.dqkv1, 2 * Hb), 3 * Hb,
                             3 * Hb, b.dvec));
        VB_TRY(attn_bwd<B16>(st, cross_call(b, B16, 2), b.d_ctx2, b.dqkv1, col<B16>(b.dqkv2, Hb), col<B16>(b.dqkv2, 2 * Hb), 3 * Hb,
                             3 * Hb, b.dvec));
    }
    if (b.dx1 != nullptr) VB_TRY(dgrad<B16>(st, M1, b.qkv1, b.dqkv1, b.dx1, b.dres1, nullptr));
    wq.push(M1, b.qkv1, b.dqkv1, b.x1);
    if (b.n2 != 0) {
        const long M2 = (long)b.batch * b.n2;
        if (b.dx2 != nullptr) VB_TRY(dgrad<B16>(st, M2, b.qkv2, b.dqkv2, b.dx2, b.dres2, nullptr));
        wq.push(M2, b.qkv2, b.dqkv2, b.x2);
    }
    return 0;
}

bool bad_linear(const vb_layer_linear& L, bool b16) {
    if (L.nseg < 1 || L.nseg > VB_MAX_SEGMENTS || L.seg_n <= 0 || L.K <= 0) return true;
    if (b16) return L.w16 == nullptr;
    for (int s = 0; s < L.nseg; ++s)
        if (L.w[s] == nullptr) return true;
    return false;
}

int check(const vb_layer_args* a, bool backward) {
    if (a == nullptr) return VB_E_BADARG;
    if (a->dtype != VB_DT_F32 && a->dtype != VB_DT_BF16) return VB_E_BADARG;
    const bool b16 = a->dtype == VB_DT_BF16;
    const vb_attn_block& b = a->attn;
    const bool has_attn = b.batch != 0;         // (a call may carry only the attention block, or only output + FFN blocks)
    if (!has_attn && a->s1.M == 0 && a->s2.M == 0) return VB_E_BADARG;
    if (has_attn) {
        if (b.batch < 0 || b.heads <= 0 || b.head_dim <= 0 || b.n1 <= 0 || b.n2 < 0) return VB_E_BADARG;
        if (b.x1 == nullptr || b.qkv1_out == nullptr || b.ctx1 == nullptr || bad_linear(b.qkv1, b16)) return VB_E_BADARG;
        if ((long)b.qkv1.nseg * b.qkv1.seg_n != 3L * b.heads * b.head_dim) return VB_E_SEGMENT;
        if (b.n2 > 0) {
            if (b.x2 == nullptr || b.qkv2_out == nullptr || b.ctx2 == nullptr || bad_linear(b.qkv2, b16)) return VB_E_BADARG;
            if ((long)b.qkv2.nseg * b.qkv2.seg_n != 3L * b.heads * b.head_dim) return VB_E_SEGMENT;
        }
    }
    for (const vb_ffn_block* f : {&a->s1, &a->s2}) {
        if (f->M == 0) continue;
        if (f->M < 0 || f->H <= 0 || f->I <= 0 || f->Hc <= 0) return VB_E_BADARG;
        if (bad_linear(f->o, b16) || bad_linear(f->f1, b16) || bad_linear(f->f2, b16)) return VB_E_BADARG;
        if (f->o.K != f->Hc || f->o.nseg * f->o.seg_n != f->H || f->f1.K != f->H || f->f1.nseg * f->f1.seg_n != f->I ||
            f->f2.K != f->I || f->f2.nseg * f->f2.seg_n != f->H)
            return VB_E_SEGMENT;
        if (f->ctx == nullptr || f->x == nullptr || f->sum1 == nullptr || f->a1 == nullptr || f->h == nullptr || f->sum2 == nullptr ||
            f->y == nullptr || f->ln1.gamma == nullptr || f->ln2.gamma == nullptr)
            return VB_E_BADARG;
        if (backward) {
            if (f->dy == nullptr || f->d_sum2 == nullptr || f->d_pre == nullptr || f->d_a1 == nullptr || f->d_sum1 == nullptr ||
                f->d_ctx == nullptr || f->ln_ws == nullptr || f->dact == nullptr || f->mean1 == nullptr || f->mean2 == nullptr ||
                f->ln1.dgamma == nullptr || f->ln2.dgamma == nullptr)
                return VB_E_BADARG;
            if ((f->p_f > 0.f && f->d_sum2_drop == nullptr) || (f->p_o > 0.f && f->d_sum1_drop == nullptr)) return VB_E_BADARG;
        }
    }
    if (backward && has_attn &&
        (b.dqkv1 == nullptr || b.dvec == nullptr || b.d_ctx1 == nullptr || b.lse1 == nullptr ||
         (b.n2 > 0 && (b.dqkv2 == nullptr || b.d_ctx2 == nullptr || b.lse2 == nullptr))))
        return VB_E_BADARG;
    return 0;
}

template <bool B16>
int layer_fwd(void* st, const vb_layer_args* a) {
    if (a->attn.batch != 0) VB_TRY(attn_block_fwd<B16>(st, a->attn));
    VB_TRY(ffn_block_fwd<B16>(st, a->s1, a->training != 0));
    return ffn_block_fwd<B16>(st, a->s2, a->training != 0);
}

template <bool B16>
int layer_bwd(void* st, const vb_layer_args* a) {
    // VB_WGRAD_FORK=each: the round-6 first form, one fork per weight gradient right behind its input gradient (A/B only)
    static const bool each = [] { const char* e = getenv("VB_WGRAD_FORK"); return e != nullptr && e[0] == 'e'; }();
    WgradQueue wq;
    VB_TRY(ffn_block_bwd<B16>(st, wq, a->s1));
    if (each) { VB_TRY(wgrad_flush<B16>(st, a->wgrad_stream, wq)); wq.n = 0; }
    VB_TRY(ffn_block_bwd<B16>(st, wq, a->s2));
    if (each) { VB_TRY(wgrad_flush<B16>(st, a->wgrad_stream, wq)); wq.n = 0; }
    if (a->attn.batch != 0) VB_TRY(attn_block_bwd<B16>(st, wq, a->attn));
    return wgrad_flush<B16>(st, a->wgrad_stream, wq);
}

}  // namespace

extern "C" int vb_layer_fwd(void* stream, const vb_layer_args* a) {
    VB_TRY(check(a, false));
    return a->dtype == VB_DT_BF16 ? layer_fwd<kB16>(stream, a) : layer_fwd<kF32>(stream, a);
}

extern "C" int vb_layer_bwd(void* stream, const vb_layer_args* a) {
    VB_TRY(check(a, true));
    return a->dtype == VB_DT_BF16 ? layer_bwd<kB16>(stream, a) : layer_bwd<kF32>(stream, a);
}
