// Stand-alone laboratory of the bf16 training kernels (csrc/gemm_bf16.hip) over the C ABI, torch-free:
//     make -C tools bf16_lab && tools/bf16_lab [diag|check|time|all]
// diag:  what ds_read_b64_tr_b16 returns for a known LDS image (the transposing read the weight-gradient kernel is built on)
// check: vb_linear_bf16 (every epilogue the model launches: plain / bias, GELU + derivative, residual, dropout + residual,
//        multiplier, fp32 output) and vb_wgrad_bf16 (stacked segments, ragged M, several contraction splits),
//        vb_weight_shadow_bf16, vb_colsum_bf16 against fp64 host sums over the SAME bf16 operand values, asymmetric random data
// time:  the encoder's GEMM shapes at batch 256 (M = 9216 text / 9472 image rows): forward, dgrad and wgrad, us and TFLOP/s
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <functional>
#include <string>
#include <vector>

#include "../include/vilbert_hip.h"

#define CK(x)                                                                         \
    do {                                                                              \
        hipError_t e_ = (x);                                                          \
        if (e_ != hipSuccess) {                                                       \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
            exit(1);                                                                  \
        }                                                                             \
    } while (0)
#define VB(x)                                                                  \
    do {                                                                       \
        int e_ = (x);                                                          \
        if (e_ != 0) {                                                         \
            fprintf(stderr, "%s:%d vb error %d\n", __FILE__, __LINE__, e_);    \
            exit(1);                                                           \
        }                                                                      \
    } while (0)

static uint32_t rng_state = 2463534242u;
static inline uint32_t rnd() {
    rng_state ^= rng_state << 13; rng_state ^= rng_state >> 17; rng_state ^= rng_state << 5;
    return rng_state;
}
static inline float urand() { return (float)(rnd() >> 8) * (1.0f / 16777216.0f) * 2.0f - 1.0f; }

static uint16_t bf16_of(float v) {
    uint32_t u; memcpy(&u, &v, 4);
    return (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}
static float bf16_val(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }

template <class T>
static T* to_dev(const std::vector<T>& v) {
    T* p;
    CK(hipMalloc(&p, std::max<size_t>(v.size(), 1) * sizeof(T)));
    CK(hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    return p;
}
template <class T>
static std::vector<T> to_host(const T* p, size_t n) {
    std::vector<T> v(n);
    CK(hipMemcpy(v.data(), p, n * sizeof(T), hipMemcpyDeviceToHost));
    return v;
}
static std::vector<uint16_t> rand_bf16(size_t n, float scale) {
    std::vector<uint16_t> v(n);
    for (auto& x : v) x = bf16_of(urand() * scale);
    return v;
}

// ---- the dropout mask of csrc/rng.h -----------------------------------------------------------------------------
static bool keep_host(uint64_t seed, uint64_t idx, float p) {
    uint64_t z = seed + idx * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z = z ^ (z >> 31);
    return (float)((uint32_t)(z >> 32) >> 8) * (1.0f / 16777216.0f) >= p;
}
static double gelu_host(double x) { return 0.5 * x * (1.0 + erf(x / sqrt(2.0))); }
static double gelu_grad_host(double x) { return 0.5 * (1.0 + erf(x / sqrt(2.0))) + x * exp(-0.5 * x * x) / sqrt(2.0 * M_PI); }

// ---- diag ---------------------------------------------------------------------------------------------------------
typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
__global__ void tr_diag_kernel(unsigned short* out, int row_stride_bytes) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    unsigned short* s = (unsigned short*)smem;
    for (int i = threadIdx.x; i < 8192; i += 64) s[i] = (unsigned short)i;      // element value = its element index
    __syncthreads();
    const int lane = threadIdx.x, i16 = lane & 15, g = lane >> 4;
    // lane 4 r + q of a 16-lane group: row r, columns 4 q .. 4 q + 3 of the group's 4 x 16 block; group g starts 16 columns on
    typedef __attribute__((address_space(3))) bf16x4_t lds4;
    const unsigned addr = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem + (i16 >> 2) * row_stride_bytes +
                          (i16 & 3) * 8 + g * 32;
    const bf16x4_t v = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds4*)addr);
    const unsigned short* u = (const unsigned short*)&v;
    for (int j = 0; j < 4; ++j) out[lane * 4 + j] = u[j];
}

static int diag() {
    printf("---- ds_read_b64_tr_b16: LDS element e holds the value e; lane 4 r + q of a 16-lane group reads row r (row stride 256 B =\n"
           "     128 elements), columns 4 q .. 4 q + 3, group g starts at column 16 g. Expected: lane i of group g receives\n"
           "     (row j, column 16 g + i) = 128 j + 16 g + i for j = 0..3\n");
    unsigned short* d;
    CK(hipMalloc(&d, 64 * 4 * 2));
    hipLaunchKernelGGL(tr_diag_kernel, dim3(1), dim3(64), 16384, 0, d, 256);
    CK(hipDeviceSynchronize());
    auto h = to_host(d, 256);
    int bad = 0;
    for (int lane = 0; lane < 64; ++lane) {
        const int g = lane >> 4, i = lane & 15;
        for (int j = 0; j < 4; ++j) bad += h[lane * 4 + j] != 128 * j + 16 * g + i;
        if (lane < 18 || lane == 63)
            printf("lane %2d: %4d %4d %4d %4d\n", lane, h[lane * 4], h[lane * 4 + 1], h[lane * 4 + 2], h[lane * 4 + 3]);
    }
    printf("transposing read %s\n", bad ? "DIFFERS from the assumed semantics - FAIL" : "matches the assumed semantics: ok");
    return bad != 0;
}

// ---- checks -------------------------------------------------------------------------------------------------------
struct Variant { const char* name; int act; bool bias, res, mul, drop, f32, dgrad_out; };

static int check_linear(int M, int N, int K) {
    printf("---- vb_linear_bf16 M=%d N=%d K=%d (%d tiles)\n", M, N, K, ((M + 255) / 256) * (N / 128));
    auto A = rand_bf16((size_t)M * K, 1.0f), W = rand_bf16((size_t)N * K, 0.05f), R = rand_bf16((size_t)M * N, 2.0f);
    std::vector<float> bias(N);
    for (auto& v : bias) v = urand();
    uint16_t *dA = to_dev(A), *dW = to_dev(W), *dR = to_dev(R);
    float* dbias = to_dev(bias);
    uint16_t *dC, *dD;
    float* dC32;
    CK(hipMalloc(&dC, (size_t)M * N * 2)); CK(hipMalloc(&dD, (size_t)M * N * 2)); CK(hipMalloc(&dC32, (size_t)M * N * 4));
    std::vector<int> rows;
    for (int r = 0; r < M; r += std::max(1, M / 37)) rows.push_back(r);
    for (int s = 256; s < M; s += 256) { rows.push_back(s - 1); rows.push_back(s); }
    rows.push_back(M - 1);
    std::sort(rows.begin(), rows.end());
    rows.erase(std::unique(rows.begin(), rows.end()), rows.end());
    if (rows.size() > 120) {   // keep the host sums short: every third seam
        std::vector<int> r2;
        for (size_t i = 0; i < rows.size(); ++i) if (i % 3 == 0 || rows[i] >= M - 2) r2.push_back(rows[i]);
        rows.swap(r2);
    }
    std::vector<double> acc(rows.size() * (size_t)N), mag(rows.size() * (size_t)N);
    std::vector<float> Wf((size_t)N * K);
    for (size_t i = 0; i < Wf.size(); ++i) Wf[i] = bf16_val(W[i]);
    for (size_t ri = 0; ri < rows.size(); ++ri) {
        std::vector<float> a(K);
        for (int k = 0; k < K; ++k) a[k] = bf16_val(A[(size_t)rows[ri] * K + k]);
        for (int n = 0; n < N; ++n) {
            double s = 0, m = 0;
            const float* w = &Wf[(size_t)n * K];
            for (int k = 0; k < K; ++k) { const double p = (double)a[k] * w[k]; s += p; m += fabs(p); }
            acc[ri * N + n] = s; mag[ri * N + n] = m;
        }
    }
    const uint64_t seed = 0x1234567890abcdefull;
    const float drop_p = 0.1f;
    const Variant variants[] = {
        {"plain bf16", VB_ACT_NONE, false, false, false, false, false, false},
        {"bias bf16", VB_ACT_NONE, true, false, false, false, false, false},
        {"bias + GELU + derivative", VB_ACT_GELU, true, false, false, false, false, false},
        {"bias + residual", VB_ACT_NONE, true, true, false, false, false, false},
        {"bias + dropout + residual", VB_ACT_NONE, true, true, false, true, false, false},
        {"x multiplier (dgrad through an activation)", VB_ACT_NONE, false, false, true, false, false, false},
        {"bias, fp32 out", VB_ACT_NONE, true, false, false, false, true, false},
        {"bias + ReLU, fp32 out", VB_ACT_RELU, true, false, false, false, true, false},
    };
    int bad = 0;
    for (const Variant& v : variants) {
        vb_linear_bf16_args g{};
        g.A = dA; g.lda = K; g.W = dW; g.ldw = K; g.M = M; g.N = N; g.K = K;
        g.bias[0] = v.bias ? dbias : nullptr;
        if (v.f32) { g.C32 = dC32; g.ldc32 = N; } else { g.C = dC; g.ldc = N; }
        if (v.res) { g.residual = dR; g.ldr = N; }
        if (v.mul) { g.mul = dR; g.ldm = N; }
        g.act = v.act;
        if (v.act == VB_ACT_GELU) { g.act_grad = dD; g.ldg = N; }
        if (v.drop) { g.dropout_p = drop_p; g.seed = seed; }
        CK(hipMemset(dC, 0xff, (size_t)M * N * 2)); CK(hipMemset(dC32, 0xff, (size_t)M * N * 4)); CK(hipMemset(dD, 0xff, (size_t)M * N * 2));
        VB(vb_linear_bf16(nullptr, &g));
        CK(hipDeviceSynchronize());
        auto c16 = to_host(dC, (size_t)M * N);
        auto c32 = to_host(dC32, (size_t)M * N);
        auto d16 = to_host(dD, (size_t)M * N);
        double worst = 0, worst_d = 0;
        long nbad = 0;
        for (size_t ri = 0; ri < rows.size(); ++ri)
            for (int n = 0; n < N; ++n) {
                const size_t o = (size_t)rows[ri] * N + n;
                double want = acc[ri * N + n] + (v.bias ? bias[n] : 0.0);
                double wd = 0;
                if (v.act == VB_ACT_GELU) { wd = gelu_grad_host(want); want = gelu_host(want); }
                if (v.act == VB_ACT_RELU) want = std::max(want, 0.0);
                if (v.drop) want = keep_host(seed, (uint64_t)o, drop_p) ? want / (1.0 - drop_p) : 0.0;
                if (v.res) want += bf16_val(R[o]);
                if (v.mul) want *= bf16_val(R[o]);
                const double got = v.f32 ? c32[o] : bf16_val(c16[o]);
                const double rmul = v.mul ? fabs(bf16_val(R[o])) : 1.0;
                const double tol = 2e-6 * mag[ri * N + n] * rmul + (v.f32 ? 1e-6 : fabs(want) / 256 + 1e-30) + 1e-5;
                const double e = fabs(got - want);
                if (!(e <= tol)) ++nbad;
                worst = std::max(worst, e / tol);
                if (v.act == VB_ACT_GELU) {
                    const double ed = fabs(bf16_val(d16[o]) - wd), told = fabs(wd) / 256 + 2e-5 + 2e-6 * mag[ri * N + n];
                    if (!(ed <= told)) ++nbad;
                    worst_d = std::max(worst_d, ed / told);
                }
            }
        // rows past M must not be touched, unchecked rows must have been written
        printf("  %-44s worst err / tol %.3f%s  %s\n", v.name, worst, v.act == VB_ACT_GELU ? (" (derivative " + std::to_string(worst_d) + ")").c_str() : "",
               nbad ? "FAIL" : "ok");
        bad += nbad != 0;
    }
    CK(hipFree(dA)); CK(hipFree(dW)); CK(hipFree(dR)); CK(hipFree(dbias)); CK(hipFree(dC)); CK(hipFree(dD)); CK(hipFree(dC32));
    return bad;
}

static int check_wgrad(int M, int nseg, int seg_n, int K) {
    const int N = nseg * seg_n;
    printf("---- vb_wgrad_bf16 M=%d dY [M, %d x %d] X [M, %d]\n", M, nseg, seg_n, K);
    auto Y = rand_bf16((size_t)M * N, 1.0f), X = rand_bf16((size_t)M * K, 1.0f);
    uint16_t *dY = to_dev(Y), *dX = to_dev(X);
    std::vector<float> init((size_t)N * K);
    for (auto& v : init) v = urand();       // the kernel ADDS: start from a non-zero gradient
    float* dW = to_dev(init);
    vb_wgrad_bf16_args g{};
    g.dY = dY; g.ldy = N; g.X = dX; g.ldx = K; g.M = M; g.K = K; g.nseg = nseg; g.seg_n = seg_n; g.ldw = K;
    std::vector<float> binit(N, 0.25f);
    float* dB = to_dev(binit);
    for (int s = 0; s < nseg; ++s) { g.dW[s] = dW + (size_t)s * seg_n * K; g.dbias[s] = dB + (size_t)s * seg_n; }
    VB(vb_wgrad_bf16(nullptr, &g));
    CK(hipDeviceSynchronize());
    auto got = to_host(dW, (size_t)N * K);
    auto gotb = to_host(dB, (size_t)N);
    long nbad = 0;
    double worst = 0;
    const int samples = 6000;
    for (int t = 0; t < samples; ++t) {
        int n = rnd() % N, k = rnd() % K;
        if (t < 64) { n = (t & 1) ? N - 1 - (t >> 1) % N : (t >> 1) % N; k = (t & 2) ? K - 1 - (t * 7) % K : (t * 5) % K; }
        double s = 0, m = 0;
        for (int r = 0; r < M; ++r) {
            const double p = (double)bf16_val(Y[(size_t)r * N + n]) * bf16_val(X[(size_t)r * K + k]);
            s += p; m += fabs(p);
        }
        const double want = init[(size_t)n * K + k] + s, e = fabs(got[(size_t)n * K + k] - want), tol = 3e-6 * m + 1e-5;
        if (!(e <= tol)) { if (nbad < 5) printf("    dW[%d][%d] = %g, want %g\n", n, k, got[(size_t)n * K + k], want); ++nbad; }
        worst = std::max(worst, e / tol);
    }
    // bias gradient of the same dY
    float *dcs, *dws;
    std::vector<float> cinit(N, 0.5f);
    dcs = to_dev(cinit);
    CK(hipMalloc(&dws, vb_colsum_bf16_workspace(N) * 4));
    VB(vb_colsum_bf16(nullptr, M, N, dY, N, dcs, dws));
    CK(hipDeviceSynchronize());
    auto cs = to_host(dcs, N);
    long cbad = 0, fbad = 0;
    for (int n = 0; n < N; ++n) {
        double s = 0, m = 0;
        for (int r = 0; r < M; ++r) { s += bf16_val(Y[(size_t)r * N + n]); m += fabs(bf16_val(Y[(size_t)r * N + n])); }
        if (!(fabs(cs[n] - (s + 0.5)) <= 3e-6 * m + 1e-5)) ++cbad;
        if (!(fabs(gotb[n] - (s + 0.25)) <= 3e-6 * m + 1e-5)) { if (fbad < 3) printf("    dbias[%d] = %g, want %g\n", n, gotb[n], s + 0.25); ++fbad; }
    }
    printf("  %d sampled dW elements: worst err / tol %.3f %s; fused bias gradient %s; vb_colsum_bf16 %s\n", samples, worst,
           nbad ? "FAIL" : "ok", fbad ? "FAIL" : "ok", cbad ? "FAIL" : "ok");
    CK(hipFree(dB));
    nbad += fbad;
    CK(hipFree(dY)); CK(hipFree(dX)); CK(hipFree(dW)); CK(hipFree(dcs)); CK(hipFree(dws));
    return (nbad != 0) + (cbad != 0);
}

static int check_shadow(int rows, int cols) {
    std::vector<float> w((size_t)rows * cols);
    for (auto& v : w) v = urand() * 0.1f;
    float* dw = to_dev(w);
    uint16_t *d16, *dt;
    CK(hipMalloc(&d16, (size_t)rows * cols * 2)); CK(hipMalloc(&dt, (size_t)rows * cols * 2));
    VB(vb_weight_shadow_bf16(nullptr, rows, cols, dw, cols, d16, cols, dt, rows));
    CK(hipDeviceSynchronize());
    auto h16 = to_host(d16, (size_t)rows * cols), ht = to_host(dt, (size_t)rows * cols);
    long bad = 0;
    for (int r = 0; r < rows; ++r)
        for (int c = 0; c < cols; ++c) {
            const uint16_t want = bf16_of(w[(size_t)r * cols + c]);
            bad += h16[(size_t)r * cols + c] != want;
            bad += ht[(size_t)c * rows + r] != want;
        }
    // casts
    float* back;
    CK(hipMalloc(&back, (size_t)rows * cols * 4));
    VB(vb_cast_bf16_f32(nullptr, (int64_t)rows * cols - 3, d16, back));
    uint16_t* again;
    CK(hipMalloc(&again, (size_t)rows * cols * 2));
    VB(vb_cast_f32_bf16(nullptr, (int64_t)rows * cols - 3, dw, again));
    CK(hipDeviceSynchronize());
    auto hb = to_host(back, (size_t)rows * cols - 3);
    auto ha = to_host(again, (size_t)rows * cols - 3);
    for (size_t i = 0; i < hb.size(); ++i) bad += (hb[i] != bf16_val(h16[i])) + (ha[i] != h16[i]);
    printf("---- weight shadow + casts %d x %d: %ld mismatches %s\n", rows, cols, bad, bad ? "FAIL" : "ok");
    CK(hipFree(dw)); CK(hipFree(d16)); CK(hipFree(dt)); CK(hipFree(back)); CK(hipFree(again));
    return bad != 0;
}

// ---- timing -------------------------------------------------------------------------------------------------------
static double time_us(const std::function<void()>& fn, int iters) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) fn();
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < iters; ++i) fn();
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return 1e3 * ms / iters;
}

static void time_all() {
    struct S { const char* what; int M, N, K; int epi; };   // epi: 0 plain, 1 gelu + derivative, 2 dropout + residual, 3 multiplier, 4 residual
    const S shapes[] = {
        {"text q|k|v fwd", 9216, 2304, 768, 0}, {"text attn-out fwd (+drop+res)", 9216, 768, 768, 2}, {"text FFN up fwd (GELU)", 9216, 3072, 768, 1},
        {"text FFN down fwd (+drop+res)", 9216, 768, 3072, 2}, {"text FFN down dgrad (x gelu')", 9216, 3072, 768, 3},
        {"text FFN up dgrad (+res)", 9216, 768, 3072, 4}, {"text q|k|v dgrad", 9216, 768, 2304, 0},
        {"image q|k|v fwd", 9472, 3072, 1024, 0}, {"image 1024 fwd (+drop+res)", 9472, 1024, 1024, 2}, {"image FFN up fwd (GELU)", 9472, 1024, 1024, 1},
        {"image q|k|v dgrad", 9472, 1024, 3072, 0}, {"co-attn text q|k|v fwd", 9216, 3072, 768, 0}, {"co-attn out -> text (+drop+res)", 9216, 768, 1024, 2},
        {"image features fwd", 9472, 1024, 2048, 0}, {"B=64 text q|k|v fwd", 2304, 2304, 768, 0}, {"B=64 image 1024 fwd", 2368, 1024, 1024, 2},
    };
    size_t maxA = 0, maxW = 0, maxC = 0;
    for (const S& s : shapes) { maxA = std::max(maxA, (size_t)s.M * s.K); maxW = std::max(maxW, (size_t)s.N * s.K); maxC = std::max(maxC, (size_t)s.M * s.N); }
    uint16_t *dA = to_dev(rand_bf16(maxA, 1.0f)), *dW = to_dev(rand_bf16(maxW, 0.05f)), *dR = to_dev(rand_bf16(maxC, 1.0f)), *dC, *dD;
    CK(hipMalloc(&dC, maxC * 2)); CK(hipMalloc(&dD, maxC * 2));
    std::vector<float> bias(4096, 0.1f);
    float* dbias = to_dev(bias);
    printf("---- vb_linear_bf16 (forward / dgrad launches), random data\n");
    for (const S& s : shapes) {
        vb_linear_bf16_args g{};
        g.A = dA; g.lda = s.K; g.W = dW; g.ldw = s.K; g.M = s.M; g.N = s.N; g.K = s.K; g.C = dC; g.ldc = s.N; g.bias[0] = dbias;
        if (s.epi == 1) { g.act = VB_ACT_GELU; g.act_grad = dD; g.ldg = s.N; }
        if (s.epi == 2) { g.residual = dR; g.ldr = s.N; g.dropout_p = 0.1f; g.seed = 7; }
        if (s.epi == 3) { g.mul = dR; g.ldm = s.N; g.bias[0] = nullptr; }
        if (s.epi == 4) { g.residual = dR; g.ldr = s.N; g.bias[0] = nullptr; }
        const double us = time_us([&] { VB(vb_linear_bf16(nullptr, &g)); }, 20);
        printf("  %-34s %5d x %4d x %4d  %3d tiles  %8.1f us  %7.1f TFLOP/s\n", s.what, s.M, s.N, s.K, ((s.M + 255) / 256) * (s.N / 128), us,
               2.0 * s.M * s.N * s.K / us * 1e-6);
    }
    struct Wg { const char* what; int M, nseg, seg_n, K; };
    const Wg wg[] = {{"text q|k|v", 9216, 3, 768, 768}, {"text attn-out", 9216, 1, 768, 768}, {"text FFN up", 9216, 1, 3072, 768},
                     {"text FFN down", 9216, 1, 768, 3072}, {"image q|k|v", 9472, 3, 1024, 1024}, {"image 1024", 9472, 1, 1024, 1024},
                     {"co-attn text q|k|v", 9216, 3, 1024, 768}, {"co-attn out -> text", 9216, 1, 768, 1024}, {"image features", 9472, 1, 1024, 2048},
                     {"B=64 text FFN up", 2304, 1, 3072, 768}, {"B=64 image 1024", 2368, 1, 1024, 1024}};
    float* dWg;
    CK(hipMalloc(&dWg, (size_t)3072 * 3072 * 4));
    CK(hipMemset(dWg, 0, (size_t)3072 * 3072 * 4));
    // the product default: deterministic weight gradient (partials to a registered workspace + ordered reduce); LAB_DET=0 = atomics
    const char* det_env = getenv("LAB_DET");
    const bool det = det_env == nullptr || atoi(det_env) != 0;
    void* det_ws = nullptr;
    if (det) {
        CK(hipMalloc(&det_ws, (size_t)2048 << 20));
        VB(vb_set_deterministic(1, det_ws, (int64_t)2048 << 20));
    }
    printf("---- vb_wgrad_bf16 (%s)\n", det ? "deterministic: partials + ordered reduce launch" : "fp32 atomics");
    for (const Wg& s : wg) {
        vb_wgrad_bf16_args g{};
        g.dY = dR; g.ldy = (int64_t)s.nseg * s.seg_n; g.X = dA; g.ldx = s.K; g.M = s.M; g.K = s.K; g.nseg = s.nseg; g.seg_n = s.seg_n; g.ldw = s.K;
        for (int i = 0; i < s.nseg; ++i) g.dW[i] = dWg + (size_t)i * s.seg_n * s.K;
        const double us = time_us([&] { VB(vb_wgrad_bf16(nullptr, &g)); }, 20);
        printf("  %-34s %5d x %4d x %4d  %8.1f us  %7.1f TFLOP/s\n", s.what, s.M, s.nseg * s.seg_n, s.K, us,
               2.0 * s.M * s.nseg * s.seg_n * s.K / us * 1e-6);
    }
}

int main(int argc, char** argv) {
    const std::string mode = argc > 1 ? argv[1] : "all";
    int bad = 0;
    if (mode == "diag" || mode == "all") bad += diag();
    if (mode == "check" || mode == "all") {
        bad += check_shadow(768, 3072);
        bad += check_shadow(1024, 64);
        bad += check_linear(256, 128, 64);
        bad += check_linear(300, 768, 768);
        bad += check_linear(77, 256, 128);
        bad += check_linear(1, 128, 192);
        bad += check_linear(9216, 768, 3072);     // 216 tiles: one round
        bad += check_linear(9472, 1024, 1024);    // 296 tiles: a second round on 40 blocks, ragged last row tile
        bad += check_linear(18432, 2304, 768);    // 1296 tiles: five rounds
        bad += check_wgrad(64, 1, 256, 128);
        bad += check_wgrad(16, 1, 256, 128);
        bad += check_wgrad(100, 1, 256, 256);
        bad += check_wgrad(1000, 3, 256, 128);
        bad += check_wgrad(9216, 3, 768, 768);
        bad += check_wgrad(9472 + 40, 1, 1024, 2048);
        bad += check_wgrad(2304, 1, 3072, 768);
        printf("==== %s\n", bad ? "CHECK FAILED" : "all checks ok");
    }
    if (mode == "time" || mode == "all") time_all();
    return bad != 0;
}
