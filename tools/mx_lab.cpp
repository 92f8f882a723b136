// Stand-alone laboratory of the MX e4m3 path (csrc/mx8.hip) over the C ABI, torch-free (a fresh GPU box spends 1-2
// minutes importing torch; this starts in a second):
//     make -C tools mx_lab && tools/mx_lab [check|time|all]
// check: vb_quantize_rows_mx bit-exact against a host restatement of the format (scale byte from the bits of amax, e4m3fn
//        round-to-nearest-even), vb_linear_fwd_mx against fp64 sums over the dequantised operands on sampled rows - fp32
//        output (+ bias, + residual), bf16 output, MX output (+ GELU) - with per-block magnitudes spread over 2^-6 .. 2^6
//        so that a wrong scale-operand lane map or byte order cannot hide.
// time:  the encoder's forward shapes at batch 512 (M = 18432), microseconds and TFLOP/s per launch.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
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

static uint32_t rng_state = 12345u;
static inline uint32_t rnd() {
    rng_state ^= rng_state << 13; rng_state ^= rng_state >> 17; rng_state ^= rng_state << 5;
    return rng_state;
}
static inline float urand() { return (float)(rnd() >> 8) * (1.0f / 16777216.0f) * 2.0f - 1.0f; }

// ---- host restatement of the format -------------------------------------------------------------------------------
static unsigned scale_byte(float amax) {
    uint32_t u; memcpy(&u, &amax, 4);
    int b = (int)(u >> 23) - 8 + ((u & 0x7fffffu) > 0x600000u ? 1 : 0);
    return (unsigned)std::max(b, 0);
}
static float scale_of(unsigned byte) { return ldexpf(1.0f, (int)byte - 127); }
static uint8_t e4m3(float x) {
    const uint8_t sign = signbit(x) ? 0x80 : 0;
    float a = fabsf(x);
    if (!(a > 0.f)) return sign;
    if (a >= 448.f) return sign | 0x7e;
    if (a < 0.015625f) {   // below the smallest normal 2^-6: subnormals, step 2^-9
        const int q = (int)nearbyintf(a * 512.f);
        return sign | (uint8_t)q;   // q == 8 is the code of 2^-6
    }
    int e; const float m = frexpf(a, &e);   // a = m 2^e, m in [0.5, 1)
    e -= 1;                                  // a = (2 m) 2^e, 2 m in [1, 2)
    int q = (int)nearbyintf((2.f * m - 1.f) * 8.f);
    if (q == 8) { q = 0; e += 1; }
    if (e > 8 || (e == 8 && q > 6)) return sign | 0x7e;
    return sign | (uint8_t)(((e + 7) << 3) | q);
}
static float e4m3_value(uint8_t c) {
    const int s = c >> 7, e = (c >> 3) & 15, q = c & 7;
    const float v = e == 0 ? ldexpf((float)q, -9) : ldexpf(1.f + q / 8.f, e - 7);
    return s ? -v : v;
}
static void quant_host(const float* x, long rows, int K, long srows, std::vector<uint8_t>& q, std::vector<uint32_t>& sc) {
    q.assign((size_t)rows * K, 0);
    sc.assign((size_t)(K / 128) * srows, 0);
    for (long r = 0; r < rows; ++r)
        for (int b = 0; b < K / 32; ++b) {
            float amax = 0.f;
            for (int k = 0; k < 32; ++k) amax = std::max(amax, fabsf(x[r * K + 32 * b + k]));
            const unsigned byte = scale_byte(amax);
            const float inv = ldexpf(1.0f, 127 - (int)byte);
            for (int k = 0; k < 32; ++k) q[r * K + 32 * b + k] = e4m3(x[r * K + 32 * b + k] * inv);
            sc[(size_t)(b / 4) * srows + r] |= byte << (8 * (b % 4));
        }
}
// the epilogue's GELU (csrc/mx8.h mx_gelu): odd polynomial of Phi on the clamped argument
static double gelu_ref(double x) {
    static const double c[6] = {0.398325773, -0.0648922966, 0.00876406186, -0.000774126121, 3.89366778e-05, -8.3218473e-07};
    const double t = std::min(3.5, std::max(-3.5, x)), u = t * t;
    double q = 0;
    for (int k = 5; k >= 0; --k) q = q * u + c[k];
    return x * (0.5 + t * q);
}

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

static float bf16_value(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }

static int check_case(int M, int N, int K) {
    printf("---- check M=%d N=%d K=%d\n", M, N, K);
    const long a_srows = (M + 255) / 256 * 256, w_srows = (N + 3) / 4 * 4, c_srows = (M + 3) / 4 * 4;
    std::vector<float> A((size_t)M * K), W((size_t)N * K), bias(N), R((size_t)M * N);
    // per (row, block) magnitudes 2^-6 .. 2^6; a few all-zero blocks and rows
    // MX_LAB_MAG: 3 (default) magnitudes vary per (row, block); 0 uniform; 1 per row only; 2 per K block only;
    // +4: the same for W only (A uniform), +8: for A only (W uniform)
    const int magmode = getenv("MX_LAB_MAG") ? atoi(getenv("MX_LAB_MAG")) : 3;
    auto mag_of = [&](int mode, long r, int b, int span) -> float {
        const uint32_t h = (mode & 1 ? (uint32_t)r * 2654435761u : 0u) ^ (mode & 2 ? (uint32_t)(b + 1) * 40503u : 0u);
        if ((mode & 3) == 0) return 1.f;
        return ldexpf(1.f, (int)((h >> 7) % span) - 6);
    };
    for (long r = 0; r < M; ++r)
        for (int b = 0; b < K / 32; ++b) {
            float mag = (magmode & 4) ? 1.f : mag_of(magmode, r, b, 13);
            if (magmode == 3 && rnd() % 29 == 0) mag = 0.f;
            for (int k = 0; k < 32; ++k) A[r * K + 32 * b + k] = (0.25f + 0.75f * fabsf(urand())) * (urand() < 0 ? -1.f : 1.f) * mag;
        }
    for (long r = 0; r < N; ++r)
        for (int b = 0; b < K / 32; ++b) {
            const float mag = (magmode & 8) ? 1.f : mag_of(magmode, r, b, 9);
            for (int k = 0; k < 32; ++k) W[r * K + 32 * b + k] = (0.25f + 0.75f * fabsf(urand())) * (urand() < 0 ? -1.f : 1.f) * mag;
        }
    for (auto& v : bias) v = urand();
    for (auto& v : R) v = urand() * 4.f;
    float *dA = to_dev(A), *dW = to_dev(W), *dbias = to_dev(bias), *dR = to_dev(R);
    uint8_t *qA, *qW;
    uint32_t *sA, *sW;
    CK(hipMalloc(&qA, (size_t)M * K)); CK(hipMalloc(&qW, (size_t)N * K));
    CK(hipMalloc(&sA, (size_t)(K / 128) * a_srows * 4)); CK(hipMalloc(&sW, (size_t)(K / 128) * w_srows * 4));
    CK(hipMemset(sA, 0, (size_t)(K / 128) * a_srows * 4));
    VB(vb_quantize_rows_mx(nullptr, M, K, dA, K, qA, K, sA, a_srows));
    VB(vb_quantize_rows_mx(nullptr, N, K, dW, K, qW, K, sW, w_srows));
    CK(hipDeviceSynchronize());
    int bad = 0;
    std::vector<uint8_t> hqA, hqW;
    std::vector<uint32_t> hsA, hsW;
    quant_host(A.data(), M, K, a_srows, hqA, hsA);
    quant_host(W.data(), N, K, w_srows, hqW, hsW);
    {
        auto gq = to_host(qA, (size_t)M * K);
        auto gs = to_host(sA, (size_t)(K / 128) * a_srows);
        long dq = 0, ds = 0;
        for (size_t i = 0; i < gq.size(); ++i) dq += gq[i] != hqA[i];
        for (int kt = 0; kt < K / 128; ++kt)
            for (long r = 0; r < M; ++r) ds += gs[kt * a_srows + r] != hsA[kt * a_srows + r];
        auto gqw = to_host(qW, (size_t)N * K);
        for (size_t i = 0; i < gqw.size(); ++i) dq += gqw[i] != hqW[i];
        printf("quantiser: %ld code mismatches, %ld scale-word mismatches (A) %s\n", dq, ds, dq + ds ? "FAIL" : "ok");
        bad += dq + ds != 0;
    }
    // fp64 reference on sampled rows from the HOST codes
    std::vector<int> rows;
    for (int r = 0; r < M; r += std::max(1, M / 61)) rows.push_back(r);
    rows.push_back(M - 1);
    if (M > 300) { rows.push_back(255); rows.push_back(256); rows.push_back(M - 2); }
    std::vector<double> ref(rows.size() * (size_t)N), mag(rows.size() * (size_t)N);
    std::vector<float> dqW((size_t)N * K);
    for (long n = 0; n < N; ++n)
        for (int k = 0; k < K; ++k)
            dqW[n * K + k] = e4m3_value(hqW[n * K + k]) * scale_of((hsW[(size_t)(k / 128) * w_srows + n] >> (8 * ((k / 32) % 4))) & 0xff);
    for (size_t ri = 0; ri < rows.size(); ++ri) {
        const long r = rows[ri];
        std::vector<float> a(K);
        for (int k = 0; k < K; ++k)
            a[k] = e4m3_value(hqA[r * K + k]) * scale_of((hsA[(size_t)(k / 128) * a_srows + r] >> (8 * ((k / 32) % 4))) & 0xff);
        for (int n = 0; n < N; ++n) {
            double s = 0, m = 0;
            for (int k = 0; k < K; ++k) { const double p = (double)a[k] * dqW[(size_t)n * K + k]; s += p; m += fabs(p); }
            ref[ri * N + n] = s; mag[ri * N + n] = m;
        }
    }
    float* dC; uint16_t* dCb; uint8_t* dCq; uint32_t* dCs;
    CK(hipMalloc(&dC, (size_t)M * N * 4)); CK(hipMalloc(&dCb, (size_t)M * N * 2)); CK(hipMalloc(&dCq, (size_t)M * N));
    CK(hipMalloc(&dCs, (size_t)(N / 128) * c_srows * 4));
    vb_linear_mx_args g{};
    g.A = qA; g.lda = K; g.a_scales = sA; g.a_srows = a_srows;
    g.W = qW; g.ldw = K; g.w_scales = sW; g.w_srows = w_srows;
    g.M = M; g.N = N; g.K = K;
    // (1) fp32 + bf16 output, bias + residual
    g.bias = dbias; g.residual = dR; g.ldr = N; g.C = dC; g.ldc = N; g.act = VB_ACT_NONE;
    CK(hipMemset(dC, 0xff, (size_t)M * N * 4));
    VB(vb_linear_fwd_mx(nullptr, &g));
    g.C = nullptr; g.Cb = dCb; g.ldb16 = N;       // one output form per launch
    VB(vb_linear_fwd_mx(nullptr, &g));
    CK(hipDeviceSynchronize());
    {
        auto c = to_host(dC, (size_t)M * N);
        auto cb = to_host(dCb, (size_t)M * N);
        double worst = 0, worst_b = 0;
        long nbad = 0;
        for (size_t ri = 0; ri < rows.size(); ++ri)
            for (int n = 0; n < N; ++n) {
                const long r = rows[ri];
                const double want = ref[ri * N + n] + bias[n] + R[r * N + n];
                const double tol = 2e-5 * mag[ri * N + n] + 1e-5 * fabs(want) + 1e-6;
                const double err = fabs((double)c[r * N + n] - want);
                if (!(err <= tol)) { if (nbad < 5) printf("  fp32 out (%ld,%d): got %g want %g\n", r, n, c[r * N + n], want); ++nbad; }
                worst = std::max(worst, err / tol);
                const double eb = fabs((double)bf16_value(cb[r * N + n]) - want);
                if (!(eb <= fabs(want) / 256.0 + tol)) { if (nbad < 5) printf("  bf16 out (%ld,%d): got %g want %g\n", r, n, bf16_value(cb[r * N + n]), want); ++nbad; }
                worst_b = std::max(worst_b, eb / (fabs(want) / 256.0 + tol));
            }
        printf("fp32 / bf16 output, bias + residual: %ld bad of %zu (worst err / tol %.3f, bf16 %.3f) %s\n", nbad, rows.size() * (size_t)N,
               worst, worst_b, nbad ? "FAIL" : "ok");
        bad += nbad != 0;
    }
    // (2) MX output, bias + GELU
    g.residual = nullptr; g.C = nullptr; g.Cb = nullptr; g.Cq = dCq; g.ldq = N; g.c_scales = dCs; g.c_srows = c_srows; g.act = VB_ACT_GELU;
    CK(hipMemset(dCs, 0, (size_t)(N / 128) * c_srows * 4));
    VB(vb_linear_fwd_mx(nullptr, &g));
    CK(hipDeviceSynchronize());
    {
        auto cq = to_host(dCq, (size_t)M * N);
        auto cs = to_host(dCs, (size_t)(N / 128) * c_srows);
        long nbad = 0, smis = 0, blocks = 0;
        for (size_t ri = 0; ri < rows.size(); ++ri) {
            const long r = rows[ri];
            for (int b = 0; b < N / 32; ++b) {
                double amax = 0;
                for (int k = 0; k < 32; ++k) amax = std::max(amax, fabs(gelu_ref(ref[ri * N + 32 * b + k] + bias[32 * b + k])));
                const unsigned want_byte = scale_byte((float)amax);
                const unsigned got_byte = (cs[(size_t)(b / 4) * c_srows + r] >> (8 * (b % 4))) & 0xff;
                ++blocks;
                if (want_byte != got_byte) {
                    ++smis;
                    if (abs((int)want_byte - (int)got_byte) > 1) { if (nbad < 5) printf("  scale (%ld, block %d): got %u want %u\n", r, b, got_byte, want_byte); ++nbad; }
                }
                const double sc = scale_of(got_byte);
                for (int k = 0; k < 32; ++k) {
                    const int n = 32 * b + k;
                    const double want = gelu_ref(ref[ri * N + n] + bias[n]);
                    const double got = e4m3_value(cq[r * N + n]) * sc;
                    const double tol = fabs(want) / 16.0 + sc * (1.0 / 1024.0) * 1.01 + 4e-5 * mag[ri * N + n] + 1e-6;
                    if (!(fabs(got - want) <= tol)) { if (nbad < 5) printf("  MX out (%ld,%d): got %g want %g (scale %g)\n", r, n, got, want, sc); ++nbad; }
                }
            }
        }
        printf("MX output, bias + GELU: %ld bad, %ld of %ld scale bytes differ by one (amax at a binade edge) %s\n", nbad, smis, blocks,
               (nbad || smis * 50 > blocks) ? "FAIL" : "ok");
        bad += (nbad != 0) || (smis * 50 > blocks);
    }
    hipFree(dA); hipFree(dW); hipFree(dbias); hipFree(dR); hipFree(qA); hipFree(qW); hipFree(sA); hipFree(sW);
    hipFree(dC); hipFree(dCb); hipFree(dCq); hipFree(dCs);
    return bad;
}

template <class F>
static double time_us(F fn, int iters) {
    for (int i = 0; i < 3; ++i) fn();
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < iters; ++i) fn();
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1000.0 / iters;
}

static void time_case(int M, int N, int K, int mode) {
    // mode 0: fp32 out + bias; 1: fp32 out + bias + residual; 2: MX out + GELU; 3: bf16 out + bias
    const long a_srows = (M + 255) / 256 * 256, w_srows = N, c_srows = M;
    uint8_t *qA, *qW, *dCq; uint32_t *sA, *sW, *dCs; float *dC, *dR, *dbias; uint16_t* dCb;
    CK(hipMalloc(&qA, (size_t)M * K)); CK(hipMalloc(&qW, (size_t)N * K));
    CK(hipMemset(qA, 0x38, (size_t)M * K)); CK(hipMemset(qW, 0x30, (size_t)N * K));
    CK(hipMalloc(&sA, (size_t)(K / 128) * a_srows * 4)); CK(hipMalloc(&sW, (size_t)(K / 128) * w_srows * 4));
    CK(hipMemset(sA, 0x7f, (size_t)(K / 128) * a_srows * 4)); CK(hipMemset(sW, 0x7f, (size_t)(K / 128) * w_srows * 4));
    CK(hipMalloc(&dC, (size_t)M * N * 4)); CK(hipMalloc(&dR, (size_t)M * N * 4)); CK(hipMalloc(&dbias, N * 4));
    CK(hipMalloc(&dCb, (size_t)M * N * 2)); CK(hipMalloc(&dCq, (size_t)M * N)); CK(hipMalloc(&dCs, (size_t)(N / 128) * c_srows * 4));
    CK(hipMemset(dR, 0, (size_t)M * N * 4)); CK(hipMemset(dbias, 0, N * 4));
    vb_linear_mx_args g{};
    g.A = qA; g.lda = K; g.a_scales = sA; g.a_srows = a_srows; g.W = qW; g.ldw = K; g.w_scales = sW; g.w_srows = w_srows;
    g.M = M; g.N = N; g.K = K; g.bias = dbias;
    if (mode == 0 || mode == 1) { g.C = dC; g.ldc = N; }
    if (mode == 1) { g.residual = dR; g.ldr = N; }
    if (mode == 2) { g.Cq = dCq; g.ldq = N; g.c_scales = dCs; g.c_srows = c_srows; g.act = VB_ACT_GELU; }
    if (mode == 3) { g.Cb = dCb; g.ldb16 = N; }
    const double us = time_us([&] { VB(vb_linear_fwd_mx(nullptr, &g)); }, 20);
    static const char* names[] = {"fp32 out", "fp32 out + residual", "MX out + GELU", "bf16 out"};
    printf("M=%6d N=%5d K=%5d  %-20s %8.1f us  %7.1f TF\n", M, N, K, names[mode], us, 2.0 * M * N * K / us * 1e-6);
    hipFree(qA); hipFree(qW); hipFree(sA); hipFree(sW); hipFree(dC); hipFree(dR); hipFree(dbias); hipFree(dCb); hipFree(dCq); hipFree(dCs);
}

int main(int argc, char** argv) {
    const std::string what = argc > 1 ? argv[1] : "all";
    int bad = 0;
    if (what == "diag") {
        for (int mode : {0, 1 + 8, 2 + 8, 1 + 4, 2 + 4, 3}) {
            char buf[8]; snprintf(buf, sizeof buf, "%d", mode); setenv("MX_LAB_MAG", buf, 1);
            printf("==== MX_LAB_MAG=%d\n", mode);
            check_case(256, 128, 128);
            check_case(256, 128, 256);
        }
        return 0;
    }
    if (what == "check" || what == "all") {
        bad += check_case(256, 128, 128);        // one tile, one K tile
        bad += check_case(512, 256, 768);        // 4 tiles, 6 K tiles
        bad += check_case(1000, 384, 1024);      // ragged M
        bad += check_case(2304, 1024, 384);      // 72 tiles; odd number of K tiles
        bad += check_case(9216, 768, 256);       // 216 tiles: more than 256 CUs' worth with N = 768? (36 x 6)
        bad += check_case(4096, 3072, 128);      // 384 tiles: two rounds, 4 x 8 patches
        printf(bad ? "CHECK FAILED (%d)\n" : "CHECK OK\n", bad);
    }
    if (what == "order") {     // the same two launches alternately: order / clock effects of the harness
        for (int rep = 0; rep < 3; ++rep) { time_case(18432, 3072, 768, 0); time_case(18432, 3072, 768, 2); time_case(18432, 3072, 768, 3); }
        return 0;
    }
    if (what == "time" || what == "all") {
        const int M = 18432;
        time_case(M, 2304, 768, 0);
        time_case(M, 2304, 768, 3);
        time_case(M, 768, 768, 1);
        time_case(M, 3072, 768, 2);
        time_case(M, 3072, 768, 0);
        time_case(M, 768, 3072, 1);
        time_case(M, 3072, 1024, 0);
        time_case(M, 1024, 1024, 1);
        time_case(M, 1024, 1024, 2);
        time_case(M, 1024, 2048, 0);
    }
    return bad ? 1 : 0;
}
