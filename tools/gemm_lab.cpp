// Stand-alone GEMM laboratory: times (and checks) the three vb_linear_* entry points of libvilbert_hip.so on the
// model's shapes without Python. Build + run (GPU box):
//     make -C tools gemm_lab && [VB_GEMM_V2=0|1] [VB_GEMM_TILE=33|34|43|44|22] [VB_GEMM_ABL=1|2] tools/gemm_lab [quick|check]
// Prints one line per shape: kind, M, N, K, nseg, microseconds, TFLOP/s (algorithmic 2MNK), max relative error
// against an fp64-accumulating reference kernel.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <map>
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

__global__ void fill_kernel(float* p, long n, uint32_t seed, float scale) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t x = (uint32_t)(i * 2654435761u) ^ seed;
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    p[i] = ((float)(x >> 8) * (1.0f / 16777216.0f) - 0.5f) * 2.0f * scale;
}

// C[m][n] = sum_k A(m,k) B(n,k) with generic strides (fp64 accumulate); sampled rows only
__global__ void ref_kernel(int M, int N, int K, const float* A, long a_rs, long a_cs, const float* B, long b_rs, long b_cs,
                           double* C, int row_step) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    const int m = blockIdx.y * row_step;
    if (n >= N || m >= M) return;
    double s = 0.0;
    for (int k = 0; k < K; ++k) s += (double)A[m * a_rs + k * a_cs] * (double)B[n * b_rs + k * b_cs];
    C[(long)blockIdx.y * N + n] = s;
}

static float* dev_rand(long n, uint32_t seed, float scale) {
    float* p;
    CK(hipMalloc(&p, n * sizeof(float)));
    hipLaunchKernelGGL(fill_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, p, n, seed, scale);
    return p;
}

template <class F>
static double time_us(F fn, int iters) {
    for (int i = 0; i < 3; ++i) fn();
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < iters; ++i) fn();
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1e3 / iters;
}

// compares sampled rows of C (row-major [M, N], ld) against the fp64 reference
static double check(int M, int N, int K, const float* A, long a_rs, long a_cs, const float* B, long b_rs, long b_cs,
                    const float* C, long ldc, const float* bias) {
    const int row_step = M > 64 ? M / 61 : 1;
    const int rows = (M + row_step - 1) / row_step;
    double* ref;
    CK(hipMalloc(&ref, (size_t)rows * N * sizeof(double)));
    hipLaunchKernelGGL(ref_kernel, dim3((N + 127) / 128, rows), dim3(128), 0, 0, M, N, K, A, a_rs, a_cs, B, b_rs, b_cs, ref,
                       row_step);
    std::vector<double> h((size_t)rows * N);
    std::vector<float> c((size_t)N), hb(bias ? N : 0);
    CK(hipMemcpy(h.data(), ref, h.size() * sizeof(double), hipMemcpyDeviceToHost));
    if (bias) CK(hipMemcpy(hb.data(), bias, N * sizeof(float), hipMemcpyDeviceToHost));
    double worst = 0.0, scale = 1e-30;
    for (size_t i = 0; i < h.size(); ++i) scale = fmax(scale, fabs(h[i]));
    for (int r = 0; r < rows; ++r) {
        CK(hipMemcpy(c.data(), C + (long)r * row_step * ldc, N * sizeof(float), hipMemcpyDeviceToHost));
        for (int n = 0; n < N; ++n) {
            const double want = h[(size_t)r * N + n] + (bias ? hb[n] : 0.0);
            worst = fmax(worst, fabs(c[n] - want) / scale);
        }
    }
    CK(hipFree(ref));
    return worst;
}

extern "C" void vblab_gemm_cycles(unsigned long long* dev_buf);
// LAB_TIMELINE=1 (library built with LAB=1): one instrumented launch per shape / kind; every block records realtime
// (100 MHz) stamps {start, after prologue, after K loop, after epilogue} + shader cycles; summary per launch.
static const int TL_MAX_BLOCKS = 1 << 16;
static unsigned long long* g_tl = nullptr;
template <class F>
static void timeline(const char* kind, F fn) {
    if (!getenv("LAB_TIMELINE")) return;
    if (!g_tl) CK(hipMalloc(&g_tl, (size_t)TL_MAX_BLOCKS * 64 + 256 * 128 + 3000 * 8));
    for (int rep = 0; rep < 2; ++rep) {
    CK(hipMemset(g_tl, 0, (size_t)TL_MAX_BLOCKS * 64 + 256 * 128 + 3000 * 8));
    fn(); fn();
    CK(hipDeviceSynchronize());
    vblab_gemm_cycles(g_tl);
    fn();
    CK(hipDeviceSynchronize());
    vblab_gemm_cycles(nullptr);
    std::vector<unsigned long long> h((size_t)TL_MAX_BLOCKS * 8);
    CK(hipMemcpy(h.data(), g_tl, h.size() * 8, hipMemcpyDeviceToHost));
    std::vector<double> st, pro, loop, epi, en, clk;
    unsigned long long t0 = ~0ull, t1 = 0;
    int per_xcd[8] = {0};
    for (int b = 0; b < TL_MAX_BLOCKS; ++b) {
        const unsigned long long* r = &h[(size_t)b * 8];
        if (r[0] == 0 || r[3] == 0) continue;
        if (r[0] < t0) t0 = r[0];
        if (r[3] > t1) t1 = r[3];
    }
    for (int b = 0; b < TL_MAX_BLOCKS; ++b) {
        const unsigned long long* r = &h[(size_t)b * 8];
        if (r[0] == 0 || r[3] == 0) continue;
        st.push_back((r[0] - t0) * 0.01); pro.push_back((r[1] - r[0]) * 0.01); loop.push_back((r[2] - r[1]) * 0.01);
        epi.push_back((r[3] - r[2]) * 0.01); en.push_back((r[3] - t0) * 0.01);
        if (r[3] > r[0]) clk.push_back((double)(r[5] - r[4]) / ((r[3] - r[0]) * 10.0));
        per_xcd[(r[6] >> 32) & 7]++;
    }
    if (st.empty()) { printf("  timeline %s: no stamps (library not built with LAB=1?)\n", kind); return; }
    auto stat = [](std::vector<double> v, const char* name) {
        std::sort(v.begin(), v.end());
        double s = 0; for (double x : v) s += x;
        printf("    %-9s min %7.2f  p10 %7.2f  med %7.2f  p90 %7.2f  max %7.2f  mean %7.2f\n", name, v.front(), v[v.size() / 10],
               v[v.size() / 2], v[v.size() * 9 / 10], v.back(), s / v.size());
    };
    printf("  timeline %s rep %d: %zu blocks, span %.2f us (first start -> last end), blocks per XCD %d %d %d %d %d %d %d %d\n", kind, rep, st.size(),
           (t1 - t0) * 0.01, per_xcd[0], per_xcd[1], per_xcd[2], per_xcd[3], per_xcd[4], per_xcd[5], per_xcd[6], per_xcd[7]);
    stat(st, "start us"); stat(pro, "prologue"); stat(loop, "K loop"); stat(epi, "epilogue"); stat(en, "end us"); stat(clk, "clk GHz");
    {   // persistent kernel (gemm_v4.h): the loader waves' cycle split
        std::vector<unsigned long long> ls(256 * 16);
        CK(hipMemcpy(ls.data(), g_tl + (size_t)TL_MAX_BLOCKS * 8, ls.size() * 8, hipMemcpyDeviceToHost));
        double a = 0, v = 0, bb = 0, n = 0, steps = 0;
        for (int i = 0; i < 256; ++i) if (ls[16 * i + 3]) { a += ls[16 * i]; v += ls[16 * i + 1]; bb += ls[16 * i + 2]; steps += ls[16 * i + 3]; n++; }
        {   // block 0, MFMA wave 0: per K step cycles and clock
            std::vector<unsigned long long> sp(3000);
            CK(hipMemcpy(sp.data(), g_tl + (size_t)TL_MAX_BLOCKS * 8 + 16 * 256, sp.size() * 8, hipMemcpyDeviceToHost));
            int ns = 0;
            while (ns < 1000 && sp[3 * ns + 1]) ++ns;
            if (ns > 8) {
                printf("    block 0 wave 0, %d steps: [step: cycles since previous barrier exit | cycles waiting at the barrier | clock GHz]\n     ", ns);
                for (int i = 1; i < ns; ++i) {
                    const double dt = (sp[3 * i + 2] - sp[3 * (i - 1) + 2]) * 10.0;   // ns
                    if (i < 12 || i % 8 == 0 || i > ns - 6)
                        printf(" [%d: %llu | %llu | %.2f]", i, sp[3 * i + 1] - sp[3 * (i - 1) + 1], sp[3 * i + 1] - sp[3 * i], dt > 0 ? (sp[3 * i + 1] - sp[3 * (i - 1) + 1]) / dt : 0.0);
                }
                printf("\n");
            }
        }
        for (int blk : {0, 100}) {
            if (!ls[16 * blk + 3]) continue;
            printf("    block %d rounds (us since block start): ", blk);
            for (int r = 0; r < 6 && ls[16 * blk + 4 + 2 * r]; ++r)
                printf(" [K loop end %.2f, epilogue end %.2f]", (ls[16 * blk + 4 + 2 * r] - h[(size_t)blk * 8]) * 0.01, (ls[16 * blk + 5 + 2 * r] - h[(size_t)blk * 8]) * 0.01);
            printf("\n");
        }
        if (n > 0) printf("    loader waves (%g): per K step %.0f cycles issuing, %.0f waiting for DMA, %.0f at the barrier (total %.0f)\n", n,
                          a / steps, v / steps, bb / steps, (a + v + bb) / steps);
    }
    // per compute unit (key = XCC id, HW_ID bits [15:8] = se / sh / cu): blocks placed there, when its last block ended,
    // and the K-loop time of its blocks summed (= how long the unit's matrix pipes had this launch's work queued)
    struct Cu { int n = 0; double last = 0, first_end = 1e30, ksum = 0; };
    std::map<unsigned, Cu> cus;
    for (int b = 0; b < TL_MAX_BLOCKS; ++b) {
        const unsigned long long* r = &h[(size_t)b * 8];
        if (r[0] == 0 || r[3] == 0) continue;
        Cu& c = cus[(unsigned)(((r[6] >> 32) & 7) << 8 | ((r[6] >> 8) & 0xff))];
        c.n++;
        c.last = std::max(c.last, (r[3] - t0) * 0.01);
        c.first_end = std::min(c.first_end, (r[3] - t0) * 0.01);
        c.ksum += (r[2] - r[1]) * 0.01;
    }
    std::map<int, int> hist;
    std::vector<double> last, first;
    for (auto& kv : cus) { hist[kv.second.n]++; last.push_back(kv.second.last); first.push_back(kv.second.first_end); }
    printf("    %zu compute units; blocks per unit:", cus.size());
    for (auto& kv : hist) printf("  %d x%d", kv.first, kv.second);
    printf("\n");
    stat(first, "CU 1st end"); stat(last, "CU last end");
    for (auto& kv : hist) {
        std::vector<double> v;
        for (auto& c : cus) if (c.second.n == kv.first) v.push_back(c.second.last);
        char nm[32]; snprintf(nm, sizeof nm, "last|n=%d", kv.first);
        stat(v, nm);
    }
    }
}

extern "C" int vb_set_gemm_v4(int mode);
// LAB_V4_AB=1: the 4-wave kernels (mode 0) against the persistent kernel forced (mode 2), alternating in one process
template <class F>
static void v4_ab(const char* kind, double fl, F fn) {
    if (!getenv("LAB_V4_AB")) return;
    double best[2] = {1e30, 1e30};
    const int mode_a = getenv("LAB_AB_MODE_A") ? atoi(getenv("LAB_AB_MODE_A")) : 0;
    for (int rep = 0; rep < 3; ++rep)
        for (int m = 0; m < 2; ++m) {
            vb_set_gemm_v4(m ? 2 : mode_a);
            best[m] = std::min(best[m], time_us(fn, 20));
        }
    vb_set_gemm_v4(1);
    printf("  A/B %-5s 4-wave %8.1f us %6.1f TF | persistent %8.1f us %6.1f TF | %+5.1f %%\n", kind, best[0], fl / best[0] / 1e6, best[1],
           fl / best[1] / 1e6, 100.0 * (best[0] / best[1] - 1.0));
}

struct Shape { int M, N, K, nseg; };

int main(int argc, char** argv) {
    setvbuf(stdout, nullptr, _IONBF, 0);
    bool quick = false, do_check = true;
    for (int i = 1; i < argc; ++i) {
        if (!strcmp(argv[i], "quick")) quick = true;
        if (!strcmp(argv[i], "nocheck")) do_check = false;
    }
    const int Mr = getenv("LAB_M") ? atoi(getenv("LAB_M")) : 9216;
    std::vector<Shape> shapes = {{Mr, 768, 768, 1}, {Mr, 768, 768, 3}, {Mr, 3072, 768, 1}, {Mr, 768, 3072, 1},
                                 {Mr, 1024, 1024, 1}, {Mr, 1024, 1024, 3}, {Mr, 1024, 768, 3}, {Mr, 1024, 2048, 1}};
    if (getenv("LAB_LARGE")) shapes = {{Mr, 1024, 1024, 1}, {Mr, 1024, 1024, 3}, {Mr, 4096, 1024, 1}, {Mr, 1024, 4096, 1}};
    if (quick) shapes = {{Mr, 768, 768, 1}, {Mr, 3072, 768, 1}, {Mr, 1024, 1024, 3}};
    if (const char* e = getenv("LAB_SHAPES")) {       // "N,K,nseg;N,K,nseg;..." at LAB_M rows
        shapes.clear();
        std::string str(e);
        size_t pos = 0;
        while (pos < str.size()) {
            size_t end = str.find(';', pos);
            if (end == std::string::npos) end = str.size();
            int n = 0, k = 0, sg = 1;
            if (sscanf(str.substr(pos, end - pos).c_str(), "%d,%d,%d", &n, &k, &sg) >= 2) shapes.push_back({Mr, n, k, sg});
            pos = end + 1;
        }
    }
    const bool no_wgrad = getenv("LAB_NOWGRAD") != nullptr;
    printf("VB_GEMM_V2=%s VB_GEMM_TILE=%s VB_GEMM_ABL=%s M=%d\n", getenv("VB_GEMM_V2") ? getenv("VB_GEMM_V2") : "-",
           getenv("VB_GEMM_TILE") ? getenv("VB_GEMM_TILE") : "-", getenv("VB_GEMM_ABL") ? getenv("VB_GEMM_ABL") : "-", Mr);
    double tot_f = 0, tot_t = 0;
    {   // clock warm-up: ~0.3 s of back-to-back GEMMs before anything is timed (the shader clock ramps over milliseconds)
        const int M = 9216, N = 3072, K = 768;
        float* x = dev_rand((long)M * K, 1, 1.0f); float* w = dev_rand((long)N * K, 2, 0.05f); float* y = dev_rand((long)M * N, 4, 0.0f);
        vb_linear_args a; memset(&a, 0, sizeof(a));
        a.M = M; a.K = K; a.nseg = 1; a.seg_n = N; a.A = x; a.lda = K; a.ldw = K; a.C = y; a.ldc = N; a.W[0] = w;
        for (int i = 0; i < (getenv("LAB_WARM") ? atoi(getenv("LAB_WARM")) : 800); ++i) vb_linear_fwd(nullptr, &a);
        CK(hipDeviceSynchronize());
        for (float* p : {x, w, y}) CK(hipFree(p));
    }
    for (const Shape& s : shapes) {
        const int M = s.M, n = s.N, K = s.K, nseg = s.nseg, N = n * nseg;
        float* x = dev_rand((long)M * K, 1, 1.0f);
        float* w = dev_rand((long)N * K, 2, 0.05f);     // the nseg weight blocks, contiguous here
        float* b = dev_rand(N, 3, 1.0f);
        float* y = dev_rand((long)M * N, 4, 0.0f);
        float* dy = dev_rand((long)M * N, 5, 1.0f);
        float* dx = dev_rand((long)M * K, 6, 0.0f);
        float* dw = dev_rand((long)N * K, 7, 0.0f);
        float* db = dev_rand(N, 8, 0.0f);
        const double fl = 2.0 * M * N * K;
        const int iters = 20;
        // forward
        vb_linear_args a;
        memset(&a, 0, sizeof(a));
        a.M = M; a.K = K; a.nseg = nseg; a.seg_n = n; a.A = x; a.lda = K; a.ldw = K; a.C = y; a.ldc = N;
        for (int i = 0; i < nseg; ++i) { a.W[i] = w + (long)i * n * K; a.bias[i] = b + (long)i * n; }
        double us = time_us([&] { int e = vb_linear_fwd(nullptr, &a); if (e) { fprintf(stderr, "fwd err %d\n", e); exit(1); } }, iters);
        double err = do_check ? check(M, N, K, x, K, 1, w, K, 1, y, N, b) : -1;
        printf("fwd   M=%5d N=%5d K=%5d nseg=%d %9.1f us %6.1f TF  err %.1e\n", M, n, K, nseg, us, fl / us / 1e6, err);
        timeline("fwd", [&] { vb_linear_fwd(nullptr, &a); });
        v4_ab("fwd", fl, [&] { vb_linear_fwd(nullptr, &a); });
        tot_f += fl; tot_t += us;
        // dgrad: dx[M,K] = dy[M,N] . W[N,K]
        vb_linear_bwd_input_args g;
        memset(&g, 0, sizeof(g));
        g.M = M; g.K = K; g.nseg = nseg; g.seg_n = n; g.dY = dy; g.ldy = N; g.ldw = K; g.dX = dx; g.ldx = K;
        for (int i = 0; i < nseg; ++i) g.W[i] = w + (long)i * n * K;
        us = time_us([&] { int e = vb_linear_bwd_input(nullptr, &g); if (e) { fprintf(stderr, "dgrad err %d\n", e); exit(1); } }, iters);
        err = do_check ? check(M, K, N, dy, N, 1, w, 1, K, dx, K, nullptr) : -1;
        printf("dgrad M=%5d N=%5d K=%5d nseg=%d %9.1f us %6.1f TF  err %.1e\n", M, n, K, nseg, us, fl / us / 1e6, err);
        timeline("dgrad", [&] { vb_linear_bwd_input(nullptr, &g); });
        v4_ab("dgrad", fl, [&] { vb_linear_bwd_input(nullptr, &g); });
        tot_f += fl; tot_t += us;
        if (no_wgrad) { for (float* p : {x, w, b, y, dy, dx, dw, db}) CK(hipFree(p)); continue; }
        // wgrad: dw[N,K] = dy^T . x (+ bias gradient)
        vb_linear_bwd_weight_args wg;
        memset(&wg, 0, sizeof(wg));
        wg.M = M; wg.K = K; wg.nseg = nseg; wg.seg_n = n; wg.dY = dy; wg.ldy = N; wg.X = x; wg.ldx = K; wg.ldw = K;
        for (int i = 0; i < nseg; ++i) { wg.dW[i] = dw + (long)i * n * K; wg.dbias[i] = db + (long)i * n; }
        us = time_us([&] { int e = vb_linear_bwd_weight(nullptr, &wg); if (e) { fprintf(stderr, "wgrad err %d\n", e); exit(1); } }, iters);
        err = do_check ? check(N, K, M, dy, 1, N, x, 1, K, dw, K, nullptr) : -1;
        printf("wgrad M=%5d N=%5d K=%5d nseg=%d %9.1f us %6.1f TF  err %.1e\n", M, n, K, nseg, us, fl / us / 1e6, err);
        timeline("wgrad", [&] { vb_linear_bwd_weight(nullptr, &wg); });
        wg.accumulate = 1;
        v4_ab("wgrad", fl, [&] { vb_linear_bwd_weight(nullptr, &wg); });
        wg.accumulate = 0;
        tot_f += fl; tot_t += us;
        if (getenv("LAB_SPLIT")) {
            // forward as ONE launch over M rows vs TWO launches over M / 2 rows each on two streams (micro-batch halves)
            hipStream_t s0, s1;
            CK(hipStreamCreate(&s0));
            CK(hipStreamCreate(&s1));
            vb_linear_args h0 = a, h1 = a;
            h0.M = M / 2; h1.M = M - M / 2;
            h1.A = x + (long)(M / 2) * K; h1.C = y + (long)(M / 2) * N;
            for (int two = 0; two < 2; ++two) {
                auto run = [&] {
                    if (!two) { if (vb_linear_fwd(s0, &a)) exit(1); }
                    else if (vb_linear_fwd(s0, &h0) || vb_linear_fwd(s1, &h1)) exit(1);
                };
                for (int i = 0; i < 3; ++i) run();
                CK(hipDeviceSynchronize());
                hipEvent_t e0, e1, j;
                CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&j));
                CK(hipEventRecord(e0, s0));
                CK(hipStreamWaitEvent(s1, e0, 0));
                for (int i = 0; i < iters; ++i) run();
                CK(hipEventRecord(j, s1));
                CK(hipStreamWaitEvent(s0, j, 0));
                CK(hipEventRecord(e1, s0));
                CK(hipEventSynchronize(e1));
                float ms;
                CK(hipEventElapsedTime(&ms, e0, e1));
                printf("  fwd x %d as %s: %9.1f us  %6.1f TF\n", iters, two ? "two half-batches on two streams" : "one launch",
                       ms * 1e3 / iters, fl / (ms * 1e3 / iters) / 1e6);
            }
            CK(hipStreamDestroy(s0));
            CK(hipStreamDestroy(s1));
        }
        if (getenv("LAB_STREAMS")) {
            // backward-shaped concurrency experiment: 20 x (dgrad, wgrad) on ONE stream vs dgrad on stream 0 and the
            // (off-critical-path) wgrad on stream 1 - does a second independent kernel fill the tails / prologues?
            hipStream_t s0, s1;
            CK(hipStreamCreate(&s0));
            CK(hipStreamCreate(&s1));
            wg.accumulate = 1;
            for (int two = 0; two < 2; ++two) {
                auto pair = [&] {
                    if (vb_linear_bwd_input(s0, &g) || vb_linear_bwd_weight(two ? s1 : s0, &wg)) exit(1);
                };
                for (int i = 0; i < 3; ++i) pair();
                CK(hipDeviceSynchronize());
                hipEvent_t e0, e1, j;
                CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&j));
                CK(hipEventRecord(e0, s0));
                CK(hipStreamWaitEvent(s1, e0, 0));
                for (int i = 0; i < iters; ++i) pair();
                CK(hipEventRecord(j, s1));
                CK(hipStreamWaitEvent(s0, j, 0));
                CK(hipEventRecord(e1, s0));
                CK(hipEventSynchronize(e1));
                float ms;
                CK(hipEventElapsedTime(&ms, e0, e1));
                printf("  (dgrad + wgrad) x %d on %d stream(s): %9.1f us per pair  %6.1f TF\n", iters, two + 1,
                       ms * 1e3 / iters, 2 * fl / (ms * 1e3 / iters) / 1e6);
            }
            CK(hipStreamDestroy(s0));
            CK(hipStreamDestroy(s1));
        }
        for (float* p : {x, w, b, y, dy, dx, dw, db}) CK(hipFree(p));
    }
    printf("aggregate: %.1f TF\n", tot_f / tot_t / 1e6);
    return 0;
}
