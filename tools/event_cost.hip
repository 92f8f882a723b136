// What does a cross-stream rendezvous cost the PRODUCER stream on this runtime? (round 6, DESIGN.md section 6: a stream that only
// ever waits on events of the compute stream made the bf16 step 10.9 ms slower)
//   hipcc --offload-arch=gfx950 -O2 tools/event_cost.hip -o tools/event_cost && tools/event_cost
// Stream A runs N short kernels back to back; every `every`-th kernel is followed by hipEventRecord on A and, depending on the
// scenario, hipStreamWaitEvent on stream B (idle otherwise, or followed by a tiny kernel on B). Wall time of A per kernel.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void work(float* p, int iters) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    float v = p[i];
    for (int k = 0; k < iters; ++k) v = v * 1.0001f + 0.5f;
    p[i] = v;
}
__global__ void tiny(float* p) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1.f; }

int main() {
    const int N = 2000, every = 10;
    float *a, *b;
    CK(hipMalloc(&a, 1024 * 256 * 4)); CK(hipMalloc(&b, 1024));
    CK(hipMemset(a, 0, 1024 * 256 * 4)); CK(hipMemset(b, 0, 1024));
    hipStream_t A, B;
    CK(hipStreamCreateWithFlags(&A, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&B, hipStreamNonBlocking));
    struct F { const char* name; unsigned flags; };
    const F flagsets[] = {{"default", hipEventDefault}, {"disable-timing", hipEventDisableTiming},
                          {"disable-timing | release-to-device", hipEventDisableTiming | hipEventReleaseToDevice},
                          {"disable-timing | release-to-system", hipEventDisableTiming | hipEventReleaseToSystem}};
    for (int iters : {200, 2000}) {
        for (int sc = 0; sc < 5; ++sc) {
            for (const F& f : flagsets) {
                if (sc == 0 && f.flags != hipEventDefault) continue;
                std::vector<hipEvent_t> ev(N / every);
                for (auto& e : ev) CK(hipEventCreateWithFlags(&e, f.flags));
                for (int rep = 0; rep < 2; ++rep) {                       // rep 0 = warm-up
                    CK(hipDeviceSynchronize());
                    const auto t0 = std::chrono::steady_clock::now();
                    for (int i = 0; i < N; ++i) {
                        hipLaunchKernelGGL(work, dim3(1024), dim3(256), 0, A, a, iters);
                        if (sc >= 1 && i % every == every - 1) {
                            hipEvent_t e = ev[i / every];
                            CK(hipEventRecord(e, A));
                            if (sc >= 2) CK(hipStreamWaitEvent(B, e, 0));
                            if (sc == 3) hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, B, b);
                            if (sc == 4) {                                 // B joins back into A (fork + join)
                                hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, B, b);
                                CK(hipEventRecord(e, B));
                                CK(hipStreamWaitEvent(A, e, 0));
                            }
                        }
                    }
                    const auto t1 = std::chrono::steady_clock::now();
                    CK(hipDeviceSynchronize());
                    const auto t2 = std::chrono::steady_clock::now();
                    if (rep == 1) {
                        const char* names[] = {"A alone", "A + event record every 10th kernel (no waiter)", "+ idle stream B waits on each event",
                                               "+ B waits and runs a tiny kernel", "+ B waits, runs a tiny kernel, A waits for B (fork + join)"};
                        printf("kernel ~%s  %-62s %-38s  %.2f us per kernel wall, %.2f us host enqueue\n", iters == 200 ? "short" : "long ",
                               names[sc], sc ? f.name : "", std::chrono::duration<double, std::micro>(t2 - t0).count() / N,
                               std::chrono::duration<double, std::micro>(t1 - t0).count() / N);
                    }
                }
                for (auto& e : ev) CK(hipEventDestroy(e));
            }
        }
    }
    return 0;
}
