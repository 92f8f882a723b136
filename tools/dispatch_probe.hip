// Where does the hardware dispatcher put the workgroups of a grid that does not fill the chip?
// Each block records its XCC id, HW_ID (SE / CU) and start/end time.  Build: hipcc -shared -fPIC.
#include <hip/hip_runtime.h>
#include <stdint.h>

__global__ __launch_bounds__(256) void probe_kernel(uint32_t* out, int spin) {
    extern __shared__ float lds[];
    uint32_t xcc, hwid;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    const uint64_t t0 = __builtin_readcyclecounter();
    float v = threadIdx.x;
    for (int i = 0; i < spin; ++i) v = v * 1.0001f + 0.5f;
    lds[threadIdx.x] = v;
    __syncthreads();
    const uint64_t t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) {
        uint32_t* o = out + blockIdx.x * 6;
        o[0] = xcc; o[1] = hwid; o[2] = (uint32_t)t0; o[3] = (uint32_t)(t0 >> 32); o[4] = (uint32_t)(t1 - t0);
        o[5] = (uint32_t)lds[(threadIdx.x + 1) & 255];
    }
}

extern "C" int run_probe(void* stream, uint32_t* out, int blocks, int lds_bytes, int spin) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(probe_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    hipLaunchKernelGGL(probe_kernel, dim3(blocks), dim3(256), lds_bytes, static_cast<hipStream_t>(stream), out, spin);
    return (int)hipGetLastError();
}
