// Kernels of the second-generation fp32 GEMM main loop (gemm_v2.h), one object per operand layout:
//   -DVB_V2_LAYOUT=0  forward  (NT)  A k-contiguous,   B k-contiguous
//   -DVB_V2_LAYOUT=1  dgrad    (NN)  A k-contiguous,   B row-contiguous
//   -DVB_V2_LAYOUT=2  wgrad    (TN)  A row-contiguous, B row-contiguous
// Each object instantiates the tile menu {64x64, 96x96, 96x128, 128x96, 128x128} (block tile = 32 TM x 32 TN).
#include "gemm_v2.h"

#ifndef VB_V2_LAYOUT
#error "compile with -DVB_V2_LAYOUT=0|1|2"
#endif

namespace {

using namespace vbgemm;

constexpr bool A_KC = VB_V2_LAYOUT != 2;
constexpr bool B_KC = VB_V2_LAYOUT == 0;

template <int TM, int TN, int ABL>
__global__ __launch_bounds__(256, (V2Cfg<TM, TN, A_KC, B_KC>::OCC)) void gemm_v2_kernel(const GemmP p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    // XCD-aware tile order: block b runs on XCD b % 8; each XCD walks a contiguous run of tiles (N fastest), so the
    // blocks that share an A panel share one L2
    const int t = xcd_swizzle(blockIdx.x, gridDim.x);
    gemm_tile_v2<TM, TN, A_KC, B_KC, ABL>(p, smem, (t / p.tiles_n) * (32 * TM), (t % p.tiles_n) * (32 * TN));
}

template <int TM, int TN>
int launch(hipStream_t st, const GemmP& p, int tiles, int splits) {
    using Cfg = V2Cfg<TM, TN, A_KC, B_KC>;
    static const int abl = [] { const char* e = getenv("VB_GEMM_ABL"); return e ? atoi(e) : 0; }();
    dim3 grid(tiles, splits), block(256);
    if (abl == 1) hipLaunchKernelGGL((gemm_v2_kernel<TM, TN, 1>), grid, block, Cfg::LDS_BYTES, st, p);
    else if (abl == 2) hipLaunchKernelGGL((gemm_v2_kernel<TM, TN, 2>), grid, block, Cfg::LDS_BYTES, st, p);
    else if (abl == 3) hipLaunchKernelGGL((gemm_v2_kernel<TM, TN, 3>), grid, block, Cfg::LDS_BYTES, st, p);
    else if (abl == 4) hipLaunchKernelGGL((gemm_v2_kernel<TM, TN, 4>), grid, block, Cfg::LDS_BYTES, st, p);
    else if (abl == 5) hipLaunchKernelGGL((gemm_v2_kernel<TM, TN, 5>), grid, block, Cfg::LDS_BYTES, st, p);
    else if (abl == 6) hipLaunchKernelGGL((gemm_v2_kernel<TM, TN, 6>), grid, block, Cfg::LDS_BYTES, st, p);
    else hipLaunchKernelGGL((gemm_v2_kernel<TM, TN, 0>), grid, block, Cfg::LDS_BYTES, st, p);
    VB_LAUNCH_CHECK();
    return 0;
}

int dispatch(hipStream_t st, const GemmP& p, int tm, int tn, int tiles, int splits) {
    if (tm == 2 && tn == 2) return launch<2, 2>(st, p, tiles, splits);
    if (tm == 3 && tn == 3) return launch<3, 3>(st, p, tiles, splits);
    if (tm == 3 && tn == 4) return launch<3, 4>(st, p, tiles, splits);
    if (tm == 4 && tn == 3) return launch<4, 3>(st, p, tiles, splits);
    if (tm == 4 && tn == 4) return launch<4, 4>(st, p, tiles, splits);
    return VB_E_RANGE;
}

}  // namespace

namespace vbgemm {
#if VB_V2_LAYOUT == 0
int launch_gemm_v2_nt(hipStream_t st, const GemmP& p, int tm, int tn, int tiles, int splits) { return dispatch(st, p, tm, tn, tiles, splits); }
#elif VB_V2_LAYOUT == 1
int launch_gemm_v2_nn(hipStream_t st, const GemmP& p, int tm, int tn, int tiles, int splits) { return dispatch(st, p, tm, tn, tiles, splits); }
#else
int launch_gemm_v2_tn(hipStream_t st, const GemmP& p, int tm, int tn, int tiles, int splits) { return dispatch(st, p, tm, tn, tiles, splits); }
#endif
}  // namespace vbgemm
