// Kernels of the second-generation fp32 GEMM main loop (gemm_v2.h), one object per operand layout:
//   -DVB_V2_LAYOUT=0  forward  (NT)  A k-contiguous,   B k-contiguous
//   -DVB_V2_LAYOUT=1  dgrad    (NN)  A k-contiguous,   B row-contiguous
//   -DVB_V2_LAYOUT=2  wgrad    (TN)  A row-contiguous, B row-contiguous
// Each object instantiates the tile menu {64x64, 96x96, 96x128, 128x96, 128x128} (block tile = 32 TM x 32 TN) and the
// mixed-height launches {128 | 96} x {128, 96}, {96 | 64} x {128, 96}.
#include "gemm_v4w.h"

#ifndef VB_V2_LAYOUT
#error "compile with -DVB_V2_LAYOUT=0|1|2"
#endif

namespace {

using namespace vbgemm;

constexpr bool A_KC = VB_V2_LAYOUT != 2;
constexpr bool B_KC = VB_V2_LAYOUT == 0;

// One launch may mix two tile heights (same width): blocks [0, n_big) cut rows [0, m_split) into (32 TM1)-row tiles,
// the rest cuts the remaining rows into (32 TM2)-row tiles. With 37 regions per sample the image stream has M =
// 9472 = 8 x 128 + 88 x 96 rows: 96 row tiles x 8 column tiles = 768 blocks, exactly 3 per CU, where a single
// height gives 592 (128) or 792 (96) blocks and a mostly empty last round.
template <int TM1, int TM2, int TN, int ABL>
__global__ __launch_bounds__(256, (V2Cfg<TM1, TN, A_KC, B_KC>::OCC)) void gemm_v2_kernel(const GemmP p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    // XCD-aware tile order: block b runs on XCD b % 8; each XCD walks a contiguous run of tiles (N fastest), so the
    // blocks that share an A panel share one L2
    const int b = blockIdx.x;
    if (TM1 == TM2 || b < p.n_big) {
        const int t = xcd_swizzle(b, TM1 == TM2 ? (int)gridDim.x : p.n_big);
        gemm_tile_v2<TM1, TN, A_KC, B_KC, ABL>(p, smem, (t / p.tiles_n) * (32 * TM1), (t % p.tiles_n) * (32 * TN));
    } else {
        const int t = xcd_swizzle(b - p.n_big, (int)gridDim.x - p.n_big);
        gemm_tile_v2<TM2, TN, A_KC, B_KC, ABL>(p, smem, p.m_split + (t / p.tiles_n) * (32 * TM2), (t % p.tiles_n) * (32 * TN));
    }
}

template <int TM1, int TM2, int TN>
int launch(hipStream_t st, const GemmP& p, int tiles, int splits) {
    using Cfg = V2Cfg<TM1, TN, A_KC, B_KC>;   // TM1 >= TM2: the taller tile sets the LDS size and the register budget
    dim3 grid(tiles, splits), block(256);
    // lab knob: extra dynamic LDS per block = fewer co-resident blocks per CU (occupancy experiments)
    static const int lds_pad = [] { const char* e = getenv("VB_GEMM_LDS_PAD"); return e ? atoi(e) : 0; }();
    const int lds_bytes = Cfg::LDS_BYTES + lds_pad;
#ifdef VB_GEMM_LAB
    // ablation variants for tools/gemm_lab (make LAB=1): 1 = no staging, 2 = no barrier, 4 = no global loads,
    // 5 = loads do not advance, 6 = contiguous load pattern - they time parts of the K loop and give WRONG results
    static const int abl = [] { const char* e = getenv("VB_GEMM_ABL"); return e ? atoi(e) : 0; }();
    if (abl == 1) hipLaunchKernelGGL((gemm_v2_kernel<TM1, TM2, TN, 1>), grid, block, lds_bytes, st, p);
    else if (abl == 2) hipLaunchKernelGGL((gemm_v2_kernel<TM1, TM2, TN, 2>), grid, block, lds_bytes, st, p);
    else if (abl == 4) hipLaunchKernelGGL((gemm_v2_kernel<TM1, TM2, TN, 4>), grid, block, lds_bytes, st, p);
    else if (abl == 5) hipLaunchKernelGGL((gemm_v2_kernel<TM1, TM2, TN, 5>), grid, block, lds_bytes, st, p);
    else if (abl == 6) hipLaunchKernelGGL((gemm_v2_kernel<TM1, TM2, TN, 6>), grid, block, lds_bytes, st, p);
    else if (abl == 3) hipLaunchKernelGGL((gemm_v2_kernel<TM1, TM2, TN, 3>), grid, block, lds_bytes, st, p);
    else if (abl == 7) hipLaunchKernelGGL((gemm_v2_kernel<TM1, TM2, TN, 7>), grid, block, lds_bytes, st, p);
    else
#endif
        hipLaunchKernelGGL((gemm_v2_kernel<TM1, TM2, TN, 0>), grid, block, lds_bytes, st, p);
    VB_LAUNCH_CHECK();
    return 0;
}

#if VB_V2_LAYOUT != 2
// persistent one-block-per-CU kernel (gemm_v4.h). TM2 != 0: two tile heights in one launch (blocks whose row tile is one of
// the first p.n_small run TM, the others TM2 - a block-uniform branch; registers and LDS are those of the taller tile)
template <int WM, int TM, int TM2, int TN>
__global__ __launch_bounds__((V4Cfg<WM, TM, TN, B_KC>::THREADS), (WM == 6 ? 4 : 3)) void gemm_v4_kernel(const GemmP p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    if (TM2 == 0) {
        gemm_block_v4<WM, TM, TN, B_KC, false>(p, smem);
    } else {
        const int r = 4 * (blockIdx.x & 7) + (blockIdx.x >> 6);
        if (r < p.n_small) gemm_block_v4<WM, TM, TN, B_KC, true>(p, smem);
        else gemm_block_v4<WM, (TM2 == 0 ? TM : TM2), TN, B_KC, true>(p, smem);
    }
}

template <int WM, int TM, int TM2, int TN>
int launch_v4(hipStream_t st, GemmP p) {
    using Cfg = V4Cfg<WM, TM, TN, B_KC>;
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_v4_kernel<WM, TM, TM2, TN>),
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS_BYTES);
    if (attr != hipSuccess) return (int)attr;
    // laboratory: VB_GEMM_V4_EXCL=0 lets the small-tile configurations share a CU with another block (no LDS padding)
    static const bool excl = [] { const char* e = getenv("VB_GEMM_V4_EXCL"); return e == nullptr || atoi(e) != 0; }();
    const int lds_bytes = excl ? Cfg::LDS_BYTES : Cfg::RING_BYTES;
    p.tiles_n = p.N / Cfg::BN;
    int grid;
    if (TM2 != 0) {
        // 32 row tiles: n_tall of 16 TM WM rows, the rest of 16 TM2 WM rows, covering M exactly (checked by plan_v4)
        constexpr int H1 = Cfg::BM, H2 = 16 * TM2 * WM;
        const int n_tall = (p.M - 32 * H2) / (H1 - H2);
        p.n_small = n_tall;
        p.m_split = n_tall * H1;
        p.n_big = 32 * p.tiles_n;
        grid = 256;
    } else {
        p.n_big = ((p.M + Cfg::BM - 1) / Cfg::BM) * p.tiles_n;
        grid = p.n_big < 256 ? p.n_big : 256;
    }
    hipLaunchKernelGGL((gemm_v4_kernel<WM, TM, TM2, TN>), dim3(grid), dim3(Cfg::THREADS), lds_bytes, st, p);
    VB_LAUNCH_CHECK();
    return 0;
}

// cfg = WM * 1000 + TM * 100 + TM2 * 10 + TN (plan_v4 in gemm.hip)
int dispatch_v4(hipStream_t st, const GemmP& p, int cfg) {
    switch (cfg) {
        case 6303: return launch_v4<6, 3, 0, 3>(st, p);
        case 6304: return launch_v4<6, 3, 0, 4>(st, p);
        case 4544: return launch_v4<4, 5, 4, 4>(st, p);     // 320 | 256 x 128 mixed: M = 9472
        case 4543: return launch_v4<4, 5, 4, 3>(st, p);
        case 6204: return launch_v4<6, 2, 0, 4>(st, p);     // 192 x 128
        case 6103: return launch_v4<6, 1, 0, 3>(st, p);     // 96 x 96
        case 6104: return launch_v4<6, 1, 0, 4>(st, p);     // 96 x 128
        case 4202: return launch_v4<4, 2, 0, 2>(st, p);     // 128 x 64
        case 4104: return launch_v4<4, 1, 0, 4>(st, p);     // 64 x 128
        default: return VB_E_BADARG;
    }
}
#endif

#if VB_V2_LAYOUT == 2
// persistent weight-gradient kernel (gemm_v4w.h): 384 x 96 tiles x K splits as work units
template <int WM, int TM, int TN>
__global__ __launch_bounds__((V4WCfg<WM, TM, TN>::THREADS), (WM == 6 ? 4 : 3)) void gemm_v4w_kernel(const GemmP p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    gemm_block_v4w<WM, TM, TN>(p, smem);
}
#endif

int dispatch(hipStream_t st, const GemmP& p, int tm1, int tm2, int tn, int tiles, int splits) {
    const int code = tm1 * 100 + tm2 * 10 + tn;
    switch (code) {
        case 222: return launch<2, 2, 2>(st, p, tiles, splits);
        case 333: return launch<3, 3, 3>(st, p, tiles, splits);
        case 334: return launch<3, 3, 4>(st, p, tiles, splits);
        case 443: return launch<4, 4, 3>(st, p, tiles, splits);
        case 444: return launch<4, 4, 4>(st, p, tiles, splits);
        case 434: return launch<4, 3, 4>(st, p, tiles, splits);
        case 433: return launch<4, 3, 3>(st, p, tiles, splits);
        case 324: return launch<3, 2, 4>(st, p, tiles, splits);
        case 323: return launch<3, 2, 3>(st, p, tiles, splits);
        default: return VB_E_RANGE;
    }
}

}  // namespace

namespace vbgemm {
#if VB_V2_LAYOUT == 2
template <int WM, int TM, int TN>
int launch_v4w(hipStream_t st, const GemmP& p) {
    using Cfg = V4WCfg<WM, TM, TN>;
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_v4w_kernel<WM, TM, TN>),
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS_BYTES);
    if (attr != hipSuccess) return (int)attr;
    const int grid = p.n_big < 256 ? p.n_big : 256;
    hipLaunchKernelGGL((gemm_v4w_kernel<WM, TM, TN>), dim3(grid), dim3(Cfg::THREADS), Cfg::LDS_BYTES, st, p);
    VB_LAUNCH_CHECK();
    return 0;
}

// cfg: 0 = 384 x 96 (12 MFMA waves), 1 = 256 x 128, 2 = 256 x 96 (8 MFMA waves)
int launch_gemm_v4_tn(hipStream_t st, const GemmP& p, int cfg) {
    return cfg == 0 ? launch_v4w<6, 4, 3>(st, p) : cfg == 1 ? launch_v4w<4, 4, 4>(st, p) : launch_v4w<4, 4, 3>(st, p);
}
#endif
#if VB_V2_LAYOUT == 0
int launch_gemm_v4_nt(hipStream_t st, const GemmP& p, int cfg) { return dispatch_v4(st, p, cfg); }
#elif VB_V2_LAYOUT == 1
int launch_gemm_v4_nn(hipStream_t st, const GemmP& p, int cfg) { return dispatch_v4(st, p, cfg); }
#endif
#if VB_V2_LAYOUT == 0
int launch_gemm_v2_nt(hipStream_t st, const GemmP& p, int tm1, int tm2, int tn, int tiles, int splits) { return dispatch(st, p, tm1, tm2, tn, tiles, splits); }
#elif VB_V2_LAYOUT == 1
int launch_gemm_v2_nn(hipStream_t st, const GemmP& p, int tm1, int tm2, int tn, int tiles, int splits) { return dispatch(st, p, tm1, tm2, tn, tiles, splits); }
#else
int launch_gemm_v2_tn(hipStream_t st, const GemmP& p, int tm1, int tm2, int tn, int tiles, int splits) { return dispatch(st, p, tm1, tm2, tn, tiles, splits); }
#endif
}  // namespace vbgemm
