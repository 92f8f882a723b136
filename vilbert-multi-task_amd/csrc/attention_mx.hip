// Short-sequence attention of the MX e4m3 forward path (BASELINE configs[4] "fp8 MFMA co-attention path", round 4):
//   ctx = softmax(Q K^T * scale + mask) V   per (sample, head)   - reference vilbert.py:429-449, 588-608, 768-809 in eval mode
// with bf16 operands (the q | k | v projection leaves its MX GEMM as bf16, vb_linear_fwd_mx Cb), bf16 MFMA for both
// contractions, fp32 scores / softmax / accumulators, and the context written STRAIGHT as MX codes + scale words for the
// output projection that consumes it (csrc/mx8.hip) - no fp32 q | k | v, no fp32 context, no quantiser pass.
//
// One wave owns one (sample, head); sequences of at most 48 queries and 48 keys (three 16-row tiles; the pre-training and
// task shapes of the north-star configs: 36 / 37 tokens and regions). Longer sequences / probabilities wanted / dropout
// stay on the fp32 kernels of attention.hip.
//   scores   S^T = K Q^T on v_mfma_f32_16x16x32_bf16 (A = key rows, B = query rows, both read straight from HBM as 16-byte
//            pieces with the SAME (lane, slot) -> head-dim map, so the instruction's internal K order is irrelevant):
//            lane (c, g) holds S[q = 16 qt + c][key = 16 kt + 4 g + r] - a fixed query per lane: the softmax reductions
//            are in-register plus two cross-lane steps (xor 16, xor 32), as in attention.hip;
//   context  O^T = V^T P^T on v_mfma_f32_16x16x16_bf16: B = P^T is exactly the lane's own probabilities (slot j of lane
//            group g <-> key 16 kt + 4 g + j), A = V^T tile = four 2-byte LDS reads of the wave's V block (row-major, rows
//            padded by 16 bytes) - again the same (lane group, slot) -> key map on both sides; the result leaves lane (c, g)
//            with O[q = 16 qt + c][d = 16 dt + 4 g + r]: a fixed query per lane, 4 consecutive head dims per tile;
//   MX out   a 32-column block of a context row = two d tiles = 8 in-lane values x 4 lane groups: amax with two xor steps,
//            codes packed 4 per lane and tile, scale bytes assembled by lane group 0 (head_dim 128: one whole word per
//            head; head_dim 64: half a word).
#include "common.h"
#include "mx8.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef int v4i __attribute__((ext_vector_type(4)));

struct AttnMxP {
    int batch, heads, n_q, n_k, n_qt, n_kt;
    long q_bstride, kv_bstride;          // rows per sample (0 = broadcast)
    const unsigned short* Q; long ldq;   // bf16 bits, row strides in elements
    const unsigned short* K; long ldk;
    const unsigned short* V; long ldv;
    const float* mask;                   // additive, [kv batch][n_k], may be null
    float scale;
    unsigned char* Oq; long ldo;         // codes [batch * n_q][heads * D]
    unsigned* os; long os_rows;          // scale words [heads * D / 128][os_rows]
    int v_rows;                          // V rows each wave stages in LDS (= n_k: tile rows past it are read as row n_k - 1)
};

__device__ __forceinline__ unsigned short bf16_bits(float v) {
    const unsigned u = __float_as_uint(v);
    return (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}

constexpr int AMX_ROWS = 48;           // longest query / key sequence (three 16-row tiles)

template <int D>
// four waves per SIMD (<= 128 registers): 16 waves x 256 CUs = the 4096 (sample, head) pairs of the image stream at batch 512 in one
// round (head_dim 64 compiles to 108 registers without a bound; five or six waves per SIMD would spill)
__global__ __launch_bounds__(256, 4) void attn_mx_kernel(const AttnMxP p) {
    constexpr int LDV = D + 8;                       // LDS row stride in bf16 elements (+ 16 bytes)
    extern __shared__ __attribute__((aligned(16))) unsigned short smem_v[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long item = (long)blockIdx.x * 4 + wave;
    if (item >= (long)p.batch * p.heads) return;
    const int b = (int)(item / p.heads), h = (int)(item % p.heads);
    const int c = lane & 15, g = lane >> 4;
    unsigned short* __restrict__ vs = smem_v + wave * (p.v_rows * LDV);
    const unsigned short* __restrict__ Qg = p.Q + (long)b * p.q_bstride * p.ldq + h * D;
    const unsigned short* __restrict__ Kg = p.K + (long)b * p.kv_bstride * p.ldk + h * D;
    const unsigned short* __restrict__ Vg = p.V + (long)b * p.kv_bstride * p.ldv + h * D;
    const int n_qt = p.n_qt, n_kt = p.n_kt;

    // ---- stage V: exactly n_k rows (this wave's region). Tile rows past n_k are READ as row n_k - 1 below: they only ever
    // meet a probability of exactly 0. Staging n_k instead of 48 rows keeps the block at 4 x n_k x (D + 8) x 2 bytes - four
    // blocks per CU at head_dim 128 and n_k <= 37 (16 waves x 256 CUs = the 4096 (sample, head) pairs of the image stream at
    // batch 512 in ONE round instead of 1.33).
    {
        constexpr int CH = D / 8;                    // 16-byte chunks per row
        const int total = p.n_k * CH;
        for (int idx = lane; idx < total; idx += 64) {
            const int row = idx / CH, ch = idx % CH;
            const v4i val = *reinterpret_cast<const v4i*>(Vg + (long)row * p.ldv + ch * 8);
            *reinterpret_cast<v4i*>(vs + row * LDV + ch * 8) = val;
        }
    }

    // ---- scores: s[kt][qt][r] = S[q = 16 qt + c][key = 16 kt + 4 g + r]
    f32x4 s[3][3];
#pragma unroll
    for (int kt = 0; kt < 3; ++kt)
#pragma unroll
        for (int qt = 0; qt < 3; ++qt) s[kt][qt] = f32x4{0.f, 0.f, 0.f, 0.f};
    const unsigned short* krow[3];
    const unsigned short* qrow[3];
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        krow[t] = Kg + (long)min(16 * t + c, p.n_k - 1) * p.ldk + 8 * g;
        qrow[t] = Qg + (long)min(16 * t + c, p.n_q - 1) * p.ldq + 8 * g;
    }
#pragma unroll
    for (int st = 0; st < D / 32; ++st) {
        bf16x8 kf[3], qf[3];
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            kf[t] = t < n_kt ? *reinterpret_cast<const bf16x8*>(krow[t] + 32 * st) : bf16x8{};
            qf[t] = t < n_qt ? *reinterpret_cast<const bf16x8*>(qrow[t] + 32 * st) : bf16x8{};
        }
#pragma unroll
        for (int kt = 0; kt < 3; ++kt)
#pragma unroll
            for (int qt = 0; qt < 3; ++qt)
                if (kt < n_kt && qt < n_qt) s[kt][qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[kt], qf[qt], s[kt][qt], 0, 0, 0);
    }

    // ---- softmax over the keys of each of the lane's (up to three) queries
    float mk[3][4];
    const float* __restrict__ mrow = p.mask != nullptr ? p.mask + (long)(p.kv_bstride ? b : 0) * p.n_k : nullptr;
#pragma unroll
    for (int kt = 0; kt < 3; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int key = 16 * kt + 4 * g + r;
            mk[kt][r] = key < p.n_k ? (mrow != nullptr ? mrow[key] : 0.f) : -INFINITY;
        }
    s16x4 pb[3][3];      // [kt][qt]: the lane's probabilities as the B operand of the context product
#pragma unroll
    for (int qt = 0; qt < 3; ++qt) {
        float mx = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < 3; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                s[kt][qt][r] = fmaf(s[kt][qt][r], p.scale, mk[kt][r]);
                mx = fmaxf(mx, s[kt][qt][r]);
            }
        mx = fmaxf(mx, __shfl_xor(mx, 16));
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        float sum = 0.f;
#pragma unroll
        for (int kt = 0; kt < 3; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                s[kt][qt][r] = __expf(s[kt][qt][r] - mx);       // exp(-inf) = 0 for the padded keys
                sum += s[kt][qt][r];
            }
        sum += __shfl_xor(sum, 16);
        sum += __shfl_xor(sum, 32);
        const float inv = 1.0f / sum;
#pragma unroll
        for (int kt = 0; kt < 3; ++kt) {
            s16x4 v;
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = (short)bf16_bits(s[kt][qt][r] * inv);
            pb[kt][qt] = v;
        }
    }

    // ---- context, 64 head dims (4 d tiles = two 32-column MX blocks) at a time: o[dl][qt][r] = O[q = 16 qt + c][d = 64 half +
    // 16 dl + 4 g + r]. Half the accumulators of a whole head at head_dim 128 (48 instead of 96 registers): the kernel fits
    // four waves per SIMD, the MX blocks of a half are complete in it.
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // this wave's V rows are in LDS (wave-private region)
    unsigned word[3] = {0u, 0u, 0u};     // per query tile: the scale bytes of the head's blocks
#pragma unroll
    for (int half = 0; half < D / 64; ++half) {
        f32x4 o[4][3];
#pragma unroll
        for (int dl = 0; dl < 4; ++dl)
#pragma unroll
            for (int qt = 0; qt < 3; ++qt) o[dl][qt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kt = 0; kt < 3; ++kt) {
            if (kt < n_kt) {
                int voff[4];   // LDS offsets of the lane group's four key rows (clamped to the last staged row)
#pragma unroll
                for (int j = 0; j < 4; ++j) voff[j] = min(16 * kt + 4 * g + j, p.n_k - 1) * LDV + c + 64 * half;
#pragma unroll
                for (int dl = 0; dl < 4; ++dl) {
                    s16x4 va;   // A operand: row d = 64 half + 16 dl + c, slot j <-> key 16 kt + 4 g + j
#pragma unroll
                    for (int j = 0; j < 4; ++j) va[j] = (short)vs[voff[j] + 16 * dl];
#pragma unroll
                    for (int qt = 0; qt < 3; ++qt)
                        if (qt < n_qt) o[dl][qt] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(va, pb[kt][qt], o[dl][qt], 0, 0, 0);
                }
            }
        }
        // MX codes of this half's two blocks
#pragma unroll
        for (int qt = 0; qt < 3; ++qt) {
            if (qt < n_qt) {
                const int q = 16 * qt + c;
                const bool live = q < p.n_q;
                const long row = (long)b * p.n_q + (live ? q : 0);
#pragma unroll
                for (int bl = 0; bl < 2; ++bl) {
                    const int blk = 2 * half + bl;
                    float amax = fmaxf(amax4(o[2 * bl][qt]), amax4(o[2 * bl + 1][qt]));
                    amax = fmaxf(amax, __shfl_xor(amax, 16));
                    amax = fmaxf(amax, __shfl_xor(amax, 32));
                    const unsigned byte = mx_scale_byte(amax);
                    const float inv = mx_inv_scale(byte);
                    word[qt] |= byte << (8 * blk);
                    if (live) {
                        unsigned char* __restrict__ dst = p.Oq + row * p.ldo + h * D + 32 * blk + 4 * g;
                        *reinterpret_cast<unsigned*>(dst) = mx_pack4(o[2 * bl][qt], inv);
                        *reinterpret_cast<unsigned*>(dst + 16) = mx_pack4(o[2 * bl + 1][qt], inv);
                    }
                }
            }
        }
    }
    // scale words of the context rows
#pragma unroll
    for (int qt = 0; qt < 3; ++qt) {
        if (qt < n_qt) {
            const int q = 16 * qt + c;
            if (q < p.n_q && g == 0) {
                const long row = (long)b * p.n_q + q;
                if (D == 128) p.os[(long)h * p.os_rows + row] = word[qt];
                else reinterpret_cast<unsigned short*>(p.os + (long)(h >> 1) * p.os_rows + row)[h & 1] = (unsigned short)word[qt];
            }
        }
    }
}

}  // namespace

extern "C" int vb_attention_fwd_mx(void* stream, const vb_attention_mx_args* a) {
    if (a == nullptr || a->Q == nullptr || a->K == nullptr || a->V == nullptr || a->Oq == nullptr || a->o_scales == nullptr)
        return VB_E_BADARG;
    if (a->batch <= 0 || a->heads <= 0 || a->n_q <= 0 || a->n_k <= 0) return VB_E_BADARG;
    if (a->n_q > AMX_ROWS || a->n_k > AMX_ROWS) return VB_E_RANGE;
    if (a->head_dim != 64 && a->head_dim != 128) return VB_E_RANGE;
    if ((a->q_batch != a->batch && a->q_batch != 1) || (a->kv_batch != a->batch && a->kv_batch != 1)) return VB_E_BADARG;
    const long H = (long)a->heads * a->head_dim;
    if (H % 128 != 0 || a->ldo < H || a->ldo % 4 != 0 || a->o_srows < (long)a->batch * a->n_q) return VB_E_RANGE;
    if ((a->ldq | a->ldk | a->ldv) % 8 != 0 || !vb_aligned16(a->Q) || !vb_aligned16(a->K) || !vb_aligned16(a->V) ||
        (reinterpret_cast<uintptr_t>(a->Oq) & 3u) != 0 || (reinterpret_cast<uintptr_t>(a->o_scales) & 3u) != 0)
        return VB_E_ALIGN;
    AttnMxP p{};
    p.batch = a->batch; p.heads = a->heads; p.n_q = a->n_q; p.n_k = a->n_k;
    p.n_qt = (a->n_q + 15) / 16; p.n_kt = (a->n_k + 15) / 16;
    p.q_bstride = a->q_batch == 1 && a->batch > 1 ? 0 : a->n_q;
    p.kv_bstride = a->kv_batch == 1 && a->batch > 1 ? 0 : a->n_k;
    p.Q = a->Q; p.ldq = a->ldq; p.K = a->K; p.ldk = a->ldk; p.V = a->V; p.ldv = a->ldv;
    p.mask = a->mask_add; p.scale = a->scale;
    p.Oq = a->Oq; p.ldo = a->ldo; p.os = a->o_scales; p.os_rows = a->o_srows;
    const long items = (long)a->batch * a->heads;
    const dim3 grid((unsigned)((items + 3) / 4)), block(256);
    hipStream_t st = static_cast<hipStream_t>(stream);
    p.v_rows = a->n_k;
    if (a->head_dim == 128) {
        const int lds = 4 * p.v_rows * (128 + 8) * 2;
        hipLaunchKernelGGL(attn_mx_kernel<128>, grid, block, lds, st, p);
    } else {
        const int lds = 4 * p.v_rows * (64 + 8) * 2;
        hipLaunchKernelGGL(attn_mx_kernel<64>, grid, block, lds, st, p);
    }
    VB_LAUNCH_CHECK();
    return 0;
}
