// Device-side finishing of a Conceptual-Captions pre-training batch.
// The reference does this on the host in numpy for every step (vilbert/datasets/concept_cap_dataset.py:
// 241-282: global mean-region row, [0,0,1,1,1] box, mask column) and with a handful of torch ops on the
// GPU (train_concap.py:535-540: objective-1 label masking). Here the raw worker output is copied to the
// device as it is and ONE pass writes the model's input tensors: the [B, R, 2048] feature block is read
// once and written once into rows 1..R of [B, R+1, 2048] while its column sums become row 0.
// HBM-bound: 8 (R + 1/2) F bytes per sample.
#include "common.h"

namespace {

// One thread owns 4 feature columns of one sample: out[b, r + 1, c] = in[b, r, c], out[b, 0, c] = mean.
// The sum runs over r = 0 .. R-1 in fp32 in that order (numpy's np.sum(x, axis=1) for a C-contiguous
// [B, R, F] array adds the R rows in order) and the division is done in fp64 and rounded to fp32 exactly
// like `np.sum(..) / sum_count` (float32 / int64 -> float64) followed by the float32 cast (:251-256).
__global__ __launch_bounds__(256) void concap_feat_kernel(int R, int F, const float* __restrict__ in,
                                                          const int64_t* __restrict__ masked_label,
                                                          float* __restrict__ out) {
    const int b = blockIdx.y;
    const int c = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (c >= F) return;
    int cnt = 0;
    for (int r = 0; r < R; ++r) cnt += masked_label[(long)b * R + r] == 0 ? 1 : 0;
    if (cnt == 0) cnt = 1;
    const float* src = in + (long)b * R * F + c;
    float* dst = out + (long)b * (R + 1) * F + c;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
    for (int r = 0; r < R; ++r) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(src + (long)r * F);
        *reinterpret_cast<f32x4*>(dst + (long)(r + 1) * F) = v;
        s[0] = __fadd_rn(s[0], v[0]);
        s[1] = __fadd_rn(s[1], v[1]);
        s[2] = __fadd_rn(s[2], v[2]);
        s[3] = __fadd_rn(s[3], v[3]);
    }
    f32x4 g;
#pragma unroll
    for (int e = 0; e < 4; ++e) g[e] = (float)((double)s[e] / (double)cnt);
    *reinterpret_cast<f32x4*>(dst) = g;
}

// One block per sample: boxes, region mask, and the objective-1 label edit
//   label = label * (is_next == 0);  label[label == 0] = -1        (train_concap.py:535-540)
__global__ __launch_bounds__(256) void concap_meta_kernel(int R, int T, int objective, const float* __restrict__ loc,
                                                          const int64_t* __restrict__ mask,
                                                          const int64_t* __restrict__ is_next,
                                                          const int64_t* __restrict__ image_label,
                                                          const int64_t* __restrict__ lm_label,
                                                          float* __restrict__ out_loc, int64_t* __restrict__ out_mask,
                                                          int64_t* __restrict__ out_image_label,
                                                          int64_t* __restrict__ out_lm_label) {
    const long b = blockIdx.x;
    const int tid = threadIdx.x;
    for (int i = tid; i < (R + 1) * 5; i += 256) {
        const int r = i / 5, j = i % 5;
        out_loc[b * (R + 1) * 5 + i] = r == 0 ? (j < 2 ? 0.f : 1.f) : loc[b * R * 5 + (r - 1) * 5 + j];
    }
    for (int r = tid; r <= R; r += 256) out_mask[b * (R + 1) + r] = r == 0 ? 1 : mask[b * R + r - 1];
    const int64_t keep = (objective == 1 && is_next[b] != 0) ? 0 : 1;
    for (int r = tid; r < R; r += 256) {
        int64_t v = image_label[b * R + r];
        if (objective == 1) { v *= keep; if (v == 0) v = -1; }
        out_image_label[b * R + r] = v;
    }
    for (int t = tid; t < T; t += 256) {
        int64_t v = lm_label[b * T + t];
        if (objective == 1) { v *= keep; if (v == 0) v = -1; }
        out_lm_label[b * T + t] = v;
    }
}

}  // namespace

extern "C" int vb_concap_finish_batch(void* stream, const vb_concap_batch* a) {
    if (a == nullptr || a->batch <= 0 || a->regions <= 0 || a->tokens <= 0 || a->feat_dim <= 0) return VB_E_BADARG;
    if (a->feat_dim % 4 != 0) return VB_E_ALIGN;
    if (!a->image_feat || !a->image_loc || !a->image_mask || !a->masked_label || !a->is_next || !a->image_label ||
        !a->lm_label_ids || !a->out_image_feat || !a->out_image_loc || !a->out_image_mask || !a->out_image_label ||
        !a->out_lm_label_ids)
        return VB_E_BADARG;
    if (!vb_aligned16(a->image_feat) || !vb_aligned16(a->out_image_feat)) return VB_E_ALIGN;
    hipStream_t st = (hipStream_t)stream;
    dim3 grid((unsigned)((a->feat_dim / 4 + 255) / 256), (unsigned)a->batch);
    hipLaunchKernelGGL(concap_feat_kernel, grid, dim3(256), 0, st, a->regions, a->feat_dim, a->image_feat,
                       a->masked_label, a->out_image_feat);
    VB_LAUNCH_CHECK();
    hipLaunchKernelGGL(concap_meta_kernel, dim3((unsigned)a->batch), dim3(256), 0, st, a->regions, a->tokens,
                       a->objective, a->image_loc, a->image_mask, a->is_next, a->image_label, a->lm_label_ids,
                       a->out_image_loc, a->out_image_mask, a->out_image_label, a->out_lm_label_ids);
    VB_LAUNCH_CHECK();
    return 0;
}
