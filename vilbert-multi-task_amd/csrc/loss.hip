// Row losses of the pre-training heads, forward and backward:
//  * cross-entropy with an ignore index, mean over the counted rows - nn.CrossEntropyLoss(ignore_index=-1)
//    on the masked-LM logits [rows, 30522] and the alignment logits [B, 2] (vilbert.py:1453,1578-1585);
//  * KL divergence between a target distribution and log_softmax(scores), summed over the rows and divided
//    by a caller-given count - nn.KLDivLoss(reduction="none")(log_softmax(pred), target) (:1454,1516-1522).
// One 256-thread block per row: max, sum-exp and the label / target terms are reduced in registers + LDS;
// the row (<= 122 KB) is re-read from L2, never from HBM. The softmax is never materialised in the
// forward; the backward writes the gradient of the logits directly from the saved log-sum-exp.
// HBM-bound: forward reads 4 n bytes per row (KL: 8 n), backward reads 4 n (8 n) and writes 4 n.
#include "common.h"

namespace {

constexpr int LOSS_THREADS = 256;

__device__ __forceinline__ float block_reduce(float v, bool is_max, float* scratch) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const float o = __shfl_xor(v, off, 64);
        v = is_max ? fmaxf(v, o) : v + o;
    }
    const int wave = threadIdx.x >> 6;
    __syncthreads();                       // scratch may still be read from the previous reduction
    if ((threadIdx.x & 63) == 0) scratch[wave] = v;
    __syncthreads();
    float r = scratch[0];
#pragma unroll
    for (int w = 1; w < LOSS_THREADS / 64; ++w) r = is_max ? fmaxf(r, scratch[w]) : r + scratch[w];
    return r;
}

__device__ __forceinline__ float row_lse(const float* x, int n, float* scratch) {
    float mx = -INFINITY;
    for (int j = threadIdx.x; j < n; j += LOSS_THREADS) mx = fmaxf(mx, x[j]);
    mx = block_reduce(mx, true, scratch);
    float s = 0.f;
    for (int j = threadIdx.x; j < n; j += LOSS_THREADS) s += expf(x[j] - mx);
    s = block_reduce(s, false, scratch);
    return mx + logf(s);
}

__global__ __launch_bounds__(LOSS_THREADS) void xent_fwd_kernel(int n, const float* __restrict__ logits, long ld,
                                                                const int64_t* __restrict__ labels, int64_t ignore,
                                                                float* __restrict__ row_loss, float* __restrict__ lse_out) {
    __shared__ float scratch[LOSS_THREADS / 64];
    const long r = blockIdx.x;
    const int64_t lab = labels[r];
    if (lab == ignore) {                   // block-uniform
        if (threadIdx.x == 0) { row_loss[r] = 0.f; lse_out[r] = 0.f; }
        return;
    }
    const float* x = logits + r * ld;
    const float lse = row_lse(x, n, scratch);
    if (threadIdx.x == 0) {
        lse_out[r] = lse;
        // a label outside [0, n) (the reference's CrossEntropyLoss raises a device assert) poisons the loss
        // with NaN instead of reading past the row
        row_loss[r] = (lab >= 0 && lab < n) ? lse - x[lab] : NAN;
    }
}

// loss = sum(row_loss) / count, count = number of rows with label != ignore (xent) or `divisor` (KL)
__global__ __launch_bounds__(LOSS_THREADS) void loss_mean_kernel(long rows, const float* __restrict__ row_loss,
                                                                 const int64_t* __restrict__ labels, int64_t ignore,
                                                                 float divisor, const float* __restrict__ divisor_dev,
                                                                 float* __restrict__ loss,
                                                                 float* __restrict__ count_out) {
    __shared__ float scratch[LOSS_THREADS / 64];
    float s = 0.f, c = 0.f;
    for (long r = threadIdx.x; r < rows; r += LOSS_THREADS) {
        s += row_loss[r];
        if (labels != nullptr) c += labels[r] != ignore ? 1.f : 0.f;
    }
    s = block_reduce(s, false, scratch);
    c = labels != nullptr ? block_reduce(c, false, scratch) : (divisor_dev != nullptr ? divisor_dev[0] : divisor);
    if (threadIdx.x == 0) {
        loss[0] = s / c;                   // 0 / 0 = NaN when nothing is labelled, like the reference
        count_out[0] = c;
    }
}

__global__ __launch_bounds__(LOSS_THREADS) void xent_bwd_kernel(int n, const float* __restrict__ logits, long ld,
                                                                const int64_t* __restrict__ labels, int64_t ignore,
                                                                const float* __restrict__ lse, const float* __restrict__ gout,
                                                                const float* __restrict__ count, float* __restrict__ dlogits,
                                                                long ldd) {
    const long r = blockIdx.x;
    const int64_t lab = labels[r];
    float* d = dlogits + r * ldd;
    if (lab == ignore) {
        for (int j = threadIdx.x; j < n; j += LOSS_THREADS) d[j] = 0.f;
        return;
    }
    const float* x = logits + r * ld;
    const float l = lse[r], g = gout[0] / count[0];
    for (int j = threadIdx.x; j < n; j += LOSS_THREADS) d[j] = (expf(x[j] - l) - (j == lab ? 1.f : 0.f)) * g;
}

__global__ __launch_bounds__(LOSS_THREADS) void kl_fwd_kernel(int n, const float* __restrict__ scores, long ld,
                                                              const float* __restrict__ target, long ldt,
                                                              float* __restrict__ row_loss, float* __restrict__ lse_out,
                                                              float* __restrict__ tsum_out) {
    __shared__ float scratch[LOSS_THREADS / 64];
    const long r = blockIdx.x;
    const float* x = scores + r * ld;
    const float* t = target + r * ldt;
    const float lse = row_lse(x, n, scratch);
    float acc = 0.f, ts = 0.f;
    for (int j = threadIdx.x; j < n; j += LOSS_THREADS) {
        const float tj = t[j];
        // KLDivLoss pointwise term t (log t - input), 0 where t == 0 (xlogy), input = x - lse
        acc += tj > 0.f ? tj * (logf(tj) - (x[j] - lse)) : -tj * (x[j] - lse);
        ts += tj;
    }
    acc = block_reduce(acc, false, scratch);
    ts = block_reduce(ts, false, scratch);
    if (threadIdx.x == 0) {
        row_loss[r] = acc;
        lse_out[r] = lse;
        tsum_out[r] = ts;
    }
}

__global__ __launch_bounds__(LOSS_THREADS) void kl_bwd_kernel(int n, const float* __restrict__ scores, long ld,
                                                              const float* __restrict__ target, long ldt,
                                                              const float* __restrict__ lse, const float* __restrict__ tsum,
                                                              const float* __restrict__ gout, float divisor,
                                                              const float* __restrict__ divisor_dev,
                                                              float* __restrict__ dscores, long ldd) {
    const long r = blockIdx.x;
    const float* x = scores + r * ld;
    const float* t = target + r * ldt;
    float* d = dscores + r * ldd;
    const float l = lse[r], ts = tsum[r], g = gout[0] / (divisor_dev != nullptr ? divisor_dev[0] : divisor);
    // d/dx_j of -sum_k t_k (x_k - lse) = softmax_j * sum(t) - t_j
    for (int j = threadIdx.x; j < n; j += LOSS_THREADS) d[j] = (expf(x[j] - l) * ts - t[j]) * g;
}

}  // namespace

extern "C" int vb_xent_fwd(void* stream, int64_t rows, int32_t n, const float* logits, int64_t ld,
                           const int64_t* labels, int64_t ignore_index, float* row_loss, float* lse,
                           float* loss, float* count) {
    if (rows < 0 || n <= 0 || ld < n) return VB_E_BADARG;
    if (!logits || !labels || !row_loss || !lse || !loss || !count) return VB_E_BADARG;
    hipStream_t st = (hipStream_t)stream;
    if (rows > 0) {
        hipLaunchKernelGGL(xent_fwd_kernel, dim3((unsigned)rows), dim3(LOSS_THREADS), 0, st, n, logits, ld, labels,
                           ignore_index, row_loss, lse);
        VB_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(loss_mean_kernel, dim3(1), dim3(LOSS_THREADS), 0, st, rows, row_loss, labels, ignore_index, 0.f,
                       (const float*)nullptr, loss, count);
    VB_LAUNCH_CHECK();
    return 0;
}

extern "C" int vb_xent_bwd(void* stream, int64_t rows, int32_t n, const float* logits, int64_t ld,
                           const int64_t* labels, int64_t ignore_index, const float* lse, const float* grad_loss,
                           const float* count, float* dlogits, int64_t ldd) {
    if (rows < 0 || n <= 0 || ld < n || ldd < n) return VB_E_BADARG;
    if (!logits || !labels || !lse || !grad_loss || !count || !dlogits) return VB_E_BADARG;
    if (rows == 0) return 0;
    hipLaunchKernelGGL(xent_bwd_kernel, dim3((unsigned)rows), dim3(LOSS_THREADS), 0, (hipStream_t)stream, n, logits, ld,
                       labels, ignore_index, lse, grad_loss, count, dlogits, ldd);
    VB_LAUNCH_CHECK();
    return 0;
}

extern "C" int vb_kl_fwd(void* stream, int64_t rows, int32_t n, const float* scores, int64_t ld, const float* target,
                         int64_t ldt, float divisor, float* row_loss, float* lse, float* tsum, float* loss,
                         const float* divisor_dev) {
    if (rows < 0 || n <= 0 || ld < n || ldt < n) return VB_E_BADARG;
    if (!scores || !target || !row_loss || !lse || !tsum || !loss) return VB_E_BADARG;
    hipStream_t st = (hipStream_t)stream;
    if (rows > 0) {
        hipLaunchKernelGGL(kl_fwd_kernel, dim3((unsigned)rows), dim3(LOSS_THREADS), 0, st, n, scores, ld, target, ldt,
                           row_loss, lse, tsum);
        VB_LAUNCH_CHECK();
    }
    // loss points to TWO floats: {sum(row_loss) / divisor, divisor}
    hipLaunchKernelGGL(loss_mean_kernel, dim3(1), dim3(LOSS_THREADS), 0, st, rows, row_loss, (const int64_t*)nullptr,
                       (int64_t)0, divisor, divisor_dev, loss, loss + 1);
    VB_LAUNCH_CHECK();
    return 0;
}

extern "C" int vb_kl_bwd(void* stream, int64_t rows, int32_t n, const float* scores, int64_t ld, const float* target,
                         int64_t ldt, const float* lse, const float* tsum, const float* grad_loss, float divisor,
                         float* dscores, int64_t ldd, const float* divisor_dev) {
    if (rows < 0 || n <= 0 || ld < n || ldt < n || ldd < n) return VB_E_BADARG;
    if (!scores || !target || !lse || !tsum || !grad_loss || !dscores) return VB_E_BADARG;
    if (rows == 0) return 0;
    hipLaunchKernelGGL(kl_bwd_kernel, dim3((unsigned)rows), dim3(LOSS_THREADS), 0, (hipStream_t)stream, n, scores, ld,
                       target, ldt, lse, tsum, grad_loss, divisor, divisor_dev, dscores, ldd);
    VB_LAUNCH_CHECK();
    return 0;
}
