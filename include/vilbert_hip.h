/*
 * vilbert_hip.h - C ABI of libvilbert_hip.so, the MI355X (gfx950) native layer under the
 * ViLBERT two-stream encoder.
 *
 * The reference (facebookresearch/vilbert-multi-task) has no FFI / plugin interface: its hot
 * path is a Python nn.Module tree (vilbert/vilbert.py) that calls torch ops. The drop-in boundary
 * is therefore the Python class API (vilbert.vilbert.BertConfig / BertModel /
 * BertForMultiModalPreTraining / VILBertForVLTasks); this header is the native layer *below* it.
 * Each entry point replaces the group of torch calls cited next to it (file:line into
 * /root/reference/vilbert/vilbert.py).
 *
 * Conventions
 *  - every pointer is a DEVICE pointer to fp32 (ids / masks: int64) unless stated; row-major;
 *    leading dimensions (ld*) and sizes are in ELEMENTS;
 *  - `stream` is a hipStream_t passed as void*; every call only enqueues work on it: no
 *    allocation, no synchronisation, graph-capture safe;
 *  - return value: 0 = ok, >0 = hipError_t from the launch, <0 = VB_E_* argument error.
 *    Nothing throws across the ABI. vb_error_string() names a code;
 *  - all arithmetic is fp32 (v_mfma_f32_32x32x2_f32 / v_mfma_f32_16x16x4_f32 contractions,
 *    fp32 accumulate): results agree with the fp32 reference to rounding (parity bar 1e-4).
 */
#ifndef VILBERT_HIP_H
#define VILBERT_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VB_ABI_VERSION 1

/* argument errors (negative) */
#define VB_E_BADARG   (-1)  /* null pointer / non-positive size */
#define VB_E_ALIGN    (-2)  /* pointer or leading dimension not usable by the kernel */
#define VB_E_RANGE    (-3)  /* size outside the compiled range (e.g. keys > VB_MAX_KEYS) */
#define VB_E_SEGMENT  (-4)  /* bad weight-segment description */

/* epilogue activations of vb_linear_fwd */
#define VB_ACT_NONE 0
#define VB_ACT_GELU 1  /* x*0.5*(1+erf(x/sqrt(2))) - vilbert.py:111-117 */
#define VB_ACT_RELU 2  /* poolers, vilbert.py:1114,1129 */

#define VB_MAX_SEGMENTS 4
#define VB_MAX_KEYS 320     /* longest key sequence one attention launch handles */
#define VB_MAX_LN_COLS 8192

int vb_abi_version(void);
const char* vb_error_string(int code);

/* ------------------------------------------------------------------------------------------
 * vb_linear_fwd:  C[M, nseg*seg_n] = act( A[M,K] . W^T + bias ) (+ residual)
 *
 * Replaces nn.Linear (+ gelu / relu, + residual add) - vilbert.py:425-427,471,501,514,573-575,
 * 630,662,675,749-751,760-762,846,849,1116-1122,1131-1137,1416-1417.
 * W is given as `nseg` row blocks ("segments") of seg_n rows each, W[s] = [seg_n, K] row-major
 * with leading dimension ldw, so that q/k/v projections that share an input run as ONE launch
 * writing C = [q | k | v] without packing the reference's separate nn.Parameters.
 * bias[s] (seg_n floats) may be NULL. residual (ldr) may be NULL; when given it is added AFTER
 * the activation (the `dense(x) + input_tensor` of BertSelfOutput/BertOutput).
 * preact (ldp) may be NULL; when given the pre-activation (A.W^T + bias) is also stored
 * (saved for backward). N = nseg*seg_n; when nseg > 1, seg_n must be a multiple of 128.
 * ------------------------------------------------------------------------------------------ */
typedef struct {
    int32_t M, K;
    int32_t nseg, seg_n;
    const float* A;            int64_t lda;
    const float* W[VB_MAX_SEGMENTS]; int64_t ldw;
    const float* bias[VB_MAX_SEGMENTS];
    float* C;                  int64_t ldc;
    const float* residual;     int64_t ldr;
    float* preact;             int64_t ldp;
    int32_t act;
} vb_linear_args;

int vb_linear_fwd(void* stream, const vb_linear_args* a);

/* ------------------------------------------------------------------------------------------
 * vb_layernorm_fwd:  y = gamma * (x - mean) / sqrt(var + eps) + beta   per row of n_cols
 *
 * Replaces BertLayerNorm (TF style: biased variance, eps inside the sqrt) - vilbert.py:297-317.
 * `x2` (same shape, may be NULL) is added to x first. mean / rstd (rows floats each, may be NULL)
 * are saved for backward.
 * ------------------------------------------------------------------------------------------ */
int vb_layernorm_fwd(void* stream, int64_t rows, int32_t n_cols, const float* x, const float* x2,
                     const float* gamma, const float* beta, float eps, float* y,
                     float* mean, float* rstd);

/* ------------------------------------------------------------------------------------------
 * vb_text_embed_ln_fwd: LayerNorm(word[ids] + pos[arange + pos_offset] + type[segment_ids])
 *
 * Replaces BertEmbeddings.forward - vilbert.py:346-367 (pos_offset = 2 restates
 * RobertaEmbeddings :379-393). ids / seg: int64 [batch, n_tok]. When task_ids != NULL
 * (config.task_specific_tokens, :358-362) the row task_emb[task_ids[b]] is inserted at output
 * position 1 (it receives no position / type embedding) and the output has n_tok + 1 rows per
 * sample. out: [batch, n_tok (+1), hidden].
 * ------------------------------------------------------------------------------------------ */
int vb_text_embed_ln_fwd(void* stream, int32_t batch, int32_t n_tok, int32_t hidden,
                         const int64_t* ids, const int64_t* seg, int32_t pos_offset,
                         const float* word_emb, const float* pos_emb, const float* type_emb,
                         const int64_t* task_ids, const float* task_emb,
                         const float* gamma, const float* beta, float eps, float* out,
                         float* mean, float* rstd);

/* ------------------------------------------------------------------------------------------
 * vb_image_embed_ln_fwd: LayerNorm(feat_proj + loc . Wloc^T + bloc)
 *
 * Second half of BertImageEmbeddings.forward - vilbert.py:1421-1432: feat_proj [rows, hidden] is
 * the 2048->hidden projection (already holding its bias; produced by vb_linear_fwd), loc is
 * [rows, 5], Wloc is [hidden, 5]. The pre-norm sum is stored to `presum` when non-NULL.
 * ------------------------------------------------------------------------------------------ */
int vb_image_embed_ln_fwd(void* stream, int64_t rows, int32_t hidden, const float* feat_proj,
                          const float* loc, const float* w_loc, const float* b_loc,
                          const float* gamma, const float* beta, float eps, float* out,
                          float* mean, float* rstd);

/* ------------------------------------------------------------------------------------------
 * vb_additive_mask: out[i] = (1 - mask[i]) * -10000      (int64 or fp32 -> fp32)
 * Replaces the extended-mask arithmetic of BertModel.forward - vilbert.py:1341-1362.
 * mask points to int64 values when mask_is_f32 == 0, to fp32 values otherwise.
 * ------------------------------------------------------------------------------------------ */
int vb_additive_mask(void* stream, int64_t n, const void* mask, int32_t mask_is_f32, float* out);

/* ------------------------------------------------------------------------------------------
 * vb_attention_fwd: O = softmax(Q K^T * scale + mask) V   per (sample, head), heads merged
 *
 * Replaces the score / softmax / context block of BertSelfAttention (vilbert.py:429-449),
 * BertImageSelfAttention (:588-608) and each direction of BertBiAttention (:768-809) including
 * transpose_for_scores (:416-422) and the head merge: Q/K/V are token-major [batch*S, ld] views
 * (typically slices of a fused [q|k|v] projection, hence the separate ld*), head h occupies
 * columns [h*head_dim, (h+1)*head_dim). mask_add: fp32 [kv_batch, n_k] additive mask (may be
 * NULL). q_batch / kv_batch: number of samples behind Q and behind K/V/mask - either equal to
 * `batch` or 1 (broadcast; the 1-caption x N-images case of eval_retrieval, :1042-1053).
 * probs (may be NULL): [batch, heads, n_q, n_k] softmax output (`visualization` /
 * output_all_attention_masks, :451-458). head_dim in {64, 128} (32 also compiled for unit
 * tests); n_k <= VB_MAX_KEYS.
 * ------------------------------------------------------------------------------------------ */
typedef struct {
    int32_t batch, heads, head_dim, n_q, n_k;
    int32_t q_batch, kv_batch;
    const float* Q; int64_t ldq;
    const float* K; int64_t ldk;
    const float* V; int64_t ldv;
    const float* mask_add;
    float* O;       int64_t ldo;
    float* probs;
    float scale;
} vb_attention_args;

int vb_attention_fwd(void* stream, const vb_attention_args* a);

#ifdef __cplusplus
}
#endif
#endif /* VILBERT_HIP_H */
