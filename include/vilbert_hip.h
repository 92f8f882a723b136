/*
 * vilbert_hip.h - C ABI of libvilbert_hip.so, the MI355X (gfx950) native layer under the
 * ViLBERT two-stream encoder.
 *
 * The reference (facebookresearch/vilbert-multi-task) has no FFI / plugin interface: its hot
 * path is a Python nn.Module tree (vilbert/vilbert.py) that calls torch ops. The drop-in boundary
 * is therefore the Python class API (vilbert.vilbert.BertConfig / BertModel /
 * BertForMultiModalPreTraining / VILBertForVLTasks); this header is the native layer *below* it.
 * Each entry point replaces the group of torch calls cited next to it (file:line into
 * /root/reference/vilbert/vilbert.py); the *_bwd entry points replace what torch autograd runs for
 * the same lines under loss.backward() (train_concap.py:572, train_tasks.py:540).
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

#define VB_ABI_VERSION 18

/* argument errors (negative) */
#define VB_E_BADARG   (-1)  /* null pointer / non-positive size */
#define VB_E_ALIGN    (-2)  /* pointer or leading dimension not usable by the kernel */
#define VB_E_RANGE    (-3)  /* size outside the compiled range (e.g. keys > VB_MAX_KEYS) */
#define VB_E_SEGMENT  (-4)  /* bad weight-segment description */
#define VB_E_WORKSPACE (-5) /* deterministic mode: the registered workspace is too small for this launch */

/* epilogue activations of vb_linear_fwd */
#define VB_ACT_NONE 0
#define VB_ACT_GELU 1  /* x*0.5*(1+erf(x/sqrt(2))) - vilbert.py:111-117 */
#define VB_ACT_RELU 2  /* poolers, vilbert.py:1114,1129 */
#define VB_ACT_SWISH 3 /* x*sigmoid(x) - vilbert.py:120-121 (ACT2FN["swish"]; no shipped config uses it) */

#define VB_MAX_SEGMENTS 4
#define VB_MAX_KEYS 320     /* longest key sequence one attention launch handles */
#define VB_MAX_LN_COLS 8192

int vb_abi_version(void);
const char* vb_error_string(int code);

/* Arithmetic of every vb_linear_* GEMM: 0 = v_mfma_f32_32x32x2_f32 (exact fp32 products, the default),
 * 3 = "bf16x6": each fp32 operand split exactly into three bf16 planes and the six largest partial
 * products accumulated in fp32 on v_mfma_f32_32x32x16_bf16 (fp32-class result, ~1.2x faster),
 * 2 = "bf16x3": two planes, three products (error ~2^-16 per product), 1 = "bf16": operands rounded to bf16
 * (round-to-nearest-even) on their way into LDS, ONE product per element on v_mfma_f32_32x32x16_bf16, fp32
 * accumulate, fp32 tensors in memory - the reduced-precision throughput mode (relative error ~2^-9 per operand:
 * outside the 1e-4 parity bar, reported under its own tolerance). Returns the previous mode; an unknown value only
 * queries. Initial value from the environment variable VB_GEMM_MODE (f32 | bf16x6 | bf16x3 | bf16). */
int vb_set_gemm_mode(int planes);

/* Device-side step counter of the dropout masks. Every dropout mask is keep(seed, element index) with the seed given by
 * the caller; when a counter is registered here each launch mixes *device_counter into its seed inside the kernel.
 * A training step captured once into a HIP graph replays with constant host seeds: the graph contains one
 * vb_bump_counter() node, so every replay draws fresh masks while forward and backward of one step still agree.
 * device_counter: device pointer to one uint64 (NULL unregisters). The pointer must stay valid while registered. */
int vb_set_seed_epoch(const uint64_t* device_counter);
int vb_bump_counter(void* stream, uint64_t* device_counter);   /* *device_counter += 1 on the stream */

/* Tile selection of the fp32 GEMM kernels. Aligned launches (16-byte pointers, K % 16 == 0, N % 4 == 0) run the
 * second-generation kernel (v_mfma_f32_16x16x4_f32, block tile 32 TM x 32 TN from the menu 64x64, 96x96, 96x128,
 * 128x96, 128x128, chosen by a cost model so that the tile count fills whole rounds of the 256 CUs); everything else
 * runs the round-1 kernel (v_mfma_f32_32x32x2_f32, 128x128 + 64x64 tail tiles, any alignment / ragged K).
 * One launch may mix two tile heights (e.g. M = 9472 rows = 8 x 128 + 88 x 96: exactly 3 blocks per CU).
 * code: 0 = cost model (default), 10 TM + TN in {22, 33, 34, 43, 44} = force that tile, 100 TM1 + 10 TM2 + TN in
 * {434, 433, 324, 323} = force that mixed-height pair, -1 = round-1 kernel only.
 * Returns the previous code; an unknown value only queries. Environment: VB_GEMM_TILE=<code>, VB_GEMM_V2=0. */
int vb_set_gemm_tile(int code);

/* Deterministic weight gradients (round 3; per-device workspaces since ABI 13). By default a split-K launch
 * (vb_linear_bwd_weight: the contraction runs over the token rows; vb_linear_bwd_input with a small output and a long
 * contraction, e.g. the MLM decoder) adds the partial products of its splits into the result with fp32 atomics: the
 * value depends in the last bits on the order the blocks finish in, where the reference's cuBLAS path (autograd of the
 * nn.Linear call sites, vilbert.py:425-427, 471, 501, 514) is repeatable. With on = 1 every split stores its partial
 * product into a workspace and a second kernel adds the partials in split order: bit-identical from run to run
 * (measured cost: DESIGN.md). `workspace` (device memory, 16-byte aligned, owned by the caller, must outlive the setting
 * AND every captured graph that was recorded under it) serves the DEVICE IT LIVES ON: call once per device a process
 * drives (nn.DataParallel replicas - the reference's non-distributed multi-GPU path, train_concap.py:513-515); a launch
 * on a device without a workspace uses the atomics. A workspace is cut into 8 equal slices, one per stream of that
 * device that issues split launches (launches on different streams never share memory); a slice has to hold
 * splits x (M x N + M) floats of the largest split launch (8 x 256 MiB is ample for the two-stream models). A launch
 * that finds no free slice (a ninth stream) or whose partials do not fit falls back to the atomics for that launch -
 * never an error - and is counted by vb_deterministic_fallbacks(). Registering the same buffer again keeps the
 * stream -> slice assignment; on = 0 drops every device's registration. The embedding-table gradients
 * (vb_text_embed_bwd) still use atomics. Returns the previous setting (0 / 1) or a negative error. */
int vb_set_deterministic(int on, void* workspace, int64_t workspace_bytes);
/* Split launches since the last vb_set_deterministic(1, ...) that wanted the ordered reduce and ran with atomics. */
int64_t vb_deterministic_fallbacks(void);

/* Persistent one-block-per-CU fp32 GEMM (round 3; forward and dgrad layouts of vb_linear_fwd / vb_linear_bwd_input,
 * replaces the same nn.Linear call sites - vilbert.py:425-427,471,501,514): 12 MFMA waves + 1 LDS-DMA loader wave per
 * compute unit, 288 x 96 / 288 x 128 output tiles, the K tiles of all output tiles of a block streamed through a 4-stage
 * LDS ring. mode: 0 = never, 1 = wherever its tiles fill whole rounds of the 256 compute units (default:
 * >= 90 % of the launched tile slots useful, e.g. M = 9216 or 18432 rows), 2 = every eligible launch. Returns the previous mode; an unknown value only
 * queries. Environment: VB_GEMM_V4=<mode>. */
int vb_set_gemm_v4(int mode);

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
 * (saved for backward). N = nseg*seg_n. dropout_p > 0 applies nn.Dropout to the activated value BEFORE
 * the residual is added (the `dropout(dense(x)) + input_tensor` of :471-473, 514-516, ...): keep mask =
 * f(seed, row * N + col), the same function vb_dropout uses, so vb_dropout(dY, seed) is its backward
 * (C must be contiguous, ldc == N, in that case).
 * act_grad (ldg) may be NULL; when given it receives act'(A.W^T + bias), the derivative of the epilogue
 * activation (gelu: 0.5 (1 + erf(x / sqrt 2)) + x exp(-x^2 / 2) / sqrt(2 pi); relu: x > 0) - saved by the
 * training forward so that the backward of `act` is ONE multiply in the epilogue of vb_linear_bwd_input
 * (its `mul` argument) instead of a separate erf / exp pass over [M, N].
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
    float* act_grad;           int64_t ldg;
    int32_t act;
    float dropout_p;
    uint64_t seed;
} vb_linear_args;

int vb_linear_fwd(void* stream, const vb_linear_args* a);

/* ------------------------------------------------------------------------------------------
 * vb_linear_bwd_input:  dX[M,K] (+)= dY[M, nseg*seg_n] . stack(W) + residual
 * Gradient of nn.Linear w.r.t. its input (autograd of the lines listed for vb_linear_fwd); the
 * stacked segments are contracted in one launch. accumulate != 0 adds into dX.
 * residual (ldr, may be NULL): a gradient of the same shape arriving over a skip connection (the `+ x` of
 * vilbert.py:516 and its twins), added in the epilogue instead of by a separate pass.
 * mul (ldm, may be NULL): [M, K] elementwise multiplier of the result, dX = (dY . stack(W) + residual) * mul -
 * the act_grad saved by the forward of the Linear that PRODUCED this layer's input (gelu backward fused here).
 * ------------------------------------------------------------------------------------------ */
typedef struct {
    int32_t M, K;
    int32_t nseg, seg_n;
    const float* dY;           int64_t ldy;
    const float* W[VB_MAX_SEGMENTS]; int64_t ldw;
    float* dX;                 int64_t ldx;
    int32_t accumulate;
    const float* residual;     int64_t ldr;
    const float* mul;          int64_t ldm;
} vb_linear_bwd_input_args;

int vb_linear_bwd_input(void* stream, const vb_linear_bwd_input_args* a);

/* ------------------------------------------------------------------------------------------
 * vb_linear_bwd_weight:  dW[s][seg_n,K] (+)= dY[:, s*seg_n:(s+1)*seg_n]^T . X[M,K]
 *                        dbias[s][seg_n] (+)= column sums of the same slice of dY
 * Gradient of nn.Linear w.r.t. weight and bias of the nseg stacked segments (dY is [M, nseg*seg_n]
 * with row stride ldy). The contraction over the M rows is split over workgroups and combined with
 * fp32 atomics; the bias gradient is fused into the same launch. dbias[s] may be NULL.
 * accumulate == 0 zero-fills dW / dbias first.
 * ------------------------------------------------------------------------------------------ */
typedef struct {
    int32_t M, K;
    int32_t nseg, seg_n;
    const float* dY;           int64_t ldy;
    const float* X;            int64_t ldx;
    float* dW[VB_MAX_SEGMENTS]; int64_t ldw;
    float* dbias[VB_MAX_SEGMENTS];
    int32_t accumulate;
} vb_linear_bwd_weight_args;

int vb_linear_bwd_weight(void* stream, const vb_linear_bwd_weight_args* a);

/* ---- FP8 forward path (BASELINE.json configs[4]: "fp8 MFMA co-attention path"). Not a reference interface: the
 * reference's reduced-precision switch is apex fp16 (train_tasks.py:168-171 --fp16, vilbert/optimization.py); on
 * MI355X the matching hardware path is OCP e4m3 on the block-scaled MFMA. The numerics are defined by
 * oracle/fp8_oracle.py (row-wise amax scaling, round-to-nearest-even e4m3fn, fp32 accumulation).
 *
 * vb_quantize_rows_fp8: q[r][0..K) = e4m3(x[r][k] * (448 / amax_r)), scale[r] = amax_r / 448 (1 for an all-zero
 * row). K % 4 == 0; ldx in floats, ldq in bytes (% 4 == 0); x 16-byte aligned. Used for activations (one scale per
 * row = per token / region) and weights (one scale per out-feature; segments of a stacked weight are quantised into
 * row ranges of one [N][K] buffer). */
int vb_quantize_rows_fp8(void* stream, int64_t rows, int32_t K, const float* x, int64_t ldx, uint8_t* q, int64_t ldq,
                         float* scale);

/* nn.Linear forward on quantised operands (the fp8 form of vb_linear_fwd, same epilogues):
 * C[M][N] = act((A . W^T)[m][n] * a_scale[m] * w_scale[n] + bias[n]), then dropout, then + residual.
 * A [M][K] and W [N][K] e4m3 bytes (K % 128 == 0, lda / ldw in bytes % 16 == 0, 16-byte aligned). preact / act_grad
 * (may be NULL, at most one of them): receive the pre-activation / act'(pre-activation) as in vb_linear_args. */
typedef struct {
    const uint8_t* A;
    int64_t lda;
    const float* a_scale;
    const uint8_t* W;
    int64_t ldw;
    const float* w_scale;
    const float* bias;       /* [N] or NULL */
    float* C;
    int64_t ldc;
    const float* residual;   /* [M][N] or NULL */
    int64_t ldr;
    float* preact;           /* [M][N] or NULL: receives the pre-activation (exclusive with act_grad) */
    int64_t ldp;
    float* act_grad;
    int64_t ldg;
    int32_t M, N, K;
    int32_t act;             /* VB_ACT_* */
    float dropout_p;
    uint64_t seed;
} vb_linear_fp8_args;

int vb_linear_fwd_fp8(void* stream, const vb_linear_fp8_args* a);

/* ---- MX (OCP microscaling) e4m3 forward path, round 4 (csrc/mx8.hip; numerics: oracle/fp8_oracle.py `mx_*`). Same place
 * in the model as the row-scaled path above (every forward nn.Linear of vilbert.py:424-517, 571-694, 738-900 in the fp8
 * mode), different format: one power-of-two scale per 32 consecutive K elements, applied by the scaled MFMA itself - so
 * a producer that holds 32 consecutive columns of an output row (a GEMM epilogue, LayerNorm) emits the codes the next
 * linear consumes and no fp32 tensor / separate quantiser pass sits between two linears.
 *
 * Format: codes [rows][K] e4m3fn bytes (K % 128 == 0); scales as uint32 words S[K / 128][scale_rows]: word (kt, r) holds
 * the E8M0 bytes of row r's blocks 4 kt .. 4 kt + 3 (byte b = block 4 kt + b; value 2^(byte - 127); the smallest power of
 * two with amax_block / scale <= 448, byte 0 for an all-zero block). scale_rows >= rows is the row stride of a scale plane.
 *
 * vb_quantize_rows_mx: fp32 rows -> that format (weights: once per optimizer step; activations no producer quantises).
 * vb_quantize_rows_mx_bf16: the same for bfloat16 rows (ABI 16: the context of the key-tiled bf16 attention kernel, which
 * serves the rows the MX attention kernel does not - more than 48 queries / keys, e.g. 101 regions; x 8-byte aligned). Bit-
 * identical to vb_quantize_rows_mx on the rows widened to fp32. */
int vb_quantize_rows_mx(void* stream, int64_t rows, int32_t K, const float* x, int64_t ldx, uint8_t* q, int64_t ldq,
                        uint32_t* scales, int64_t scale_rows);
int vb_quantize_rows_mx_bf16(void* stream, int64_t rows, int32_t K, const uint16_t* x, int64_t ldx, uint8_t* q, int64_t ldq,
                             uint32_t* scales, int64_t scale_rows);

/* nn.Linear forward on MX operands: v = act(A W^T + bias) (+ residual), A [M][K] and W [N][K] in the format above,
 * K % 128 == 0, N % 128 == 0; a_srows >= M rounded up to 256, w_srows >= N (both % 4 == 0; the kernel fetches the scale
 * words of a 256-row / 128-row tile as one piece). EXACTLY ONE of three output forms of the values v per launch:
 *   C  fp32 [M][N] (ldc floats);  Cb bf16 [M][N] (round to nearest even, ldb16 elements);
 *   Cq + c_scales: v re-quantised to MX along N (the K of the next linear), ldq bytes % 16 == 0, c_srows >= M.
 * act: VB_ACT_NONE or VB_ACT_GELU. No dropout: this is the inference path of BASELINE configs[4]. Built combinations:
 * C / Cb with or without residual (bf16 residual: without GELU), Cq with GELU or with an fp32 residual; others VB_E_BADARG. */
typedef struct {
    const uint8_t* A;
    int64_t lda;
    const uint32_t* a_scales;
    int64_t a_srows;
    const uint8_t* W;
    int64_t ldw;
    const uint32_t* w_scales;
    int64_t w_srows;
    const float* bias;       /* [N] or NULL */
    const float* residual;   /* fp32 [M][N] or NULL */
    int64_t ldr;
    const uint16_t* residual_bf16; /* bf16 [M][N] or NULL (at most one of the two residuals) */
    int64_t ldr16;
    float* C;                /* or NULL */
    int64_t ldc;
    uint16_t* Cb;            /* or NULL */
    int64_t ldb16;
    uint8_t* Cq;             /* or NULL */
    int64_t ldq;
    uint32_t* c_scales;
    int64_t c_srows;
    int64_t M, N, K;
    int32_t act;
} vb_linear_mx_args;

int vb_linear_fwd_mx(void* stream, const vb_linear_mx_args* a);

/* BertLayerNorm forward of the MX path's bf16 residual stream: x bf16 [rows][n_cols] (the pre-LayerNorm sum a
 * vb_linear_fwd_mx launch wrote through Cb) -> y bf16 (the next residual) AND its MX codes + scale words; statistics in fp32
 * exactly as vb_layernorm_fwd (vilbert.py:313-317) on the bf16 values. n_cols % 128 == 0. The codes are those of the fp32
 * result BEFORE it is rounded to bf16. */
int vb_layernorm_fwd_mx16(void* stream, int64_t rows, int32_t n_cols, const uint16_t* x, const float* gamma, const float* beta,
                          float eps, uint16_t* y, uint8_t* q, int64_t ldq, uint32_t* scales, int64_t scale_rows);

/* Attention of the MX path (csrc/attention_mx.hip): ctx = softmax(Q K^T * scale + mask_add) V per (sample, head) - the
 * eval-mode arithmetic of BertSelfAttention / BertImageSelfAttention / BertBiAttention (vilbert.py:429-449, 588-608,
 * 768-809) - on bf16 operands (Q, K, V: bf16 bit patterns, row strides in ELEMENTS % 8 == 0, 16-byte aligned; typically
 * column slices of the [q | k | v] projection written by vb_linear_fwd_mx's Cb output), bf16 MFMA for both contractions,
 * fp32 softmax, and the context written in the MX format above for the output projection: Oq [batch * n_q][heads *
 * head_dim] codes (ldo bytes), o_scales words [heads * head_dim / 128][o_srows]. n_q, n_k <= 48; head_dim 64 or 128;
 * q_batch / kv_batch = batch or 1 (broadcast); mask_add [kv_batch][n_k] fp32 or NULL. No dropout, no probabilities. */
typedef struct {
    int32_t batch, heads, head_dim, n_q, n_k;
    int32_t q_batch, kv_batch;
    const uint16_t* Q;
    int64_t ldq;
    const uint16_t* K;
    int64_t ldk;
    const uint16_t* V;
    int64_t ldv;
    const float* mask_add;
    float scale;
    uint8_t* Oq;
    int64_t ldo;
    uint32_t* o_scales;
    int64_t o_srows;
} vb_attention_mx_args;

int vb_attention_fwd_mx(void* stream, const vb_attention_mx_args* a);

/* BertLayerNorm forward (vilbert.py:313-317, as vb_layernorm_fwd: y = LN(x (+ x2))) that ALSO emits every output row in
 * the MX format above for the linears consuming it (n_cols % 128 == 0; scale_rows >= rows). Bit-identical to
 * vb_layernorm_fwd followed by vb_quantize_rows_mx on y. */
int vb_layernorm_fwd_mx(void* stream, int64_t rows, int32_t n_cols, const float* x, const float* x2,
                        const float* gamma, const float* beta, float eps, float* y, uint8_t* q, int64_t ldq,
                        uint32_t* scales, int64_t scale_rows);

/* BertLayerNorm forward (vilbert.py:313-317, as vb_layernorm_fwd: y = LN(x (+ x2))) that ALSO emits the e4m3 codes and
 * scale of every output row for the fp8 linears consuming it (inference in fp8 mode): q[r][0..n_cols), row stride ldq
 * bytes (% 4 == 0), qscale[rows]. Bit-identical to vb_layernorm_fwd followed by vb_quantize_rows_fp8 on y - it saves
 * that second pass (one read of the row). */
int vb_layernorm_fwd_fp8(void* stream, int64_t rows, int32_t n_cols, const float* x, const float* x2,
                         const float* gamma, const float* beta, float eps, float* y, uint8_t* q, int64_t ldq,
                         float* qscale);

/* dx = dy * act'(preact) elementwise (n % 4 == 0), act in {VB_ACT_GELU, VB_ACT_RELU}: the backward of
 * the GEMM epilogue activation (gelu vilbert.py:111-117, ReLU :1114,1129). */
int vb_act_bwd(void* stream, int64_t n, int32_t act, const float* dy, const float* preact, float* dx);

/* y = x * keep(seed, i) / (1 - p) (+ residual): nn.Dropout (vilbert.py:365,472,515,631,676,847,850,
 * 1430 ...) optionally fused with the `+ input_tensor` that follows it (:473,516,632,677,852-853).
 * residual may be NULL. The mask is a pure function of (seed, element index), so calling it again on
 * the output gradient with the same seed (and no residual) IS the backward. */
int vb_dropout(void* stream, int64_t n, const float* x, const float* residual, float* y, float p,
               uint64_t seed);

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

/* vb_layernorm_bwd: dx, dgamma, dbeta of the above given dy, the normalised INPUT x (= x + x2 of the
 * forward), and the saved mean / rstd. Deterministic two-stage column reduction through `workspace`
 * (vb_layernorm_bwd_workspace(rows, n_cols) floats). dgamma / dbeta are overwritten. */
int64_t vb_layernorm_bwd_workspace(int64_t rows, int32_t n_cols);
int vb_layernorm_bwd(void* stream, int64_t rows, int32_t n_cols, const float* dy, const float* x,
                     const float* mean, const float* rstd, const float* gamma, float* dx,
                     float* dgamma, float* dbeta, float* workspace);
/* The same with the dropout mask of the layer IN FRONT of the LayerNorm fused in (round 3): y = LayerNorm(dropout(
 * dense(h), p) + x) is the shape of every BertSelfOutput / BertOutput / BertBiOutput (vilbert.py:471-473, 514-516, 850-
 * 855); its backward needs dx (for the skip connection) AND dropout-masked dx (for the dense layer's GEMMs). dx_dropped
 * [rows, n_cols] receives keep(seed, row * n_cols + col) ? dx / (1 - p) : 0 - the mask function of vb_linear_fwd's
 * dropout epilogue / vb_dropout - in the same pass. n_cols <= 4096. */
int vb_layernorm_bwd_drop(void* stream, int64_t rows, int32_t n_cols, const float* dy, const float* x,
                          const float* mean, const float* rstd, const float* gamma, float* dx, float* dgamma,
                          float* dbeta, float* workspace, float* dx_dropped, float dropout_p, uint64_t seed);

/* ------------------------------------------------------------------------------------------
 * vb_text_embed_ln_fwd: LayerNorm(word[ids] + pos[arange + pos_offset] + type[segment_ids])
 *
 * Replaces BertEmbeddings.forward - vilbert.py:346-367. pos_offset is 0 for every reference model:
 * RobertaEmbeddings (:370-393) computes offset ids that its base class then overwrites (:349-352).
 * ids / seg: int64 [batch, n_tok]. When task_ids != NULL (config.task_specific_tokens, :358-362) the
 * row task_emb[task_ids[b]] is inserted at output position 1 (it receives no position / type
 * embedding) and the output has n_tok + 1 rows per sample. out: [batch, n_tok (+1), hidden].
 * presum (may be NULL) receives the pre-LayerNorm sum (saved for backward).
 * vocab / n_types / n_tasks are the row counts of word_emb / type_emb / task_emb (type_vocab_size is 1 in
 * roberta_base_6layer_6connect.json): an id outside its table contributes a zero row instead of reading
 * past the allocation (the reference's nn.Embedding raises a device assert there).
 * ------------------------------------------------------------------------------------------ */
int vb_text_embed_ln_fwd(void* stream, int32_t batch, int32_t n_tok, int32_t hidden, int32_t vocab,
                         int32_t n_types, int32_t n_tasks,
                         const int64_t* ids, const int64_t* seg, int32_t pos_offset,
                         const float* word_emb, const float* pos_emb, const float* type_emb,
                         const int64_t* task_ids, const float* task_emb,
                         const float* gamma, const float* beta, float eps, float* out,
                         float* mean, float* rstd, float* presum);

/* vb_text_embed_bwd: scatter-add (fp32 atomics) of dx [batch, n_tok (+1), hidden] - the gradient of
 * the pre-LayerNorm sum - into the ZERO-FILLED (or accumulating) tables dword / dpos / dtype / dtask.
 * Word row 0 is nn.Embedding's padding_idx (vilbert.py:330-332) and receives nothing; ids outside
 * [0, vocab) / [0, n_types) / [0, n_tasks) are skipped (dtype holds n_types rows - possibly one). */
int vb_text_embed_bwd(void* stream, int32_t batch, int32_t n_tok, int32_t hidden, int32_t vocab,
                      int32_t n_types, int32_t n_tasks, const int64_t* ids, const int64_t* seg,
                      const int64_t* task_ids, const float* dx, float* dword, float* dpos, float* dtype,
                      float* dtask);

/* ------------------------------------------------------------------------------------------
 * vb_image_embed_ln_fwd: LayerNorm(feat_proj + loc . Wloc^T + bloc)
 *
 * Second half of BertImageEmbeddings.forward - vilbert.py:1421-1432: feat_proj [rows, hidden] is
 * the 2048->hidden projection (already holding its bias; produced by vb_linear_fwd), loc is
 * [rows, 5], Wloc is [hidden, 5]. presum (may be NULL) receives the pre-LayerNorm sum.
 * ------------------------------------------------------------------------------------------ */
int vb_image_embed_ln_fwd(void* stream, int64_t rows, int32_t hidden, const float* feat_proj,
                          const float* loc, const float* w_loc, const float* b_loc,
                          const float* gamma, const float* beta, float eps, float* out,
                          float* mean, float* rstd, float* presum);

/* ------------------------------------------------------------------------------------------
 * vb_additive_mask: out[i] = (1 - mask[i]) * -10000      (int64 or fp32 -> fp32)
 * Replaces the extended-mask arithmetic of BertModel.forward - vilbert.py:1341-1362.
 * mask points to int64 values when mask_is_f32 == 0, to fp32 values otherwise.
 * ------------------------------------------------------------------------------------------ */
int vb_additive_mask(void* stream, int64_t n, const void* mask, int32_t mask_is_f32, float* out);

/* ------------------------------------------------------------------------------------------
 * vb_attention_fwd: O = dropout(softmax(Q K^T * scale + mask)) V   per (sample, head), heads merged
 *
 * Replaces the score / softmax / dropout / context block of BertSelfAttention (vilbert.py:429-449),
 * BertImageSelfAttention (:588-608) and each direction of BertBiAttention (:768-809) including
 * transpose_for_scores (:416-422) and the head merge: Q/K/V are token-major [batch*S, ld] views
 * (typically slices of a fused [q|k|v] projection, hence the separate ld*), head h occupies
 * columns [h*head_dim, (h+1)*head_dim). mask_add: fp32 [kv_batch, n_k] additive mask (may be
 * NULL). q_batch / kv_batch: number of samples behind Q and behind K/V/mask - either equal to
 * `batch` or 1 (broadcast; the 1-caption x N-images case of eval_retrieval, :1042-1053).
 * probs (may be NULL): [batch, heads, n_q, n_k] attention probabilities after dropout
 * (`visualization`, :451-458). lse (may be NULL): [batch, heads, n_q] log-sum-exp of the masked
 * scores (saved for backward). dropout_p in [0, 1): keep mask = f(seed, element index of probs).
 * head_dim in {32, 64, 128}; n_k <= VB_MAX_KEYS.
 *
 * vb_attention_bwd: given the same arguments (lse filled by the forward, probs ignored) and dO,
 * writes dQ / dK / dV (each element exactly once - they may be column slices of one fused gradient
 * buffer) and uses dvec [batch, heads, n_q] as scratch. Broadcast batches are not supported.
 *
 * More than VB_MAX_KEYS keys (round 6; stacked retrieval options under autograd, reference vilbert.py:1008-1040): the
 * caller cuts the keys into chunks of <= VB_MAX_KEYS, passes `lse` = the log-sum-exp over ALL keys (the merge of the
 * chunk forwards) and calls the backward per chunk twice: dvec_mode = VB_DVEC_ACCUMULATE adds this chunk's share of
 * D = rowsum(P dP) to `dvec` (zeroed by the caller; nothing else is written), then dvec_mode = VB_DVEC_GIVEN computes
 * this chunk's dK / dV and its share of dQ from the complete D (the caller sums the dQ shares).
 * ------------------------------------------------------------------------------------------ */
#define VB_DVEC_COMPUTE 0
#define VB_DVEC_ACCUMULATE 1
#define VB_DVEC_GIVEN 2
typedef struct {
    int32_t batch, heads, head_dim, n_q, n_k;
    int32_t q_batch, kv_batch;
    const float* Q; int64_t ldq;
    const float* K; int64_t ldk;
    const float* V; int64_t ldv;
    const float* mask_add;
    float* O;       int64_t ldo;
    float* probs;
    float* lse;
    float scale;
    float dropout_p;
    uint64_t seed;
} vb_attention_args;

typedef struct {
    const float* dO; int64_t lddo;
    float* dQ;       int64_t lddq;
    float* dK;       int64_t lddk;
    float* dV;       int64_t lddv;
    float* dvec;
    int32_t dvec_mode;   /* VB_DVEC_* (round 6): 0 = the whole backward of n_k <= VB_MAX_KEYS keys in one call */
} vb_attention_grads;

int vb_attention_fwd(void* stream, const vb_attention_args* a);
int vb_attention_bwd(void* stream, const vb_attention_args* a, const vb_attention_grads* g);

/* ------------------------------------------------------------------------------------------
 * vb_adamw_step: one launch = AdamW update of every tensor listed in `table` (device array).
 *
 * Replaces `optimizer.step()` of pytorch-transformers 1.0.0 AdamW (reference train_concap.py:465-470,583-585
 * with betas (0.9, 0.98); train_tasks.py:426,550 with correct_bias=False): per tensor
 *   m = beta1 m + (1-beta1) g;  v = beta2 v + (1-beta2) g^2;  p -= step_size m / (sqrt(v) + eps);
 *   p -= decay p        (decoupled weight decay on the updated value, decay = lr * weight_decay)
 * step_size = lr * sqrt(1-beta2^t)/(1-beta1^t) (correct_bias) or lr, computed by the caller.
 * Work is cut into chunks of chunk_elems (multiple of 4) elements: chunk c updates elements
 * [chunk_off[c], chunk_off[c] + chunk_elems) of tensor chunk_tensor[c]. All three tables live in device
 * memory; tensors must be 16-byte aligned.
 * ------------------------------------------------------------------------------------------ */
typedef struct {
    float* param;
    const float* grad;
    float* exp_avg;
    float* exp_avg_sq;
    int64_t numel;
    float step_size, beta1, beta2, eps, decay, reserved;
} vb_adamw_tensor;

int vb_adamw_step(void* stream, int32_t n_chunks, const vb_adamw_tensor* table, const int32_t* chunk_tensor,
                  const int64_t* chunk_off, int32_t chunk_elems);

/* ------------------------------------------------------------------------------------------
 * Losses of the pre-training heads, one scalar each.
 *
 * vb_xent_fwd: nn.CrossEntropyLoss(ignore_index) on logits [rows, n] (leading dimension ld) with int64
 * labels [rows] - vilbert.py:1453 (`loss_fct`), :1578-1585 (masked-LM and alignment losses):
 *   row_loss[r] = logsumexp(logits[r]) - logits[r][label[r]]   (0 for ignored rows),  lse[r] saved,
 *   loss[0] = sum(row_loss) / count[0],  count[0] = number of rows whose label != ignore_index
 *   (0 / 0 = NaN when every row is ignored, like torch).
 * vb_xent_bwd: dlogits[r][j] = (exp(logits[r][j] - lse[r]) - [j == label[r]]) * grad_loss[0] / count[0];
 * ignored rows get zeros. grad_loss and count are DEVICE scalars (no host sync).
 *
 * vb_kl_fwd: sum over rows and classes of nn.KLDivLoss(reduction="none")(log_softmax(scores), target)
 * divided by `divisor` - vilbert.py:1454,1516-1522 (the caller passes only the labelled region rows, so
 * the reference's `* (image_label == 1)` mask is the row selection and divisor = their count).
 *   row_loss[r] = sum_j t_j (log t_j - (s_j - lse[r]))  (terms with t_j == 0 are 0),  tsum[r] = sum_j t_j;
 *   loss[0] = sum(row_loss) / divisor, loss[1] = divisor  (loss points to TWO floats).
 * vb_kl_bwd: dscores[r][j] = (exp(s_j - lse[r]) tsum[r] - t_j) * grad_loss[0] / divisor.
 * divisor_dev (may be NULL): DEVICE scalar that replaces `divisor` - the row count of a fixed-capacity gather that
 * the host never learns (sync-free / graph-captured training step); rows whose target is all zero contribute 0.
 * ------------------------------------------------------------------------------------------ */
int vb_xent_fwd(void* stream, int64_t rows, int32_t n, const float* logits, int64_t ld, const int64_t* labels,
                int64_t ignore_index, float* row_loss, float* lse, float* loss, float* count);
int vb_xent_bwd(void* stream, int64_t rows, int32_t n, const float* logits, int64_t ld, const int64_t* labels,
                int64_t ignore_index, const float* lse, const float* grad_loss, const float* count,
                float* dlogits, int64_t ldd);
int vb_kl_fwd(void* stream, int64_t rows, int32_t n, const float* scores, int64_t ld, const float* target,
              int64_t ldt, float divisor, float* row_loss, float* lse, float* tsum, float* loss,
              const float* divisor_dev);
int vb_kl_bwd(void* stream, int64_t rows, int32_t n, const float* scores, int64_t ld, const float* target,
              int64_t ldt, const float* lse, const float* tsum, const float* grad_loss, float divisor,
              float* dscores, int64_t ldd, const float* divisor_dev);

/* ------------------------------------------------------------------------------------------
 * vb_concap_finish_batch: device-side finishing of a Conceptual-Captions pre-training batch
 * (SURVEY.md section 8(f) row f3).
 *
 * Replaces the per-step numpy work of ConceptCapLoaderTrain.__iter__ (reference
 * vilbert/datasets/concept_cap_dataset.py:241-282: global mean-region feature row, its [0,0,1,1,1]
 * box and mask entry are prepended) and the objective-1 label edit of the training loop
 * (train_concap.py:535-540). Inputs are the worker's raw arrays copied to the device unchanged:
 *   image_feat [B,R,F] f32, image_loc [B,R,5] f32, image_mask [B,R] i64, masked_label [B,R] i64,
 *   is_next [B] i64, image_label [B,R] i64, lm_label_ids [B,T] i64.
 * Outputs: out_image_feat [B,R+1,F] (row 0 = sum over the R rows / max(#(masked_label == 0), 1),
 *   summed in fp32 in row order, divided in fp64 and rounded to fp32 like the reference's numpy
 *   expression), out_image_loc [B,R+1,5], out_image_mask [B,R+1] i64, out_image_label [B,R],
 *   out_lm_label_ids [B,T]; with objective == 1 both label tensors are multiplied by (is_next == 0) and
 *   zeros become -1, otherwise they are copied. F % 4 == 0; feature pointers 16-byte aligned.
 * ------------------------------------------------------------------------------------------ */
typedef struct {
    int32_t batch, regions, tokens, feat_dim, objective;
    const float* image_feat;
    const float* image_loc;
    const int64_t* image_mask;
    const int64_t* masked_label;
    const int64_t* is_next;
    const int64_t* image_label;
    const int64_t* lm_label_ids;
    float* out_image_feat;
    float* out_image_loc;
    int64_t* out_image_mask;
    int64_t* out_image_label;
    int64_t* out_lm_label_ids;
} vb_concap_batch;

int vb_concap_finish_batch(void* stream, const vb_concap_batch* a);

/* ------------------------------------------------------------------------------------------
 * bf16 TRAINING path (round 5; csrc/gemm_bf16.hip, rowops16.hip, attention.hip). Replaces the reference's reduced-precision
 * training mode - `model.half()` + apex FP16_Optimizer, /root/reference/train_concap.py:443-461,504-505 (train_tasks.py has
 * the same block) - the way gfx950 wants it: bf16 activations / saved tensors / activation gradients in HBM (no loss scaling:
 * bf16 has fp32's exponent range), v_mfma_f32_32x32x16_bf16 products with fp32 accumulation, fp32 LayerNorm statistics and
 * softmax, fp32 master weights updated by vb_adamw_step with fp32 gradients accumulated straight into the gradient arena,
 * and a bf16 shadow of every weight - row-major for the forward, transposed for the dgrad - refreshed after each optimizer
 * step (vb_weight_shadow_bf16). All bf16 arguments are bit patterns (uint16_t), leading dimensions in ELEMENTS.
 *
 * vb_linear_bf16:  C[M][N] = epilogue(A[M][K] W[N][K]^T), both operands contraction-contiguous. Forward: W = the shadow of
 *   nn.Linear.weight (stacked segments are stacked in the shadow). Input gradient: A = dY [M][N'], W = the TRANSPOSED shadow
 *   [K'][N'] -> dX [M][K'] (the same kernel). K % 64 == 0, N % 128 == 0; A / W 16-byte aligned, lda / ldw % 8 == 0.
 *   v = acc + bias[n] (fp32, in up to VB_MAX_SEGMENTS equal column segments; NULL = none), then exactly one of:
 *     act = GELU: gelu(v) (vilbert.py:111-117), and gelu'(v) to act_grad if given (bf16 [M][N]);  act = RELU: max(v, 0);
 *     residual (bf16 [M][N]):  dropout(v, dropout_p, seed) + residual   (mask element index = m * N + n, rng.h - the
 *         index vb_layernorm_bwd_bf16 regenerates it from; dropout only together with a residual);
 *     mul (bf16 [M][N]):       v * mul   (dgrad through an activation whose derivative the forward saved);
 *   written as bf16 to C (ldc % 2 == 0) or as fp32 to C32 (plain / residual / RELU only). Rows >= M are never stored.
 * vb_wgrad_bf16:   dW_s[seg_n][K] += dY[:, s seg_n : (s + 1) seg_n]^T X   for the nseg stacked segments of dY [M][nseg seg_n]
 *   (contraction over the M rows: row-major tiles in LDS, fragments by the transposing LDS read ds_read_b64_tr_b16), fp32
 *   atomics into dW (the gradient-arena slices: zero-filled once per backward pass, or holding an earlier contribution);
 *   dbias_s[seg_n] += the column sums of the segment's dY (the bias gradient, from the fragments the kernel holds anyway).
 *   seg_n % 256 == 0, K % 128 == 0, any M; dY / X 16-byte aligned, ldy / ldx % 8 == 0.
 * vb_colsum_bf16:  out[c] += sum_m x[m][c] (the bias gradient of the same dY), two deterministic stages through `workspace`
 *   (vb_colsum_bf16_workspace(cols) floats). cols % 4 == 0.
 * vb_weight_shadow_bf16: w fp32 [rows][cols] (ldw) -> w16 bf16 [rows][cols] (ld16; may be NULL) and wt16 = its transpose
 *   [cols][rows] (ldt; may be NULL), round to nearest even. rows, cols % 64 == 0. Stacked segments: call once per segment with
 *   w16 / wt16 offset to the segment's rows / columns.
 * vb_cast_f32_bf16 / vb_cast_bf16_f32: n elements, round to nearest even / exact; pointers 16-byte aligned.
 * vb_layernorm_fwd_bf16 / vb_layernorm_bwd_bf16: BertLayerNorm (vilbert.py:313-317) and its autograd on bf16 rows - the
 *   statistics (saved as fp32 mean / rstd per row), the normalisation and the gradient sums in fp32 as vb_layernorm_fwd /
 *   vb_layernorm_bwd; dgamma / dbeta fp32, OVERWRITTEN (two deterministic stages through `workspace` =
 *   vb_layernorm_bwd_bf16_workspace(rows, n_cols) floats); dx_dropped (optional, with 0 < dropout_p < 1): dx under the
 *   dropout mask (seed, row * n_cols + col) of the dense layer in front, written in the same pass. n_cols % 4 == 0, <= 1024.
 * ------------------------------------------------------------------------------------------ */
/* Attention of the bf16 training path: exactly vb_attention_fwd / vb_attention_bwd (same kernels, csrc/attention.hip
 * compiled with -DVB_ATTN_BF16; same argument meaning, same limits) with Q, K, V, O and dO, dQ, dK, dV as bf16 bit patterns
 * (row strides in ELEMENTS % 4 == 0, pointers 8-byte aligned); mask_add, probs, lse and dvec stay fp32; the operands are
 * widened to fp32 on their way into registers / LDS, products on the exact-fp32 MFMA, fp32 softmax. */
typedef struct {
    int32_t batch, heads, head_dim, n_q, n_k;
    int32_t q_batch, kv_batch;
    const uint16_t* Q;
    int64_t ldq;
    const uint16_t* K;
    int64_t ldk;
    const uint16_t* V;
    int64_t ldv;
    const float* mask_add;
    uint16_t* O;
    int64_t ldo;
    float* probs;
    float* lse;
    float scale;
    float dropout_p;
    uint64_t seed;
} vb_attention_bf16_args;

typedef struct {
    const uint16_t* dO;
    int64_t lddo;
    uint16_t* dQ;
    int64_t lddq;
    uint16_t* dK;
    int64_t lddk;
    uint16_t* dV;
    int64_t lddv;
    float* dvec;
    int32_t dvec_mode;   /* VB_DVEC_*, as in vb_attention_grads */
} vb_attention_bf16_grads;

int vb_attention_fwd_bf16(void* stream, const vb_attention_bf16_args* a);
int vb_attention_bwd_bf16(void* stream, const vb_attention_bf16_args* a, const vb_attention_bf16_grads* gr);

typedef struct {
    const uint16_t* A;
    int64_t lda;
    const uint16_t* W;
    int64_t ldw;
    const float* bias[VB_MAX_SEGMENTS]; /* bias of output columns [s N / bias_segments, (s + 1) N / bias_segments), or NULL */
    int32_t bias_segments;      /* 0 or 1: one bias of N values; stacked weights pass their nn.Linear biases unpacked */
    uint16_t* C;                /* bf16 out, or NULL */
    int64_t ldc;
    float* C32;                 /* fp32 out, or NULL (exactly one of C / C32) */
    int64_t ldc32;
    const uint16_t* residual;   /* or NULL */
    int64_t ldr;
    const uint16_t* mul;        /* or NULL (at most one of residual / mul) */
    int64_t ldm;
    uint16_t* act_grad;         /* GELU only: gelu'(v) out, or NULL */
    int64_t ldg;
    int64_t M, N, K;
    int32_t act;                /* VB_ACT_NONE | VB_ACT_GELU | VB_ACT_RELU */
    float dropout_p;
    uint64_t seed;
} vb_linear_bf16_args;

int vb_linear_bf16(void* stream, const vb_linear_bf16_args* a);

typedef struct {
    const uint16_t* dY;
    int64_t ldy;
    const uint16_t* X;
    int64_t ldx;
    float* dW[VB_MAX_SEGMENTS];
    int64_t ldw;
    float* dbias[VB_MAX_SEGMENTS];  /* [seg_n] each, ADDED into; NULL = no bias gradient for that segment */
    int64_t M, K;
    int32_t nseg, seg_n;
    int32_t n_valid;                /* round 6: rows of dW / entries of dbias that EXIST (nseg == 1): seg_n is the tile-padded
                                       width of dY (zero columns past n_valid), nothing past row n_valid is written; 0 = seg_n */
    int32_t reserved;
} vb_wgrad_bf16_args;

int vb_wgrad_bf16(void* stream, const vb_wgrad_bf16_args* a);

int64_t vb_colsum_bf16_workspace(int32_t cols);
int vb_colsum_bf16(void* stream, int64_t rows, int32_t cols, const uint16_t* x, int64_t ldx, float* out, float* workspace);
int vb_weight_shadow_bf16(void* stream, int32_t rows, int32_t cols, const float* w, int64_t ldw, uint16_t* w16, int64_t ld16,
                          uint16_t* wt16, int64_t ldt);
/* every registered weight in ONE launch (after an optimizer step): `table` = n_segs DEVICE records, one per weight segment -
 * w fp32 [rows][cols] contiguous, w16 / wt16 as vb_weight_shadow_bf16 (both required), tile0 = the number of 64 x 64 tiles of
 * all earlier records (ascending); total_tiles = the sum over all records of (rows / 64) (cols / 64). */
typedef struct {
    const float* w;
    uint16_t* w16;
    uint16_t* wt16;
    int32_t rows, cols;
    int64_t ld16, ldt;
    int64_t tile0;
} vb_shadow_seg;
int vb_weight_shadow_multi(void* stream, int32_t n_segs, const vb_shadow_seg* table, int64_t total_tiles);
int vb_cast_f32_bf16(void* stream, int64_t n, const float* x, uint16_t* y);
/* fp32 [rows][n] with row stride ldx -> bf16 [rows][ldy], the columns n .. ldy - 1 ZERO (ldx % 4 == 0, ldy % 8 == 0) */
int vb_cast_rows_f32_bf16(void* stream, int64_t rows, int32_t n, const float* x, int64_t ldx, uint16_t* y, int64_t ldy);
int vb_cast_bf16_f32(void* stream, int64_t n, const uint16_t* x, float* y);
int vb_layernorm_fwd_bf16(void* stream, int64_t rows, int32_t n_cols, const uint16_t* x, const float* gamma, const float* beta,
                          float eps, uint16_t* y, float* mean, float* rstd);
int64_t vb_layernorm_bwd_bf16_workspace(int64_t rows, int32_t n_cols);
int vb_layernorm_bwd_bf16(void* stream, int64_t rows, int32_t n_cols, const uint16_t* dy, const uint16_t* x, const float* mean,
                          const float* rstd, const float* gamma, uint16_t* dx, float* dgamma, float* dbeta, float* workspace,
                          uint16_t* dx_dropped, float dropout_p, uint64_t seed);

/* ------------------------------------------------------------------------------------------
 * Whole-layer entry points (round 6; SURVEY.md section 8(b) "vb_text_layer_fwd ... vb_connection_layer_fwd"): ONE call
 * enqueues the kernel sequence of a BertLayer / BertImageLayer (reference vilbert.py:527-533, 688-694) or of a
 * BertConnectionLayer (:871-900) - forward, or backward including every weight gradient - into caller-allocated buffers.
 * Same kernels, same arithmetic and same order as the per-op entry points above (the launcher only calls those); what it
 * removes is the host work between them: at the reference's per-GPU batch 64 the eager step was bound by ~1,400 Python ->
 * ctypes launches per step.
 *
 * dtype: VB_DT_F32 = fp32 tensors (vb_linear_fwd / vb_attention_fwd / vb_layernorm_fwd ...), VB_DT_BF16 = the bf16
 * training path (vb_linear_bf16 on the weight shadows, ...); every `void*` activation below has that element type,
 * leading dimension = its width (contiguous rows). Layer = attention block + one ("self" layers) or two (connection
 * layer: stream 1 = image regions, stream 2 = text tokens) output + feed-forward blocks:
 *
 *   attention block, self (n2 == 0):  qkv1_out = x1 . [Wq|Wk|Wv]^T + b;  ctx1 = attn(q, k, v | mask1, p1, seed1)
 *   attention block, co-attention:    qkv1_out = x1 . W1^T, qkv2_out = x2 . W2^T;
 *                                     ctx1 = attn(q2; k1, v1 | mask1, p1) [batch n2 rows: the TEXT stream's context],
 *                                     ctx2 = attn(q1; k2, v2 | mask2, p2) [batch n1 rows: the IMAGE stream's] (:768-809)
 *   output + FFN block:  sum1 = dropout(ctx . Wo^T + bo, p_o, seed_o) + x;   a1 = LayerNorm(sum1)
 *                        h = gelu(a1 . W1^T + b1), dact = gelu'(.) (training);  sum2 = dropout(h . W2^T + b2, p_f, seed_f) + a1
 *                        y = LayerNorm(sum2)
 * The caller wires the blocks: s1.ctx = the attention block's ctx1 (self) / ctx2 (co-attention), s2.ctx = ctx1; in backward
 * attn.d_ctx* = the blocks' d_ctx and attn.dres* = the blocks' d_sum1 (the gradient arriving over the skip connection, added
 * in the epilogue of the q|k|v input-gradient GEMM: dx = dqkv . W + dres).
 * Backward: weight / bias gradients are ADDED into dw / dbias (gradient-arena semantics; all of a linear's targets given, or
 * none: NULL skips that weight gradient), dgamma / dbeta are OVERWRITTEN. wgrad_stream != NULL: the weight-gradient launches
 * go to that stream behind an event of the launch stream (the caller joins it before the gradients are read).
 * ------------------------------------------------------------------------------------------ */
#define VB_DT_F32 0
#define VB_DT_BF16 1

typedef struct {
    int32_t nseg, seg_n, K;                    /* nseg stacked nn.Linear weights [seg_n, K] */
    const float* w[VB_MAX_SEGMENTS];           /* fp32 weights (the GEMM operands of the fp32 path) */
    const float* bias[VB_MAX_SEGMENTS];
    const uint16_t* w16;                       /* bf16 path: row-major shadow [nseg seg_n, K] */
    const uint16_t* wt16;                      /* bf16 path: transposed shadow [K, nseg seg_n] */
    float* dw[VB_MAX_SEGMENTS];                /* backward targets, ADDED into */
    float* dbias[VB_MAX_SEGMENTS];
} vb_layer_linear;

typedef struct {
    const float* gamma;
    const float* beta;
    float* dgamma;                             /* backward, OVERWRITTEN */
    float* dbeta;
} vb_layer_norm;

typedef struct {
    int64_t M;                                 /* rows; 0 = block absent */
    int32_t Hc, H, I;                          /* context width, hidden size, intermediate size */
    const void* ctx;                           /* [M, Hc] */
    const void* x;                             /* [M, H] the layer's input (residual) */
    vb_layer_linear o, f1, f2;
    vb_layer_norm ln1, ln2;
    float eps, p_o, p_f;
    uint64_t seed_o, seed_f;
    void* sum1; void* a1; void* h; void* dact; void* sum2; void* y;      /* [M,H] [M,H] [M,I] [M,I] [M,H] [M,H] */
    float* mean1; float* rstd1; float* mean2; float* rstd2;              /* [M] each (training) */
    const void* dy;                            /* backward: [M, H] */
    void* d_sum2; void* d_sum2_drop; void* d_pre; void* d_a1; void* d_sum1; void* d_sum1_drop; void* d_ctx;
    float* ln_ws;                              /* vb_layernorm_bwd(_bf16)_workspace(M, H) floats */
} vb_ffn_block;

typedef struct {
    int32_t batch, heads, head_dim, n1, n2;    /* n2 == 0: self-attention */
    const void* x1; const void* x2;            /* [batch n1, K1], [batch n2, K2] */
    const float* mask1; const float* mask2;    /* additive fp32 [batch, n1] / [batch, n2] */
    vb_layer_linear qkv1, qkv2;
    float p1, p2;
    uint64_t seed1, seed2;
    void* qkv1_out; void* qkv2_out;            /* [batch n1, 3 Hb], [batch n2, 3 Hb], Hb = heads head_dim */
    float* lse1; float* lse2;                  /* [batch, heads, n_q] (training) */
    void* ctx1; void* ctx2;
    const void* d_ctx1; const void* d_ctx2;    /* backward */
    void* dqkv1; void* dqkv2;
    float* dvec;                               /* scratch [batch, heads, max(n1, n2)] */
    const void* dres1; const void* dres2;
    void* dx1; void* dx2;
} vb_attn_block;

typedef struct {
    int32_t dtype;                             /* VB_DT_F32 | VB_DT_BF16 */
    int32_t training;                          /* forward: store what backward needs (lse, mean / rstd, dact) */
    void* wgrad_stream;                        /* backward: stream of the weight-gradient launches, or NULL */
    vb_attn_block attn;
    vb_ffn_block s1;
    vb_ffn_block s2;
} vb_layer_args;

int vb_layer_fwd(void* stream, const vb_layer_args* a);
int vb_layer_bwd(void* stream, const vb_layer_args* a);

#ifdef __cplusplus
}
#endif
#endif /* VILBERT_HIP_H */
