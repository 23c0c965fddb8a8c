/* C-ABI of libbtvae_hip.so: gfx950 (MI355X) kernels for the conditional-VAE training and Monte-Carlo
 * scoring path of kabolat/behavio-temporal-vae.
 *
 * The reference is pure Python/PyTorch and has no FFI of its own; each entry point below names the
 * reference op sequence (file:line under the reference root) it replaces.  All pointers are DEVICE
 * pointers to contiguous row-major fp32 arrays unless stated; `stream` is a hipStream_t.  Every function
 * returns 0 on success, non-zero on failure with the reason in btvae_last_error().  Nothing here
 * allocates or frees device memory or synchronises the device; the optional event timing (btvae_prof_enable) creates
 * HIP events on first use and recycles them.
 */
#ifndef BTVAE_HIP_H
#define BTVAE_HIP_H
#ifdef __cplusplus
extern "C" {
#endif

/* ---- library state ---- */
const char* btvae_last_error(void);
int btvae_version(void);
/* Verify that `device` is gfx950 and set up the library's per-device state on it.  Call once per device before anything
 * else.  The caller's current device is left unchanged; every other call runs on the device that is current. */
int btvae_init(int device);
/* Event timing per kernel family (0 gemm_fwd, 1 gemm_dgrad, 2 gemm_wgrad, 3 gemm_mvn, 4 bn, 5 mvn_rows,
 * 6 reduce, 7 optim); collect after a device synchronisation into arrays of 8. */
int btvae_prof_enable(int on);
int btvae_prof_enabled(void);
/* Time only the families in kind_mask (bit k = family k) and launches of at least min_flops; (~0u, 0) = everything. */
int btvae_prof_filter(unsigned kind_mask, double min_flops);
int btvae_prof_collect(double* ms, double* flops, double* bytes, long long* launches, int n);
/* Before btvae_prof_collect: the launches of one family with the largest FLOP count (its dominant shape).  kind = family
 * + 256 * selection: 0 all of them, 1 only those on fp32 / bf16 operands, 2 only those with an operand in the pair format. */
int btvae_prof_dominant(int kind, double* ms, double* flops_per_launch, long long* count);

/* Operand precision of the matrix products.  0 = the fp32 MFMA instruction (the mode the library starts in; the
 * precision torch.nn.Linear runs at in the reference, src/submodels.py:15-17; the Python host mirror selects mode 3 and watches it with the guard below
 * unless a model is configured otherwise).  1 = operands of the Linear-layer products
 * (btvae_linear_*_fwd, _dgrad, _wgrad) rounded to bf16 on the way into LDS, fp32 accumulation; the covariance products
 * stay fp32.  2 = every product fp32-accurate on the bf16 matrix pipe: operands split exactly into three bf16 pieces,
 * six bf16 MFMAs per product (error against fp64 at or below mode 0's).  3 = as 2, except that a Linear-layer product
 * given the absolute maxima of its two operand tensors (btvae_gemm_operand_maxima) runs on two fp16 pieces per value
 * of the tensor scaled by a power of two that puts its maximum just below 2^15: three fp16 MFMAs per product; the
 * representation error of an element is ~2^-22 of its own magnitude down to 2^-18 of the tensor's maximum and ~2^-40
 * of that maximum below it.  Process-wide. */
int btvae_set_gemm_precision(int mode);
int btvae_get_gemm_precision(void);

/* Pre-split weights (mode 2).  Every workgroup of a mode-2 product splits the operand tiles it stages; a weight matrix
 * W[N,K] (torch.nn.Linear.weight, src/submodels.py:15-17) is re-split by all of them.  btvae_weight_planes splits it once
 * into `planes`: three bf16 planes [N padded to 16][K padded to 16] (high, middle, low pieces, zero padded, 16-byte
 * aligned base, btvae_weight_planes_elems(N, K) 2-byte elements in all).  The *_pw variants of the forward and dX
 * products take the planes next to W and read the weight operand from them in mode 2 (other modes, unaligned operands
 * and Wplanes == NULL use W as the plain entry points do).  The planes hold one tile-ordered image per direction;
 * `which` = 1 writes the forward image, 2 the dX image, 3 both, and a product must be given planes whose image for its
 * direction is current.  The planes are a copy: call btvae_weight_planes again whenever W has changed. */
long long btvae_weight_planes_elems(int N, int K);
int btvae_weight_planes(const float* W, int ldw, int N, int K, void* planes, int which, void* stream);
/* The same for `count` contiguous matrices W[i][N[i]][K[i]] (host arrays of length count) in one launch per 16 entries. */
int btvae_weight_planes_multi(const float* const* W, const int* N, const int* K, void* const* planes, const int* which,
                              int count, void* stream);

/* Mode 3 (no counterpart in the reference: torch.nn.Linear multiplies fp32 values, src/submodels.py:15-17).
 * A maximum "slot" is TWO floats in device memory: [0] = max |x|, [1] = a range statistic the same kernels maintain for the
 * guard below (the largest of 0x7f800000 - bits(m) over the regional maxima m > 0 of the waves / workgroups that wrote
 * the tensor); both start at zero.  Every amax / amax_out pointer below points at such a pair.
 * btvae_absmax folds max|x| over n floats into the slot *out, which must hold zeros or a maximum to extend;
 * btvae_slots_zero clears n neighbouring slots with a kernel launch (use it rather than hipMemsetAsync on a stream that may
 * be under capture: on memory that belongs to the graph's own allocation node a captured memset node has no effect, ROCm 7.2).
 * btvae_gemm_operand_maxima names the device scalars holding the maxima of the A operand (activations or gradients)
 * and the B operand (weight, or activations for a weight gradient) of the NEXT product launched from the calling
 * thread; that launch consumes them.  Each must be final on the product's stream and cover every element the product
 * reads (a larger value costs precision, a smaller one overflows fp16).  btvae_weight_planes_f16 writes the planes of
 * W as two fp16 pieces scaled by *amax (same buffer size and image selection as btvae_weight_planes); a *_pw product
 * reads such planes only when it was also given the maxima, the weight's being the value the planes were made with.
 * btvae_weight_planes_f16_multi computes each matrix's maximum into amax[i] (device scalars) and writes its planes.
 * btvae_output_maximum asks the next launch, from the calling thread, of a kernel that writes an operand of a later
 * Linear product (btvae_linear_relu_stats_fwd*, btvae_linear_relu_affine_fwd*, btvae_bn_apply_fwd,
 * btvae_linear_dgrad_bnbwd*, btvae_bn_relu_bwd, btvae_bn_relu_bwd_apply, btvae_relu_bwd, btvae_bn_fold_linear,
 * btvae_reparam_kl_fwd) to fold max|Y / Out / dPre / Wf / zc| into *amax_out as btvae_absmax would, without a pass of
 * its own. */
int btvae_absmax(const float* x, long long n, float* out, void* stream);
/* n floats (two per slot) */
int btvae_slots_zero(float* slots, int n, void* stream);
/* Guard of the mode-3 products (no counterpart in the reference).  Two scaled fp16 pieces represent an element to ~2^-23
 * of its magnitude only while it is within ~2^17 of its tensor's maximum, and a maximum that understates the data
 * overflows fp16.  A product launched after btvae_gemm_guard_site(s) (consumed by that launch; -1 = none) records in the
 * host-visible array btvae_guard_flags() (2 * btvae_guard_sites() ints in mapped host memory, set up by the first
 * btvae_init; NULL before): [2 s] = 1 when an operand's slot shows a region more than 2^16 below the tensor's maximum
 * (btvae_guard_range_log2 sets another threshold, process-wide; a region that far down holds elements ~2^19 below the
 * maximum, still represented to ~2^-21 of their magnitude) -- run that product without maxima (three exact pieces) from
 * then on; [2 s + 1] = 1 when its accumulators came out non-finite -- an error.  What the host reads is final for
 * launches that have completed; it clears what it has handled.  btvae_guard_check makes the range test as a launch of
 * its own (for callers that synchronise and decide before they launch the product). */
int btvae_guard_sites(void);
int btvae_guard_range_log2(int log2_range);
int* btvae_guard_flags(void);
int btvae_gemm_guard_site(int site);
int btvae_guard_check(const float* amax_a, const float* amax_b, int site, void* stream);
int btvae_output_maximum(float* amax_out);
int btvae_gemm_operand_maxima(const float* amax_a, const float* amax_b);
int btvae_weight_planes_f16(const float* W, int ldw, int N, int K, void* planes, int which, const float* amax, void* stream);
int btvae_weight_planes_f16_multi(const float* const* W, const int* N, const int* K, void* const* planes, const int* which,
                                  float* const* amax, int count, void* stream);

/* ---- Linear layers on MFMA (src/submodels.py:15-17 Linear+ReLU, :20-21 BatchNorm1d, :28-35 NNBlock.forward) ---- */
/* Y[M,N] = relu(X[M,K] W[N,K]^T + b); stat_partials[ceil(M/128)][2][N] = per-row-tile column (mean, M2). */
int btvae_linear_relu_stats_fwd(const float* X, int ldx, const float* W, const float* b, float* Y,
                                float* stat_partials, int M, int K, int N, void* stream);
int btvae_linear_relu_stats_fwd_pw(const float* X, int ldx, const float* W, const void* Wplanes, const float* b, float* Y,
                                   float* stat_partials, int M, int K, int N, void* stream);
/* Eval mode: Out = relu(X W^T + b) * scale + shift (+ R); scale/shift/R may be NULL. */
int btvae_linear_relu_affine_fwd(const float* X, int ldx, const float* W, const float* b, const float* scale,
                                 const float* shift, const float* R, float* Out, int M, int K, int N, void* stream);
int btvae_linear_relu_affine_fwd_pw(const float* X, int ldx, const float* W, const void* Wplanes, const float* b, const float* scale,
                                    const float* shift, const float* R, float* Out, int M, int K, int N, void* stream);
/* The same with row pitches for the residual input and the output (ldr, ldo >= N): a chain of eval-mode layers keeps its
 * activations in buffers whose rows start on 128-byte lines. */
int btvae_linear_relu_affine_fwd_pw_ld(const float* X, int ldx, const float* W, const void* Wplanes, const float* b, const float* scale,
                                       const float* shift, const float* R, int ldr, float* Out, int ldo, int M, int K, int N, void* stream);
/* Head output Linear + non-linearity (src/submodels.py:102-104,156-157): act 0 clamp(raw,lo,hi),
 * 1 clamp(softplus(raw, threshold 5),lo,hi) (src/utils.py:10), 2 clamp(relu(raw),lo,hi).
 * Dfac = d Out / d raw and Sq = Out^2 are optional outputs (NULL to skip). */
int btvae_linear_head_fwd(const float* X, int ldx, const float* W, const float* b, int act, float lo, float hi,
                          float* Out, float* Dfac, float* Sq, int M, int K, int N, void* stream);
/* The same with a workspace (ws_floats floats; NULL / 0 = none): a product with at most 128 output tiles and K >= 256
 * splits its reduction over up to 512 workgroups into slabs held in ws and applies the non-linearity where they are
 * summed (small batches: 8 workgroups walking K = 1000 alone take 66 us, 16 us this way). */
int btvae_linear_head_fwd_ws(const float* X, int ldx, const float* W, const float* b, int act, float lo, float hi,
                             float* Out, float* Dfac, float* Sq, float* ws, long long ws_floats, int M, int K, int N, void* stream);
/* dX[M,K] = dY[M,N] W[N,K] (+ R[M,K], same leading dimension lddx).  Replaces autograd's Linear dgrad
 * (src/vae_models.py:322). */
int btvae_linear_dgrad(const float* dY, const float* W, const float* R, float* dX, int lddx, int M, int N, int K, void* stream);
int btvae_linear_dgrad_pw(const float* dY, const float* W, const void* Wplanes, const float* R, float* dX, int lddx, int M, int N, int K, void* stream);
/* dW[N,K] (+)= dY[M,N]^T X[M,K]; ws = split-reduction workspace of ws_floats floats (>= N*K to split). */
int btvae_linear_wgrad(const float* dY, const float* X, int ldx, float* dW, float* ws, long long ws_floats,
                       int M, int N, int K, int accumulate, void* stream);
/* C[M,N] (+)= A^T B with A stored [K][M], B stored [K][N]. prof_kind: timing family to book under. */
int btvae_matmul_tn(const float* A, int lda, const float* B, int ldb, float* C, float* ws, long long ws_floats,
                    int M, int N, int K, int accumulate, int prof_kind, void* stream);
/* C[M,N] = A[M,K] B[N,K]^T + bias[N]   (covariance build, src/submodels.py:176-179). */
int btvae_matmul_nt_bias(const float* A, int lda, const float* B, int ldb, const float* bias, float* C, int ldc,
                         int M, int N, int K, void* stream);
/* C[M,N] = A[M,K] B[N,K]^T + R[M,N]    (correlated sampling mu + Uhat (eps*sigma), src/submodels.py:160-164). */
int btvae_matmul_nt_add(const float* A, int lda, const float* B, int ldb, const float* R, int ldr, float* C, int ldc,
                        int M, int N, int K, void* stream);
/* C[M,N] = (A[M,K] B[K,N]) * E1[M,N] * E2[M,N]. */
int btvae_matmul_nn_mul(const float* A, int lda, const float* B, int ldb, const float* E1, const float* E2, int lde,
                        float* C, int ldc, int M, int N, int K, void* stream);

/* Activation and gradient matrices with a row pitch (no counterpart in the reference, whose tensors are dense).  The
 * next launch, from the calling thread, of btvae_linear_relu_stats_fwd* (Y), btvae_bn_apply_fwd (Y, R, Out),
 * btvae_linear_dgrad* (dY), btvae_linear_dgrad_bnbwd* / _gstats* (dY; R, Yb, dPre / dX), btvae_linear_wgrad / _affine (dY),
 * btvae_bn_relu_bwd / _sums / _apply and btvae_relu_bwd (G, Y, dPre) or btvae_colsum / btvae_colsum_noalloc (G) takes the rows of those [M, N]
 * matrices to be ld_n floats apart and the rows of the [M, K] ones ld_k apart (0 = dense; at least the matrix's width;
 * arguments that have a leading dimension of their own -- ldx, ldy, lddx -- keep it; dropout keep flags stay dense).
 * Width-1000 rows that start on 128-byte lines (pitch 1024) make the products that read them 6-11 % faster. */
int btvae_next_row_pitches(int ld_n, int ld_k);

/* bf16 storage of hidden activations and gradients (mode 1 only; no counterpart in the reference, whose tensors are fp32:
 * src/submodels.py:28-35).  The [rows, hidden width] matrices between the Linear layers may live in HBM as bf16 -- the
 * values mode 1 multiplies anyway.  btvae_next_bf16(mask) names which matrices of the NEXT launch from the calling
 * thread are bf16 (pitches then count bf16 elements): bit 0 the A operand (X of a forward product, dY of a dX or
 * weight-gradient product), bit 1 the output (Y / Out / dX / dPre), bit 2 the add-in R together with the BatchNorm input
 * Yb, bit 3 the B operand X of a weight gradient; for btvae_bn_apply_fwd, btvae_bn_relu_bwd*, btvae_relu_bwd any non-zero
 * mask means all their [M, N] matrices.  Widths must be multiples of 8 (4 for outputs), bases 16-byte aligned.  Producers
 * write zeros into the pad columns between the width and the pitch.
 * A forward / dX product whose A operand is bf16 with a pitch that is a multiple of 32 and whose Wplanes argument holds
 * the weight's stage images (btvae_weight_image_bf16: forward image at the start, dX image
 * btvae_weight_image_bf16_bytes(N, K, 0) bytes further; `which` 1 / 2 / 3 as for btvae_weight_planes; buffer of
 * _bytes(N, K, 0) + _bytes(N, K, 1) bytes, 16-byte aligned) runs on a 256-wide kernel that copies both operands
 * global -> LDS by DMA (780 TFLOP/s at M = 65536, N = K = 1000 against 360 for fp32 tensors); without images the
 * 128-wide kernel reads the fp32 weight. */
int btvae_next_bf16(int mask);
long long btvae_weight_image_bf16_bytes(int N, int K, int transpose);
int btvae_weight_image_bf16(const float* W, int ldw, int N, int K, int which, void* image, void* stream);
/* The same for `count` contiguous matrices (host arrays of length count) in one launch per 32 images. */
int btvae_weight_image_bf16_multi(const float* const* W, const int* N, const int* K, void* const* image, const int* which,
                                  int count, void* stream);

/* Pair-format activations (mode 3, eval-mode chains; no counterpart in the reference, whose tensors are fp32:
 * src/submodels.py:28-35).  A matrix X[M][N] in the pair format occupies the bytes of the fp32 matrix of the same pitch
 * (a multiple of 16 floats): per row and 16 columns 64 bytes = 16 fp16 high pieces p = fp16(s x), then 16 low pieces
 * q = fp16(s x - p), with one power-of-two scale s per 128 x 128 block taken from the block's own maximum
 * (sc[ceil(M/128)][ceil(N/128)] floats).  x = (p + q) / s to 2^-23 of its magnitude for elements within 2^17 of their
 * block's maximum; the pieces are the operands of the two-piece products as they stand, so a consumer needs no arithmetic
 * on them and no per-tensor maximum.  Columns between N and the end of the last 16-group are zeros.
 * btvae_pairs_encode / _decode convert whole matrices (N % 8 == 0; in place when the pitches agree).
 * btvae_weight_image_pq writes the stage images of W[R][C] (transpose 0: output columns = rows of W, the forward
 * products) that the 256-wide kernel copies into LDS; *amax = max |W| (a slot made by btvae_absmax), buffer of
 * btvae_weight_image_pq_bytes(R, C, transpose) bytes, 16-byte aligned.
 * btvae_linear_relu_affine_fwd_pairs is btvae_linear_relu_affine_fwd_pw_ld for either format on either side (residual != 0:
 * the unit's input, in the pair format and of the output's width, is added to the result): x_sc NULL = X is fp32 (Wsplit = the weight's fp16 planes or NULL, operand maxima through
 * btvae_gemm_operand_maxima), else X is in the pair format with block scales x_sc and Wsplit = the weight's stage
 * images made with the maximum *wmax; out_sc NULL = Out is fp32, else Out is written in the pair format (ldo % 16 == 0)
 * and out_sc receives its block scales.  Pitches count floats in both formats.
 * Training-side forms (two-piece mode): btvae_linear_relu_stats_fwd_pairs = btvae_linear_relu_stats_fwd_pw on a pair-format X
 * (Y fp32 with pitch ldy, the same statistics partials); btvae_linear_wgrad_pairs = btvae_linear_wgrad on a pair-format X (the
 * maximum of dY through btvae_gemm_operand_maxima(slot, NULL)); btvae_bn_apply_fwd_pairs = btvae_bn_apply_fwd without dropout
 * writing Out in the pair format, its add-in R fp32 (r_sc NULL) or in the pair format. */
int btvae_linear_relu_stats_fwd_pairs(const void* Xp, int ldx, const float* x_sc, const void* Wimage, const float* wmax, const float* b,
                                      float* Y, int ldy, float* stat_partials, int M, int K, int N, void* stream);
int btvae_linear_wgrad_pairs(const float* dY, const void* Xp, int ldx, const float* x_sc, float* dW, float* ws, long long ws_floats,
                             int M, int N, int K, int accumulate, void* stream);
int btvae_bn_apply_fwd_pairs(const float* Y, int ldy, const float* mean, const float* invstd, const float* gamma, const float* beta,
                             const void* R, int ldr, const float* r_sc, void* Out, int ldo, float* out_sc, int M, int N, void* stream);
int btvae_pairs_encode(const float* X, int ldx, void* P, int ldp, float* sc, int M, int N, void* stream);
int btvae_pairs_decode(const void* P, int ldp, const float* sc, float* X, int ldx, int M, int N, void* stream);
long long btvae_weight_image_pq_bytes(int R, int C, int transpose);
int btvae_weight_image_pq(const float* W, int ldw, int R, int C, int transpose, void* image, const float* amax, void* stream);
int btvae_linear_relu_affine_fwd_pairs(const void* X, int ldx, const float* x_sc, const float* W, const void* Wsplit, const float* wmax,
                                       const float* b, const float* scale, const float* shift, int residual, void* Out, int ldo,
                                       float* out_sc, int M, int K, int N, void* stream);

/* ---- BatchNorm1d (torch.nn.BatchNorm1d as used at src/submodels.py:20-21,50) ---- */
int btvae_strip_chunks(int M);
/* Merge the GEMM's per-tile partials: mean, invstd = 1/sqrt(biased var + eps); running buffers (may both be
 * NULL) move with `momentum` using the unbiased variance. */
int btvae_bn_stats_finalize(const float* partials, int ntiles, int M, int N, float* mean, float* invstd,
                            float* running_mean, float* running_var, float momentum, float eps, void* stream);
/* Statistics that span ranks (torch.nn.SyncBatchNorm semantics for the BatchNorm1d of src/submodels.py:20-21,50):
 * btvae_bn_stats_local leaves this rank's local[0] = mean, local[1] = M2; after an all-gather,
 * btvae_bn_stats_merge combines `nparts` such pairs of `rows_per_part` rows each. */
int btvae_bn_stats_local(const float* partials, int ntiles, int M, int N, float* local, void* stream);
int btvae_bn_stats_merge(const float* parts, int nparts, int rows_per_part, int N, float* mean, float* invstd,
                         float* running_mean, float* running_var, float momentum, float eps, void* stream);
/* Eval mode: scale = gamma/sqrt(running_var+eps), shift = beta - running_mean*scale. */
int btvae_bn_fold_eval(const float* gamma, const float* beta, const float* running_mean, const float* running_var,
                       float eps, float* scale, float* shift, int N, void* stream);
/* keep[i] = 1 with probability 1-p from Philox(seed, offset): the Bernoulli flags of one train-mode
 * torch.nn.Dropout(p) call (src/submodels.py:21). */
int btvae_dropout_keep_mask(unsigned char* keep, long long n, float p, unsigned long long seed,
                            unsigned long long offset, void* stream);
/* Out = dropout((Y-mean)*invstd*gamma + beta) (+ R): normalisation, optional dropout (keep NULL = none;
 * kept elements are scaled by keep_scale = 1/(1-p)) and the residual add of src/submodels.py:21,32.
 * mean = invstd = gamma = beta = NULL skips the normalisation (units built without BatchNorm). */
int btvae_bn_apply_fwd(const float* Y, const float* mean, const float* invstd, const float* gamma, const float* beta,
                       const float* R, const unsigned char* keep, float keep_scale, float* Out, int M, int N,
                       void* stream);
/* Backward of ReLU -> BatchNorm(train) (-> Dropout): from G = dL/dOut, the saved Y (post-ReLU) and the
 * forward's keep flags (NULL = no dropout) produce dgamma, dbeta, dbias (column sums of dPre) and
 * dPre = dL/d(Linear output).  scratch: btvae_bn_relu_bwd_scratch(M,N) floats. */
long long btvae_bn_relu_bwd_scratch(int M, int N);
int btvae_bn_relu_bwd(const float* G, const float* Y, const float* mean, const float* invstd, const float* gamma,
                      const unsigned char* keep, float keep_scale, float* dgamma, float* dbeta, float* dbias, float* dbias_cur,
                      float* dPre, float* scratch, int M, int N, int accumulate, void* stream);
/* btvae_bn_relu_bwd in two halves with a sum-all-reduce of sums[2][N] between them (statistics spanning ranks):
 * _sums leaves this rank's (sum g, sum g*xhat) and adds them into dbeta / dgamma; _apply takes the batch-wide sums
 * and the global row count.  scratch as for btvae_bn_relu_bwd. */
int btvae_bn_relu_bwd_sums(const float* G, const float* Y, const float* mean, const float* invstd,
                           const unsigned char* keep, float keep_scale, float* dgamma, float* dbeta, float* sums,
                           float* scratch, int M, int N, int accumulate, void* stream);
int btvae_bn_relu_bwd_apply(const float* G, const float* Y, const float* mean, const float* invstd, const float* gamma,
                            const unsigned char* keep, float keep_scale, const float* sums, long long total_rows,
                            float* dbias, float* dbias_cur, float* dPre, float* scratch, int M, int N, int accumulate, void* stream);
/* Backward of Linear -> ReLU (-> Dropout) units built without BatchNorm (batch_normalization: false,
 * src/submodels.py:15-25): dPre = (Y>0) * dropout(G), dbias (+)= column sums.  scratch: btvae_strip_chunks(M)*N. */
int btvae_relu_bwd(const float* G, const float* Y, const unsigned char* keep, float keep_scale, float* dbias,
                   float* dbias_cur, float* dPre, float* scratch, int M, int N, int accumulate, void* stream);
/* out[N] (+)= column sums of G[M,N], cur[N] = the sums alone (either may be null); scratch: btvae_strip_chunks(M)*N floats.
 * The backward entry points above take the same optional `dbias_cur`: this call's bias gradient regardless of
 * `accumulate`, which btvae_linear_wgrad_affine needs. */
int btvae_colsum(const float* G, float* out, float* cur, float* scratch, int M, int N, int accumulate, void* stream);
/* btvae_colsum for callers that cannot allocate (code under HIP graph capture, tight serving loops, harnesses that forbid
 * hipMalloc).  No counterpart in the reference.  Takes no workspace and allocates nothing: the strip partials live in a
 * static 4 MiB device array inside the library, one copy per GPU, whose address btvae_init(device) resolves.  Rejects
 * btvae_strip_chunks(M) * N > btvae_colsum_noalloc_max_floats() (1,048,576).  All calls on one GPU share the array: calls
 * that overlap on different streams of the same GPU race; successive calls on one stream are safe.  Like btvae_colsum it
 * consumes a pending btvae_next_row_pitches request (ld_n = row pitch of G). */
long long btvae_colsum_noalloc_max_floats(void);
int btvae_colsum_noalloc(const float* G, float* out, float* cur, int M, int N, int accumulate, void* stream);
/* A Linear[N,K] fed by a train-mode BatchNorm (src/submodels.py:15-21 followed by the next block's Linear, :30-34,
 * :63-68) computes (Y*sc + sh) W^T + b = Y (W*sc)^T + (b + W sh): Wf, bf are those weights, sc/sh[K] the BatchNorm's
 * column scale and shift.  The normalised activation is then never written; the weight gradient follows from Y by
 * btvae_linear_wgrad_affine: dW (+)= sc[k] (dY^T Y)[n,k] + sh[k] dbias_cur[n]. */
int btvae_bn_fold_linear(const float* W, const float* b, const float* mean, const float* invstd, const float* gamma,
                         const float* beta, float* Wf, float* bf, float* scale, float* shift, int N, int K, void* stream);
int btvae_linear_wgrad_affine(const float* dY, const float* Y, int ldy, const float* scale, const float* shift,
                              const float* dbias_cur, const float* W, float* col_wp, float* col_wb, int cols_accumulate,
                              float* dW, float* ws, long long ws_floats, int M, int N, int K, int accumulate, void* stream);
/* With W, col_wp, col_wb given, btvae_linear_wgrad_affine also leaves col_wp[k] (+)= sum_n W[n,k] (dY^T Y)[n,k] and
 * col_wb[k] (+)= sum_n dbias_cur[n] W[n,k]: the sums over rows of G*Y and G for G = dY W, the gradient that reaches
 * the BatchNorm output (src/submodels.py:19-21 backward).  From them the BatchNorm-backward column sums follow without
 * a pass over the activations: sums[0] = col_wb, sums[1] = invstd (col_wp - mean col_wb); then btvae_bn_relu_bwd_apply. */
/* The consumer's dX product fused with that BatchNorm's backward: dPre of the producing unit is written where the gradient
 * G = dY W (+R) would have been, plus per-128-row-tile column sums of dPre (colpart[ceil(M/128)][K], finished into the bias
 * gradient by btvae_colsum_partials).  sums[2][K] from btvae_bn_bwd_sums_from_products (all-reduced when statistics span
 * ranks; total_rows = rows they span).  Needs K % 4 == 0. */
int btvae_linear_dgrad_bnbwd(const float* dY, const float* W, const float* R, const float* Yb, const float* mean,
                             const float* invstd, const float* gamma, const float* sums, long long total_rows,
                             float* dPre, float* colpart, int M, int N, int K, void* stream);
int btvae_linear_dgrad_bnbwd_pw(const float* dY, const float* W, const void* Wplanes, const float* R, const float* Yb, const float* mean,
                                const float* invstd, const float* gamma, const float* sums, long long total_rows,
                                float* dPre, float* colpart, int M, int N, int K, void* stream);
/* dX product that also leaves the reduction pass of the BatchNorm backward of the unit whose output it differentiates
 * (units read through a residual connection, whose sums do not follow from weight gradients): gpart[ceil(M/128)][2][K],
 * finished by btvae_bn_bwd_sums_from_partials into sums[2][K] (+ dgamma, dbeta); then btvae_bn_relu_bwd_apply. */
int btvae_linear_dgrad_gstats(const float* dY, const float* W, const float* R, const float* Yb, const float* mean,
                              const float* invstd, float* dX, float* gpart, int M, int N, int K, void* stream);
int btvae_linear_dgrad_gstats_pw(const float* dY, const float* W, const void* Wplanes, const float* R, const float* Yb, const float* mean,
                                 const float* invstd, float* dX, float* gpart, int M, int N, int K, void* stream);
int btvae_bn_bwd_sums_from_partials(const float* partials, int nchunks, int N, float* dgamma, float* dbeta, float* sums,
                                    int accumulate, void* stream);
int btvae_colsum_partials(const float* partials, int nchunks, int N, float* out, float* cur, int accumulate, void* stream);
int btvae_bn_bwd_sums_from_products(const float* col_wp, const float* col_wb, const float* mean, const float* invstd,
                                    float* dgamma, float* dbeta, float* sums, int N, int accumulate, void* stream);

/* ---- reparameterisation, KL, likelihoods ---- */
/* z = mu + sigma*eps (src/submodels.py:108-109) written next to the broadcast condition as decoder input rows
 * zc[S*R, L+C] (src/vae_models.py:264,274), and klrow[R] = sum_l KL(N(mu,sigma)||N(0,1)) (src/utils.py:29-33,
 * src/vae_models.py:66-68).  Posterior row r, draw s <-> row s*R+r.  Condition of row r starts at
 * cond + (r/cdiv)*cs_hi + (r%cdiv)*cs_lo.  eps NULL: Philox(seed, offset) draws, stored to eps_out if given. */
int btvae_reparam_kl_fwd(const float* mu, const float* sigma, const float* eps, float* eps_out, const float* cond,
                         float* zc, float* klrow, int R, int S, int L, int C, int cdiv, int cs_hi, int cs_lo,
                         unsigned long long seed, unsigned long long offset, void* stream);
/* Gradients wrt the encoder's raw head outputs from dzc and the KL term (weight kl_weight = beta / R). */
int btvae_reparam_kl_bwd(const float* dzc, const float* eps, const float* mu, const float* sigma, const float* dfac_mu,
                         const float* dfac_sigma, float* draw_mu, float* draw_sigma, int R, int S, int L, int C,
                         float kl_weight, void* stream);
/* logp[row] = sum_d log N(x[row % B][d]; mu, sigma) (src/submodels.py:114-115, src/utils.py:21-23); optional
 * gradients wrt raw head outputs scaled by g (g_rows[row] or g_const). */
int btvae_diag_gauss_loglik(const float* x, const float* mu, const float* sigma, float* logp, const float* dfac_mu,
                            const float* dfac_sigma, float* draw_mu, float* draw_sigma, const float* g_rows, float g_const,
                            int rows, int B, int D, void* stream);
/* Uhat = U/||U[:,v]|| (src/utils.py:18-19), UU[p][v] = Uhat[d,v]*Uhat[e,v] for packed p=d(d+1)/2+e, diag[p] = m^2
 * on diagonal entries.  D <= 128 here and in the three entry points below: up to 64 the per-row kernels hold the covariance
 * in registers, one wave (or lane) per row; beyond that one workgroup per row works on it in LDS, far slower per row (the
 * reference, src/submodels.py:169-179, has no limit; its shipped configurations stop at 48). */
int btvae_dict_prepare(const float* U, float* Uhat, float* norms, float* UU, float* diag, int D, int V,
                       float marginal_std_lim, void* stream);
/* dU (+)= gradient of the dictionary from M1[p][v] = sum_rows Pg[row][p]*sigma^2[row][v]. */
int btvae_dict_grad(const float* M1, const float* Uhat, const float* norms, float* dU, int D, int V, int accumulate, void* stream);
/* log N(x[row % B]; mu[row], Sigma[row]) from packed covariances (MultivariateNormal.log_prob, src/submodels.py:169-171). */
int btvae_mvn_rows_fwd(const float* Sp, const float* mu, const float* x, float* logp, int rows, int B, int D, void* stream);
/* Same plus backward seeds: dmu_raw = g*Sigma^-1(x-mu)*dfac_mu, Pg = g*(alpha alpha^T - Sigma^-1) packed with
 * off-diagonals doubled. */
int btvae_mvn_rows_fwd_bwd(const float* Sp, const float* mu, const float* x, const float* dfac_mu, const float* g_rows,
                           float g_const, float* logp, float* dmu_raw, float* Pg, int rows, int B, int D, void* stream);
/* out4 = [loss, elbo, rll, kl] with rll = mean(logp), kl = mean(klrow), loss = -(rll - beta*kl) (src/vae_models.py:70-74). */
int btvae_elbo_scalars(const float* logp, int nlog, const float* klrow, int nkl, float beta, float* out4, void* stream);
/* Streaming K-sample importance-weighted logsumexp (src/testing_lib.py:10-14): fold kc more samples into the
 * running (max, sum) per profile; finalize writes ll = logsumexp - log K. */
int btvae_iw_lse_update(const float* rll, const float* eps, const float* mu, const float* sigma, float* m_run, float* s_run,
                        int kc, int N, int L, int first, void* stream);
int btvae_iw_lse_finalize(const float* m_run, const float* s_run, float* ll, int N, int K, void* stream);

/* ---- optimiser (src/vae_models.py:103,323-324) ---- */
int btvae_grad_sumsq(const float* g, long long n, double* partials512, void* stream);
int btvae_clip_adam_step(float* p, const float* g, float* m, float* v, long long n, const double* partials512,
                         float max_norm, float lr, float beta1, float beta2, float eps, float weight_decay,
                         long long step, float* norm_out, void* stream);
/* The same step with its scalars in device memory, so that a launch replayed from a HIP graph follows the learning rate
 * and the step count: `step` is an int64 counter on the device, hyper = [lr, beta1, beta2, eps, weight_decay, max_norm
 * (<= 0: no clipping), bc1, sqrt(bc2)].  btvae_adam_tick adds one to the counter and rewrites bc1 = 1 - beta1^step and
 * sqrt(bc2) = sqrt(1 - beta2^step) in `rows` consecutive hyper blocks; the caller writes hyper[0..5] of each. */
int btvae_adam_tick(long long* step, float* hyper, int rows, void* stream);
int btvae_clip_adam_step_dev(float* p, const float* g, float* m, float* v, long long n, const double* partials512,
                             const float* hyper, float* norm_out, void* stream);
int btvae_scale(float* x, long long n, float a, void* stream);

/* ---- conditional imputation (src/imputation_lib.py:16-57,92,230-231) ----
 * btvae_cond_mvn_sample: D <= 64, bit d of obs_mask = coordinate d observed.  btvae_cond_mvn_sample_wide: D <= 128, the
 * observed set as a host array of two 64-bit words (bit d % 64 of word d / 64); up to 64 it forwards to the first. */
int btvae_cond_mvn_sample(const float* Sp, const float* mu, const float* x, const float* eps, unsigned long long obs_mask,
                          const int* un_pos, float* samples, float* cond_mean, float* cond_cov, int rows, int D, int U,
                          void* stream);
int btvae_cond_mvn_sample_wide(const float* Sp, const float* mu, const float* x, const float* eps, const unsigned long long* obs_mask2,
                               const int* un_pos, float* samples, float* cond_mean, float* cond_cov, int rows, int D, int U,
                               void* stream);
int btvae_masked_scatter(float* dst, const long long* idx, const float* src, long long n, void* stream);
int btvae_lse_over_k(const float* logp, float* out, int K, int B, void* stream);
int btvae_pack_lower(const float* full, float* packed, long long rows, int D, void* stream);

/* Streaming accumulation of the l1 / l2 sample errors of scripts/collect_testing_results.py:104-113 over chunks of
 * draws xs[kc,N,D] against x[N,D]: e1[n] (+)= sum |xs-x|, e2[n] (+)= sum (xs-x)^2 (first != 0 overwrites). */
int btvae_sample_error_update(const float* xs, const float* x, float* e1, float* e2, int kc, int N, int D, int first,
                              void* stream);

/* ---- condition feed ---- */
/* Dirichlet draws of DirichletTransformer.transform(style "sample") (src/utils.py:112-129: gamma(alpha,1) draws
 * normalised per row) made on the device and written into the [B, S, ldo] condition tensor at column col0:
 * out[(b*S+s)*ldo + col0 + t], alpha = gamma[rows[b]*T + t] (rows NULL: b).  Philox(seed, offset..offset+63). */
int btvae_dirichlet_sample(const float* gamma, const long long* rows, float* out, int B, int S, int T, int ldo,
                           int col0, unsigned long long seed, unsigned long long offset, void* stream);

#ifdef __cplusplus
}
#endif
#endif
