/* libneurst_hip.so - C ABI of the MI355X (gfx950) kernels behind the neurst-ST
 * TransformerTPE train/infer hot path.
 *
 * The reference (dqqcasia/neurst) has NO plugin / custom-op / FFI interface on this
 * path: every op is a TensorFlow-1.15 library call made from Python.  Each entry point
 * below therefore cites the reference Python function (file:line under the reference
 * root) whose arithmetic it replaces; the Python classes in neurst_amd/st/ that keep
 * the reference's names are the binding a maintainer swaps in.
 *
 * Conventions
 *  - every function returns 0 on success or a negative NEURST_E_* code; a thread-local
 *    message is available from neurst_last_error();
 *  - all pointers are DEVICE pointers owned by the caller (no allocation inside), all
 *    calls are asynchronous on `stream` (a hipStream_t passed as void*);
 *  - tensors are dense row-major; `dtype` is NEURST_F32 or NEURST_BF16 for activations
 *    (bf16 operands always accumulate in fp32); lengths / ids are int32 on device;
 *  - dropout uses the counter hash of neurst_amd/csrc/common.h keyed by a 32-bit
 *    call-site `key`; rate == 0 disables it.  Backward kernels regenerate the mask.
 */
#ifndef NEURST_HIP_H_
#define NEURST_HIP_H_
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { NEURST_F32 = 0, NEURST_BF16 = 1 };
enum { NEURST_OK = 0, NEURST_E_INVALID = -1, NEURST_E_LAUNCH = -2, NEURST_E_UNSUPPORTED = -3 };
enum { NEURST_MASK_NONE = 0, NEURST_MASK_KEYPAD = 1, NEURST_MASK_CAUSAL = 2 };

/* Version of this interface.  It changes whenever an existing entry point changes its argument list or the layout of a
 * buffer it reads or writes (version 2: neurst_conv2d_ln_relu_bwd gained col_ws / col_ws_bytes and attention's keep_bits
 * became a u32 row bitmap; version 3: the neurst_comm_* entry points and the bf16 / deferred-scale arguments of this round).
 * neurst_abi_version() returns the value the library was built with: a host checks it against the header it compiled with
 * (neurst_amd/lib.py does at load time and refuses a mismatch). */
#define NEURST_ABI_VERSION 3

const char* neurst_last_error(void);
int neurst_abi_version(void);

/* Kernel-selection switches for tests and A/B timing (no reference counterpart; defaults come from the environment
 * variables of the same upper-case name with the NEURST_ prefix): "gemm256" 1/0 = large bf16 GEMMs may use the
 * 256 x 256-tile kernel; "gemm256_min_tiles" = smallest number of 256 x 256 tiles for which it is chosen (128);
 * "gemm256_bn" 0 / 192 / 256 = column width of its tiles (0: chosen per problem); "gemm256_max_grid" n >= 8 = its persistent
 * launches use at most n workgroups (a launch beside a latency-bound chain on another stream leaves that chain CUs), 0 = all; "mha_fused" 0 / 1 / 2 = the wide
 * attention forward and the one-kernel backward of attention2.hip never / for encoder-size shapes / whenever they can
 * take the shape (bf16, dh = 96 or 64, no causal mask); "mha_small_fused" 1/0 = dQ and dK / dV of short-query shapes
 * (Tq <= 256) in one launch; "mha_bwd_waves" 4 / 8 = waves per workgroup of the one-kernel backward; "mha_split" bits (default 3):
 * with fewer (batch, head) pairs than CUs, 1 = the wide forward hands a pair's query groups to several workgroups, 2 = the
 * one-kernel backward its key blocks (dQ summed by a second launch); "conv_fast" 1/0 =
 * bf16 fast paths of the conv front end; "ln_prefetch" bits (default 53): 16 / 32 = LayerNorm forward / backward of bf16 rows of
 * 768, 512 or 256 columns without a row mask through the straight-line kernels, 1 / 2 = the general forward / backward row
 * loops request the next row ahead of the current row's stores, 4 / 8 = they take 8-byte lanes where a row is whole wave
 * passes of them.  Unknown names: NEURST_E_INVALID. */
int neurst_set_option(const char* name, int value);

/* a1: splice(left=0,right=num_context) + down_sample(rate)
 * st/tools/tftools/tfAudioTools.py:69-98, st/tools/tftools/tfRecord.py:827-845.
 * x [B,Traw,F] fp32 -> out [B,ceil(Traw/rate),F*(num_context+1)] (out_dtype);
 * len_out[b] = ceil(len_raw[b]/rate). Frames at or beyond len_raw[b] read as zero. */
int neurst_splice_downsample(const float* x, const int32_t* len_raw, void* out, int32_t* len_out,
                             int64_t B, int64_t Traw, int64_t F, int num_context, int rate,
                             int out_dtype, void* stream);

/* a3: SpecAugment, st/tools/tftools/tfAudioTools.py:134-212 (2 freq + 2 time masks).
 * _params samples (f,f0,f',f0',t,t0,t',t0') per utterance with the build's hash RNG
 * (TF's stream is not reproducible); _apply zeroes the bands in place. x [B,T,v] fp32. */
int neurst_specaug_params(const float* x, int32_t* params, int64_t B, int64_t T, int64_t v,
                          int F, int Tmax, float p, uint32_t key, void* stream);
int neurst_specaug_apply(float* x, const int32_t* params, int64_t B, int64_t T, int64_t v, void* stream);

/* a5 + dropout: y = dropout(x + signal[t,c]); st/layers/common_attention_v1.py:28-70,
 * st/models/encoders/transformer_encoder.py:110-115. signal [T,C] fp32 (host-built table). */
int neurst_add_timing_dropout_fwd(const void* x, int x_dtype, const float* signal, void* y, int y_dtype,
                                  int64_t B, int64_t T, int64_t C, float rate, uint32_t key, void* stream);
/* backward of any elementwise dropout: dx = dy * mask * 1/(1-rate) (flat index). */
int neurst_dropout_bwd(const void* dy, void* dx, int64_t n, int dtype, float rate, uint32_t key, void* stream);

/* a8/a9/a10/a16 GEMMs: C[M,N] = op(A)[M,K] . op(B)[K,N] (+bias[N]) (relu) (* (relu_src>0)) (+C)
 * replaces `dense` (st/models/tools/utils.py:367-398), tf.layers.dense
 * (transformer_encoder.py:117-122, fc_decoder.py:33-38, transformer_decoder.py:42-46)
 * and their autodiff gradients.  trans_a: A stored [K,M]; trans_b: B stored [N,K].
 * A,B have dtype `ab_dtype`; C has `c_dtype`; bias fp32; relu_src (dtype ab_dtype, ldc)
 * gates the result by (relu_src > 0) (ReLU backward).  split_k > 1 requires fp32 C and
 * accumulate (atomic adds into C). */
int neurst_gemm_bias_act(const void* A, const void* B, void* C, const float* bias, const void* relu_src,
                         int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldb, int64_t ldc,
                         int trans_a, int trans_b, int ab_dtype, int c_dtype, int relu, int accumulate,
                         int split_k, void* stream);

/* a8/a9 backward: weight gradient AND bias gradient of one `dense` layer (autodiff of
 * st/models/tools/utils.py:367-398: dW = dY^T.X, db = reduce_sum(dY, rows)) in one launch.
 * C[M,N] += A^T . B with A stored [K,M], B stored [K,N], fp32 C (split_k > 1: atomic adds);
 * sum_operand 1: opsum[M] += sum_k A[k][m] (A is dY: '[out,in]' kernels);
 * sum_operand 2: opsum[N] += sum_k B[k][n] (B is dY: '[in,out]' kernels).  opsum fp32, accumulated. */
int neurst_gemm_wgrad_bias(const void* A, const void* B, float* C, float* opsum, int sum_operand, int64_t M, int64_t N,
                           int64_t K, int64_t lda, int64_t ldb, int64_t ldc, int ab_dtype, int split_k, void* stream);
/* The same for up to 8 layers in ONE launch (host arrays of n pointers / sizes): problem i is C_i[M_i,N_i] += A_i^T . B_i
 * with A_i [K_i,M_i] = dY, B_i [K_i,N_i] = X and opsum_i (may be NULL) += column sums of A_i.  bf16 operands run as whole-K
 * 256 x 192 tiles, one workgroup each (no split-K, no atomics on C): a single encoder-size gradient has 12-48 tiles, so
 * the caller packs gradients until they fill the 256 CUs (st/models/graph.py).  Other cases: one launch per problem. */
int neurst_gemm_wgrad_bias_grouped(int n, const void* const* A, const void* const* B, float* const* C, float* const* opsum,
                                   const int64_t* M, const int64_t* N, const int64_t* K, const int64_t* lda,
                                   const int64_t* ldb, const int64_t* ldc, int ab_dtype, void* stream);
/* The grouped launch as a QUEUE of output tiles (same reference functions as above: the weight / bias gradients of
 * `dense`, st/models/tools/utils.py:367-398): only tiles tile_begin[i] .. tile_end[i] - 1 of problem i are computed (a
 * problem has neurst_gemm_wgrad_group_tiles(M_i, N_i) tiles of 256 x 192, independent of each other), so a caller can cut
 * the tiles of many layers into launches of exactly 256 - one workgroup per CU - with a layer straddling two launches
 * (st/models/graph.py).  Up to neurst_gemm_wgrad_group_max() problems.  Whether the tile kernel takes a problem is decided
 * PER PROBLEM from its own operands (bf16, 16-byte aligned rows, M % 8 == N % 8 == 0, 32-bit window), never from what
 * else is queued in the call: eligible problems share one grouped launch, and a problem the tile kernel cannot take is
 * computed whole by the call whose range starts at tile 0 - so the calls that hold the parts of a straddling problem agree
 * as long as the `gemm256` option is not changed between them. */
int neurst_gemm_wgrad_bias_grouped_tiles(int n, const void* const* A, const void* const* B, float* const* C,
                                         float* const* opsum, const int64_t* M, const int64_t* N, const int64_t* K,
                                         const int64_t* lda, const int64_t* ldb, const int64_t* ldc, const int* tile_begin,
                                         const int* tile_end, int ab_dtype, void* stream);
int64_t neurst_gemm_wgrad_group_tiles(int64_t M, int64_t N);
int neurst_gemm_wgrad_group_max(void);
/* column sums (bias gradients): out[n] (+)= sum_m x[m,n]; out fp32. */
int neurst_colsum(const void* x, float* out, int64_t M, int64_t N, int64_t ldx, int dtype, int accumulate,
                  void* stream);

/* a8: fused multi-head attention, st/models/tools/utils.py:545-553 +
 * st/layers/common_attention_v1.py:180-204,244-266,325-365.  q [B,Tq,*], k/v [B,Tk,*]
 * with row strides ldq/ldk/ldv (so packed qkv / kv projections are consumed in place),
 * head h occupies columns [h*dh,(h+1)*dh).  softmax((q*dh^-0.5).k^T + bias) with the
 * reference's additive -1e9 bias (KEYPAD: keys >= len_k[b]; CAUSAL: key > query),
 * dropout on the weights, .v, heads merged into o [B,Tq,ldo].  lse [B,H,Tq] fp32 is
 * saved for backward.  The [B,H,Tq,Tk] matrix is never materialised. */
int neurst_mha_fwd(const void* q, const void* k, const void* v, void* o, float* lse,
                   int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo,
                   int64_t B, int64_t H, int64_t Tq, int64_t Tk, int64_t dh,
                   int mask_mode, const int32_t* len_k, float rate, uint32_t key, int dtype, void* stream);
/* Same with explicit batch strides (elements) for q/k/v/o, used by the cached greedy decoder
 * (a21) where K/V live in a [B,Lmax,2d] cache of which only the first Tk rows are valid. */
int neurst_mha_fwd_strided(const void* q, const void* k, const void* v, void* o, float* lse,
                           int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo,
                           int64_t bsq, int64_t bsk, int64_t bsv, int64_t bso,
                           int64_t B, int64_t H, int64_t Tq, int64_t Tk, int64_t dh,
                           int mask_mode, const int32_t* len_k, float rate, uint32_t key, int dtype, void* stream);
/* delta [B,H,Tq] fp32 workspace; dq/dk/dv written with strides lddq/lddk/lddv. */
int neurst_mha_bwd(const void* q, const void* k, const void* v, const void* o, const void* d_o,
                   const float* lse, float* delta, void* dq, void* dk, void* dv,
                   int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo,
                   int64_t lddq, int64_t lddk, int64_t lddv,
                   int64_t B, int64_t H, int64_t Tq, int64_t Tk, int64_t dh,
                   int mask_mode, const int32_t* len_k, float rate, uint32_t key, int dtype, void* stream);

/* Training variants of the two calls above that carry the attention-dropout decisions from forward to backward:
 * keep_bits (neurst_mha_keep_bits_bytes(B,H,Tq,Tk) bytes, 4-byte aligned, may be NULL = hash again in backward) receives
 * the keep decisions as a row-major bitmap, uint32 [B*H][Tq][ceil(Tk/32)]: bit (k & 31) of word (k >> 5) of row
 * (b*H + h)*Tq + q is 1 when weight (q, k) is kept (words of tiles the forward skips as fully masked stay unwritten);
 * the backward kernels read it back instead of re-hashing.  Same results bit for bit
 * (st/layers/common_attention_v1.py:357-359: TF keeps the mask tensor). */
int64_t neurst_mha_keep_bits_bytes(int64_t B, int64_t H, int64_t Tq, int64_t Tk);
int neurst_mha_fwd_train(const void* q, const void* k, const void* v, void* o, float* lse, uint16_t* keep_bits,
                         int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo,
                         int64_t B, int64_t H, int64_t Tq, int64_t Tk, int64_t dh,
                         int mask_mode, const int32_t* len_k, float rate, uint32_t key, int dtype, void* stream);
/* The same decisions ahead of the forward: neurst_mha_keep_bits fills keep_bits from (key, shape) alone - it can run on
 * another stream while the projection in front of the attention still works - and neurst_mha_fwd_pregen is
 * neurst_mha_fwd_train reading them instead of hashing (bit-identical outputs). */
int neurst_mha_keep_bits(uint16_t* keep_bits, int64_t B, int64_t H, int64_t Tq, int64_t Tk, float rate, uint32_t key, void* stream);
int neurst_mha_fwd_pregen(const void* q, const void* k, const void* v, void* o, float* lse, const uint16_t* keep_bits,
                          int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo,
                          int64_t B, int64_t H, int64_t Tq, int64_t Tk, int64_t dh,
                          int mask_mode, const int32_t* len_k, float rate, uint32_t key, int dtype, void* stream);
int neurst_mha_bwd_train(const void* q, const void* k, const void* v, const void* o, const void* d_o,
                         const float* lse, const uint16_t* keep_bits, float* delta, void* dq, void* dk, void* dv,
                         int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo,
                         int64_t lddq, int64_t lddk, int64_t lddv,
                         int64_t B, int64_t H, int64_t Tq, int64_t Tk, int64_t dh,
                         int mask_mode, const int32_t* len_k, float rate, uint32_t key, int dtype, void* stream);

/* neurst_mha_bwd_train with a caller-provided workspace (neurst_mha_bwd_workspace_bytes(B,H,Tq,Tk,dh) bytes, 16-byte
 * aligned; 0 bytes are needed when Tk <= 256).  With it, encoder-size bf16 problems (dh = 96, no causal mask, 16-byte aligned
 * rows, stored keep_bits when rate > 0) run ONE fused kernel - five MFMA products per tile instead of the seven of the
 * dq + dkv kernel pair, dQ summed over key blocks inside one workgroup through the workspace (no atomics, deterministic).
 * Every other problem, or a NULL / too small workspace, takes the two-kernel path: same results up to bf16 rounding.
 * Same reference arithmetic as neurst_mha_bwd (common_attention_v1.py:180-204,325-365 differentiated). */
int64_t neurst_mha_bwd_workspace_bytes(int64_t B, int64_t H, int64_t Tq, int64_t Tk, int64_t dh);
int neurst_mha_bwd_train_ws(const void* q, const void* k, const void* v, const void* o, const void* d_o,
                            const float* lse, const uint16_t* keep_bits, float* delta, void* dq, void* dk, void* dv,
                            int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo,
                            int64_t lddq, int64_t lddk, int64_t lddv,
                            int64_t B, int64_t H, int64_t Tq, int64_t Tk, int64_t dh,
                            int mask_mode, const int32_t* len_k, float rate, uint32_t key, int dtype,
                            void* workspace, int64_t workspace_bytes, void* stream);

/* a7 (+a6): out = LN(x + dropout(y)) * (t < len[b]) ; st/models/tools/utils.py:324-337,
 * st/layers/common_layers_v1.py:342-347 (eps 1e-6 inside rsqrt, biased variance).
 * z = x + dropout(y) is written over y (in place) for backward; mean/rstd [rows] fp32.
 * lens (may be NULL) zeroes rows t >= lens[b] (transformer_encoder.py:151), T = rows per b. */
int neurst_dropout_add_layernorm_fwd(const void* x, void* y_inout_z, const float* gamma, const float* beta,
                                     void* out, float* mean, float* rstd, int64_t rows, int64_t C,
                                     float rate, uint32_t key, const int32_t* lens, int64_t T,
                                     int dtype, void* stream);
/* dz (gradient wrt x, and wrt y when rate==0) and dy = dz*mask*scale (dy may be NULL or == dz
 * when rate == 0); dgamma/dbeta accumulate (+=) in fp32; partial: NULL = workgroups add their column partials to dgamma/dbeta with fp32 atomics (fast, summation order
 * varies); or a [2*nblk*C] fp32 workspace = deterministic two-pass reduction
 * where nblk = neurst_layernorm_bwd_blocks(rows).  dbias (may be NULL): += column sums of dy (fp32) - the gradient of the
 * bias of the dense layer that produced y, so that layer's weight-gradient GEMM can skip its operand sum. */
int64_t neurst_layernorm_bwd_blocks(int64_t rows);
int neurst_dropout_add_layernorm_bwd(const void* dout, const void* z, const float* gamma,
                                     const float* mean, const float* rstd, void* dz, void* dy,
                                     float* dgamma, float* dbeta, float* partial, float* dbias,
                                     int64_t rows, int64_t C, float rate, uint32_t key,
                                     const int32_t* lens, int64_t T, int dtype, void* stream);

/* Allocation-free variant of neurst_dropout_add_layernorm_bwd.  NO counterpart in the reference: it exists for callers
 * that cannot allocate device memory - code under HIP graph capture, tight serving loops, harnesses that forbid
 * hipMalloc - and therefore cannot bring the `partial` workspace of the deterministic two-pass reduction.  The block
 * partials live in a 4 MiB (1 Mi floats) array inside the library, one copy per GPU, held until the process ends; the
 * call is rejected (NEURST_E_INVALID) when 2 * neurst_layernorm_bwd_blocks(rows) * C exceeds 1 Mi floats (e.g. C > 2048 at
 * 256 blocks).  Same arguments (minus `partial`) and bit-identical results as the call above with a workspace.  It
 * calls no hipMalloc / hipFree.  All calls on one GPU share the array: calls that overlap on one GPU (two streams) race. */
int neurst_dropout_add_layernorm_bwd_noalloc(const void* dout, const void* z, const float* gamma,
                                             const float* mean, const float* rstd, void* dz, void* dy,
                                             float* dgamma, float* dbeta, float* dbias,
                                             int64_t rows, int64_t C, float rate, uint32_t key,
                                             const int32_t* lens, int64_t T, int dtype, void* stream);

/* gradient accumulation helper (a tensor consumed twice): dst += src, same dtype. */
int neurst_add_inplace(void* dst, const void* src, int64_t n, int dtype, void* stream);

/* a4 lengths: out[b] = ceil(in[b] / div) (transformer_encoder.py:91-92). */
int neurst_lens_ceil_div(const int32_t* in, int32_t* out, int64_t B, int div, void* stream);

/* a6 standalone: x[b,t,:] *= (t < lens[b]) in place. */
int neurst_mask_rows(void* x, const int32_t* lens, int64_t B, int64_t T, int64_t C, int dtype, void* stream);

/* a15: decoder inputs, st/models/decoders/decoder.py:63-92 + tfTools.py:195-198 +
 * TransformerTPE.py:335-338: inp[b,0]=sos, inp[b,l]=labels[b,l-1], zeroed at l >= len[b]. */
int neurst_build_decoder_input(const int32_t* labels, const int32_t* len_labels, int32_t* inp,
                               int64_t B, int64_t L, int32_t sos, void* stream);
/* a16 embedding: out = dropout(2*table[ids] + signal[l]) (decoder.py:198-204,
 * transformer_decoder.py:193-197); pos0 offsets the signal row (cached decoding). table fp32. */
int neurst_embedding_fwd(const int32_t* ids, const float* table, const float* signal, void* out,
                         int64_t B, int64_t L, int64_t D, int64_t V, float rate, uint32_t key,
                         int out_dtype, void* stream);
/* dtable[ids] += 2 * dropout_bwd(dout) (fp32 atomics; dtable pre-initialised by the caller). */
int neurst_embedding_bwd(const int32_t* ids, const void* dout, float* dtable, int64_t B, int64_t L,
                         int64_t D, int64_t V, float rate, uint32_t key, int dtype, void* stream);

/* a11: tf.nn.ctc_loss(ctc_merge_repeated=True, ignore_longer_outputs_than_inputs=True,
 * time_major=False) forward + gradient, call site st/models/TransformerTPE.py:74-104.
 * logits [B,T,V] fp32 (blank = V-1, softmax internal). loss [B] fp32.  grad [B,T,V]
 * (grad_dtype; may alias logits when fp32) = grad_scale * d loss_b / d logits.
 * argmax [B,T] int32 (first maximum) is produced on the way (used by a12/a13).
 * workspace: neurst_ctc_workspace_bytes(B,T,Umax). */
int64_t neurst_ctc_workspace_bytes(int64_t B, int64_t T, int64_t Umax);
int neurst_ctc_loss_fwd_bwd(const float* logits, const int32_t* len_logits, const int32_t* labels,
                            const int32_t* len_labels, float* loss, void* grad, int grad_dtype,
                            int32_t* argmax, void* workspace, int64_t B, int64_t T, int64_t V,
                            int64_t Umax, float grad_scale, void* stream);
/* The same in two parts that may be issued on different streams: phases 1 = per-frame softmax / arg-max / gathered
 * log-probabilities (everything a13's shrink layer waits for), 2 = alpha-beta recursion + sparse gradient correction
 * (needs part 1 complete), 3 = both. */
int neurst_ctc_loss_fwd_bwd_phased(const float* logits, const int32_t* len_logits, const int32_t* labels,
                                   const int32_t* len_labels, float* loss, void* grad, int grad_dtype,
                                   int32_t* argmax, void* workspace, int64_t B, int64_t T, int64_t V,
                                   int64_t Umax, float grad_scale, int phases, void* stream);
/* per-frame argmax only (inference path of a12/a13). */
int neurst_argmax_rows(const float* logits, int32_t* argmax, int64_t rows, int64_t V, void* stream);
/* a12: tf.nn.ctc_greedy_decoder(merge_repeated=True) + sparse_to_dense(default 0),
 * TransformerTPE.py:106-125,251-259.  out [B,T] (zero padded), out_len [B]. */
int neurst_ctc_greedy(const int32_t* argmax, const int32_t* len_logits, int32_t* out, int32_t* out_len,
                      int64_t B, int64_t T, int32_t blank, void* stream);

/* a13: shrink_layer, st/models/tools/blocks.py:8-38, in two steps so the host can size
 * the output like the reference does (max(max count, 1)):
 *  _index: keep frame t iff argmax != blank && argmax != argmax[t-1] && t < lens[b];
 *          index [B,T] = source frame of the j-th kept frame (-1 beyond the count), new_len [B];
 *  _fwd:   out [B,Tout,D] = gathered rows, zero padded (Tout <= T);
 *  _bwd:   dx [B,T,D] = scatter of dout [B,Tout,D] (rows not kept get zero). */
int neurst_shrink_index(const int32_t* argmax, const int32_t* lens, int32_t* index, int32_t* new_len,
                        int64_t B, int64_t T, int32_t blank, void* stream);
int neurst_shrink_fwd(const void* x, const int32_t* index, void* out, int64_t B, int64_t T, int64_t Tout,
                      int64_t D, int dtype, void* stream);
int neurst_shrink_bwd(const void* dout, const int32_t* index, void* dx, int64_t B, int64_t T, int64_t Tout,
                      int64_t D, int dtype, void* stream);
/* a13 without the device -> host read that sizes the reference's output (st/models/tools/blocks.py:26-36: zero-pad to the
 * longest kept sequence): the host allocates Tout = `bound` rows per utterance from earlier steps' maxima; this call
 * clamps new_len[b] to bound, writes state[1] = max_b new_len[b] (before clamping) and state[0] = 1 if that exceeded the
 * bound, else 0.  Any bound >= the true maximum gives the reference's result on every valid position (padding rows are
 * zero and masked by length); on overflow the caller discards the step (neurst_adam_step_guarded) and repeats it. */
int neurst_shrink_clamp(int32_t* new_len, int64_t B, int bound, int32_t* state, void* stream);

/* a17: smoothing_cross_entropy + mask + per-utterance mean, st/models/model_tools.py:17-38,
 * st/models/seq2seqModel.py:157-191.  logits [B,L,V] fp32; xent [B,L] fp32 (masked, divided
 * by len); loss_b = sum_l xent[b,l].  grad [B,L,V] (grad_dtype) = grad_scale * d loss_b/d logits. */
int neurst_ls_xent_fwd_bwd(const float* logits, const int32_t* labels, const int32_t* len_labels,
                           float* xent, void* grad, int grad_dtype, int64_t B, int64_t L, int64_t V,
                           float confidence, float grad_scale, void* stream);

/* a18: tf.losses.mean_squared_error (st/models/kd_layer.py:9-11): partial sums of
 * (pred-label)^2 into partial[nblk] (nblk = neurst_mse_blocks(n)); dpred = grad_scale*2*(pred-label)/n. */
int64_t neurst_mse_blocks(int64_t n);
int neurst_mse_fwd_bwd(const void* pred, const void* label, int label_dtype, float* partial, void* dpred,
                       int64_t n, float grad_scale, int dtype, void* stream);

/* a2 loss mix (TransformerTPE.py:385-418, :471): out[0]=mean_b(w_ctc*ctc_b + w_kd*mse + w_ast*ce_b),
 * out[1]=mean ce_b, out[2]=mean ctc_b, out[3]=mse.  Any input may be NULL (treated as 0). */
int neurst_loss_combine(const float* xent, const float* ctc, const float* mse_partial, int64_t n_partial,
                        int64_t mse_numel, int64_t B, int64_t L, float w_ast, float w_ctc, float w_kd,
                        float* out4, void* stream);

/* a4: tf.layers.conv2d(SAME) -> layer_norm over channels -> ReLU
 * (st/models/encoders/transformer_encoder.py:18-32,80-93) and the 3x3/1 SAME max-pool (:96-100).
 * x [B,H,W,Cin] NHWC, w [kh,kw,Cin,Cout] fp32, out [B,Ho,Wo,Cout]; pre-LN conv output is
 * kept in `conv_out` (same shape) for backward. */
int neurst_conv2d_ln_relu_fwd(const void* x, const float* w, const float* bias, const float* gamma,
                              const float* beta, void* conv_out, void* out, float* mean, float* rstd,
                              int64_t B, int64_t H, int64_t W, int64_t Cin, int64_t Cout,
                              int kh, int kw, int sh, int sw, int dtype, void* stream);
/* dgamma/dbeta/dw/dbias accumulate (+=, fp32 atomics); dx may be NULL (first layer).  dconv_ws: workspace of conv_out's
 * shape and type.  col_ws (may be NULL: the general weight-gradient kernel then runs, much slower): workspace of
 * neurst_conv2d_bwd_workspace_bytes() bytes - the im2col image [pixels][kh kw Cin] that turns the weight gradient of a
 * bf16 3x3 layer with Cin % 8 == 0 into one split-K GEMM over the pixels (0 bytes for other shapes and for fp32).
 * bf16 layers take implicit-GEMM MFMA kernels (3x3, Cin and Cout in {32, 64}) or a registers-resident small-K kernel
 * (kh kw Cin <= 32, the first layer); neurst_set_option("conv_fast", 0) forces the general kernels. */
int64_t neurst_conv2d_bwd_workspace_bytes(int64_t B, int64_t H, int64_t W, int64_t Cin, int64_t Cout, int kh, int kw,
                                          int sh, int sw, int dtype);
int neurst_conv2d_ln_relu_bwd(const void* dout, const void* x, const float* w, const float* gamma,
                              const void* conv_out, const void* out, const float* mean, const float* rstd,
                              void* dconv_ws, void* col_ws, int64_t col_ws_bytes, void* dx, float* dw, float* dbias,
                              float* dgamma, float* dbeta,
                              int64_t B, int64_t H, int64_t W, int64_t Cin, int64_t Cout,
                              int kh, int kw, int sh, int sw, int dtype, void* stream);
int neurst_maxpool3x3_fwd(const void* x, void* out, int8_t* argidx, int64_t B, int64_t H, int64_t W, int64_t C,
                          int dtype, void* stream);
int neurst_maxpool3x3_bwd(const void* dout, const int8_t* argidx, void* dx, int64_t B, int64_t H, int64_t W,
                          int64_t C, int dtype, void* stream);

/* a20: tf.train.AdamOptimizer(beta1=.9,beta2=.98,eps=1e-9) update on a flat arena
 * (st/models/lstmModel.py:526-530): g is scaled by grad_scale first (1/world_size after a
 * summed all-reduce, a19); lr_t = lr*sqrt(1-b2^t)/(1-b1^t) is computed by the caller.
 * shadow (may be NULL) receives the bf16 copy of the updated parameters. */
int neurst_adam_step(float* p, const float* g, float* m, float* v, void* shadow_bf16, int64_t n,
                     float lr_t, float beta1, float beta2, float eps, float grad_scale, void* stream);
/* The same update with a device-side veto (no reference counterpart): when skip_flag is not NULL and *skip_flag != 0 at
 * execution time the launch changes nothing.  Used with neurst_shrink_clamp: a step whose shrunk tensor did not fit the
 * length the host had guessed must not reach the parameters (the host repeats that batch). */
int neurst_adam_step_guarded(float* p, const float* g, float* m, float* v, void* shadow_bf16, int64_t n,
                             float lr_t, float beta1, float beta2, float eps, float grad_scale, const int32_t* skip_flag,
                             void* stream);
/* handle_gradients (st/tools/tftools/gradientTools.py:40-67), the two rules that reach this model's variables (the value and
 * per-variable norm clips only touch variables named *lstm*): L2 penalty g += lambda * v on non-bias variables = axpy on
 * the variable's slice of the gradient arena; tf.clip_by_global_norm needs sqrt(sum g^2) over all gradients: block partials
 * (fp64, neurst_sumsq_blocks(n) of them) that the caller sums; the clip factor then rides on Adam's grad_scale. */
int neurst_axpy_f32(float* y, const float* x, float alpha, int64_t n, void* stream);
int64_t neurst_sumsq_blocks(int64_t n);
int neurst_sumsq_f32(const float* x, int64_t n, void* partial_f64, void* stream);
/* dtype conversion fp32 -> bf16 (parameter shadow after loading weights). */
int neurst_cast_f32_to_bf16(const float* src, void* dst, int64_t n, void* stream);
int neurst_cast_bf16_to_f32(const void* src, float* dst, int64_t n, void* stream);

/* a19, the exchange itself, for hosts without torch.distributed: average_gradients of
 * st/tools/tftools/gradientTools.py:7-37 (the towers' gradients summed per variable; the 1/N of the mean rides on
 * neurst_adam_step's grad_scale) as in-place sums of buckets of the gradient arena over RCCL / xGMI, one process and one
 * communicator per GPU.  RCCL is loaded at run time on the first call (NEURST_RCCL_LIB, else librccl.so.1 / librccl.so);
 * NEURST_E_UNSUPPORTED when it cannot be.  Rank 0 calls neurst_comm_unique_id (neurst_comm_unique_id_bytes() = 128 bytes)
 * and hands the bytes to the other ranks by its own means; every rank then calls neurst_comm_init with its device current.
 * neurst_allreduce_bucket is asynchronous on `stream` (dtype NEURST_F32, or NEURST_BF16 for buckets cast with
 * neurst_cast_f32_to_bf16); calls of all ranks must follow the same order.  The Python host of this repository
 * (GradBucketer) drives the same collective through torch.distributed instead. */
int neurst_comm_unique_id_bytes(void);
int neurst_comm_unique_id(void* id_out);
int neurst_comm_init(int rank, int world, const void* unique_id);
int neurst_comm_world(void);
int neurst_allreduce_bucket(void* ptr, int64_t count, int dtype, void* stream);
int neurst_comm_destroy(void);

/* a21 helper: greedy step bookkeeping of decoder_with_caching
 * (st/models/decoders/transformer_decoder.py:83-90): ids = argmax(logits[b,:]) (first max),
 * preds[b,step] = ids, finished |= ids==eos, len_decoded += 1-finished, n_unfinished += !finished. */
int neurst_greedy_step(const float* logits, int32_t* preds, int32_t* finished, int32_t* len_decoded,
                       int32_t* n_unfinished, int64_t B, int64_t V, int64_t step, int64_t max_steps,
                       int32_t eos, void* stream);

/* Beam search bookkeeping of Transformer_Decoder.beam_decode_rerank
 * (st/models/decoders/transformer_decoder.py:248-431), one decoding step without a read-back.
 * neurst_beam_topk (:303-341): scores[r, 0:k] / ids[r, 0:k] = the k largest entries of log_softmax(logits[r, :]) in
 *   descending order, equal values by lower index (tf.nn.top_k); logits [rows, V] fp32, 1 <= k <= 16.
 * neurst_beam_prune (:316-373): for each utterance b with `beam` live hypotheses (rows b*beam .. b*beam+beam-1):
 *   a finished hypothesis keeps one branch that emits `eos` with branch score 0, its other branches get -1e10 (the
 *   reference's finite `inf`, :14); candidate score = (scores[j] + branch score) / ((5 + len) / 6)^length_penalty_weight
 *   with len = len_decoded[j] + (token != eos); the `beam` best of the beam^2 candidates (ties by lower candidate index)
 *   give new_scores / new_len / new_tokens / parents (row index of the hypothesis they extend) / new_finished
 *   (= finished[slot] | token == eos: the reference or-s the flags by slot without gathering them by parent, :368-369);
 *   *n_unfinished += number of unfinished survivors.  prev_unfinished (may be NULL): the counter the previous step
 *   wrote; when it is 0 the reference's while_loop (:375-383) has already ended, and a step the host enqueued before
 *   reading that back copies the state through unchanged (parents = identity, tokens = eos).  1 <= beam <= 8; outputs
 *   must not alias inputs.
 * neurst_gather_rows (:361-362): dst[r] = src[idx[r]] for rows of row_bytes bytes at the given byte strides (predictions,
 *   caches re-ordered by parent); rows < 65536; not in place. */
int neurst_beam_topk(const float* logits, float* scores, int32_t* ids, int64_t rows, int64_t V, int k, void* stream);
int neurst_beam_prune(const float* next_scores, const int32_t* next_ids, const float* scores, const int32_t* len_decoded,
                      const int32_t* finished, float* new_scores, int32_t* new_len, int32_t* new_finished,
                      int32_t* new_tokens, int32_t* parents, int32_t* n_unfinished, const int32_t* prev_unfinished,
                      int64_t batch, int beam, int32_t eos, float length_penalty_weight, void* stream);
int neurst_gather_rows(const void* src, void* dst, const int32_t* idx, int64_t rows, int64_t row_bytes, int64_t src_stride,
                       int64_t dst_stride, void* stream);

/* Input pipeline, host side (pure host code, callable without a GPU).  Replaces what the reference
 * gets from TensorFlow's tf.TFRecordReader / tf.parse_single_example / tf.python_io.TFRecordWriter
 * (st/tools/tftools/tfRecord.py:107-150 writer, :460-824 readers; `*.recode` files, :848-855).
 * neurst_crc32c: CRC-32C (Castagnoli) of a byte range, unmasked; the container stores rotr(crc, 15) + 0xa282ead8.
 * neurst_tfrecord_scan: walks a whole file image (e.g. a read-only mapping): record r has its payload at
 *   buf + offsets[r], lengths[r] bytes.  Returns the number of records in the file (only the first max_records are
 *   written; call with max_records = 0 to count), or NEURST_E_INVALID for a truncated file or - with verify_crc - a
 *   length or payload whose checksum does not match (the message names the record and byte position).
 * neurst_example_bytes_feature: in a serialized tf.train.Example, the first value of the bytes-list feature `key`
 *   (what FixedLenFeature([], tf.string) + tf.decode_raw read, :639-672): *offset / *size relative to `record`.
 *   NEURST_E_INVALID when the key is absent or the message malformed, NEURST_E_UNSUPPORTED when it is not a bytes list. */
uint32_t neurst_crc32c(const void* data, int64_t nbytes);
int64_t neurst_tfrecord_scan(const void* buf, int64_t nbytes, int64_t* offsets, int64_t* lengths, int64_t max_records,
                             int verify_crc);
int neurst_example_bytes_feature(const void* record, int64_t nbytes, const char* key, int64_t* offset, int64_t* size);

/* Dev-loop metric (host code): Levenshtein distance between two id sequences - what
 * `editdistance.eval` gives batch_cer / batch_wer (st/tools/textTools.py:34-70).  sid (may be NULL) receives the
 * substitutions / insertions / deletions of the alignment that st/tools/wer.py:124-151 picks (smallest total per cell,
 * ties in the order substitution, insertion, deletion).  Returns the distance, or NEURST_E_INVALID. */
int64_t neurst_edit_distance(const int32_t* ref, int64_t nref, const int32_t* hyp, int64_t nhyp, int64_t* sid);

/* kernel self-test: one 16x16xK MFMA tile product through the same fragment loaders the
 * GEMM / attention kernels use (A [16,K], B^T [16,K] -> C [16,16] fp32). */
int neurst_selftest_mma(const void* A, const void* Bt, float* C, int64_t K, int dtype, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* NEURST_HIP_H_ */
