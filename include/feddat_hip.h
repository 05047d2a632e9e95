/* libfeddat_hip.so -- C ABI of the MI355X-native FedDAT hot path (gfx950 only).
 *
 * The reference (HaokunChen245/FedDAT) has no FFI: its boundary for this path is the Python module API
 * (SURVEY.md section 8b).  These entry points are what a binding for that API calls; each one cites the
 * reference code whose arithmetic it replaces (paths relative to the reference repo).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless stated; the library never allocates or frees;
 *   - all launches are asynchronous on `stream`; functions are re-entrant per stream; the only library-side state is the
 *     per-device cache described under "Devices and contexts" (thread-safe) and the optional feddat_ctx handles;
 *   - return value: FEDDAT_OK (0), FEDDAT_EINVAL (bad shape / alignment / null), FEDDAT_ELAUNCH (HIP launch error);
 *   - "bf16" buffers are passed as void* (2 bytes per element); fp32 as float*.  The 16-bit OPERAND FORMAT is a property
 *     of the library build, reported by feddat_operand_format(): libfeddat_hip.so = bfloat16 (FEDDAT_OPERANDS_BF16),
 *     libfeddat_hip_f16.so = IEEE binary16 (FEDDAT_OPERANDS_FP16: the same sources built with -DFEDDAT_OPERANDS_F16, the
 *     same entry points, v_mfma_f32_16x16x32_f16 instead of _bf16 at the same rate).  In the fp16 library every parameter
 *     or entry point named "bf16" carries fp16 values: 10 instead of 7 mantissa bits on every frozen weight and activation
 *     operand -- the reference's own GPU arithmetic (fp16 autocast, src/accelerate_config.yaml:8) -- with the 5-bit
 *     exponent's range: the caller scales the loss gradient by a power of two (as the reference's GradScaler does,
 *     task_trainer.py:302,323 via accelerator.backward) and removes the factor where the weight gradients leave
 *     (feddat_wgrad_seg.scale).  fp32 accumulation, fp32 residual / gradient streams and the split-operand weight
 *     gradients are the same in both;
 *   - matrices are row-major with an explicit leading dimension in ELEMENTS.
 */
#ifndef FEDDAT_HIP_H
#define FEDDAT_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#ifndef __HIP__
typedef struct ihipStream_t* hipStream_t;
#endif

#define FEDDAT_OK 0
#define FEDDAT_EINVAL 1
#define FEDDAT_ELAUNCH 2
#define FEDDAT_ETIMEOUT 3   /* feddat_comm_create_timeout only */

#define FEDDAT_ABI_VERSION 8   /* 8: dynamic loss scale (GradScaler semantics on the device): feddat_wgrad_seg.grad_unscale_dev, feddat_ht_job.alpha_dev, feddat_adamw_group.skip_if / bak / restore_if, feddat_adapter_wgrad_reduce_checked, feddat_dat_loss_fwd_bwd_checked, feddat_dat_step_finish; 7: feddat_operand_format (bf16 / fp16 operand builds of the same ABI); 6: feddat_attn_cls_fwd / _bwd (token-0-only attention of the last layer), fused step tail (feddat_head_gemm, feddat_head_ln_gelu, feddat_head_ln_bwd_full, feddat_dat_loss_fwd_bwd_single, feddat_adamw_multi, feddat_step_tick_multi), feddat_vqa_score_accumulate, feddat_comm_create_timeout (FEDDAT_ETIMEOUT), feddat_set_debug_flags rejects the timing-only ablation bits (production build); 5: 8-bit gelu' epilogues (FEDDAT_EPI_GELU_G8 / _MUL_G8; feddat_vilt_layer_acts.u), feddat_adapter_wgrad_partial / _reduce, wgrad_reduce_now; 4: dropout entry points (feddat_dropout, feddat_attn2_*_dropout), feddat_comm_info; 3: feddat_ctx, feddat_set_debug_flags, feddat_comm_*, feddat_fedavg_allreduce, z_save / z_saved */
int feddat_abi_version(void);
#define FEDDAT_OPERANDS_BF16 0
#define FEDDAT_OPERANDS_FP16 1
int feddat_operand_format(void);   /* which 16-bit format this library's "bf16" operands are (see Conventions) */

/* ---------------------------------------------------------------------------------------------
 * Devices and contexts.  The per-op entry points below are stateless towards the caller: whatever they cache (compute-unit
 * count, per-kernel "max dynamic LDS" attributes) is keyed by the calling thread's CURRENT HIP device and mutex-guarded,
 * so one process may drive several GPUs from several threads.  feddat_ctx makes that explicit for callers that want
 * it: create = switch to `device`, look up its CU count and set every kernel's attributes there (so that no later launch
 * pays for it, e.g. inside a stream capture); destroy frees only the handle.  The composite entry points
 * (feddat_vilt_layer_fwd/bwd) take a ctx.
 * feddat_set_debug_flags: ablation switches used by tools/ only (GEMM: 1 / 2 = everything on the two-wave-group / the
 * one-wave-per-SIMD kernel, 8 = skip epilogue, 32 / 64 = force 192- / 256-row tiles, 256 = K = 32 fp8 MFMA, 512 = deferred-
 * epilogue timing probe, bits 28..31 = cap the persistent grid at 16 x value workgroups; adapters: bits 24..26; attention
 * backward: bits 20..22 = timing-only ablations, bit 23 = one block per (sample, head) instead of the persistent grid).  The one piece
 * of process-wide mutable state in the library: 0 by default, never read from the environment, and no production path sets
 * it -- with flags = 0 every launch is a pure function of its arguments.
 * ------------------------------------------------------------------------------------------- */
typedef struct feddat_ctx feddat_ctx;
int feddat_ctx_create(int device, feddat_ctx** ctx);
int feddat_ctx_destroy(feddat_ctx* ctx);
int feddat_ctx_device(const feddat_ctx* ctx, int* device, int* compute_units);
int feddat_set_debug_flags(int flags);

/* ---------------------------------------------------------------------------------------------
 * K1  bf16 MFMA GEMM  C[M,N] = A[M,K] * B[N,K]^T  (+ fused epilogue)
 * Replaces the frozen nn.Linear calls inside HF ViltLayer (reference call site src/modeling/vilt.py:127;
 * FFN-out dense+residual: src/modeling/adaptered_output.py:74-76) and, with B = W^T, their dX-only
 * backward (autograd through frozen weights, src/train/visionlanguage_tasks/task_trainer.py:302,323).
 * Requirements: K % 64 == 0, lda/ldb % 8 == 0, and N % 192 == 0 with M >= 1024 (the two persistent kernels, 192 x 192 or
 * 256 x 192 tiles chosen per shape: one wave per SIMD with AGPR-pinned accumulators for the plain / residual epilogues, two
 * ping-pong wave groups for the GELU and gelu' epilogues; bit-identical results) or else N % 128 == 0 (128 x 128 kernel),
 * or M < 1024 with N % 64 == 0 (64 x 64 small-tile kernel).  Row strides (lda, ldr, ldo*) are in elements
 * and may exceed the row length (strided operands).
 * ------------------------------------------------------------------------------------------- */
#define FEDDAT_EPI_BF16 0       /* out_bf16 = acc + bias                                    */
#define FEDDAT_EPI_RESID_F32 1  /* out_f32  = acc + bias + resid(fp32)                      */
#define FEDDAT_EPI_GELU 2       /* out2_bf16 = u = acc + bias (optional); out_bf16 = gelu(u) */
#define FEDDAT_EPI_MUL_DGELU 3  /* out_bf16 = (acc) * gelu'(aux_bf16)                        */
#define FEDDAT_EPI_F32 4        /* out_f32  = acc + bias                                    */
/* The GELU pair with the saved tensor at 8 bits instead of 16 (the FFN pair of a ViLT layer moves 25 % fewer bytes in its
 * HBM-bound epilogues).  gelu'(u) is computed from the fp32 u in the FFN1 epilogue and stored as a uint8 code:
 *   code = round((gelu'(u) - FEDDAT_G8_LO) / FEDDAT_G8_STEP),  gelu'(u) ~ FEDDAT_G8_LO + FEDDAT_G8_STEP * code,
 * |error| <= 2.5e-3 over gelu's whole range [-0.129, 1.129] (the bf16 u of FEDDAT_EPI_GELU carries |u| 2^-9 |gelu''(u)|, the
 * same size at |u| ~ 1).  out2 / aux are uint8 [M, N] here (ldo2 / ldaux in BYTES, % 16 == 0, 16-byte aligned).  Persistent
 * kernels only: M >= 1024 and N % 192 == 0 (FEDDAT_EINVAL otherwise). */
#define FEDDAT_EPI_GELU_G8 5    /* out2 = uint8 codes of gelu'(acc + bias); out_bf16 = gelu(acc + bias) */
#define FEDDAT_EPI_MUL_G8 6     /* out_bf16 = (acc) * (FEDDAT_G8_LO + FEDDAT_G8_STEP * aux_u8)        */
/* configs[4] only (feddat_gemm_fp8_nt): the same pair with the MAIN output as e4m3 bytes instead of bf16, so that the next
 * product of the chain is an fp8 product too (out / ldo16: uint8 [M, N], ld in bytes, % 16 == 0).
 *   GELU_G8_F8: out = e4m3(gelu(u) / FEDDAT_F8_ACT_SCALE), saturating at +-448 -- a FIXED scale (e4m3 is floating point:
 *               3 mantissa bits from 2^-6 FEDDAT_F8_ACT_SCALE up to 448 FEDDAT_F8_ACT_SCALE = 56); the consumer passes a
 *               constant a_scale vector.
 *   MUL_G8_F8:  out = e4m3((A8 B8^T) * b_scale[n] * gelu' / FEDDAT_F8_GRAD_HEADROOM): row m of the output carries the row scale
 *               FEDDAT_F8_GRAD_HEADROOM * a_scale[m] of the input gradient row it came from (|sum_k g W| <= 448 * 12 in code
 *               units for ViLT's FFN2; typical 10-60: the headroom keeps outliers below 448 and typical values 6 binades above
 *               the subnormals).  The consumer multiplies its weight scales by the headroom once, at load time. */
#define FEDDAT_EPI_GELU_G8_F8 7
#define FEDDAT_EPI_MUL_G8_F8 8
#define FEDDAT_F8_ACT_SCALE 0.125f
#define FEDDAT_F8_GRAD_HEADROOM 4.0f
#define FEDDAT_G8_LO (-0.135f)     /* = -27 steps */
#define FEDDAT_G8_STEP 0.005f
int feddat_gemm_bf16_nt(const void* A, int lda, const void* B, int ldb, int M, int N, int K, int epi,
                        const float* bias, const float* resid, int ldr, const void* aux, int ldaux, float* out_f32,
                        int ldo32, void* out_bf16, int ldo16, void* out2_bf16, int ldo2, hipStream_t stream);
/* configs[4]: fp8 (OCP e4m3) MFMA for frozen linears.  A8 [M,K] and B8 [N,K] are e4m3 with per-row scales (a_scale [M],
 * b_scale [N] = per output channel): C = (A8 B8^T) * a_scale[m] * b_scale[n] (+ bias; epilogue BF16, GELU / GELU_G8 / GELU_G8_F8
 * or MUL_DGELU / MUL_G8 / MUL_G8_F8 (aux = the saved pre-GELU u, or the gelu' codes) as above -- the last group is the dX
 * product of FFN2 with A8 = the e4m3 row-quantised gradient; for the _F8 forms out_bf16 / ldo16 are the e4m3 output).
 * Same persistent kernel and data movement as the bf16 form (128 fp8 per 128-byte LDS row); the MFMA is the CDNA4
 * block-scaled v_mfma_scale_f32_16x16x128_f8f6f4 with unit block scales (twice the bf16 rate).
 * Requirements: M >= 1024, N % 192 == 0, K % 128 == 0, lda / ldb % 16 == 0, 16-byte aligned outputs.
 * feddat_quant_rows_fp8: fp32 [rows, cols] -> e4m3 + per-row scale amax / 448 (weights at load time; any activation);
 * feddat_layernorm_fwd_fp8: LayerNorm whose output leaves as e4m3 + per-row scale (optionally also bf16). */
int feddat_gemm_fp8_nt(const void* A8, int lda, const float* a_scale, const void* B8, int ldb, const float* b_scale, int M,
                       int N, int K, int epi, const float* bias, const void* aux_bf16, int ldaux, void* out_bf16, int ldo16,
                       void* out2_bf16, int ldo2,
                       hipStream_t stream);
/* the same fp8 product with an fp32 output: out_f32 = (A8 B8^T) * a_scale[m] * b_scale[n] + bias (+ resid, fp32, may be NULL):
 * FFN2 of a ViLT layer on configs[4], fed by the e4m3 gelu(u) that FEDDAT_EPI_GELU_G8_F8 leaves (adaptered_output.py:74-76). */
/* configs[4], the product whose A operand is written 64 columns at a time by different blocks (dqkv -> QKV^T): A carries true MX
 * block scales -- one E8M0 byte per (row, 32 consecutive k): a_mx [M, ld_mx], value 2^(byte - 127), A8[m][k] * 2^(a_mx[m][k / 32] -
 * 127) is the operand -- which the CDNA4 block-scaled MFMA (v_mfma_scale_f32_16x16x128_f8f6f4) applies itself; B keeps its
 * per-output-channel fp32 scale.  out_bf16 = (A . B^T) * b_scale[n] + bias.  M >= 1024, N % 192 == 0, K % 128 == 0. */
int feddat_gemm_fp8mx_nt(const void* A8, int lda, const uint8_t* a_mx, int ld_mx, const void* B8, int ldb, const float* b_scale,
                         int M, int N, int K, const float* bias, void* out_bf16, int ldo16, hipStream_t stream);
int feddat_gemm_fp8_nt_f32(const void* A8, int lda, const float* a_scale, const void* B8, int ldb, const float* b_scale, int M,
                           int N, int K, const float* bias, const float* resid, int ldr, float* out_f32, int ldo32,
                           hipStream_t stream);
int feddat_quant_rows_fp8(const float* x, long ld, int rows, int cols, void* y_fp8, float* scale, hipStream_t stream);
int feddat_layernorm_fwd_fp8(const float* x, long x_stride, const float* gamma, const float* beta, float eps, int rows,
                             int H, void* y_fp8, float* y_scale, void* y_bf16, float* stats, hipStream_t stream);
/* The same product for M <= 64 rows (the top ViLT layer only needs its 2B token-0 rows behind the attention, because the
 * pooler reads hidden_states[:, 0] only: vilt.py:127): split over (N/64) x ksplit blocks into fp32 partials in
 * `workspace` (feddat_gemm_skinny_workspace_elems(M, N, K) floats), summed in a fixed order by a second kernel that
 * applies the epilogue.  Requirements: N % 64 == 0, K % 64 == 0, lda/ldb % 8 == 0. */
long feddat_gemm_skinny_workspace_elems(int M, int N, int K);
/* ABI 8, diagnostics: workgroups per CU the runtime grants the DUAL form of the persistent GEMM (two independent 128 x 192
 * workgroups per CU, 80 KiB of LDS and 256 registers per wave each; feddat_set_debug_flags 1 | 2 [| 64]) on the current device. */
int feddat_gemm_dual_blocks_per_cu(int* out);
int feddat_gemm_bf16_nt_skinny(const void* A, int lda, const void* B, int ldb, int M, int N, int K, int epi,
                               const float* bias, const float* resid, int ldr, const void* aux, int ldaux,
                               float* out_f32, int ldo32, void* out_bf16, int ldo16, void* out2_bf16, int ldo2,
                               float* workspace, long workspace_elems, hipStream_t stream);

/* ---------------------------------------------------------------------------------------------
 * K2  fused self-attention for one ViLT layer (HF ViltSelfAttention: softmax(QK^T/8 + mask) V).
 * qkv: bf16 [B*S, 3*H] rows = tokens, columns = [Q | K | V], head h at columns h*64.. of each part.
 * key_mask: optional uint8 [B,S] (1 = attend, 0 = masked; text padding, vilt.py:98) or NULL.
 * ctx: bf16 [B*S, H]; lse: fp32 [B, heads, S] (log-sum-exp of the scaled scores, saved for backward).
 * Requirements: head_dim == 64, S <= 320 (ViLT: 40 text + 1 + up to 12 x 20 patches of a 384 x 640 image = 281).
 * ------------------------------------------------------------------------------------------- */
int feddat_attn_fwd(const void* qkv, const uint8_t* key_mask, void* ctx, float* lse, int B, int S, int heads,
                    hipStream_t stream);
/* dX-only backward: dqkv (bf16 [B*S, 3*H]) from dctx (bf16 [B*S,H]), qkv, ctx, lse. */
int feddat_attn_bwd(const void* qkv, const uint8_t* key_mask, const void* ctx, const float* lse, const void* dctx,
                    void* dqkv, int B, int S, int heads, hipStream_t stream);
/* configs[4]: feddat_attn_bwd whose dq | dk | dv leave as MX-scaled e4m3 instead of 16-bit values: dq8 [B S, 3 H] bytes (the column
 * layout of dqkv), dq_scale [B S, 3 H / 32] E8M0 bytes, element = e4m3 * 2^(scale - 127), one scale per (row, 32 columns) =
 * the A operand of feddat_gemm_fp8mx_nt (QKV^T on the block-scaled fp8 MFMA).  S <= 192. */
int feddat_attn_bwd_fp8mx(const void* qkv, const uint8_t* key_mask, const void* ctx, const float* lse, const void* dctx,
                          uint8_t* dq8, uint8_t* dq_scale, int B, int S, int heads, hipStream_t stream);
/* The same attention for a layer of which only token 0 of every sample is consumed (the LAST ViLT layer: HF ViltPooler
 * reads hidden_states[:, 0]; vilt.py:127): one query per (sample, head).  feddat_attn_cls_fwd writes row b*S of ctx and
 * lse[b, h, 0] only.  feddat_attn_cls_bwd takes dctx0 = fp32 [B, H], the gradient of those rows (all other rows of dctx
 * are zero by construction), and writes the complete dqkv [B*S, 3*H] (dQ rows other than token 0 are zeros).  fp32 VALU on
 * the bf16 operands, HBM-bound; S <= 320. */
int feddat_attn_cls_fwd(const void* qkv, const uint8_t* key_mask, void* ctx, float* lse, int B, int S, int heads,
                        hipStream_t stream);
int feddat_attn_cls_bwd(const void* qkv, const uint8_t* key_mask, const void* ctx, const float* lse, const float* dctx0,
                        void* dqkv, int B, int S, int heads, hipStream_t stream);

/* General form (the ALBEF path): separate Q / K / V operands (bf16, head h at columns 64 h .. 64 h + 63 of each, row
 * strides ld* in elements, sample b's rows at b * rows_per_sample), any S_q and S_kv (K / V stream through LDS with an
 * online softmax), optional key-padding mask (uint8 [B, S_kv], 1 = attend) and causal masking (key j <= query i).
 * Replaces: ViT-B/16 Attention.forward over 577 tokens (src/modeling/models/vit.py:60-76), BertSelfAttention self- and
 * cross-attention of the ALBEF text encoder / decoder (src/modeling/models/xbert.py; additive -10000 masks == excluded
 * keys in fp32).  ctx: bf16 [B * q_rows_per_sample, ldo]; lse: fp32 [B, heads, S_q].
 * Backward: dq / dk / dv (bf16, same layouts as q / k / v) from dctx; dsum_ws: fp32 [B, heads, S_q] scratch. */
int feddat_attn2_fwd(const void* q, long ldq, const void* k, long ldk, const void* v, long ldv, const uint8_t* key_mask,
                     int causal, void* ctx, long ldo, float* lse, int B, int Sq, int Skv, long q_rows_per_sample,
                     long kv_rows_per_sample, int heads, hipStream_t stream);
int feddat_attn2_bwd(const void* q, long ldq, const void* k, long ldk, const void* v, long ldv, const uint8_t* key_mask,
                     int causal, const void* ctx, long ldo, const float* lse, const void* dctx, long lddo, float* dsum_ws,
                     void* dq, long lddq, void* dk, long lddk, void* dv, long lddv, int B, int Sq, int Skv,
                     long q_rows_per_sample, long kv_rows_per_sample, int heads, hipStream_t stream);
/* The same with train-mode dropout on the attention probabilities (BertSelfAttention, src/modeling/models/xbert.py:333;
 * attention_probs_dropout_prob = 0.1, src/configs/model_configs.py:44): ctx = (softmax(S) . M / (1 - p)) V, and in the
 * backward dP = M / (1 - p) . (dO V^T).  The mask is COUNTER-BASED and never stored: element
 * idx = ((b * heads + h) * S_q + q) * S_kv + key of the [B, heads, S_q, S_kv] probability tensor is kept iff
 *     fmix32(fmix32(idx * 0x9E3779B1 + key0) + key1 + step * 0x632BE5AB) >= p * 2^32      (32-bit wrap-around; fmix32 =
 * the murmur3 finaliser), step = *step_ctr (device int, may be NULL = 0) so that a captured hipGraph draws a fresh mask on
 * every replay; the forward and the backward of one pass take the same (key0, key1, step).  B * heads * S_q * S_kv < 2^32. */
int feddat_attn2_fwd_dropout(const void* q, long ldq, const void* k, long ldk, const void* v, long ldv,
                             const uint8_t* key_mask, int causal, void* ctx, long ldo, float* lse, int B, int Sq, int Skv,
                             long q_rows_per_sample, long kv_rows_per_sample, int heads, float p, unsigned key0,
                             unsigned key1, const int* step_ctr, hipStream_t stream);
int feddat_attn2_bwd_dropout(const void* q, long ldq, const void* k, long ldk, const void* v, long ldv,
                             const uint8_t* key_mask, int causal, const void* ctx, long ldo, const float* lse,
                             const void* dctx, long lddo, float* dsum_ws, void* dq, long lddq, void* dk, long lddk, void* dv,
                             long lddv, int B, int Sq, int Skv, long q_rows_per_sample, long kv_rows_per_sample, int heads,
                             float p, unsigned key0, unsigned key1, const int* step_ctr, hipStream_t stream);

/* ---------------------------------------------------------------------------------------------
 * K3  LayerNorm over the last dim (H <= 2048, H % 4 == 0), fp32 in.
 * fwd: y = (x - mean) * rstd * gamma + beta; x row r at x + r * x_stride (elements); writes y as bf16
 *      (y_bf16, ld = H) and/or fp32 (y_f32, ld = H); stats[2*r] = mean, stats[2*r+1] = rstd (optional).
 * bwd (frozen gamma: dX only): dx = rstd * (g*gamma - mean(g*gamma) - xhat * mean(g*gamma*xhat)),
 *      out_f32[r] = dx + (dres ? dres[r] : 0); optional bf16 copy.  dy is bf16 or fp32 (one non-null).
 * Replaces nn.LayerNorm(768, eps=1e-12) in HF ViltLayer / ViltModel.layernorm (vilt.py:127).
 * ------------------------------------------------------------------------------------------- */
int feddat_layernorm_fwd(const float* x, long x_stride, const float* gamma, const float* beta, float eps, int rows,
                         int H, void* y_bf16, float* y_f32, float* stats, hipStream_t stream);
int feddat_layernorm_bwd_dx(const void* dy_bf16, const float* dy_f32, long dy_stride, const float* x, long x_stride,
                            const float* stats, const float* gamma, const float* dres, long dres_stride, int rows,
                            int H, float* out_f32, long out_stride, void* out_bf16, hipStream_t stream);
/* ABI 8: the same with a SPARSE residual gradient -- non-zero on rows 0, E, 2 E, ... only (E = dres_every > 0), given compact:
 * row r / E of dres (row stride dres_stride) is added to output row r when r % E == 0.  The top ViLT layer's residual gradient
 * lives on token 0 of every sample (E = S): no scatter into a dense [rows, H] buffer, no read of its zeros. */
int feddat_layernorm_bwd_dx_sparse(const void* dy_bf16, const float* dy_f32, long dy_stride, const float* x, long x_stride,
                                   const float* stats, const float* gamma, const float* dres, long dres_stride, int dres_every,
                                   int rows, int H, float* out_f32, long out_stride, void* out_bf16, hipStream_t stream);
/* The same, its result additionally leaving as e4m3 rows + per-row scale (amax / 448): configs[4], the A operand of the
 * fp8 dX product that follows (attention-output^T after layernorm_after's backward). */
int feddat_layernorm_bwd_dx_fp8(const void* dy_bf16, const float* dy_f32, long dy_stride, const float* x, long x_stride,
                                const float* stats, const float* gamma, const float* dres, long dres_stride, int rows, int H,
                                float* out_f32, long out_stride, void* out_bf16, void* out_fp8, float* out_scale,
                                hipStream_t stream);

/* ---------------------------------------------------------------------------------------------
 * K4  fused dual Pfeiffer adapter (the DAT module), src/modeling/models/adapter.py:124-163.
 * One launch handles up to two row segments of x (fp32 [T,768]); segment s covers rows
 * [row_begin, row_end) and applies n_adapters (1 = adapter.py:125-131, 2 = gating 133-146):
 *     out = x + sum_a scale[a] * (W_up[a] * relu(W_down[a] * x + b_down[a]) + b_up[a])
 * Weights are the bf16 operand copies written by feddat_adapter_pack (opaque layouts): wd, wu (+ wdT, wuT for the
 * backward); biases fp32.  H = 768, r = 48.
 * ------------------------------------------------------------------------------------------- */
typedef struct {
    int row_begin, row_end;
    int n_adapters;   /* 1 or 2 */
    int train_slot;   /* backward: which adapter slot (0/1) gets weight grads, -1 = none */
    int x_row_delta;  /* input row = output row + x_row_delta (lets two segments share one input, layer 0) */
    int reserved;
    float scale[2];
    const void* wd[2];   /* bf16 [r,H]  */
    const void* wdT[2];  /* bf16 [H,r]  (backward) */
    const void* wu[2];   /* bf16 [H,r]  */
    const void* wuT[2];  /* bf16 [r,H]  (backward) */
    const float* bd[2];  /* fp32 [r]    */
    const float* bu[2];  /* fp32 [H]    */
} feddat_adapter_seg;

/* z_save (optional, may be NULL): fp32 [T, 2, r]; row t receives relu(W_down[a] x + b_down[a]) of adapter slot a at
 * z_save[t][a][:] (slot 1 untouched for single-adapter segments).  feddat_adapter_bwd takes it back as z_saved. */
int feddat_adapter_fwd(const float* x, float* out, int T, int H, int r, const feddat_adapter_seg* segs, int nseg,
                       float* z_save, hipStream_t stream);
/* feddat_adapter_fwd that also applies the NEXT layer's layernorm_before to its output rows (HF ViltLayer,
 * layernorm_before -> attention): y_bf16[t] = LN(out[t]) * gamma + beta (bf16 [T,H]), stats[2t] = mean, stats[2t+1] = rstd.
 * Saves re-reading the fp32 output (feddat_layernorm_fwd on `out` gives the same result up to fp32 summation order). */
int feddat_adapter_fwd_ln(const float* x, float* out, int T, int H, int r, const feddat_adapter_seg* segs, int nseg,
                          const float* ln_gamma, const float* ln_beta, float eps, void* y_bf16, float* stats,
                          float* z_save, hipStream_t stream);
/* backward: dx = dy + sum_a W_down[a]^T (relu' .* (scale[a] * W_up[a]^T dy)); optional bf16 copy of dx;
 * dx may be NULL (only z/dz are produced: nothing trainable lies below the first adapter).
 * The bottleneck activations come either from z_saved (what the forward wrote into z_save; x may then be NULL and is
 * not read: 3 KB less HBM traffic per token and one K = 768 product less per adapter) or, with z_saved == NULL, are
 * recomputed from x.
 * For the segment's train_slot the kernel also writes z = relu(W_down x + b_down) and dz = scale * relu' .* (W_up^T dy)
 * (fp32 [T,r] each, rows of that segment) that feddat_sgemm_f32 turns into dW_up = scale * dy^T z,
 * db_up = scale * sum_t dy, dW_down = dz^T x, db_down = sum_t dz. */
int feddat_adapter_bwd(const float* x, const float* z_saved, const float* dy, float* dx, void* dx_bf16, float* z_out,
                       float* dz_out, int T, int H, int r, const feddat_adapter_seg* segs, int nseg, hipStream_t stream);
/* configs[4]: the same backward (saved-z form) whose dx also leaves as e4m3 rows + per-row scale (amax / 448): the A operand
 * of the fp8 FFN2^T product, quantised in the kernel that produces it (no extra pass over dx). */
int feddat_adapter_bwd_fp8(const float* z_saved, const float* dy, float* dx, void* dx_fp8, float* dx_scale, float* z_out,
                           float* dz_out, int T, int H, int r, const feddat_adapter_seg* segs, int nseg, hipStream_t stream);
/* Weight gradients of the trainable adapter of up to two row segments, from the z/dz written by
 * feddat_adapter_bwd: grad = flat fp32 [wd (r x H) | bd (r) | wu (H x r) | bu (H)] (the state-dict order of one
 * layer's adapter), fully overwritten.  x, dy: fp32 [rows, H] (row stride H); z, dz: fp32 [rows, r].
 * partials: scratch of feddat_adapter_wgrad_workspace_elems(nseg) floats.
 * grad_unscale (ABI 7): all four gradients are multiplied by it on the way out; 0 (a zero-initialised struct) = 1.  A caller
 * that runs the backward on a loss scaled by 2^k (fp16 operand build) passes 2^-k here: the factor leaves exactly. */
typedef struct {
    const float* x;
    const float* dy;
    const float* z;
    const float* dz;
    float* grad;
    int rows;
    float scale;
    float grad_unscale;
    int reserved;
    const float* grad_unscale_dev;   /* ABI 8: optional DEVICE float multiplied onto grad_unscale at run time (1 / the dynamic loss scale) */
} feddat_wgrad_seg;
long feddat_adapter_wgrad_workspace_elems(int nseg);
int feddat_adapter_wgrad(const feddat_wgrad_seg* segs, int nseg, float* partials, long partials_elems, int H, int r,
                         hipStream_t stream);
/* The same in two steps, for callers that run several of these per backward pass (one per layer): _partial leaves the
 * token-split partial sums of ONE launch in `partials` (no reduction); _reduce folds the partials of n such launches --
 * launch l's at partials + l * partials_stride floats -- into their gradient buffers in one kernel (same fixed summation
 * order as feddat_adapter_wgrad: bit-identical results).  grads_dev: DEVICE array of n * nseg gradient pointers
 * ([launch][segment]), every launch with the same nseg. */
int feddat_adapter_wgrad_partial(const feddat_wgrad_seg* segs, int nseg, float* partials, long partials_elems, int H, int r,
                                 hipStream_t stream);
int feddat_adapter_wgrad_reduce(float* const* grads_dev, int n, int nseg, const float* partials, long partials_stride,
                                hipStream_t stream);
/* ABI 8: the same reduction that also reports non-finite gradients -- nonfinite[s] (DEVICE int per segment) is OR-ed with 1
 * when any gradient element of segment s (any launch) is inf / NaN; it is never cleared here.  This is the inf check of
 * torch.cuda.amp.GradScaler.unscale_ (what accelerator.backward + optimizer.step run under fp16: task_trainer.py:302-308,
 * 323-328), made where the loss scale leaves the gradients. */
int feddat_adapter_wgrad_reduce_checked(float* const* grads_dev, int n, int nseg, const float* partials, long partials_stride,
                                        int* nonfinite, hipStream_t stream);
/* fp32 masters -> bf16 MFMA operand copies (r*H elements each).  All four are stored FRAGMENT-MAJOR -- the 64 lanes of
 * a wave read 64 consecutive 16-byte pieces, i.e. one contiguous 1 KiB burst per weight load -- with the contraction
 * slots of wd / wuT permuted along H (feature c at 32*(c/32) + 8*((c%16)/4) + 4*((c%32)/16) + c%4) so that they coincide
 * with the kernels' 16-byte residual columns (exact index maps: adapter_pack_kernel in csrc/adapter.hip, checked by
 * tests/test_ops_gpu.py::test_adapter_pack).  Treat them as opaque operands of feddat_adapter_fwd/bwd. */
int feddat_adapter_pack(const float* wd, const float* wu, void* wd_bf16, void* wdT_bf16, void* wu_bf16,
                        void* wuT_bf16, int H, int r, hipStream_t stream);

/* The same for n adapter modules in one launch (all layers of adapter_a after an optimizer step): module l reads
 * wd + l * stride_f32 / wu + l * stride_f32 (floats) and writes each bf16 copy at + l * stride_bf16 (elements). */
int feddat_adapter_pack_strided(const float* wd, const float* wu, long stride_f32, void* wd_bf16, void* wdT_bf16,
                                void* wu_bf16, void* wuT_bf16, long stride_bf16, int n, int H, int r,
                                hipStream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Composite: one HF ViltLayer with the reference's Adaptered_ViltOutput as ONE call (what feddat_amd/engine.py sequences
 * per layer).  Forward = transformers ViltLayer.forward as reached from src/modeling/vilt.py:127 with
 * src/modeling/adaptered_output.py:73-78 in place of ViltOutput; backward = dX through the frozen weights + the trainable
 * adapter's weight gradients (autograd of task_trainer.py:302,323).  rows = nb * S; every pointer is a device buffer owned
 * by the caller (bf16 as void*), hidden size = 64 * heads, bottleneck 48, S <= 320.
 *   weights: bf16 [out,in] operands, their [in,out] transposes (backward only), fp32 biases / LayerNorm.
 *   acts:    h_in (in) ... h3, z_save (saved for the backward), h_out = adapter output = the next layer's h_in; x16 / f16 are
 *            bf16 scratch ([rows,768] / [rows,3072]); with next_ln_g / next_ln_b the next layer's layernorm_before is fused
 *            into the adapter kernel: x16 then holds it and st1_next its row statistics (pass ln1_done = 1 to that layer).
 *   grads:   dh_out (in; destroyed) -> dh_in (out); dh3 .. dz are scratch of the shapes noted.
 * ------------------------------------------------------------------------------------------- */
typedef struct {
    const void *wqkv, *wo, *w1, *w2;          /* bf16 [2304,768], [768,768], [3072,768], [768,3072] */
    const void *wqkvT, *woT, *w1T, *w2T;      /* bf16 transposes (backward) */
    const float *bqkv, *bo, *b1, *b2, *ln1_g, *ln1_b, *ln2_g, *ln2_b;
    float ln_eps;
} feddat_vilt_layer_weights;
typedef struct {
    const float* h_in;   /* fp32 [rows,768] */
    float* st1;          /* fp32 [rows,2]  LN1 mean / rstd */
    void* qkv;           /* bf16 [rows,2304] */
    void* ctx;           /* bf16 [rows,768] */
    float* lse;          /* fp32 [nb,heads,S] */
    float* h2;           /* fp32 [rows,768] */
    float* st2;          /* fp32 [rows,2] */
    void* u;             /* rows >= 1024: uint8 [rows,3072] gelu'(u) codes (FEDDAT_EPI_GELU_G8); fewer rows: bf16 [rows,3072] pre-GELU u */
    float* h3;           /* fp32 [rows,768] adapter input */
    float* z_save;       /* fp32 [rows,2,48] or NULL (the backward then recomputes from h3) */
    float* h_out;        /* fp32 [rows,768] */
    void* x16;           /* bf16 [rows,768] scratch / LN1 input operand */
    void* f16;           /* bf16 [rows,3072] scratch */
    float* st1_next;     /* fp32 [rows,2] or NULL */
} feddat_vilt_layer_acts;
typedef struct {
    float* dh_out;       /* fp32 [rows,768] in (destroyed) */
    float* dh_in;        /* fp32 [rows,768] out */
    float* dh3;          /* fp32 [rows,768] */
    void* dh16;          /* bf16 [rows,768] */
    void* dU;            /* bf16 [rows,3072] */
    void* dx16;          /* bf16 [rows,768] */
    void* dctx;          /* bf16 [rows,768] */
    void* dqkv;          /* bf16 [rows,2304] */
    float* z;            /* fp32 [rows,48] */
    float* dz;           /* fp32 [rows,48] */
} feddat_vilt_layer_grads;
int feddat_vilt_layer_fwd(feddat_ctx* ctx, const feddat_vilt_layer_weights* W, const feddat_vilt_layer_acts* A, int nb,
                          int S, int heads, const uint8_t* key_mask, int ln1_done, const feddat_adapter_seg* segs, int nseg,
                          const float* next_ln_g, const float* next_ln_b, hipStream_t stream);
int feddat_vilt_layer_bwd(feddat_ctx* ctx, const feddat_vilt_layer_weights* W, const feddat_vilt_layer_acts* A,
                          const feddat_vilt_layer_grads* G, int nb, int S, int heads, const uint8_t* key_mask,
                          const feddat_adapter_seg* segs, int nseg, const feddat_wgrad_seg* wsegs, int nwseg,
                          float* wgrad_partials, long wgrad_partials_elems, int wgrad_reduce_now, hipStream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Small exact-fp32 GEMM on v_mfma_f32_16x16x4_f32 with arbitrary strides and split-K partial sums:
 *   for split s: D_s[i][j] = alpha * sum_{k in chunk s} A[i*sa_i + k*sa_k] * B[k*sb_k + j*sb_j]
 *   out[s*out_split_stride + i*ldo + j] = D_s[i][j] (+ bias_j[j] if s == 0 and bias_j) .
 * Used for the task head (vilt.py:202-209) forward/backward, the ViLT pooler, and the adapter
 * weight gradients (contraction over tokens).  colsum: optional fp32 (split s at colsum + s*colsum_split_stride) receiving
 * alpha * sum_k A[i][k] for the i-tiles (used for bias gradients), or NULL.
 * ------------------------------------------------------------------------------------------- */
int feddat_sgemm_f32(const float* A, long sa_i, long sa_k, const float* B, long sb_k, long sb_j, int I, int J, int K,
                     int ksplit, float alpha, const float* bias_j, float* out, long ldo, long out_split_stride,
                     float* colsum, long colsum_split_stride, hipStream_t stream);
/* out[i] = sum_s in[s*stride + i], i < n   (deterministic reduction of split-K partials). */
int feddat_reduce_partials(const float* in, long stride, int nsplit, long n, float* out, hipStream_t stream);

/* ---------------------------------------------------------------------------------------------
 * The serial tail of a train_step, fused (csrc/head_tail.hip): token-0 LayerNorm + ViltPooler (HF ViltModel.layernorm /
 * ViltPooler via vilt.py:127), the task head (vilt.py:202-209) forward / backward for the P0 / P1 / P2 passes, and the
 * optimizer bookkeeping of task_trainer.py:280-330,477-504.  All exact fp32.
 *
 * feddat_head_gemm: one or two INDEPENDENT small products in ONE launch, each
 *     out[i, j] = epi( alpha * sum_k pro(A)[i, k] * B[k, j] + bias_j[j] ),   colsum[i] = alpha * sum_k pro(A)[i, k]
 * with arbitrary element strides (A[i, k] at A + i sa_i + k sa_k, B[k, j] at B + k sb_k + j sb_j), fp32 MFMA, the K
 * range split over the waves of a block and summed in wave order (mode 0: no split-K partials in HBM), or -- mode 1,
 * for contractions over the batch (K ~ 32) -- the waves on consecutive 64-column groups.  Prologues on A: LN = LayerNorm
 * of row i over k with gamma = pro_a, beta = pro_b, eps (sa_k must make rows contiguous enough to re-read; statistics
 * {mean, rstd} computed in the block, optionally written to stats_out [I, 2]); TANH_BWD = A[i, k] * (1 - y^2), y = pro_a
 * indexed like A.  Epilogues: TANH; MUL_DGELU = times gelu'(aux[i, j]) (erf GELU).
 * ------------------------------------------------------------------------------------------- */
#define FEDDAT_HT_PRO_NONE 0
#define FEDDAT_HT_PRO_LN 1
#define FEDDAT_HT_PRO_TANH_BWD 2
#define FEDDAT_HT_EPI_NONE 0
#define FEDDAT_HT_EPI_TANH 1
#define FEDDAT_HT_EPI_MUL_DGELU 2
typedef struct feddat_ht_job {
    const float* A; long sa_i, sa_k;
    const float* B; long sb_k, sb_j;
    int I, J, K, mode;
    float alpha;
    const float* bias_j;       /* [J] or NULL */
    float* out; long ldo;
    float* colsum;             /* [I] or NULL */
    int pro;
    const float* pro_a; const float* pro_b; float pro_eps;
    float* stats_out;
    int epi;
    const float* aux; long ld_aux;
    const float* alpha_dev;    /* ABI 8: optional DEVICE float multiplied onto alpha at run time (the dynamic loss scale) */
} feddat_ht_job;
int feddat_head_gemm(const feddat_ht_job* jobs, int njobs, hipStream_t stream);
/* y = LayerNorm(x) (stats [rows, 2] = {mean, rstd}), gelu_out = gelu(y): clf_norm0 + clf_actv0 (vilt.py:205-206) */
int feddat_head_ln_gelu(const float* x, const float* gamma, const float* beta, float eps, int rows, int H, float* y,
                        float* stats, float* gelu_out, hipStream_t stream);
/* full LayerNorm backward with trainable affine (dx, dgamma, dbeta) in one launch; rows <= 4096, H <= 2048 */
int feddat_head_ln_bwd_full(const float* dy, const float* x, const float* stats, const float* gamma, int rows, int H,
                            float* dx, float* dgamma, float* dbeta, hipStream_t stream);
/* feddat_dat_loss_fwd_bwd (below) as ONE launch; bit-identical outputs; scalars needs only 4 floats */
int feddat_dat_loss_fwd_bwd_single(const float* logits, const float* teacher, const float* target, int B, int C, float temp,
                                   float* dlogits, float* scalars, hipStream_t stream);
/* ABI 8: the same launch; *nonfinite (DEVICE int, may be NULL) is OR-ed with 1 when the loss L is inf / NaN (GradScaler would
 * find the non-finite gradients this produces and skip the step) */
int feddat_dat_loss_fwd_bwd_checked(const float* logits, const float* teacher, const float* target, int B, int C, float temp,
                                    float* dlogits, float* scalars, int* nonfinite, hipStream_t stream);
/* feddat_adamw_flat for up to FEDDAT_ADAMW_MAX_GROUPS parameter groups in one launch (n % 4 == 0, 16-byte aligned
 * buffers).  A group reads its schedule index / Adam step count at (state[0] + d_sched, state[1] + d_adam), so two
 * updates of one group inside a step (the task head: sub-steps 2b and 2b + 1) need no counter tick between them.
 * feddat_step_tick_multi: state[k][0] += d_sched[k], state[k][1] += d_adam[k] for up to that many counters, one launch. */
#define FEDDAT_ADAMW_MAX_GROUPS 4
typedef struct feddat_adamw_group {
    float* p; const float* g; float* m; float* v; long n;
    const long* seg_off; const float* seg_wd; int nseg;
    const int* state; int d_sched, d_adam;
    /* ABI 8 (all optional, NULL / 0 = the unconditional update): GradScaler's "skip the optimizer step of an overflowed
     * backward" without leaving the hipGraph.  skip_if[k]: DEVICE ints; the group is left untouched when either is non-zero.
     * bak (3 n floats): bak_mode 1 = the group's p | m | v BEFORE this update are stored there (also when skipped);
     * bak_mode 2 = when *restore_if != 0 the group's p | m | v are restored from bak instead of being updated. */
    const int* skip_if[2];
    float* bak; int bak_mode;
    const int* restore_if;
} feddat_adamw_group;
int feddat_adamw_multi(const feddat_adamw_group* groups, int ngroups, float base_lr, int warmup, int total, float beta1,
                       float beta2, float eps, hipStream_t stream);
int feddat_step_tick_multi(int* const* states, const int* d_sched, const int* d_adam, int n, hipStream_t stream);
/* ABI 8: end of one dat train_step under a DYNAMIC loss scale (torch.cuda.amp.GradScaler as accelerate drives it for
 * mixed_precision fp16: accelerate_config.yaml:8, task_trainer.py:302-308,323-328), one 1-block launch, everything on the device:
 *   flags[0] = sub-step B (P2: adapter_0 + head) saw a non-finite gradient / loss, flags[1] = sub-step A (P1: adapter_1 + head);
 *   applied = flags[1] ? 0 : flags[0] ? 1 : 2 sub-steps took their optimizer + scheduler step (an overflow in A voids the whole
 *   batch: B's forward already used A's head update -- DESIGN.md section 5b);
 *   counters {sched_t, adam_t}: head += {applied, applied}; adapter_1 += {applied, applied >= 1}; adapter_0 += {applied, applied == 2}
 *   (a skipped optimizer step skips its scheduler tick: accelerate/scheduler.py);
 *   scaler_f = {scale, 1 / scale}: any flag -> scale *= backoff (0.5), growth tracker = 0; else tracker += 2 and, once it reaches
 *   growth_interval, scale *= growth (2), tracker = 0 (GradScaler defaults: 65536 / 2 / 0.5 / 2000); scale stays within
 *   [2^-14, 2^30]; scaler_i = {tracker, skipped sub-steps so far, batches with a skip so far, reserved};
 *   flags are cleared for the next step. */
int feddat_dat_step_finish(int* head_state, int* ad1_state, int* ad0_state, int* flags, float* scaler_f, int* scaler_i,
                           float growth, float backoff, int growth_interval, hipStream_t stream);

/* ---------------------------------------------------------------------------------------------
 * K5  loss.  L = (BCEWithLogits_mean(logits,target) * C + 9 * KL_batchmean(log_softmax(logits/3) ||
 * softmax(teacher/3))) / 2   (task_trainer.py:299-301,506-516; train_vqa_crossvqa.py:237).
 * logits/teacher/target: fp32 [B,C] (C <= 1024); dlogits: fp32 [B,C] = dL/dlogits;
 * scalars: fp32 device buffer of at least 4 + 2*B floats; scalars[0..2] = {bce*C, kl, L}, the rest is scratch.
 * ------------------------------------------------------------------------------------------- */
int feddat_dat_loss_fwd_bwd(const float* logits, const float* teacher, const float* target, int B, int C, float temp,
                            float* dlogits, float* scalars, hipStream_t stream);
/* VQA score bookkeeping of the eval path on the device (train_vqa_crossvqa.py:241-257; task_trainer.py:125-157):
 * acc[0] += sum_b target[b, argmax_j logits[b, j]] (first maximal index, like torch.argmax), acc[1] += B.
 * acc: fp32 [2] device buffer the caller zeroes before a loader and reads back ONCE after it (score = 100 acc[0] / acc[1]). */
int feddat_vqa_score_accumulate(const float* logits, const float* target, int B, int C, float* acc, hipStream_t stream);

/* ALBEF (configs[3]) loss: BertLMHeadModel's shifted next-token cross-entropy, weighted per answer
 * (src/modeling/models/xbert.py:1283-1297 with reduction='none'; ALBEF.forward: loss = sum_n weights[n] * lm_loss[n] / B,
 * src/modeling/models/albef_model.py:142-143) plus the vocabulary-axis MKD term of the dat train_step,
 * temp^2 * KLdiv_batchmean(log_softmax(logits / temp) || softmax(teacher / temp)) over the LAST axis
 * (task_trainer.py:300,320,506-516), and dL/dlogits of L = (loss + kl) / 2.
 * One row = one (answer, position) of logits[:, :-1]: logits / teacher fp32 [R, ldl] (V valid columns), labels int64 [R]
 * (-100 = ignored by the CE, still part of the KL, as in the reference), row_weight[r] = weights[n] / B,
 * kl_scale = temp^2 / N (N answers); row_kl (optional fp32 [R], NULL = all ones) multiplies a row's KL term and its gradient --
 * 0 for rows that exist only because a batch was padded to a static frame (the reference pads to the longest of the batch:
 * albef.py:56-57), N_frame / n_batch elsewhere.  dlogits_bf16 [R, ldd] (may be NULL) gets zeros in columns [V, ldd) so that it can be
 * the K-padded operand of the LM-head backward GEMM.  scalars: 4 + 2 R floats; [0] = loss, [1] = kl, [2] = L.
 * grad_scale (ABI 7): factor on dL/dlogits only (the losses in `scalars` are unscaled): the power-of-two loss scale of a
 * caller that runs the backward in the fp16 operand build (1 otherwise); it leaves through feddat_wgrad_seg.grad_unscale. */
int feddat_lm_loss_fwd_bwd(const float* logits, const float* teacher, long ldl, const long* labels,
                           const float* row_weight, const float* row_kl, int R, int V, float temp, float kl_scale, float grad_scale,
                           void* dlogits_bf16, long ldd, float* scalars, hipStream_t stream);
/* ABI 8: the same with the DYNAMIC loss scale -- grad_scale_dev (DEVICE float, may be NULL) multiplies grad_scale at run time,
 * *nonfinite (DEVICE int, may be NULL) is OR-ed with 1 when L is inf / NaN (feddat_dat_loss_fwd_bwd_checked's ALBEF counterpart). */
int feddat_lm_loss_fwd_bwd_dyn(const float* logits, const float* teacher, long ldl, const long* labels,
                               const float* row_weight, const float* row_kl, int R, int V, float temp, float kl_scale,
                               float grad_scale, const float* grad_scale_dev, int* nonfinite, void* dlogits_bf16, long ldd,
                               float* scalars, hipStream_t stream);
/* ALBEF.rank_answer's selections (src/modeling/models/albef_model.py:171-228; eval loop task_trainer.py:159-204).
 * feddat_softmax_gather_rows: out[r, j] = softmax(logits[r * row_stride + 0 .. V))[ids[j * id_stride]]  -- the probability of
 *   every candidate answer's first token after [BOS] (albef_model.py:183-186: F.softmax(logits, 1).index_select(1, answer_ids[:, 1])).
 * feddat_topk_rows: the k largest of each row of n <= 8192 values, sorted descending (equal values: lower index first), as
 *   (out_vals [rows, k] fp32, out_idx [rows, k] int64).  flags: 1 = take log(v) first; `minus` (fp32 [rows, n], may be NULL) is
 *   subtracted; 2 = softmax over the row ahead of the sort.  rank_answer: topk(prob_first, k) (flags 0), then
 *   topk(softmax(log(topk_probs) - answer_loss), k) (flags 3): albef_model.py:186,223-226. */
int feddat_softmax_gather_rows(const float* logits, long row_stride, int rows, int V, const long* ids, long id_stride, int n,
                               float* out, hipStream_t stream);
int feddat_topk_rows(const float* vals, long ld, const float* minus, int rows, int n, int k, int flags, float* out_vals,
                     long* out_idx, hipStream_t stream);


/* ---------------------------------------------------------------------------------------------
 * K6  fused multi-tensor AdamW over one flat fp32 parameter buffer (torch.optim.AdamW semantics,
 * task_trainer.py:477-504) with the HF polynomial-decay-with-warmup multiplier evaluated on device
 * (task_trainer.py:53-59).  state (device, int32[2]) = {sched_t, adam_t}: lr = base_lr * lambda(sched_t);
 * bias corrections use adam_t + 1.  The kernel does not modify state; feddat_step_tick does.
 * seg_off (device int64 [nseg+1]) / seg_wd (device fp32 [nseg]) give per-tensor weight decay.
 * ------------------------------------------------------------------------------------------- */
int feddat_adamw_flat(float* p, const float* g, float* m, float* v, long n, const long* seg_off, const float* seg_wd,
                      int nseg, const int* state, float base_lr, int warmup, int total, float beta1, float beta2,
                      float eps, hipStream_t stream);
int feddat_step_tick(int* state, int d_sched, int d_adam, hipStream_t stream);

/* ---------------------------------------------------------------------------------------------
 * K8  embeddings (HF ViltEmbeddings.forward for full pixel masks, raster patch order).
 * ------------------------------------------------------------------------------------------- */
/* text rows: h[b, s, :] = LN(word[ids] + type[tt] + pos[s]) + modality[0];  h is fp32 [B, S, H], s < Lt */
int feddat_text_embed(const int64_t* input_ids, const int64_t* token_type_ids, const float* word, const float* pos,
                      const float* type, const float* ln_g, const float* ln_b, float eps, const float* modality0,
                      float* h, int B, int Lt, int S, int H, hipStream_t stream);
/* pixels fp32 [B,3,Hi,Wi] -> bf16 patches [B*gh*gw, 3*P*P] (k = c*P*P + py*P + px, Conv2d weight order) */
int feddat_im2col_patches(const float* pixels, void* patches_bf16, int B, int C, int Hi, int Wi, int P,
                          hipStream_t stream);
/* image rows: h[b, Lt, :] = cls + pos[0] + modality[1];
 * h[b, Lt+1+p, :] = proj[b*np+p] + pos_img[b * pos_batch_stride + p*H ..] + modality[1]
 * (pos_batch_stride = 0: one resized grid shared by all samples; np*H: per-sample grids of padded images) */
int feddat_image_embed_assemble(const float* proj, const float* cls, const float* pos0, const float* pos_img,
                                long pos_batch_stride, const float* modality1, float* h, int B, int Lt, int np, int S,
                                int H, hipStream_t stream);
/* feddat_image_embed_assemble + feddat_pos_embed_resize_masked + feddat_vilt_key_mask in ONE launch: the per-sample position
 * grid (pos_grid [g, g, H] resized to the sample's valid patch rectangle, bilinear / align_corners, zero outside) is
 * interpolated where it is added -- no [B, np, H] intermediate -- and key_mask (uint8 [nrep * B, S], optional) is written
 * from attention_mask (int64 [B, Lt] or NULL = all valid) and patch_mask = pixel_mask sampled at the patch origins
 * (int64 [B, gh, gw]).  Bit-identical with the three separate calls. */
int feddat_image_embed_assemble_masked(const float* proj, const float* cls, const float* pos0, const float* pos_grid,
                                       const long* patch_mask, const long* attention_mask, const float* modality1, float* h,
                                       uint8_t* key_mask, int B, int Lt, int gh, int gw, int g, int H, int nrep,
                                       hipStream_t stream);
/* bilinear(align_corners=True) resize of the [g,g,H] position grid to [gh,gw,H] (HF visual_embed) */
int feddat_pos_embed_resize(const float* pos_grid, float* out, int g, int gh, int gw, int H, hipStream_t stream);
/* Padded images (HF ViltEmbeddings.visual_embed, transformers modeling_vilt.py, called from vilt.py:127): pixel_mask is
 * the int64 [B,Hi,Wi] mask of the HF processor.  Sample b's valid patch rectangle is vh x vw (valid patch rows of patch
 * column 0 / columns of patch row 0, mask sampled at the patch origins); out[b] = the g x g grid resized to vh x vw,
 * placed top-left in the (Hi/P) x (Wi/P) grid, zero elsewhere.  out: fp32 [B, (Hi/P)*(Wi/P), H]. */
int feddat_pos_embed_resize_masked(const float* pos_grid, const long* pixel_mask, float* out, int g, int B, int Hi,
                                   int Wi, int P, int H, hipStream_t stream);
/* Attention key mask of the [text(Lt) | CLS | patches] sequence: text keys from attention_mask (int64 [B,Lt], NULL = all
 * valid), CLS valid, patch keys from pixel_mask at the patch origins (NULL = all valid); written nrep times
 * (rows b + rep*B of key_mask, uint8 [nrep*B, S], S = Lt + 1 + (Hi/P)*(Wi/P)). */
int feddat_vilt_key_mask(const long* attention_mask, const long* pixel_mask, uint8_t* key_mask, int B, int Lt, int Hi,
                         int Wi, int P, int nrep, hipStream_t stream);
/* One launch that copies the small per-batch inputs of a step (HF ViLT encodings, reference schema of process_inputs,
 * src/modeling/vilt.py:98, + target_scores) into the caller's static buffers: input_ids / token_type_ids / attention_mask
 * int64 [B, Lt] (attention_mask NULL = all valid), target fp32 [B, n_labels] (NULL = leave d_target alone), pixel_mask int64
 * [B, Hi, Wi] sampled at the patch origins into d_patch_mask int64 [B, Hi/P, Wi/P] (NULL = all valid). */
int feddat_vilt_stage_inputs(const long* input_ids, const long* token_type_ids, const long* attention_mask, const float* target,
                             const long* pixel_mask, long* d_input_ids, long* d_token_type_ids, long* d_attention_mask,
                             float* d_target, long* d_patch_mask, int B, int Lt, int n_labels, int Hi, int Wi, int P,
                             hipStream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Input pipeline, image half: HF ViltImageProcessor as the reference calls it through
 * ViltEncoderWrapper.process_inputs (src/modeling/vilt.py:87-100), once per batch on the device instead of three times
 * per batch on the host.  images: device buffer of n packed [h][w][3] uint8 RGB images, image i at byte offsets[i];
 * heights/widths/out_h/out_w/offsets are HOST arrays (out_* = the ViLT size rule: shorter edge 384, longer <= 640, floored
 * to multiples of 32 -- feddat_amd.image_processing.resize_output_size).  Writes pixel_values fp32 [n,3,Hm,Wm] (PIL
 * BICUBIC resize, bit-exact with Pillow's 8-bit ImagingResample; * 1/255; (v - 0.5) / 0.5; zero padding) and pixel_mask
 * int64 [n,Hm,Wm] (may be NULL).  workspace: feddat_vilt_image_workspace_bytes(...) bytes of device memory.
 * ------------------------------------------------------------------------------------------- */
long feddat_vilt_image_workspace_bytes(const int* heights, const int* widths, const int* out_h, const int* out_w, int n);
int feddat_vilt_image_preprocess(const uint8_t* images, const long* offsets, const int* heights, const int* widths,
                                 const int* out_h, const int* out_w, int n, int Hm, int Wm, float* pixel_values,
                                 long* pixel_mask, void* workspace, long workspace_bytes, hipStream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Input pipeline, text half: BERT WordPiece tokenisation as the reference obtains it from
 * ViltProcessor(text=..., padding=True, truncation=True, max_length=40) (src/modeling/vilt.py:98) and
 * BertTokenizer(..., truncation=True, max_length=25) (src/modeling/albef.py:56-57), once per batch on the device.
 * Vocabulary: feddat_wordpiece_table_build (HOST function, HOST pointers) turns the vocab.txt tokens -- blob of bytes,
 * token i at [offsets[i], offsets[i+1]) -- into an open-addressing table of feddat_wordpiece_table_entries(n) entries of 16
 * bytes, which the caller uploads.  feddat_wordpiece_encode: texts = device blob of UTF-8 bytes, text t at
 * [offsets[t], offsets[t+1]) (device int64 offsets); writes int64 [n_texts, max_len] input_ids ([CLS] pieces[:max_len-2]
 * [SEP], padded with pad_id) and attention_mask, and out_len[t] = number of real tokens, or -1 for a text the kernel
 * refuses (raw control characters, more than 2048 bytes): the caller normalises such texts on the host first.
 * The kernel lower-cases ASCII, folds ASCII whitespace, splits ASCII punctuation, and treats UTF-8 multi-byte sequences as
 * word characters (cut only at character boundaries); accent stripping / CJK and non-ASCII punctuation spacing /
 * non-ASCII lower-casing (BertNormalizer) are the caller's job for non-ASCII texts (feddat_amd/tokenization.py).
 * ------------------------------------------------------------------------------------------- */
long feddat_wordpiece_table_entries(int n_vocab);
int feddat_wordpiece_table_build(const char* blob, const long* offsets, int n_vocab, void* table_host, long entries);
int feddat_wordpiece_encode(const void* text, const long* offsets, int n_texts, const void* table, long entries,
                            int unk_id, int cls_id, int sep_id, int pad_id, int max_len, long* out_ids, long* out_mask,
                            int* out_len, hipStream_t stream);

/* ---------------------------------------------------------------------------------------------
 * misc element-wise helpers
 * ------------------------------------------------------------------------------------------- */
int feddat_cvt_f32_bf16(const float* in, void* out_bf16, long n, hipStream_t stream);
/* in fp32 [R,C] -> out bf16 [C,R] */
int feddat_transpose_f32_bf16(const float* in, void* out_bf16, int R, int C, hipStream_t stream);
/* y = tanh(x) in place (fwd), dx = dy * (1 - y^2) (bwd) -- ViltPooler activation */
int feddat_tanh_fwd(float* x, long n, hipStream_t stream);
int feddat_tanh_bwd(const float* y, const float* dy, float* dx, long n, hipStream_t stream);
/* y = gelu(x); dx = dy * gelu'(x) -- task head clf_actv0 (vilt.py:206) */
int feddat_gelu_fwd(const float* x, float* y, long n, hipStream_t stream);
int feddat_gelu_bwd(const float* x, const float* dy, float* dx, long n, hipStream_t stream);
/* LayerNorm backward WITH affine gradients (task head clf_norm0, trainable): rows <= 1024 */
int feddat_layernorm_bwd_full(const float* dy, const float* x, const float* stats, const float* gamma, int rows, int H,
                              float* dx, float* dgamma, float* dbeta, hipStream_t stream);
/* out = alpha a + beta b + gamma c (b, c may be NULL), fp32 and / or bf16 copy; n % 4 == 0.  Glue around the BERT
 * double-LayerNorm adapter variant (src/modeling/models/adapter.py:97-116): y + inp = (dense + inp) + (A(x) + x) - x. */
int feddat_axpby3(const float* a, float alpha, const float* b, float beta, const float* c, float gamma, float* out_f32,
                  void* out_bf16, long n, hipStream_t stream);
/* nn.Dropout in train mode on a dense tensor of n elements (n % 4 == 0, n < 2^32), optionally fused with the residual add
 * that follows it: out = (keep(idx) ? x / (1 - p) : 0) (+ resid), keep() as documented at feddat_attn2_fwd_dropout with
 * idx = the element's linear index.  Exactly one of x_f32 / x_bf16 is given; out_f32 and / or out_bf16.  Replaces the
 * hidden-state dropouts of the ALBEF BERT towers: BertEmbeddings (src/modeling/models/xbert.py:216), BertSelfOutput (:360,
 * with resid = the block's input), BertOutput (:440, ahead of the adapter); the backward applies the same call (same keys)
 * to the incoming gradient. */
int feddat_dropout(const float* x_f32, const void* x_bf16, const float* resid, float* out_f32, void* out_bf16, long n,
                   float p, unsigned key0, unsigned key1, const int* step_ctr, hipStream_t stream);
/* dst[r] = src[idx[r]], a zero row for idx[r] < 0 (rows of `width` floats; one question's states repeated for each of its k answers,
 * albef_model.py:93-98) and its adjoint over contiguous segments: dst[s] (+)= sum of src rows [off[s], off[s+1]). */
int feddat_gather_rows(const float* src, const int* idx, float* dst_f32, void* dst_bf16, int rows, int width,
                       hipStream_t stream);
int feddat_segment_sum_rows(const float* src, const int* seg_offsets, float* dst, int nseg, int width, int accumulate,
                            hipStream_t stream);
/* scatter B rows into a zero-filled [B*S, H] fp32 buffer at token 0 of each sample (+ bf16 copy) */
int feddat_scatter_cls_rows(const float* rows, float* out_f32, void* out_bf16, int B, int S, int H,
                            hipStream_t stream);

/* ---------------------------------------------------------------------------------------------
 * K7  FedAvg helpers (src/train/main.py:50-65).  acc += x * num / total in the reference's operation order
 * (x * num, then / total); first=1 overwrites acc.  The cross-GPU sum itself is one RCCL all-reduce
 * issued by the host binding (torch.distributed, backend "nccl" == RCCL) on the same flat buffer.
 * ------------------------------------------------------------------------------------------- */
int feddat_fedavg_accumulate(float* acc, const float* x, long n, float num, float total, int first,
                             hipStream_t stream);
/* The collective itself, for callers without torch.distributed: RCCL bound at run time (dlopen of the librccl already in
 * the process, else the system one).  `comm` is an ncclComm_t (passed as void*): either the caller's own, or one made
 * here -- rank 0 calls feddat_comm_unique_id (128 bytes, host memory), ships the id to the other ranks by any host
 * channel, every rank calls feddat_comm_create on its device (collective, like ncclCommInitRank).
 * feddat_fedavg_allreduce: scratch = flat * num / total (reference op order, main.py:62); all-reduce(SUM) of scratch over
 * `comm` on `stream` (xGMI); flat <- scratch.  flat / scratch: fp32 [n] device buffers (ViLT: n = 894 528 = 3.58 MB).
 * Replaces get_average_net's host loop over K clients x 48 tensors (main.py:50-65, call site main.py:510). */
int feddat_comm_unique_id(void* id_128_bytes);
int feddat_comm_create(const void* id_128_bytes, int world, int rank, void** comm_out);
/* The same with a watchdog: ncclCommInitRank waits for ALL ranks, so one rank that died (or could not load RCCL) would
 * hang the others.  timeout_ms > 0: give up after that long with FEDDAT_ETIMEOUT and *comm_out = NULL (the bootstrap
 * attempt is abandoned on a helper thread), so the survivors can agree on another exchange (feddat_amd/train.py does,
 * over torch.distributed); timeout_ms = 0: wait forever (= feddat_comm_create).  RCCL binding order: the librccl
 * the process has already mapped (dlopen RTLD_NOLOAD -- e.g. the one PyTorch-ROCm ships), else the system one. */
int feddat_comm_create_timeout(const void* id_128_bytes, int world, int rank, int timeout_ms, void** comm_out);
int feddat_comm_destroy(void* comm);
int feddat_fedavg_allreduce(void* comm, float* flat, float* scratch, long n, float num, float total, hipStream_t stream);
/* What the communicator is: the RCCL version the library bound at run time (ncclGetVersion code, e.g. 22105 = 2.21.5),
 * the number of ranks the communicator spans and this process's rank in it.  Any out pointer may be NULL; comm may be
 * NULL when only the version is asked for.  (bench.py prints these next to the N > 1 line.) */
int feddat_comm_info(void* comm, int* rccl_version, int* n_ranks, int* rank);

/* ---------------------------------------------------------------------------------------------
 * hardware-semantics probes used by tests/ (MFMA operand pairing, ds_read_b64_tr_b16 layout)
 * ------------------------------------------------------------------------------------------- */
int feddat_probe_tr16(const void* in_bf16_64x64, void* out_bf16_64x8, hipStream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* FEDDAT_HIP_H */
