/* libseedstory_hip.so — C ABI of the MI355X-native (gfx950) SEED-Story hot path.
 *
 * The reference (TencentARC/SEED-Story) has no FFI: its hot path sits behind Python-level
 * plug points.  This header is the boundary a maintainer binds with
 * ctypes from those plug points; every entry point cites the
 * reference interface it replaces (paths relative to the reference repo root).
 *
 * Rules of the ABI
 *   - plain pointers and sizes only; pointers are DEVICE pointers unless named host_*;
 *   - every call enqueues asynchronously on the caller's `stream` (a hipStream_t passed
 *     as void*; NULL = the legacy default stream) and returns immediately; the only
 *     exceptions are named: ss_llama_generate* (return a host count), ss_llama_profile_decode
 *     and the explicit tuning calls ss_gemm_tune / ss_conv3x3_tune;
 *   - return 0 (SS_OK) or a negative SS_E* code; ss_last_error() gives the message
 *     (thread-local); nothing throws; nothing allocates or frees caller memory —
 *     workspaces are passed in, sized by the *_workspace_bytes() queries; the only
 *     library-owned objects are the engine handles (ss_llama_create / ss_vit_create …);
 *   - `dtype` is the model dtype of activations AND weights: SS_F32 (CPU-parity mode),
 *     SS_BF16 (production), SS_F16.  Accumulation is always fp32; values are rounded to
 *     the model dtype where the reference's torch graph rounds them.
 *   - there is no CPU fallback: every function fails with SS_EHIP when no gfx950 device
 *     is usable.
 */
#ifndef SEEDSTORY_HIP_H_
#define SEEDSTORY_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SS_ABI_VERSION 1

enum { SS_F32 = 0, SS_BF16 = 1, SS_F16 = 2 };
enum { SS_OK = 0, SS_EINVAL = -1, SS_EHIP = -2, SS_ENOMEM = -3, SS_ESTATE = -4 };

/* GEMM / GEMV epilogues (bit flags) */
enum {
    SS_EPI_NONE = 0,
    SS_EPI_BIAS = 1,      /* + bias[N]                                                   */
    SS_EPI_GELU = 2,      /* exact-erf GELU after bias (nn.GELU, qwen_visual.py:258-260)  */
    SS_EPI_RESIDUAL = 4,  /* out = residual + round_T(acc [+bias])                        */
    SS_EPI_SILU_MUL = 8,  /* GEMV only: y[n] = silu(acc[n]) * acc[n+N]  (LlamaMLP :190)   */
    SS_EPI_GEGLU_PAIR = 16 /* GEMM only: W rows interleaved (value_i, gate_i); C[m][i] = (acc+b)[2i] * gelu_erf((acc+b)[2i+1]),
                              C has N/2 columns (diffusers GEGLU fused into ff.net.0.proj)   */
};

const char* ss_last_error(void);
int ss_abi_version(void);
/* Device properties of the current HIP device: out[0]=CU count, out[1]=is_gfx950,
 * out[2]=total HBM MiB, out[3]=wavefront size. */
int ss_device_info(int32_t out[4]);
/* Runtime tuning knobs (grid sizes, rows per wave, …); unknown keys are accepted. */
int ss_set_tuning(const char* key, int value);
int ss_get_tuning(const char* key, int dflt);

/* LayerNorm folded into the GEMM that consumes it (the UNet's norm1 -> q|k|v, norm2 -> to_q, norm3 -> ff1; reference:
 * diffusers BasicTransformerBlock, `attn(norm(h))`): with  LN(h) = (h - mean) * rstd * gamma + beta
 *     LN(h) · W^T + b  =  rstd[m] * (h · Wg^T)[m][n]  -  rstd[m] * mean[m] * c[n]  +  d[n]
 * where Wg = gamma ∘ W (rounded to the model dtype once at load), c[n] = sum_k Wg[n][k] (fp32), d[n] = b[n] + sum_k
 * beta[k] W[n][k].  The normalised activation tensor is never written or re-read: the GEMM streams the RAW rows, the
 * epilogue applies  t = acc * rstd[m] + shift[m] * c[n] + d[n]  (shift = -mean * rstd) in fp32 and continues with GELU /
 * GEGLU pairs as ss_gemm does.  Rounding differs from the unfused form (LN output not rounded to bf16; gamma rounded
 * into the weight): same order of magnitude, covered by the bf16 parity gates.
 *   ss_rowstats     x [M, K] (16-bit, row stride ld, K <= 2048) -> rstd_out[M], shift_out[M] (fp32)
 *   ss_gemm_lnfold  A [M, K] raw rows, Wg [N, K]; `bias` carries d (model dtype); K % 64 == 0, N % 16 == 0;
 *                   epilogue flags BIAS | GELU | GEGLU_PAIR.
 *
 * Producer-carried statistics (round 3): the LayerNorm's row statistics need no pass of their own when the GEMM that
 * PRODUCES h (attn.to_out + residual, ff.net.2 + residual, proj_in) accumulates them while it stores h:
 *   ss_gemm_rowstat      ss_gemm whose epilogue also adds, per output row m, sum_n C[m][n] and sum_n C[m][n]^2 of the
 *                        values as stored (rounded to the model dtype, residual included) into rowstat_accum[2m],
 *                        rowstat_accum[2m+1] (fp64 atomics — one pair per row per column tile; the caller zeroes the
 *                        array before the call; 16-byte aligned; any epilogue except GEGLU_PAIR);
 *   ss_rowstat_finalize  rowstat[2m .. 2m+1] -> rstd_out[m], shift_out[m] (mean = s / width, var = q / width - mean^2
 *                        formed in fp64: no cancellation), the inputs of ss_gemm_lnfold; re-zeroes rowstat for the next
 *                        producer.  One accumulator per row count serves a whole forward: producer, finalize and
 *                        consumer follow each other on the stream. */
int ss_rowstats(const void* x, int64_t ld, int64_t M, int64_t K, float eps, float* rstd_out, float* shift_out, int dtype, void* stream);
/* Small-M weight-streaming GEMM (128 < M <= 512 rows against a LLaMA projection: the stacked image-token block of the
 * lock-step stories, 4 x 66 rows, and their first prompts): C = A W^T (+bias)(+residual), A [M, K], W [N, K], C [M, N]
 * contiguous.  K is split over grid.y so that every CU holds two workgroups with short K loops; fp32 partial sums go to
 * the caller's workspace (ss_gemm_splitk_workspace_bytes; 0 = shape not eligible) and a reduce pass applies the epilogue
 * in ss_gemm's order.  Falls back to ss_gemm when the shape is not eligible or the workspace is missing / too small. */
size_t ss_gemm_splitk_workspace_bytes(int64_t M, int64_t N, int64_t K);
int ss_gemm_splitk(const void* A, const void* W, void* C, int64_t M, int64_t N, int64_t K, const void* bias, const void* residual,
                   void* workspace, size_t workspace_bytes, int dtype, void* stream);
int ss_gemm_rowstat(const void* A, const void* W, void* C, int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldw,
                    int64_t ldc, const void* bias, const void* residual, int64_t ldr, int epilogue, double* rowstat_accum,
                    int dtype, void* stream);
int ss_rowstat_finalize(double* rowstat, int64_t M, int64_t width, float eps, float* rstd_out, float* shift_out, void* stream);
int ss_gemm_lnfold(const void* A, const void* Wg, void* C, int64_t M, int64_t N, int64_t K, int64_t ldc, const float* rstd,
                   const float* shift, const float* colsum, const void* bias, int epilogue, int dtype, void* stream);
/* Round 4 — the same pair WITHOUT atomics and WITHOUT a finalize launch (deterministic; what UNet2DConditionModel.enable_lnfold
 * uses): every wave column strip of the producer tile writes its (sum, sum of squares) of row m ONCE to
 * rowpart[(m * strips + strip) * 2 .. +1] (fp32; strips = ss_gemm_rowpart_strips(M, N, K, dtype) = N / strip width of the
 * tile ss_gemm_rowpart will run, 0 = shape not eligible: N must be a multiple of the tile width, operands 16-byte aligned);
 * nothing needs zeroing.  ss_gemm_lnfold_part is ss_gemm_lnfold whose epilogue sums the `strips` partials of its rows and
 * forms rstd / -mean * rstd itself (mean = s / width, var = q / width - mean^2 in fp64). */
int64_t ss_gemm_rowpart_strips(int64_t M, int64_t N, int64_t K, int dtype);
int ss_gemm_rowpart(const void* A, const void* W, void* C, int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldw,
                    int64_t ldc, const void* bias, const void* residual, int64_t ldr, int epilogue, float* rowpart,
                    int dtype, void* stream);
int ss_gemm_lnfold_part(const void* A, const void* Wg, void* C, int64_t M, int64_t N, int64_t K, int64_t ldc, const float* rowpart,
                        int64_t strips, int64_t width, float eps, const float* colsum, const void* bias, int epilogue, int dtype,
                        void* stream);

/* fp8 (OCP e4m3fn) GEMM path of the SDXL UNet's linear layers (SURVEY §8 ★ row; BASELINE configs[4]).  The reference
 * has no fp8 path; this is the bf16 GEMM  C = A · W^T (+bias)(+GELU | GEGLU)(+residual)  with both operands quantised:
 *   q = RNE_e4m3(x * 448 / amax(row)),  scale = amax(row) / 448   (activations: per token row; weights: per output channel)
 * and  C = round_bf16(acc_fp32 * scale_a[m] * scale_w[n] + bias ...).  The matrix instruction is
 * v_mfma_scale_f32_16x16x128_f8f6f4 with unit block scales (twice the bf16 MFMA rate on gfx950).
 *   ss_quantize_rows_fp8  x [M, K] (16-bit dtype, row stride ld) -> q [M, K] bytes + scale [M] fp32.  With ln_gamma /
 *       ln_beta != NULL the rows are LayerNorm-ed first (eps = ln_eps; the normalised value is rounded to `dtype` as the
 *       unfused ss_layernorm would) — the bf16 normalised tensor is never written.  Also used once per weight at load.
 *   ss_gemm_fp8  A8 [M, K], W8 [N, K] row-major bytes, K % 128 == 0; C / bias / residual are bf16.  Epilogue flags as
 *       ss_gemm (BIAS, GELU, RESIDUAL, GEGLU_PAIR).  Exact w.r.t. its quantised operands (fp32 accumulation). */
int ss_quantize_rows_fp8(const void* x, int64_t ld, int64_t M, int64_t K, void* q_out, float* scale_out, const void* ln_gamma,
                         const void* ln_beta, float ln_eps, int dtype, void* stream);
int ss_gemm_fp8(const void* A8, const float* scale_a, const void* W8, const float* scale_w, void* C, int64_t M, int64_t N,
                int64_t K, int64_t ldc, const void* bias, const void* residual, int64_t ldr, int epilogue, void* stream);

/* Device context ("no hidden global state except a per-device handle").  A process drives one GPU; the
 * library's only state besides the thread-local error string — the tuning knobs and the GEMM tile table — is
 * process-global and valid for any gfx950 device (the only architecture accepted).  ss_create validates `device`, makes
 * it the current HIP device and returns a handle; ss_destroy releases the handle and leaves the tile table alone (other
 * users of the library in the process keep their tuned tiles; ss_tune_clear drops it explicitly).  Op entry points act
 * on the current HIP device and take no handle.
 * ss_context_info: out = {device ordinal, CU count, HBM bytes}. */
typedef struct ss_context ss_context;
int ss_create(int device, ss_context** out);
int ss_context_info(const ss_context* ctx, int64_t out[3]);
void ss_destroy(ss_context* ctx);

/* RCCL exchange step of the multi-GPU slot ring (reference: single device, gen_george.py:18; the dependency between
 * story steps is `image_embeds = torch.cat((image_embeds, image_embeds_gen))`, :224): the regressed image feature and
 * the KV-cache rows a round appended travel from the round's owner to the other ranks (seedstory/parallel.py does the
 * same through torch.distributed).  librccl is resolved with dlopen at the first call (SS_ESTATE when it is absent).
 * ss_rccl_unique_id: rank 0 fills a 128-byte id and hands it to the other ranks out of band; ss_rccl_init is collective.
 * send / recv / bcast enqueue on `stream`; `count` elements of `dtype`; bcast is in place. */
typedef struct ss_rccl ss_rccl;
int ss_rccl_unique_id(void* id_out_128_bytes);
int ss_rccl_init(const void* id_128_bytes, int nranks, int rank, ss_rccl** out);
void ss_rccl_destroy(ss_rccl* c);
int ss_rccl_send(ss_rccl* c, const void* buf, int64_t count, int dtype, int peer, void* stream);
int ss_rccl_recv(ss_rccl* c, void* buf, int64_t count, int dtype, int peer, void* stream);
int ss_rccl_bcast(ss_rccl* c, void* buf, int64_t count, int dtype, int root, void* stream);

/* ---------------------------------------------------------------------------------------
 * Norms and element-wise ops
 * ------------------------------------------------------------------------------------- */

/* LlamaRMSNorm.forward — src/models_clm/modeling_llama_xformer.py:107-115.
 * y = w * round_T(x * rsqrt(mean_fp32(x^2) + eps)); x,y [rows, cols]; w [cols]. */
int ss_rmsnorm(const void* x, const void* w, void* y, int64_t rows, int64_t cols, float eps, int dtype,
               void* stream);

/* nn.LayerNorm (qwen_visual.py:103,353; resampler.py norm layers): fp32 statistics,
 * y = round_T((x-mean)*rstd*w + b). */
int ss_layernorm(const void* x, const void* w, const void* b, void* y, int64_t rows, int64_t cols, float eps,
                 int dtype, void* stream);

/* y[b, r, :] = x[b, r, :] + p[r, :]   (position-embedding adds, qwen_visual.py:147-148,387;
 * x may be NULL-batched: if x_batch_stride == 0 the same x rows are used for every b). */
int ss_add_bcast(const void* x, const void* p, void* y, int64_t batch, int64_t rows, int64_t cols,
                 int64_t x_batch_stride, int dtype, void* stream);

/* out[r, i] = silu(gu[r, i]) * gu[r, inter + i]; gu [rows, 2*inter] (LlamaMLP, :190-191). */
int ss_silu_mul(const void* gu, void* out, int64_t rows, int64_t inter, int dtype, void* stream);

/* Embedding lookup: out[i, :] = table[ids[i], :] (models.py:127); ids int32 on device. */
int ss_gather_rows(const void* table, const int32_t* ids, void* out, int64_t n, int64_t cols, int dtype,
                   void* stream);
/* Splice: dst[idx[i], :] = src[i, :] (models.py:135: input_embeds[ids_cmp_mask] = ...). */
int ss_scatter_rows(const void* src, const int32_t* idx, void* dst, int64_t n, int64_t cols, int dtype,
                    void* stream);
/* Conv2d(k=stride=patch, no bias) input re-layout for the ViT patch embed (qwen_visual.py:347,382):
 * img [B,3,S,S] (T) -> patches [B*(S/p)^2, kpad] with column order (c, dy, dx), zero padded to kpad. */
int ss_im2col_patch(const void* img, void* out, int64_t batch, int64_t size, int64_t patch, int64_t kpad,
                    int dtype, void* stream);
/* F.normalize(x) over dim=1 of [B, L, C] (resampler.py:269): x / max(||x||_2 over L, 1e-12). */
int ss_l2normalize_dim1(const void* x, void* y, int64_t batch, int64_t len, int64_t cols, int dtype,
                        void* stream);

/* ---------------------------------------------------------------------------------------
 * RoPE + KV cache
 * ------------------------------------------------------------------------------------- */

/* apply_rotary_pos_emb + KV concatenation — modeling_llama_xformer.py:158-173, 236-242.
 * qkv [M, 3*n_heads*hd] (q | k | v per token).  Rotates q in the model dtype with the
 * (model-dtype) tables cos/sin [max_pos, hd] at positions pos_ids[m] (device int32; if NULL,
 * pos_start + m), writes q_out [M, n_heads*hd], and appends rotated k and v into the caches
 * kcache/vcache [n_heads, cache_cap, hd] at slots kv_start + m (keys cached AFTER RoPE). */
int ss_rope_kv_append(const void* qkv, void* q_out, void* kcache, void* vcache, const void* cos,
                      const void* sin, const int32_t* pos_ids, int64_t pos_start, int64_t M, int64_t n_heads,
                      int64_t hd, int64_t kv_start, int64_t cache_cap, int dtype, void* stream);

/* ---------------------------------------------------------------------------------------
 * Attention
 * ------------------------------------------------------------------------------------- */

/* Fused softmax attention (flash style, MFMA) replacing
 *   xops.memory_efficient_attention(q,k,v, LowerTriangularFromBottomRightMask) (:289-295)
 *   when causal_br=1 (query i attends keys j <= i + kv_len - q_len), and the unfused
 *   bmm/softmax/bmm of VisualAttention (qwen_visual.py:207-220), nn.MultiheadAttention
 *   (:147-149) and PerceiverAttention (resampler.py:69-72) when causal_br=0.
 * Element (b,h,i,d) of q is at q + b*q_sb + h*q_sh + i*q_ss + d (strides in elements),
 * likewise k/v (k_s*, v_s*) and out (o_s*).  hd <= 128 (104 and 64 are supported);
 * softmax in fp32; P and output rounded to T.  `scale` multiplies q·k. */
int ss_attention(const void* q, const void* k, const void* v, void* out, int64_t batch, int64_t n_heads,
                 int64_t q_len, int64_t kv_len, int64_t hd, int64_t q_sb, int64_t q_sh, int64_t q_ss,
                 int64_t k_sb, int64_t k_sh, int64_t k_ss, int64_t v_sb, int64_t v_sh, int64_t v_ss,
                 int64_t o_sb, int64_t o_sh, int64_t o_ss, float scale, int causal_br, int dtype,
                 void* stream);

/* The same with one key count PER batch element (host array of `batch` <= 8 ints; strides as above): the stacked forward
 * of several story slots whose caches hold different lengths — query rows [b*q_len, (b+1)*q_len) of slot b attend to the
 * first host_kv_lens[b] entries of slot b's cache, bottom-right causal per slot (modeling_llama_xformer.py:289-295 run
 * once per story by the reference; LlamaEngine.prefill_batch issues ONE launch for the group). */
int ss_attention_ragged(const void* q, const void* k, const void* v, void* out, int64_t batch, int64_t n_heads,
                        int64_t q_len, const int32_t* host_kv_lens, int64_t hd, int64_t q_sb, int64_t q_sh, int64_t q_ss,
                        int64_t k_sb, int64_t k_sh, int64_t k_ss, int64_t v_sb, int64_t v_sh, int64_t v_ss,
                        int64_t o_sb, int64_t o_sh, int64_t o_ss, float scale, int causal_br, int dtype,
                        void* stream);

/* Single-query decode attention over the KV cache (q_len = 1 case of :289-295), split-KV.
 * q [n_heads*hd]; caches [n_heads, cache_cap, hd]; kv_len read on the device from
 * *kv_len_dev (so the launch can be replayed from a hipGraph); out [n_heads*hd].
 * workspace: ss_attn_decode_workspace_bytes(n_heads, hd). */
size_t ss_attn_decode_workspace_bytes(int64_t n_heads, int64_t hd);
int ss_attn_decode(const void* q, const void* kcache, const void* vcache, void* out, void* workspace,
                   const int32_t* kv_len_dev, int64_t n_heads, int64_t hd, int64_t cache_cap, int dtype,
                   void* stream);

/* ---------------------------------------------------------------------------------------
 * Dense contractions
 * ------------------------------------------------------------------------------------- */

/* C[M,N] = A[M,K] · W[N,K]^T (+bias) (+GELU) (+residual)   — nn.Linear everywhere on the path
 * (modeling_llama_xformer.py:228-230,297,191; qwen_visual.py:191,259; resampler.py).
 * MFMA (bf16/f16: 16x16x32, f32: 16x16x4); lda/ldw/ldc/ldr in elements; K % 8 == 0. */
int ss_gemm(const void* A, const void* W, void* C, int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldw,
            int64_t ldc, const void* bias, const void* residual, int64_t ldr, int epilogue, int dtype,
            void* stream);

/* Tile-configuration table of ss_gemm / ss_conv3x3 (16-bit dtypes, M > 128).  The best (tile, XCD group) of a shape
 * is data: ss_gemm / ss_conv3x3 only LOOK THE SHAPE UP (GEMM keys bucket M up to a multiple of 128) and otherwise use a
 * closed-form rule — they never time, allocate or synchronise, so they stay asynchronous and hipGraph-capturable.
 * Entries come from the two explicit tuning calls below or from ss_tune_import.
 *
 * ss_gemm_tune / ss_conv3x3_tune: time every candidate tile x tile order for one shape on `stream` with HIP events in
 * SUSTAINED mode (back-to-back launches over rotating weight copies that cycle through more than the Infinity Cache;
 * the way the kernel runs inside a forward), operands synthesised (pseudo-random) inside the caller's `workspace`
 * (>= *_tune_workspace_bytes), SYNCHRONISES, stores the winner in the table; *best_us_host (optional) receives its
 * time.  `epilogue` may carry SS_EPI_GELU / SS_EPI_GEGLU_PAIR (their arithmetic is timed; bias/residual are not).
 * ss_tune_lookup: out = {cfg, xcd_group} of the entry (returns 1 and out[0] = -1 when the shape has none).
 * ss_tune_export / ss_tune_import: the table as int32 records [n][10] = {dtype, M', N, K, conv_Cin,
 * 2*stride+upsample, conv_H, conv_W, cfg, xcd_group}; export returns the number of entries it holds. */
size_t ss_gemm_tune_workspace_bytes(int64_t M, int64_t N, int64_t K, int dtype);
int ss_gemm_tune(int64_t M, int64_t N, int64_t K, int epilogue, int dtype, void* workspace, size_t workspace_bytes,
                 void* stream, float* best_us_host);
size_t ss_conv3x3_tune_workspace_bytes(int64_t batch, int64_t H, int64_t W, int64_t Cin, int64_t Cout, int64_t stride,
                                       int64_t upsample2x, int dtype);
int ss_conv3x3_tune(int64_t batch, int64_t H, int64_t W, int64_t Cin, int64_t Cout, int64_t stride, int64_t upsample2x,
                    int dtype, void* workspace, size_t workspace_bytes, void* stream, float* best_us_host);
int ss_tune_lookup(int64_t M, int64_t N, int64_t K, int64_t conv_Cin, int64_t conv_H, int64_t conv_W, int64_t stride,
                   int64_t upsample2x, int dtype, int32_t out[2]);
int64_t ss_tune_export(int32_t* out, int64_t cap_entries);
int ss_tune_import(const int32_t* in, int64_t n_entries);
int ss_tune_clear(void);

/* y[N] = W[N,K] · x[K] — the batch-1 decode projection (weight streaming, HBM-bound).
 * Optional fused prologue: if norm_w != NULL, x is first RMS-normalised (as ss_rmsnorm).
 * Epilogues: SS_EPI_RESIDUAL (y = residual + round_T(acc)), SS_EPI_SILU_MUL (W is
 * [2N, K] = [gate; up], y[n] = silu(gate·x) * (up·x)), SS_EPI_BIAS.  K % 8 == 0. */
int ss_gemv(const void* W, const void* x, void* y, int64_t N, int64_t K, const void* norm_w, float eps,
            const void* bias, const void* residual, int epilogue, int dtype, void* stream);
/* The same projection for nb <= 4 rows at once (the lock-step decode of nb story slots):
 * x [nb, K], y / residual [nb, N] contiguous.  W is swept ONCE; every 16-byte weight pack is
 * dotted with all nb activation slices.  Row b equals ss_gemv on row b. */
int ss_gemv_batched(const void* W, const void* x, void* y, int64_t N, int64_t K, int64_t nb,
                    const void* norm_w, float eps, const void* bias, const void* residual, int epilogue,
                    int dtype, void* stream);

/* ---------------------------------------------------------------------------------------
 * Sampling: lm_head logits -> AutoImageTokenGenerationProcessor -> greedy argmax
 * ------------------------------------------------------------------------------------- */

/* src/models_clm/generation.py:19-31 + HF greedy argmax, on device:
 * logits [vocab] (T, modified in place like the reference does), last_id = *last_id_dev,
 * img_ids int32[n_img_ids] = ids of <img><img_00000>…</img>.  If last_id is one of
 * img_ids[:-1] the successor's score becomes max+10, else scores[img_ids[1:]] = 0.0;
 * then the first maximal index is written to *token_out_dev. */
int ss_imgproc_argmax(void* logits, int64_t vocab, const int32_t* last_id_dev, const int32_t* img_ids,
                      int64_t n_img_ids, int32_t* token_out_dev, int dtype, void* stream);

/* ---------------------------------------------------------------------------------------
 * LLaMA decoder engine (native runtime: KV cache, prefill loop, hipGraph-captured decode)
 * ------------------------------------------------------------------------------------- */

typedef struct ss_llama ss_llama;

typedef struct ss_llama_config {
    int32_t hidden, n_heads, n_layers, inter, vocab, max_pos; /* LlamaConfig */
    float rms_eps;
    int32_t dtype;
    int32_t cache_cap;   /* KV slots per head */
    int32_t max_new;     /* capacity of the generated-token / hidden-state ring */
    int32_t n_img_ids;   /* 66 for SEED-Story */
    int32_t eos_id;
    int32_t n_seq;       /* sequence slots (independent stories decoded in lock-step, sharing one
                            sweep of the weights per token); 0 or 1 = the reference's batch-1 loop */
} ss_llama_config;

/* Per-layer weights, all [out, in] row-major in the model dtype, LoRA already merged
 * (or absent): wqkv [3*hidden, hidden] = [q;k;v], wo [hidden, hidden],
 * wgu [2*inter, hidden] = [gate; up], wdown [hidden, inter], ln1, ln2 [hidden]. */
typedef struct ss_llama_layer_weights {
    const void *wqkv, *wo, *wgu, *wdown, *ln1, *ln2;
} ss_llama_layer_weights;

typedef struct ss_llama_weights {
    const void* embed;    /* [vocab, hidden] */
    const void* lm_head;  /* [vocab, hidden] */
    const void* final_norm;
    const void* rope_cos; /* [max_pos, hd] model dtype (LlamaRotaryEmbedding, :118-134) */
    const void* rope_sin;
    const ss_llama_layer_weights* layers; /* host array[n_layers] */
} ss_llama_weights;

/* Bytes of device memory the engine needs (KV cache + activations); the caller allocates it
 * (torch) and hands it over in ss_llama_create — the engine never calls hipMalloc for it. */
size_t ss_llama_workspace_bytes(const ss_llama_config* cfg, int64_t max_prefill_rows);
int ss_llama_create(const ss_llama_config* cfg, const ss_llama_weights* w, void* workspace,
                    size_t workspace_bytes, int64_t max_prefill_rows, const int32_t* host_img_ids,
                    ss_llama** out);
void ss_llama_destroy(ss_llama* h);

/* Select the sequence slot (0 <= seq < n_seq) that ss_llama_buffer / set_lengths / get_lengths /
 * kv_gather / prefill / generate address.  Slots have private KV caches, logits, token and hidden
 * rings; they share the weights and the activation scratch.  Default slot 0. */
int ss_llama_select(ss_llama* h, int32_t seq);
/* Optional second stop token of the decode loop (besides EOS and the token limit): generation ends right AFTER producing
 * `token_id` (it is in the generated ids, not yet fed).  -1 = none (default).  Used to stop at `<img>`: the 64
 * `<img_000NN>` tokens + `</img>` that follow are forced by the logits processor (generation.py:19-31), so the host
 * feeds them as ONE batched continuation (ss_llama_prefill, weights streamed once for 66 rows instead of 66 times) and
 * resumes the loop from its logits.  Applies to every sequence slot.  Range: -1 <= token_id < min(vocab, 32766) — the
 * loop's stop word holds EOS in 16 bits and this id + 1 in the 15 bits above; ss_llama_create therefore requires
 * vocab <= 65535 and eos_id < 65535 (LLaMA-2 + 66 added tokens: 32066). */
int ss_llama_set_stop_id(ss_llama* h, int32_t token_id);

/* Device pointers into the engine's workspace (views for the Python side), for the selected slot:
 * which: 0 = K cache [n_layers, n_heads, cache_cap, hd], 1 = V cache (same shape),
 * 2 = generated ids int32[max_new], 3 = hidden rows [max_new, hidden] (post final norm,
 * row j = state whose input token was generated id j; models.py:182-184),
 * 4 = last-token logits [vocab], 5 = state int32[8] {kv_len,pos,n_gen,done,last_id,…}. */
void* ss_llama_buffer(ss_llama* h, int which);

/* Set / get the live KV length and next rope position (host view).  Truncating the cache
 * (vis_george_sink.py:243) is ss_llama_set_lengths(h, new_len, new_pos). */
int ss_llama_set_lengths(ss_llama* h, int64_t kv_len, int64_t pos, void* stream);
int ss_llama_get_lengths(ss_llama* h, int64_t* kv_len, int64_t* pos);
/* KV re-pack for the multimodal attention sink (vis_george_sink.py:266-295): new cache =
 * old cache gathered at keep_idx[0..n_keep) (device int32, ascending), kv_len = n_keep. */
int ss_llama_kv_gather(ss_llama* h, const int32_t* keep_idx_dev, int64_t n_keep, void* stream);

/* LlamaModel.forward on M new rows against the cached prefix (prefill when the cache is
 * empty, bottom-right-causal continuation otherwise) — modeling_llama_xformer.py:532-666.
 * embeds [M, hidden]; pos_ids device int32[M] or NULL (pos, pos+1, …).  Writes the
 * post-final-norm hidden rows to hidden_out [M, hidden] if non-NULL, and the LAST row's
 * lm_head logits into the engine's logits buffer (the reference computes all rows, :759,
 * but only row -1 is consumed by greedy search).  Advances kv_len/pos by M. */
int ss_llama_prefill(ss_llama* h, const void* embeds, int64_t M, const int32_t* pos_ids, void* hidden_out,
                     void* stream);
/* ss_llama_prefill for several sequence slots in ONE sweep of the weights: embeds = the slots' new rows stacked
 * slot-major [sum(host_rows), hidden], host_rows[n_seq] (host array; 0 = slot sits out).  Every projection runs once on
 * the stack (the layer weights are streamed once per call, not once per slot); RoPE / KV append / bottom-right causal
 * attention run per slot against that slot's cache, positions continue from each slot's own `pos`.  Each slot's last
 * row's logits land in that slot's logits buffer; hidden_out [sum(host_rows), hidden] or NULL.  Used for the image-token
 * block continuation of lock-step stories (generation.py:19-31 forces those ids) and their prompt prefill. */
int ss_llama_prefill_batch(ss_llama* h, const void* embeds, const int64_t* host_rows, void* hidden_out, void* stream);

/* Greedy decode (HF greedy search as driven from models.py:146-153):
 * starting from the logits buffer left by prefill, runs up to n_steps iterations of
 * {processor -> argmax (or forced[i] while i < n_forced) -> append -> embed -> 32 layers ->
 * final norm -> lm_head}, entirely on device from a captured hipGraph, stopping at EOS.
 * last_prompt_id seeds the processor's "last token" for the first step.
 * host_n_generated receives the number of tokens produced (synchronises the stream). */
int ss_llama_generate(ss_llama* h, int64_t n_steps, int32_t last_prompt_id, const int32_t* host_forced,
                      int64_t n_forced, int64_t* host_n_generated, void* stream);

/* The same loop for all n_seq slots at once: every decode projection sweeps the weights ONCE and
 * dots each pack with all slots' activations (HBM bytes per generated token fall as 1/n_seq).
 * Per slot b: last_prompt_ids[b], forced tokens host_forced[b*forced_ld .. +n_forced[b]) (both
 * optional), active[b] == 0 leaves the slot untouched (NULL = all active).  Slots stop
 * independently at EOS / n_steps; host_n_generated[n_seq] receives the per-slot token counts.
 * After the call the logits buffer of a slot that stopped early is undefined. */
int ss_llama_generate_batch(ss_llama* h, int64_t n_steps, const int32_t* last_prompt_ids,
                            const int32_t* host_forced, int64_t forced_ld, const int64_t* n_forced,
                            const int32_t* active, int64_t* host_n_generated, void* stream);

/* Per-kernel-class device time of one decode token (all n_seq slots together), measured with a hipEvent pair around EVERY
 * launch of un-captured (eager) decode steps on `stream`, averaged over n_tokens:
 * out_ms[0] = sum over the K=hidden GEMV launches (qkv, o, gate|up per layer + lm_head:
 *             ss::gemv_kernel), [1] = attention (+split merge), [2] = sum over the down-projection
 *             GEMV launches (ss::gemv_ldsx_kernel), [3] = sampling + final norm, [4] = whole token;
 * out_bytes[0] = weight bytes streamed by the class-0 launches of one token, [1] = their count,
 * out_bytes[2] / [3] = the same for class 2. */
int ss_llama_profile_decode(ss_llama* h, int64_t n_tokens, float out_ms[8], double out_bytes[4],
                            void* stream);

/* ---------------------------------------------------------------------------------------
 * Learnable-query cross-attention Resampler (qwen_visual.py:95-153) — input_resampler,
 * output_resampler ("image-feature regressor") and the ViT attn_pool.
 * ------------------------------------------------------------------------------------- */
typedef struct ss_resampler_weights {
    const void* q_in;      /* [nq, E]  = ln_q(query) + pos_embed, precomputed once at load      */
    const void* pos_kv;    /* [Lkv, E] = get_abs_pos(pos_embed, Lkv) (bicubic, :23-39)           */
    const void* kv_proj;   /* [E, kv_dim] or NULL (Identity when kv_dim == E, :116-121)          */
    const void *ln_kv_w, *ln_kv_b;
    const void *in_w, *in_b;   /* nn.MultiheadAttention in_proj [3E, E], [3E] ([Q;K;V] blocks) */
    const void *out_w, *out_b; /* out_proj [E, E], [E]                                          */
    int32_t nq, embed, n_heads, kv_dim, l_kv;
    float ln_eps;
} ss_resampler_weights;
size_t ss_resampler_workspace_bytes(const ss_resampler_weights* w, int64_t batch, int dtype);
/* x [batch, l_kv, kv_dim] -> y [batch, nq, E] */
int ss_resampler_forward(const ss_resampler_weights* w, const void* x, void* y, int64_t batch, void* workspace,
                         size_t workspace_bytes, int dtype, void* stream);

/* ---------------------------------------------------------------------------------------
 * Qwen ViT-G trunk (qwen_visual.py:376-392: conv1 -> +pos -> ln_pre -> 48 blocks)
 * ------------------------------------------------------------------------------------- */
typedef struct ss_vit_layer_weights {
    const void *ln1_w, *ln1_b, *ln2_w, *ln2_b;
    const void *in_w, *in_b;   /* attn.in_proj [3*width, width], head-interleaved [h][q|k|v] (:192-199) */
    const void *out_w, *out_b;
    const void *fc_w, *fc_b, *proj_w, *proj_b;
} ss_vit_layer_weights;
typedef struct ss_vit_weights {
    const void* conv_w;   /* [width, kpad]  (conv1.weight flattened (c,dy,dx), zero padded to kpad) */
    const void* pos;      /* [tokens, width] = get_abs_pos(positional_embedding, tokens)           */
    const void *ln_pre_w, *ln_pre_b;
    const ss_vit_layer_weights* layers; /* host array[n_layers] */
    int32_t width, n_layers, n_heads, mlp_width, patch, image, kpad;
    float ln_eps;
} ss_vit_weights;
size_t ss_vit_workspace_bytes(const ss_vit_weights* w, int64_t batch, int dtype);
/* img [batch,3,image,image] (T) -> tokens [batch, (image/patch)^2, width] */
int ss_vit_forward(const ss_vit_weights* w, const void* img, void* out, int64_t batch, void* workspace,
                   size_t workspace_bytes, int dtype, void* stream);

/* The transformer trunk alone: VisualAttentionBlock layers [layer0, layer0 + n_layers) of w applied IN PLACE to
 * tokens x [batch, tokens, width] (qwen_visual.py:275-287: x += attn(ln_1(x)); x += mlp(ln_2(x))); ss_vit_forward is
 * patch-embed + pos + ln_pre + this call over all layers.  Workspace as ss_vit_workspace_bytes. */
int ss_vit_blocks(const ss_vit_weights* w, void* x, int64_t batch, int64_t tokens, int64_t layer0, int64_t n_layers,
                  void* workspace, size_t workspace_bytes, int dtype, void* stream);

/* ---------------------------------------------------------------------------------------
 * SDXL de-tokenizer half (the reference reaches these ops through diffusers:
 * src/models_ipa/adapter_modules.py:455-466, gen_george.py:60-64).
 * Activations are NHWC: [batch, H*W, C] row-major.
 * ------------------------------------------------------------------------------------- */

/* 3x3 convolution, padding 1, stride 1|2, as an implicit GEMM on the matrix cores (diffusers
 * ResnetBlock2D.conv1/conv2, Downsample2D.conv, Upsample2D.conv, conv_in/conv_out):
 * x [B,H,W,Cin] -> y [B,Ho,Wo,Cout]; w [Cout, 9*Cin] with k = (ky*3+kx)*Cin + ci (re-laid once at
 * load); upsample2x fuses F.interpolate(scale 2, nearest) of the input into the gather;
 * bias [Cout] | NULL; rowvec | NULL = per-batch vector added after the bias (ResBlock time embedding,
 * `h + temb[:, :, None, None]`): element (b, co) at rowvec[b * rowvec_stride + co] (stride 0 = Cout);
 * residual [B,Ho,Wo,Cout] | NULL is added last. Cin % 8 == 0. */
int ss_conv3x3(const void* x, const void* w, void* y, int64_t batch, int64_t H, int64_t W, int64_t Cin,
               int64_t Cout, int64_t stride, int64_t upsample2x, const void* bias, const void* rowvec,
               int64_t rowvec_stride, const void* residual, int dtype, void* stream);

/* nn.GroupNorm over NHWC (+ optional fused SiLU) — replaces diffusers' ResnetBlock2D.norm1/norm2, Transformer2DModel.norm,
 * UNet conv_norm_out and the VAE decoder's norms (reference call site: src/models_ipa/adapter_modules.py:455-466 through
 * the diffusers UNet / AutoencoderKL).  Statistics per (batch, group) are reduced in a FIXED order (block partials in fp64,
 * folded by a second pass in block order): two runs on the same input give the same bits.  stats_ws must hold
 * ss_groupnorm_workspace_bytes(batch, hw, channels, groups, dtype) bytes (final sums + block partials). */
size_t ss_groupnorm_workspace_bytes(int64_t batch, int64_t hw, int64_t channels, int64_t groups, int dtype);
int ss_groupnorm(const void* x, const void* gamma, const void* beta, void* y, void* stats_ws, int64_t batch,
                 int64_t hw, int64_t channels, int64_t groups, float eps, int fuse_silu, int dtype, void* stream);

/* ---- loss heads of the training-side forward (forward only) ------------------------------------------
 * Deterministic (fixed-order reductions, no atomics); 16-bit dtypes round where the reference's torch graph rounds.
 *
 * ss_cross_entropy_rows — the token loss of LlamaForCausalLM.forward(labels=...)
 *   (src/models_clm/modeling_llama_xformer.py:761-772: CrossEntropyLoss over logits[..., :-1, :] / labels[..., 1:]):
 *   row_loss[r] = -log_softmax(logits[r, :vocab])[labels[r]], row_valid[r] = 1; labels[r] == ignore_index -> 0 / 0.
 *   The caller passes the already shifted rows (logits row r with the label of position r + 1).
 * ss_cosine_rows — cosine_loss of src/models_clm/models.py:13-17: row_val[r] = 1 - <rec_r/|rec_r|, target_r/|target_r|>.
 * ss_masked_mean — out2[0] = sum(vals * mask) / sum(mask) (mask NULL: plain mean), out2[1] = the denominator;
 *   workspace = ss_loss_workspace_bytes(n).
 * ss_mse — F.mse_loss(a.float(), b.float(), reduction="mean") of src/models_ipa/adapter_modules.py:339 -> out2[0]. */
size_t ss_loss_workspace_bytes(int64_t n);
int ss_cross_entropy_rows(const void* logits, int64_t ld, const int64_t* labels, int64_t rows, int64_t vocab,
                          int64_t ignore_index, float* row_loss, float* row_valid, int dtype, void* stream);
int ss_cosine_rows(const void* rec, const void* target, int64_t rows, int64_t dim, float* row_val, int dtype, void* stream);
int ss_masked_mean(const float* vals, const float* mask, int64_t n, void* workspace, float* out2, void* stream);
int ss_mse(const void* a, const void* b, int64_t n, void* workspace, float* out2, int dtype, void* stream);

/* diffusers GEGLU: in [rows, 2d] = [value | gate] -> out[r,i] = value * gelu_erf(gate). */
int ss_geglu(const void* in, void* out, int64_t rows, int64_t d, int dtype, void* stream);
/* y = silu(x) (op 0) | gelu_erf(x) (op 1), element-wise (time-embedding activations). */
int ss_unary(const void* x, void* y, int64_t n, int op, int dtype, void* stream);
/* in-place row softmax of scores[rows, cols] * scale (VAE mid-block attention, head_dim 512). */
int ss_softmax_rows(void* scores, int64_t rows, int64_t cols, float scale, int dtype, void* stream);
/* out[c, r] = in[r, c] */
int ss_transpose(const void* in, void* out, int64_t rows, int64_t cols, int dtype, void* stream);
/* out[r, :] = [a[r, :c1] | b[r, :c2]]  (torch.cat([h, skip], dim=1) of the UNet up path, NHWC). */
int ss_concat_channels(const void* a, const void* b, void* out, int64_t rows, int64_t c1, int64_t c2, int dtype,
                       void* stream);
/* NCHW [B,C,HW] <-> NHWC [B,HW,cpad] (zero padded channels) for the 4-channel latents. */
int ss_layout_nchw_nhwc(const void* in, void* out, int64_t batch, int64_t channels, int64_t hw, int64_t cpad,
                        int to_nhwc, int dtype, void* stream);
/* EulerDiscreteScheduler.scale_model_input + CFG duplication: xin[0] = xin[1] = x / sqrt(sigma^2+1). */
int ss_euler_scale_dup(const void* x, void* xin, int64_t n, float sigma, int dtype, void* stream);
/* e = eps[0] + guidance * (eps[1] - eps[0]);  x += e * (sigma_next - sigma)   (adapter_modules.py:437;
 * EulerDiscreteScheduler.step, epsilon prediction). eps = [uncond; cond], n elements each. */
int ss_euler_cfg_step(void* x, const void* eps, int64_t n, float guidance, float sigma, float sigma_next,
                      int dtype, void* stream);
/* Image pre-processing on the device: the reference's `image_transform(image).to(device, dtype)`
 * (src/processer/transforms.py:4-19 = torchvision Resize [+ CenterCrop] on a PIL image, ToTensor, Normalize;
 * gen_george.py:166).  The resize is Pillow's 8-bit separable resampler restated bit-exactly: 22-bit fixed-point
 * taps, horizontal pass into a uint8 intermediate, vertical pass, then fp32 (u8/255 - mean)/std rounded to `dtype`.
 *   ss_resample_ksize / ss_resample_coeffs  HOST functions: the tap table of one axis (Pillow's precompute + normalize
 *       coefficient steps, in double): coef [out_size * ksize] int32, bounds [out_size * 2] = {first tap, tap count}.
 *   ss_image_preprocess  src uint8 HWC [H, W, 3] (device) -> dst [3, CH, CW] `dtype` = rows crop_top.., columns
 *       crop_left.. of the [OH, OW] resize (CenterCrop; pass 0, 0, OH, OW for none); dst_u8_hwc (optional, [CH, CW, 3])
 *       receives the resized uint8 pixels (what PIL would return).  coef / bounds tables are DEVICE pointers.
 *       Only source rows [first_row, first_row + n_rows) are read (the span the cropped output needs: from
 *       bounds_v[2*crop_top] to the last tap of row crop_top+CH-1); tmp_u8 holds n_rows * OW * 3 bytes. */
#define SS_FILTER_BILINEAR 0 /* torchvision's default for Resize ('clip' / 'clipa' transforms) */
#define SS_FILTER_BICUBIC 1  /* the 'sd' transform */
int ss_resample_ksize(int64_t in_size, int64_t out_size, int filter);
int ss_resample_coeffs(int64_t in_size, int64_t out_size, int filter, int32_t* coef, int32_t* bounds);
int ss_image_preprocess(const void* src_u8_hwc, int64_t H, int64_t W, void* dst_chw, void* dst_u8_hwc, int64_t OH, int64_t OW,
                        int64_t crop_top, int64_t crop_left, int64_t CH, int64_t CW, const int32_t* coef_h,
                        const int32_t* bounds_h, int ksize_h, const int32_t* coef_v, const int32_t* bounds_v, int ksize_v,
                        int64_t first_row, int64_t n_rows, void* tmp_u8, const float mean[3], const float std[3], int dtype,
                        void* stream);
/* VAE output NHWC [pixels, cpad] -> uint8 HWC: round(clamp(x/2 + 0.5, 0, 1) * 255). */
int ss_image_to_u8(const void* in, void* out_u8, int64_t pixels, int64_t cpad, int dtype, void* stream);

/* ---------------------------------------------------------------------------------------
 * Diagnostics
 * ------------------------------------------------------------------------------------- */
/* One wave executes ds_read_b64_tr_b16 with per-lane LDS offsets lane_off[64] (int32, in 16-bit
 * elements) over an 8 KB LDS image copied from src (4096 u16), and writes each lane's four 16-bit
 * results to dst[64*4].
 * Documents the transposed-read lane mapping flash_attn2_kernel relies on (tools/tr_probe.py). */
int ss_debug_tr_probe(const void* src, void* dst, const void* lane_off, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SEEDSTORY_HIP_H_ */
