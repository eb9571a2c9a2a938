// Native forward drivers for the non-LLaMA modules of the MLLM half of the hot path:
//   * learnable-query cross-attention Resampler (src/models/qwen_visual.py:95-153) — used as
//     input_resampler (256 -> 64 tokens), output_resampler (the image-feature regressor,
//     64 -> 256 tokens; src/models_clm/models.py:205) and the ViT's attn_pool;
//   * Qwen ViT-G trunk (qwen_visual.py:376-392 with VisualAttention :184-235 and
//     VisualAttentionBlock :275-287);
//   * the stand-alone logits-processor + argmax entry point.
// These are host loops over the HIP kernels of this library: one C call per module forward
// instead of hundreds of framework-level op dispatches.
#include "ss_common.h"
#include "ss_sample.h"

namespace ss {
int gemm_dev(const void* A, const void* W, void* C, int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldw,
             int64_t ldc, const void* bias, const void* residual, int64_t ldr, int epi, int dtype, hipStream_t s);

template <typename T>
__global__ __launch_bounds__(1024) void imgproc_argmax_kernel(T* logits, int vocab, const int32_t* last_id,
                                                              const int32_t* img_ids, int n_img_ids, int32_t* tok) {
    __shared__ float sv[16];
    __shared__ int si[16];
    const int t = imgproc_argmax_block<T>(logits, vocab, *last_id, img_ids, n_img_ids, sv, si);
    if (threadIdx.x == 0) *tok = t;
}

template <typename T>
int imgproc_argmax_launch(void* logits, int64_t vocab, const int32_t* last_id, const int32_t* img_ids, int64_t n,
                          int32_t* tok, hipStream_t s) {
    hipLaunchKernelGGL(imgproc_argmax_kernel<T>, dim3(1), dim3(1024), 0, s, (T*)logits, (int)vocab, last_id, img_ids,
                       (int)n, tok);
    SS_LAUNCH_CHECK("imgproc_argmax");
    return SS_OK;
}

struct Bump {
    char* p; size_t off, cap;
    void* take(size_t bytes) { void* r = p + off; off += (bytes + 255) / 256 * 256; return r; }
};
}  // namespace ss

using namespace ss;

extern "C" {

int ss_imgproc_argmax(void* logits, int64_t vocab, const int32_t* last_id_dev, const int32_t* img_ids,
                      int64_t n_img_ids, int32_t* token_out_dev, int dtype, void* stream) {
    SS_REQUIRE(logits && last_id_dev && token_out_dev && vocab > 0, "imgproc_argmax: bad arguments");
    return SS_DISPATCH(dtype, imgproc_argmax_launch, logits, vocab, last_id_dev, img_ids, n_img_ids, token_out_dev,
                       (hipStream_t)stream);
}

// ---------------------------------------------------------------------------------------------
// Resampler
// ---------------------------------------------------------------------------------------------
size_t ss_resampler_workspace_bytes(const ss_resampler_weights* w, int64_t batch, int dtype) {
    if (!w) return 0;
    const size_t e = dtype_size(dtype), E = w->embed, rows = (size_t)batch * w->l_kv, q = (size_t)w->nq;
    size_t n = 0;
    auto add = [&](size_t b) { n += (b + 255) / 256 * 256; };
    add(rows * E * e);               // xk (kv_proj output)
    add(rows * E * e);               // xln
    add(rows * E * e);               // kin
    add(rows * E * e);               // K
    add(rows * E * e);               // V
    add(q * E * e);                  // Q
    add((size_t)batch * q * E * e);  // ctx
    return n + 256;
}

int ss_resampler_forward(const ss_resampler_weights* w, const void* x, void* y, int64_t batch, void* workspace,
                         size_t workspace_bytes, int dtype, void* stream) {
    SS_REQUIRE(w && x && y && workspace && batch > 0, "resampler_forward: bad arguments");
    SS_REQUIRE(workspace_bytes >= ss_resampler_workspace_bytes(w, batch, dtype), "resampler_forward: workspace too small");
    SS_REQUIRE(w->embed % w->n_heads == 0, "resampler_forward: embed %% heads != 0");
    const size_t e = dtype_size(dtype);
    const int64_t E = w->embed, L = w->l_kv, rows = batch * L, nq = w->nq;
    const int64_t hd = E / w->n_heads;
    hipStream_t s = (hipStream_t)stream;
    Bump b{(char*)workspace, 0, workspace_bytes};
    void* xk = b.take(rows * E * e);
    void* xln = b.take(rows * E * e);
    void* kin = b.take(rows * E * e);
    void* K = b.take(rows * E * e);
    void* Vv = b.take(rows * E * e);
    void* Q = b.take(nq * E * e);
    void* ctx = b.take(batch * nq * E * e);
    int rc;
    const void* src = x;
    if (w->kv_proj) {  // x = self.kv_proj(x)  (:142)
        if ((rc = gemm_dev(x, w->kv_proj, xk, rows, E, w->kv_dim, w->kv_dim, w->kv_dim, E, nullptr, nullptr, 0,
                           SS_EPI_NONE, dtype, s)))
            return rc;
        src = xk;
    }
    // x = self.ln_kv(x)  (:143);  key input = x + pos_embed (:148)
    if ((rc = ss_layernorm(src, w->ln_kv_w, w->ln_kv_b, xln, rows, E, w->ln_eps, dtype, stream))) return rc;
    if ((rc = ss_add_bcast(xln, w->pos_kv, kin, batch, L, E, L * E, dtype, stream))) return rc;
    // nn.MultiheadAttention in-projections ([Q;K;V] row blocks of in_proj_weight)
    const char* in_w = (const char*)w->in_w;
    const char* in_b = (const char*)w->in_b;
    if ((rc = gemm_dev(w->q_in, in_w, Q, nq, E, E, E, E, E, in_b, nullptr, 0, SS_EPI_BIAS, dtype, s))) return rc;
    if ((rc = gemm_dev(kin, in_w + (size_t)E * E * e, K, rows, E, E, E, E, E, in_b + (size_t)E * e, nullptr, 0,
                       SS_EPI_BIAS, dtype, s)))
        return rc;
    if ((rc = gemm_dev(xln, in_w + (size_t)2 * E * E * e, Vv, rows, E, E, E, E, E, in_b + (size_t)2 * E * e, nullptr, 0,
                       SS_EPI_BIAS, dtype, s)))
        return rc;
    // softmax(q k^T / sqrt(hd)) v per head; the same Q for every batch element (q batch stride 0)
    if ((rc = ss_attention(Q, K, Vv, ctx, batch, w->n_heads, nq, L, hd, 0, hd, E, L * E, hd, E, L * E, hd, E, nq * E, hd,
                           E, 1.0f / sqrtf((float)hd), 0, dtype, stream)))
        return rc;
    return gemm_dev(ctx, w->out_w, y, batch * nq, E, E, E, E, E, w->out_b, nullptr, 0, SS_EPI_BIAS, dtype, s);
}

// ---------------------------------------------------------------------------------------------
// ViT trunk
// ---------------------------------------------------------------------------------------------
// VisualAttentionBlock layers [l0, l0 + nl) in place on x [batch, L, width]; y/qkv/ctx/hm are scratch
static int vit_blocks(const ss_vit_weights* w, void* x, void* y, void* qkv, void* ctx, void* hm, int64_t batch, int64_t L,
                      int64_t l0, int64_t nl, int dtype, void* stream) {
    const size_t e = dtype_size(dtype);
    const int64_t rows = batch * L, Wd = w->width, hd = Wd / w->n_heads;
    hipStream_t s = (hipStream_t)stream;
    int rc;
    for (int64_t l = l0; l < l0 + nl; ++l) {
        const ss_vit_layer_weights& Lw = w->layers[l];
        if ((rc = ss_layernorm(x, Lw.ln1_w, Lw.ln1_b, y, rows, Wd, w->ln_eps, dtype, stream))) return rc;
        if ((rc = gemm_dev(y, Lw.in_w, qkv, rows, 3 * Wd, Wd, Wd, Wd, 3 * Wd, Lw.in_b, nullptr, 0, SS_EPI_BIAS, dtype, s)))
            return rc;
        // per token the in_proj output is [head][q(hd) | k(hd) | v(hd)]  (:192-199)
        const char* q = (const char*)qkv;
        if ((rc = ss_attention(q, q + (size_t)hd * e, q + (size_t)2 * hd * e, ctx, batch, w->n_heads, L, L, hd,
                               L * 3 * Wd, 3 * hd, 3 * Wd, L * 3 * Wd, 3 * hd, 3 * Wd, L * 3 * Wd, 3 * hd, 3 * Wd, L * Wd,
                               hd, Wd, 1.0f / sqrtf((float)hd), 0, dtype, stream)))
            return rc;
        if ((rc = gemm_dev(ctx, Lw.out_w, x, rows, Wd, Wd, Wd, Wd, Wd, Lw.out_b, x, Wd, SS_EPI_BIAS | SS_EPI_RESIDUAL,
                           dtype, s)))
            return rc;
        if ((rc = ss_layernorm(x, Lw.ln2_w, Lw.ln2_b, y, rows, Wd, w->ln_eps, dtype, stream))) return rc;
        if ((rc = gemm_dev(y, Lw.fc_w, hm, rows, w->mlp_width, Wd, Wd, Wd, w->mlp_width, Lw.fc_b, nullptr, 0,
                           SS_EPI_BIAS | SS_EPI_GELU, dtype, s)))
            return rc;
        if ((rc = gemm_dev(hm, Lw.proj_w, x, rows, Wd, w->mlp_width, w->mlp_width, w->mlp_width, Wd, Lw.proj_b, x, Wd,
                           SS_EPI_BIAS | SS_EPI_RESIDUAL, dtype, s)))
            return rc;
    }
    return SS_OK;
}

size_t ss_vit_workspace_bytes(const ss_vit_weights* w, int64_t batch, int dtype) {
    if (!w) return 0;
    const size_t e = dtype_size(dtype);
    const size_t G = w->image / w->patch, rows = (size_t)batch * G * G, Wd = w->width;
    size_t n = 0;
    auto add = [&](size_t b) { n += (b + 255) / 256 * 256; };
    add(rows * w->kpad * e);       // patches
    add(rows * Wd * e);            // x
    add(rows * Wd * e);            // y
    add(rows * 3 * Wd * e);        // qkv
    add(rows * Wd * e);            // ctx
    add(rows * w->mlp_width * e);  // mlp hidden
    return n + 256;
}

int ss_vit_forward(const ss_vit_weights* w, const void* img, void* out, int64_t batch, void* workspace,
                   size_t workspace_bytes, int dtype, void* stream) {
    SS_REQUIRE(w && img && out && workspace && batch > 0, "vit_forward: bad arguments");
    SS_REQUIRE(workspace_bytes >= ss_vit_workspace_bytes(w, batch, dtype), "vit_forward: workspace too small");
    SS_REQUIRE(w->width % w->n_heads == 0 && w->image % w->patch == 0, "vit_forward: bad geometry");
    const size_t e = dtype_size(dtype);
    const int64_t G = w->image / w->patch, L = G * G, rows = batch * L, Wd = w->width, hd = Wd / w->n_heads;
    hipStream_t s = (hipStream_t)stream;
    Bump b{(char*)workspace, 0, workspace_bytes};
    void* patches = b.take(rows * w->kpad * e);
    void* x = b.take(rows * Wd * e);
    void* y = b.take(rows * Wd * e);
    void* qkv = b.take(rows * 3 * Wd * e);
    void* ctx = b.take(rows * Wd * e);
    void* hm = b.take(rows * w->mlp_width * e);
    int rc;
    // conv1 (k = stride = patch, no bias) as im2col + GEMM (:382), + interpolated pos (:387), ln_pre (:389)
    if ((rc = ss_im2col_patch(img, patches, batch, w->image, w->patch, w->kpad, dtype, stream))) return rc;
    if ((rc = gemm_dev(patches, w->conv_w, y, rows, Wd, w->kpad, w->kpad, w->kpad, Wd, nullptr, nullptr, 0, SS_EPI_NONE,
                       dtype, s)))
        return rc;
    if ((rc = ss_add_bcast(y, w->pos, y, batch, L, Wd, L * Wd, dtype, stream))) return rc;
    if ((rc = ss_layernorm(y, w->ln_pre_w, w->ln_pre_b, x, rows, Wd, w->ln_eps, dtype, stream))) return rc;
    if ((rc = vit_blocks(w, x, y, qkv, ctx, hm, batch, L, 0, w->n_layers, dtype, stream))) return rc;
    SS_HIP(hipMemcpyAsync(out, x, rows * Wd * e, hipMemcpyDeviceToDevice, s));
    return SS_OK;
}

int ss_vit_blocks(const ss_vit_weights* w, void* x, int64_t batch, int64_t tokens, int64_t layer0, int64_t n_layers,
                  void* workspace, size_t workspace_bytes, int dtype, void* stream) {
    SS_REQUIRE(w && x && workspace && batch > 0 && tokens > 0, "vit_blocks: bad arguments");
    SS_REQUIRE(layer0 >= 0 && n_layers >= 0 && layer0 + n_layers <= w->n_layers, "vit_blocks: layer range");
    SS_REQUIRE(w->width % w->n_heads == 0, "vit_blocks: bad geometry");
    const size_t e = dtype_size(dtype);
    const size_t rows = (size_t)batch * tokens, Wd = w->width;
    auto up = [](size_t b) { return (b + 255) / 256 * 256; };
    const size_t need = up(rows * Wd * e) + up(rows * 3 * Wd * e) + up(rows * Wd * e) + up(rows * w->mlp_width * e);
    SS_REQUIRE(workspace_bytes >= need, "vit_blocks: workspace too small");
    Bump b{(char*)workspace, 0, workspace_bytes};
    void* y = b.take(rows * Wd * e);
    void* qkv = b.take(rows * 3 * Wd * e);
    void* ctx = b.take(rows * Wd * e);
    void* hm = b.take(rows * w->mlp_width * e);
    return vit_blocks(w, x, y, qkv, ctx, hm, batch, tokens, layer0, n_layers, dtype, stream);
}

}  // extern "C"
