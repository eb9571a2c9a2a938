// Native LLaMA-2 decoder engine for the SEED-Story MLLM (replaces, for inference,
// LlamaModel.forward / LlamaForCausalLM.forward / prepare_inputs_for_generation of
// src/models_clm/modeling_llama_xformer.py:532-852 and the HF greedy loop around them).
//
//  * KV cache lives in one preallocated slab  K,V : [n_layers][n_heads][cache_cap][hd]  — the
//    reference's per-layer (k, v) [1, n_heads, len, hd] tuples (keys post-RoPE) are *views* of
//    it; no per-token torch.cat (:239-242 copies O(S) per token per layer).
//  * prefill / continuation (q_len = M rows against the cached prefix): host loop over layers,
//    MFMA GEMMs + flash attention with the bottom-right causal mask.
//  * decode: ONE token = sample -> embed -> 32 x {GEMV(qkv, fused RMSNorm) -> RoPE+append ->
//    split-KV attention -> GEMV(o)+residual -> GEMV(gate|up, fused RMSNorm, SiLU*mul) ->
//    GEMV(down)+residual} -> final norm -> GEMV(lm_head), captured once into a hipGraph.  All
//    per-token scalars (kv_len, rope position, last token id, EOS flag) live in device memory, so
//    the graph replays without host round trips; the reference syncs the host every token
//    (`.item()` in generation.py:22, EOS check in HF).
#include <string.h>

#include <vector>

#include "ss_common.h"
#include "ss_sample.h"

namespace ss {

int gemv_dev(const void* W, const void* x, void* y, int64_t N, int64_t K, const void* norm_w, float eps,
             const void* bias, const void* residual, int epi, const int32_t* done_flag, int dtype, hipStream_t s);
int gemm_dev(const void* A, const void* W, void* C, int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldw,
             int64_t ldc, const void* bias, const void* residual, int64_t ldr, int epi, int dtype, hipStream_t s);
int rope_kv_append_dev(const void* qkv, void* q_out, void* kc, void* vc, const void* cos_t, const void* sin_t,
                       const int32_t* pos_ids, int64_t M, int64_t n_heads, int64_t hd, const int32_t* kv_start_dev,
                       int64_t cache_cap, int dtype, hipStream_t s);
int attn_decode_fused_dev(const void* qkv_raw, void* kc, void* vc, const void* cos_t, const void* sin_t, void* out,
                          void* ws, const int32_t* kv_len_dev, const int32_t* pos_dev, const int32_t* done_flag,
                          int64_t n_heads, int64_t hd, int64_t cache_cap, int dtype, hipStream_t s);

// device state words
enum { ST_KV_LEN = 0, ST_POS = 1, ST_NGEN = 2, ST_DONE = 3, ST_LAST = 4, ST_NFORCED = 5, ST_LIMIT = 6, ST_EOS = 7 };

// ---- engine kernels ---------------------------------------------------------------------------

// sample -> (forced?) -> append -> EOS/limit check -> embed.  One block.
template <typename T>
__global__ __launch_bounds__(1024) void sample_embed_kernel(T* logits, int vocab, int32_t* st,
                                                            const int32_t* __restrict__ img_ids, int n_img_ids,
                                                            const int32_t* __restrict__ forced, int32_t* gen_ids,
                                                            const T* __restrict__ embed, T* x, int hidden) {
    __shared__ float sv[16];
    __shared__ int si[16];
    if (st[ST_DONE]) return;
    int tok = imgproc_argmax_block<T>(logits, vocab, st[ST_LAST], img_ids, n_img_ids, sv, si);
    const int n = st[ST_NGEN];
    if (n < st[ST_NFORCED]) tok = forced[n];
    const bool stop = (tok == st[ST_EOS]) || (n + 1 >= st[ST_LIMIT]);
    __syncthreads();
    if (threadIdx.x == 0) {
        gen_ids[n] = tok;
        st[ST_LAST] = tok;
        st[ST_NGEN] = n + 1;
        if (stop) st[ST_DONE] = 1;
    }
    if (stop) return;
    constexpr int V = Tr<T>::kVec;
    for (int p = threadIdx.x; p < hidden / V; p += blockDim.x)
        st16(x + (int64_t)p * V, ld16(embed + (int64_t)tok * hidden + (int64_t)p * V));
}

// final RMSNorm of the single decode row: writes the fixed lm_head input buffer AND the
// hidden-state ring row (n_gen - 1), then advances kv_len / pos.  One block of 256.
template <typename T>
__global__ __launch_bounds__(256) void final_norm_advance_kernel(const T* __restrict__ x, const T* __restrict__ w,
                                                                 T* xn, T* hid_rows, int32_t* st, int hidden,
                                                                 float eps) {
    constexpr int V = Tr<T>::kVec;
    __shared__ float red[16];
    if (st[ST_DONE]) return;
    const int npack = hidden / V;
    float ssq = 0.f;
    for (int p = threadIdx.x; p < npack; p += 256) {
        float f[V];
        unpack<T>(ld16(x + (int64_t)p * V), f);
#pragma unroll
        for (int j = 0; j < V; ++j) ssq = fmaf(f[j], f[j], ssq);
    }
    const float rstd = 1.0f / sqrtf(block_sum(ssq, red) / (float)hidden + eps);
    T* row = hid_rows + (int64_t)(st[ST_NGEN] - 1) * hidden;
    for (int p = threadIdx.x; p < npack; p += 256) {
        float f[V], g[V];
        unpack<T>(ld16(x + (int64_t)p * V), f);
        unpack<T>(ld16(w + (int64_t)p * V), g);
#pragma unroll
        for (int j = 0; j < V; ++j) f[j] = g[j] * Tr<T>::rnd(f[j] * rstd);
        const uint4 o = pack<T>(f);
        st16(xn + (int64_t)p * V, o);
        st16(row + (int64_t)p * V, o);
    }
    __syncthreads();
    if (threadIdx.x == 0) { st[ST_KV_LEN] += 1; st[ST_POS] += 1; }
}

__global__ void set_state_kernel(int32_t* st, int idx0, int v0, int idx1, int v1) {
    if (threadIdx.x == 0) { st[idx0] = v0; if (idx1 >= 0) st[idx1] = v1; }
}

// dst[h][i][:] = src[h][keep[i]][:]  (one head-plane of the cache -> packed scratch)
template <typename T>
__global__ __launch_bounds__(256) void kv_gather_kernel(const T* __restrict__ src, T* __restrict__ dst,
                                                        const int32_t* __restrict__ keep, int n_keep, int cap,
                                                        int hd, int dst_cap) {
    constexpr int V = Tr<T>::kVec;
    const int h = blockIdx.y;
    const int ppr = hd / V;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < (int64_t)n_keep * ppr;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int r = (int)(i / ppr), p = (int)(i % ppr);
        const int srow = keep ? keep[r] : r;
        st16(dst + ((int64_t)h * dst_cap + r) * hd + p * V, ld16(src + ((int64_t)h * cap + srow) * hd + p * V));
    }
}

int rmsnorm_rows(const void* x, const void* w, void* y, int64_t rows, int64_t cols, float eps, int dtype,
                 hipStream_t s) {
    return ss_rmsnorm(x, w, y, rows, cols, eps, dtype, (void*)s);
}

}  // namespace ss

using namespace ss;

struct ss_llama {
    ss_llama_config cfg;
    ss_llama_weights w;
    std::vector<ss_llama_layer_weights> layers;
    int hd;
    int64_t max_rows;
    size_t esz;
    // device buffers (carved from the caller's workspace)
    char *kc, *vc;           // [L][H][cap][hd]
    int32_t* state;          // [8]
    int32_t* gen_ids;        // [max_new]
    int32_t* forced;         // [max_new]
    int32_t* img_ids;        // [n_img_ids]
    char* hid_rows;          // [max_new][hidden]
    char* logits;            // [vocab]
    char *x, *xn, *qkv, *q, *attn, *gu, *hm;  // activations ([max_rows][..]); decode uses row 0
    float* attn_ws;
    // host mirrors
    int64_t kv_len, pos;
    hipStream_t cap_stream;
    hipGraph_t graph;
    hipGraphExec_t graph_exec;
    bool graph_ready;
    int32_t* pinned;         // 8 ints of pinned host memory for state read-back
};

static size_t align_up(size_t v, size_t a = 256) { return (v + a - 1) / a * a; }

struct Carver {
    char* base; size_t off; size_t cap;
    char* take(size_t bytes) { char* p = base ? base + off : nullptr; off += align_up(bytes); return p; }
};

static void carve(ss_llama* h, Carver& c) {
    const ss_llama_config& g = h->cfg;
    const size_t e = h->esz;
    const size_t H = g.hidden, I = g.inter, R = (size_t)h->max_rows;
    const size_t kvb = (size_t)g.n_layers * g.n_heads * g.cache_cap * h->hd * e;
    h->kc = c.take(kvb);
    h->vc = c.take(kvb);
    h->state = (int32_t*)c.take(8 * sizeof(int32_t));
    h->gen_ids = (int32_t*)c.take((size_t)g.max_new * sizeof(int32_t));
    h->forced = (int32_t*)c.take((size_t)g.max_new * sizeof(int32_t));
    h->img_ids = (int32_t*)c.take((size_t)(g.n_img_ids > 0 ? g.n_img_ids : 1) * sizeof(int32_t));
    h->hid_rows = c.take((size_t)g.max_new * H * e);
    h->logits = c.take((size_t)g.vocab * e);
    h->x = c.take(R * H * e);
    h->xn = c.take(R * H * e);
    h->qkv = c.take(R * 3 * H * e);
    h->q = c.take(R * H * e);
    h->attn = c.take(R * H * e);
    h->gu = c.take(R * 2 * I * e);
    h->hm = c.take(R * I * e);
    h->attn_ws = (float*)c.take(ss_attn_decode_workspace_bytes(g.n_heads, h->hd));
}

// one decode token (sample+forward); eager or under stream capture.  `ev` (optional) receives an
// event before/after every launch class for profiling.
struct ProfSink {
    std::vector<hipEvent_t> ev;
    std::vector<int> cls;
    hipStream_t s;
    void mark(int c) {
        hipEvent_t e;
        hipEventCreate(&e);
        hipEventRecord(e, s);
        ev.push_back(e);
        cls.push_back(c);
    }
};

static int decode_token(ss_llama* h, hipStream_t s, ProfSink* prof) {
    const ss_llama_config& g = h->cfg;
    const int dt = g.dtype;
    const int H = g.hidden, I = g.inter, hd = h->hd;
    const size_t e = h->esz;
    const int32_t* done = h->state + ST_DONE;
    const size_t plane = (size_t)g.n_heads * g.cache_cap * hd * e;
#define MARK(c) do { if (prof) prof->mark(c); } while (0)
    MARK(-1);
    if (dt == SS_BF16)
        hipLaunchKernelGGL(sample_embed_kernel<bf16_t>, dim3(1), dim3(1024), 0, s, (bf16_t*)h->logits, g.vocab, h->state,
                           h->img_ids, g.n_img_ids, h->forced, h->gen_ids, (const bf16_t*)h->w.embed, (bf16_t*)h->x, H);
    else if (dt == SS_F32)
        hipLaunchKernelGGL(sample_embed_kernel<float>, dim3(1), dim3(1024), 0, s, (float*)h->logits, g.vocab, h->state,
                           h->img_ids, g.n_img_ids, h->forced, h->gen_ids, (const float*)h->w.embed, (float*)h->x, H);
    else
        hipLaunchKernelGGL(sample_embed_kernel<f16_t>, dim3(1), dim3(1024), 0, s, (f16_t*)h->logits, g.vocab, h->state,
                           h->img_ids, g.n_img_ids, h->forced, h->gen_ids, (const f16_t*)h->w.embed, (f16_t*)h->x, H);
    SS_LAUNCH_CHECK("sample_embed");
    MARK(3);
    for (int l = 0; l < g.n_layers; ++l) {
        const ss_llama_layer_weights& L = h->layers[l];
        char* kc = h->kc + (size_t)l * plane;
        char* vc = h->vc + (size_t)l * plane;
        int rc;
        MARK(-1);
        rc = gemv_dev(L.wqkv, h->x, h->qkv, 3 * H, H, L.ln1, g.rms_eps, nullptr, nullptr, SS_EPI_NONE, done, dt, s);
        if (rc) return rc;
        MARK(0);
        // RoPE(q,k) + KV append + split-KV attention in one kernel (+ the split merge)
        rc = attn_decode_fused_dev(h->qkv, kc, vc, h->w.rope_cos, h->w.rope_sin, h->attn, h->attn_ws,
                                   h->state + ST_KV_LEN, h->state + ST_POS, done, g.n_heads, hd, g.cache_cap, dt, s);
        if (rc) return rc;
        MARK(1);
        rc = gemv_dev(L.wo, h->attn, h->xn, H, H, nullptr, 0.f, nullptr, h->x, SS_EPI_RESIDUAL, done, dt, s);
        if (rc) return rc;
        MARK(0);
        rc = gemv_dev(L.wgu, h->xn, h->hm, I, H, L.ln2, g.rms_eps, nullptr, nullptr, SS_EPI_SILU_MUL, done, dt, s);
        if (rc) return rc;
        MARK(0);
        rc = gemv_dev(L.wdown, h->hm, h->x, H, I, nullptr, 0.f, nullptr, h->xn, SS_EPI_RESIDUAL, done, dt, s);
        if (rc) return rc;
        MARK(2);
    }
    if (dt == SS_BF16)
        hipLaunchKernelGGL(final_norm_advance_kernel<bf16_t>, dim3(1), dim3(256), 0, s, (const bf16_t*)h->x,
                           (const bf16_t*)h->w.final_norm, (bf16_t*)h->xn, (bf16_t*)h->hid_rows, h->state, H, g.rms_eps);
    else if (dt == SS_F32)
        hipLaunchKernelGGL(final_norm_advance_kernel<float>, dim3(1), dim3(256), 0, s, (const float*)h->x,
                           (const float*)h->w.final_norm, (float*)h->xn, (float*)h->hid_rows, h->state, H, g.rms_eps);
    else
        hipLaunchKernelGGL(final_norm_advance_kernel<f16_t>, dim3(1), dim3(256), 0, s, (const f16_t*)h->x,
                           (const f16_t*)h->w.final_norm, (f16_t*)h->xn, (f16_t*)h->hid_rows, h->state, H, g.rms_eps);
    SS_LAUNCH_CHECK("final_norm_advance");
    MARK(3);
    int rc = gemv_dev(h->w.lm_head, h->xn, h->logits, g.vocab, H, nullptr, 0.f, nullptr, nullptr, SS_EPI_NONE, done,
                      dt, s);
    if (rc) return rc;
    MARK(0);
#undef MARK
    return SS_OK;
}

static int read_state(ss_llama* h, hipStream_t s) {
    SS_HIP(hipMemcpyAsync(h->pinned, h->state, 8 * sizeof(int32_t), hipMemcpyDeviceToHost, s));
    SS_HIP(hipStreamSynchronize(s));
    return SS_OK;
}

extern "C" {

size_t ss_llama_workspace_bytes(const ss_llama_config* cfg, int64_t max_prefill_rows) {
    if (!cfg) return 0;
    ss_llama tmp;
    tmp.cfg = *cfg;
    tmp.hd = cfg->hidden / cfg->n_heads;
    tmp.max_rows = max_prefill_rows < 1 ? 1 : max_prefill_rows;
    tmp.esz = dtype_size(cfg->dtype);
    Carver c{nullptr, 0, 0};
    carve(&tmp, c);
    return c.off + 256;
}

int ss_llama_create(const ss_llama_config* cfg, const ss_llama_weights* w, void* workspace, size_t workspace_bytes,
                    int64_t max_prefill_rows, const int32_t* host_img_ids, ss_llama** out) {
    SS_REQUIRE(cfg && w && workspace && out, "llama_create: null argument");
    SS_REQUIRE(cfg->hidden % cfg->n_heads == 0, "llama_create: hidden %% n_heads != 0");
    SS_REQUIRE(cfg->n_img_ids <= 1024 && cfg->max_new > 0 && cfg->cache_cap > 0, "llama_create: bad config");
    int32_t info[4];
    int rc = ss_device_info(info);
    if (rc) return rc;
    if (!info[1]) { set_error("llama_create: device is not gfx950"); return SS_EHIP; }
    ss_llama* h = new ss_llama();
    h->cfg = *cfg;
    h->w = *w;
    h->layers.assign(w->layers, w->layers + cfg->n_layers);
    h->w.layers = h->layers.data();
    h->hd = cfg->hidden / cfg->n_heads;
    h->max_rows = max_prefill_rows < 1 ? 1 : max_prefill_rows;
    h->esz = dtype_size(cfg->dtype);
    const size_t base = (size_t)workspace;
    const size_t skew = align_up(base) - base;
    Carver c{(char*)workspace + skew, 0, workspace_bytes - skew};
    carve(h, c);
    if (c.off > c.cap) {
        set_error("llama_create: workspace too small (%zu < %zu)", workspace_bytes, c.off + skew);
        delete h;
        return SS_ENOMEM;
    }
    h->kv_len = 0; h->pos = 0;
    h->graph_ready = false; h->graph = nullptr; h->graph_exec = nullptr;
    hipError_t e = hipStreamCreateWithFlags(&h->cap_stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipHostMalloc((void**)&h->pinned, 128, hipHostMallocDefault);
    if (e == hipSuccess) e = hipMemset(h->state, 0, 8 * sizeof(int32_t));
    if (e == hipSuccess && cfg->n_img_ids > 0)
        e = hipMemcpy(h->img_ids, host_img_ids, cfg->n_img_ids * sizeof(int32_t), hipMemcpyHostToDevice);
    if (e != hipSuccess) { rc = check_hip(e, "llama_create"); delete h; return rc; }
    *out = h;
    return SS_OK;
}

void ss_llama_destroy(ss_llama* h) {
    if (!h) return;
    if (h->graph_exec) hipGraphExecDestroy(h->graph_exec);
    if (h->graph) hipGraphDestroy(h->graph);
    if (h->cap_stream) hipStreamDestroy(h->cap_stream);
    if (h->pinned) hipHostFree(h->pinned);
    delete h;
}

void* ss_llama_buffer(ss_llama* h, int which) {
    if (!h) return nullptr;
    switch (which) {
        case 0: return h->kc;
        case 1: return h->vc;
        case 2: return h->gen_ids;
        case 3: return h->hid_rows;
        case 4: return h->logits;
        case 5: return h->state;
        default: return nullptr;
    }
}

int ss_llama_set_lengths(ss_llama* h, int64_t kv_len, int64_t pos, void* stream) {
    SS_REQUIRE(h && kv_len >= 0 && kv_len <= h->cfg.cache_cap && pos >= 0 && pos < h->cfg.max_pos,
               "llama_set_lengths: out of range (kv_len=%lld pos=%lld)", (long long)kv_len, (long long)pos);
    hipLaunchKernelGGL(set_state_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, h->state, (int)ST_KV_LEN,
                       (int)kv_len, (int)ST_POS, (int)pos);
    SS_LAUNCH_CHECK("set_state");
    h->kv_len = kv_len;
    h->pos = pos;
    return SS_OK;
}

int ss_llama_get_lengths(ss_llama* h, int64_t* kv_len, int64_t* pos) {
    SS_REQUIRE(h, "llama_get_lengths: null handle");
    if (kv_len) *kv_len = h->kv_len;
    if (pos) *pos = h->pos;
    return SS_OK;
}

int ss_llama_kv_gather(ss_llama* h, const int32_t* keep_idx_dev, int64_t n_keep, void* stream) {
    SS_REQUIRE(h && keep_idx_dev && n_keep >= 0 && n_keep <= h->kv_len, "llama_kv_gather: bad arguments");
    const ss_llama_config& g = h->cfg;
    const size_t e = h->esz;
    // scratch = qkv activation buffer: [max_rows][3*hidden] elements >= n_heads * n_keep * hd = n_keep * hidden
    SS_REQUIRE(n_keep <= 3 * h->max_rows, "llama_kv_gather: n_keep %lld exceeds scratch (3 x %lld rows)",
               (long long)n_keep, (long long)h->max_rows);
    hipStream_t s = (hipStream_t)stream;
    const size_t plane = (size_t)g.n_heads * g.cache_cap * h->hd * e;
    const int V = g.dtype == SS_F32 ? 4 : 8;
    dim3 grid((unsigned)cdiv(n_keep * (h->hd / V), 256) > 0 ? (unsigned)cdiv(n_keep * (h->hd / V), 256) : 1,
              (unsigned)g.n_heads);
    if (n_keep > 0) {
        for (int l = 0; l < g.n_layers; ++l) {
            for (int kv = 0; kv < 2; ++kv) {
                char* plane_p = (kv ? h->vc : h->kc) + (size_t)l * plane;
#define GATHER(T)                                                                                                  \
    hipLaunchKernelGGL(kv_gather_kernel<T>, grid, dim3(256), 0, s, (const T*)plane_p, (T*)h->qkv, keep_idx_dev,     \
                       (int)n_keep, g.cache_cap, h->hd, (int)n_keep);                                               \
    hipLaunchKernelGGL(kv_gather_kernel<T>, grid, dim3(256), 0, s, (const T*)h->qkv, (T*)plane_p,                   \
                       (const int32_t*)nullptr, (int)n_keep, (int)n_keep, h->hd, g.cache_cap)
                if (g.dtype == SS_BF16) { GATHER(bf16_t); }
                else if (g.dtype == SS_F32) { GATHER(float); }
                else { GATHER(f16_t); }
#undef GATHER
                SS_LAUNCH_CHECK("kv_gather");
            }
        }
    }
    return ss_llama_set_lengths(h, n_keep, h->pos, stream);
}

int ss_llama_prefill(ss_llama* h, const void* embeds, int64_t M, const int32_t* pos_ids, void* hidden_out,
                     void* stream) {
    SS_REQUIRE(h && embeds && M > 0, "llama_prefill: bad arguments");
    SS_REQUIRE(M <= h->max_rows, "llama_prefill: M=%lld exceeds max_prefill_rows=%lld", (long long)M,
               (long long)h->max_rows);
    SS_REQUIRE(h->kv_len + M <= h->cfg.cache_cap, "llama_prefill: KV cache overflow (%lld + %lld > %d)",
               (long long)h->kv_len, (long long)M, h->cfg.cache_cap);
    SS_REQUIRE(pos_ids || h->pos + M <= h->cfg.max_pos, "llama_prefill: position overflow");
    const ss_llama_config& g = h->cfg;
    hipStream_t s = (hipStream_t)stream;
    const int dt = g.dtype;
    const int64_t H = g.hidden, I = g.inter;
    const int hd = h->hd;
    const size_t e = h->esz;
    const size_t plane = (size_t)g.n_heads * g.cache_cap * hd * e;
    const int64_t kv0 = h->kv_len, kv1 = h->kv_len + M;
    int rc;
    SS_HIP(hipMemcpyAsync(h->x, embeds, (size_t)M * H * e, hipMemcpyDeviceToDevice, s));
    for (int l = 0; l < g.n_layers; ++l) {
        const ss_llama_layer_weights& L = h->layers[l];
        char* kc = h->kc + (size_t)l * plane;
        char* vc = h->vc + (size_t)l * plane;
        if ((rc = rmsnorm_rows(h->x, L.ln1, h->xn, M, H, g.rms_eps, dt, s))) return rc;
        if ((rc = gemm_dev(h->xn, L.wqkv, h->qkv, M, 3 * H, H, H, H, 3 * H, nullptr, nullptr, 0, SS_EPI_NONE, dt, s)))
            return rc;
        if ((rc = ss_rope_kv_append(h->qkv, h->q, kc, vc, h->w.rope_cos, h->w.rope_sin, pos_ids, h->pos, M, g.n_heads,
                                    hd, kv0, g.cache_cap, dt, stream)))
            return rc;
        if ((rc = ss_attention(h->q, kc, vc, h->attn, 1, g.n_heads, M, kv1, hd, 0, hd, H, 0, (int64_t)g.cache_cap * hd,
                               hd, 0, (int64_t)g.cache_cap * hd, hd, 0, hd, H, 1.0f / sqrtf((float)hd), 1, dt, stream)))
            return rc;
        if ((rc = gemm_dev(h->attn, L.wo, h->x, M, H, H, H, H, H, nullptr, h->x, H, SS_EPI_RESIDUAL, dt, s))) return rc;
        if ((rc = rmsnorm_rows(h->x, L.ln2, h->xn, M, H, g.rms_eps, dt, s))) return rc;
        if ((rc = gemm_dev(h->xn, L.wgu, h->gu, M, 2 * I, H, H, H, 2 * I, nullptr, nullptr, 0, SS_EPI_NONE, dt, s)))
            return rc;
        if ((rc = ss_silu_mul(h->gu, h->hm, M, I, dt, stream))) return rc;
        if ((rc = gemm_dev(h->hm, L.wdown, h->x, M, H, I, I, I, H, nullptr, h->x, H, SS_EPI_RESIDUAL, dt, s))) return rc;
    }
    // final norm (:652) for all rows, lm_head for the last row only (greedy consumes logits[:, -1])
    void* hid = hidden_out ? hidden_out : (void*)h->xn;
    if ((rc = rmsnorm_rows(h->x, h->w.final_norm, hid, M, H, g.rms_eps, dt, s))) return rc;
    const char* last = (const char*)hid + (size_t)(M - 1) * H * e;
    if ((rc = gemv_dev(h->w.lm_head, last, h->logits, g.vocab, H, nullptr, 0.f, nullptr, nullptr, SS_EPI_NONE, nullptr,
                       dt, s)))
        return rc;
    const int64_t new_pos = pos_ids ? h->pos : h->pos + M;  // explicit pos_ids: caller sets pos afterwards
    return ss_llama_set_lengths(h, kv1, new_pos, stream);
}

int ss_llama_generate(ss_llama* h, int64_t n_steps, int32_t last_prompt_id, const int32_t* host_forced,
                      int64_t n_forced, int64_t* host_n_generated, void* stream) {
    SS_REQUIRE(h && n_steps > 0, "llama_generate: bad arguments");
    const ss_llama_config& g = h->cfg;
    hipStream_t s = (hipStream_t)stream;
    const int64_t limit = n_steps < g.max_new ? n_steps : g.max_new;
    SS_REQUIRE(n_forced >= 0 && n_forced <= g.max_new, "llama_generate: n_forced out of range");
    SS_REQUIRE(h->kv_len + limit <= g.cache_cap, "llama_generate: KV cache overflow (%lld + %lld > %d)",
               (long long)h->kv_len, (long long)limit, g.cache_cap);
    int64_t eff_limit = limit;
    for (int64_t i = 0; i < n_forced && i < eff_limit; ++i)
        if (host_forced[i] == g.eos_id) { eff_limit = i + 1; break; }  // the host already knows where it stops
    if (n_forced > 0)
        SS_HIP(hipMemcpyAsync(h->forced, host_forced, (size_t)n_forced * sizeof(int32_t), hipMemcpyHostToDevice, s));
    // state upload staged in pinned memory (words 8..15); consumed before this call returns (read_state syncs)
    int32_t* init = h->pinned + 8;
    init[ST_KV_LEN] = (int32_t)h->kv_len; init[ST_POS] = (int32_t)h->pos; init[ST_NGEN] = 0; init[ST_DONE] = 0;
    init[ST_LAST] = last_prompt_id; init[ST_NFORCED] = (int32_t)n_forced; init[ST_LIMIT] = (int32_t)limit;
    init[ST_EOS] = g.eos_id;
    SS_HIP(hipMemcpyAsync(h->state, init, 8 * sizeof(int32_t), hipMemcpyHostToDevice, s));
    const bool use_graph = tuning_get("llama_graph", 1) != 0;
    if (use_graph && !h->graph_ready) {
        SS_HIP(hipStreamBeginCapture(h->cap_stream, hipStreamCaptureModeThreadLocal));
        int rc = decode_token(h, h->cap_stream, nullptr);
        hipError_t ce = hipStreamEndCapture(h->cap_stream, &h->graph);
        if (rc) return rc;
        SS_HIP(ce);
        SS_HIP(hipGraphInstantiate(&h->graph_exec, h->graph, nullptr, nullptr, 0));
        h->graph_ready = true;
    }
    const int chunk = tuning_get("llama_done_poll", 8);
    int64_t launched = 0;
    while (launched < eff_limit) {
        const int64_t n = (eff_limit - launched) < chunk ? (eff_limit - launched) : chunk;
        for (int64_t i = 0; i < n; ++i) {
            if (use_graph) SS_HIP(hipGraphLaunch(h->graph_exec, s));
            else { int rc = decode_token(h, s, nullptr); if (rc) return rc; }
        }
        launched += n;
        int rc = read_state(h, s);
        if (rc) return rc;
        if (h->pinned[ST_DONE]) break;
    }
    h->kv_len = h->pinned[ST_KV_LEN];
    h->pos = h->pinned[ST_POS];
    if (host_n_generated) *host_n_generated = h->pinned[ST_NGEN];
    return SS_OK;
}

int ss_llama_profile_decode(ss_llama* h, int64_t n_tokens, float out_ms[8], double out_bytes[4], void* stream) {
    SS_REQUIRE(h && n_tokens > 0 && out_ms && out_bytes, "llama_profile_decode: bad arguments");
    const ss_llama_config& g = h->cfg;
    hipStream_t s = (hipStream_t)stream;
    SS_REQUIRE(h->kv_len + n_tokens + 1 <= g.cache_cap, "llama_profile_decode: KV cache too full");
    int32_t* init = h->pinned + 8;
    init[ST_KV_LEN] = (int32_t)h->kv_len; init[ST_POS] = (int32_t)h->pos; init[ST_NGEN] = 0; init[ST_DONE] = 0;
    init[ST_LAST] = 0; init[ST_NFORCED] = 0; init[ST_LIMIT] = g.max_new; init[ST_EOS] = -1;
    SS_HIP(hipMemcpyAsync(h->state, init, 8 * sizeof(int32_t), hipMemcpyHostToDevice, s));
    for (int i = 0; i < 8; ++i) out_ms[i] = 0.f;
    double cnt[4] = {0, 0, 0, 0};
    for (int64_t t = 0; t < n_tokens && t < g.max_new - 1; ++t) {
        ProfSink p;
        p.s = s;
        int rc = decode_token(h, s, &p);
        hipError_t e = hipStreamSynchronize(s);
        if (!rc && e == hipSuccess) {
            for (size_t i = 1; i < p.ev.size(); ++i) {
                float ms = 0.f;
                hipEventElapsedTime(&ms, p.ev[i - 1], p.ev[i]);
                if (p.cls[i] >= 0 && p.cls[i] < 4) { out_ms[p.cls[i]] += ms; cnt[p.cls[i]] += 1; }
            }
            float tot = 0.f;
            hipEventElapsedTime(&tot, p.ev.front(), p.ev.back());
            out_ms[4] += tot;
        }
        for (hipEvent_t ev : p.ev) hipEventDestroy(ev);
        if (rc) return rc;
        SS_HIP(e);
    }
    for (int i = 0; i < 8; ++i) out_ms[i] /= (float)n_tokens;
    const double H = g.hidden, I = g.inter;
    // class 0 = every GEMV except the down projection (qkv, o, gate|up per layer + lm_head); class 2 = down
    out_bytes[0] = ((double)g.n_layers * (4.0 * H * H + 2.0 * H * I) + (double)g.vocab * H) * (double)h->esz;
    out_bytes[1] = cnt[0] / (double)n_tokens;
    out_bytes[2] = (double)g.n_layers * H * I * (double)h->esz;
    out_bytes[3] = cnt[2] / (double)n_tokens;
    int rc = read_state(h, s);
    if (rc) return rc;
    h->kv_len = h->pinned[ST_KV_LEN];
    h->pos = h->pinned[ST_POS];
    return SS_OK;
}

}  // extern "C"
