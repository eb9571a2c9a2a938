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
int gemv_batched_dev(const void* W, const void* x, void* y, int64_t N, int64_t K, const void* norm_w, float eps,
                     const void* bias, const void* residual, int epi, const int32_t* done_flag, int done_stride,
                     int nb, int64_t x_ld, int64_t y_ld, int64_t res_ld, int dtype, hipStream_t s);
size_t gemm_splitk_workspace_bytes(int64_t M, int64_t N, int64_t K);
int gemm_splitk_dev(const void* A, const void* W, void* C, int64_t M, int64_t N, int64_t K, const void* bias,
                    const void* residual, void* ws, size_t ws_bytes, int dtype, hipStream_t s);
int gemm_dev(const void* A, const void* W, void* C, int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldw,
             int64_t ldc, const void* bias, const void* residual, int64_t ldr, int epi, int dtype, hipStream_t s);
int rope_kv_append_dev(const void* qkv, void* q_out, void* kc, void* vc, const void* cos_t, const void* sin_t,
                       const int32_t* pos_ids, int64_t M, int64_t n_heads, int64_t hd, const int32_t* kv_start_dev,
                       int64_t cache_cap, int dtype, hipStream_t s);
int attn_decode_fused_dev(const void* qkv_raw, void* kc, void* vc, const void* cos_t, const void* sin_t, void* out,
                          void* ws, const int32_t* kv_len_dev, const int32_t* pos_dev, const int32_t* done_flag,
                          int64_t n_heads, int64_t hd, int64_t cache_cap, int nb, int state_stride,
                          int64_t cache_stride, int dtype, hipStream_t s);

// device state words
enum { ST_KV_LEN = 0, ST_POS = 1, ST_NGEN = 2, ST_DONE = 3, ST_LAST = 4, ST_NFORCED = 5, ST_LIMIT = 6, ST_EOS = 7 };

// ---- engine kernels ---------------------------------------------------------------------------

// sample -> (forced?) -> append -> EOS/limit check -> embed.  One block per sequence (blockIdx.x).
template <typename T>
__global__ __launch_bounds__(1024) void sample_embed_kernel(T* logits, int vocab, int32_t* st,
                                                            const int32_t* __restrict__ img_ids, int n_img_ids,
                                                            const int32_t* __restrict__ forced, int32_t* gen_ids,
                                                            const T* __restrict__ embed, T* x, int hidden,
                                                            int max_new) {
    __shared__ float sv[16];
    __shared__ int si[16];
    {
        const int b = blockIdx.x;
        logits += (int64_t)b * vocab;
        st += b * 8;
        forced += (int64_t)b * max_new;
        gen_ids += (int64_t)b * max_new;
        x += (int64_t)b * hidden;
    }
    if (st[ST_DONE]) return;
    int tok = imgproc_argmax_block<T>(logits, vocab, st[ST_LAST], img_ids, n_img_ids, sv, si);
    if ((unsigned)tok >= (unsigned)vocab) tok = 0;     // all-NaN logits have no maximum (torch.argmax: the first NaN): never
                                                       // index the embedding table with the sentinel
    const int n = st[ST_NGEN];
    if (n < st[ST_NFORCED]) tok = forced[n];
    // ST_EOS packs two stop ids: the EOS token in the low half and an optional second stop token + 1 in the high half
    // (ss_llama_set_stop_id: the drivers stop at <img> to run the processor-forced image tokens as one batched forward)
    const int eos_word = st[ST_EOS];
    const int eos_id = eos_word & 0xFFFF, stop2 = (eos_word >> 16) - 1;
    const bool stop = (tok == eos_id) || (tok == stop2) || (n + 1 >= st[ST_LIMIT]);
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
// hidden-state ring row (n_gen - 1), then advances kv_len / pos.  One block of 256 per sequence.
template <typename T>
__global__ __launch_bounds__(256) void final_norm_advance_kernel(const T* __restrict__ x, const T* __restrict__ w,
                                                                 T* xn, T* hid_rows, int32_t* st, int hidden,
                                                                 float eps, int max_new) {
    constexpr int V = Tr<T>::kVec;
    __shared__ float red[16];
    {
        const int b = blockIdx.x;
        x += (int64_t)b * hidden;
        xn += (int64_t)b * hidden;
        hid_rows += (int64_t)b * max_new * hidden;
        st += b * 8;
    }
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

struct ss_llama;
// one projection of the prefill paths: C [M, N] = A [M, K] W^T (+ residual).  128 < M <= 512 rows (the stacked image-token
// block, the first prompts) take the split-K weight-streaming path when the engine carved a partial-sum workspace for it.
static int prefill_proj(void* splitk_ws, size_t splitk_bytes, const void* A, const void* W, void* C, int64_t M, int64_t N,
                        int64_t K, const void* residual, int dt, hipStream_t s) {
    if (splitk_ws && M > 128 && M <= 512)
        return gemm_splitk_dev(A, W, C, M, N, K, nullptr, residual, splitk_ws, splitk_bytes, dt, s);
    return gemm_dev(A, W, C, M, N, K, K, K, N, nullptr, residual, N, residual ? SS_EPI_RESIDUAL : SS_EPI_NONE, dt, s);
}

struct SeqGraph {
    int seq0, nb;
    int mode;                // arithmetic knobs the captured kernels were chosen under (gemm_f32_split): part of the cache key
    hipGraph_t graph;
    hipGraphExec_t exec;
};

struct ss_llama {
    ss_llama_config cfg;
    ss_llama_weights w;
    std::vector<ss_llama_layer_weights> layers;
    int hd;
    int n_seq;               // sequence slots (independent stories sharing one sweep of the weights)
    int cur;                 // slot addressed by the single-sequence entry points
    int stop2 = -1;          // optional second stop token of the decode loop (-1 = none), ss_llama_set_stop_id
    int64_t max_rows;
    size_t esz;
    // device buffers (carved from the caller's workspace); every per-sequence array is [n_seq][...]
    char *kc, *vc;           // [n_seq][L][H][cap][hd]
    int32_t* state;          // [n_seq][8]
    int32_t* gen_ids;        // [n_seq][max_new]
    int32_t* forced;         // [n_seq][max_new]
    int32_t* img_ids;        // [n_img_ids]
    char* hid_rows;          // [n_seq][max_new][hidden]
    char* logits;            // [n_seq][vocab]
    char *x, *xn, *qkv, *q, *attn, *gu, *hm;  // activations ([max_rows][..]); decode uses rows 0..nb-1
    float* attn_ws;          // [n_seq] split-KV partial slabs
    void* splitk_ws;         // fp32 partial sums of the small-M split-K projections (128 < rows <= 512)
    size_t splitk_bytes;
    // host mirrors
    std::vector<int64_t> kv_len, pos;
    hipStream_t cap_stream;
    std::vector<SeqGraph> graphs;
    int32_t* pinned;         // [2][n_seq][8] ints of pinned host memory: state read-back | state upload
    size_t seq_kv_bytes() const { return (size_t)cfg.n_layers * cfg.n_heads * cfg.cache_cap * hd * esz; }
};

static size_t align_up(size_t v, size_t a = 256) { return (v + a - 1) / a * a; }

struct Carver {
    char* base; size_t off; size_t cap;
    char* take(size_t bytes) { char* p = base ? base + off : nullptr; off += align_up(bytes); return p; }
};

static void carve(ss_llama* h, Carver& c) {
    const ss_llama_config& g = h->cfg;
    const size_t e = h->esz;
    const size_t H = g.hidden, I = g.inter, R = (size_t)h->max_rows, S = (size_t)h->n_seq;
    h->kc = c.take(S * h->seq_kv_bytes());
    h->vc = c.take(S * h->seq_kv_bytes());
    h->state = (int32_t*)c.take(S * 8 * sizeof(int32_t));
    h->gen_ids = (int32_t*)c.take(S * (size_t)g.max_new * sizeof(int32_t));
    h->forced = (int32_t*)c.take(S * (size_t)g.max_new * sizeof(int32_t));
    h->img_ids = (int32_t*)c.take((size_t)(g.n_img_ids > 0 ? g.n_img_ids : 1) * sizeof(int32_t));
    h->hid_rows = c.take(S * (size_t)g.max_new * H * e);
    h->logits = c.take(S * (size_t)g.vocab * e);
    h->x = c.take(R * H * e);
    h->xn = c.take(R * H * e);
    h->qkv = c.take(R * 3 * H * e);
    h->q = c.take(R * H * e);
    h->attn = c.take(R * H * e);
    h->gu = c.take(R * 2 * I * e);
    h->hm = c.take(R * I * e);
    h->attn_ws = (float*)c.take(S * ss_attn_decode_workspace_bytes(g.n_heads, h->hd));
    // split-K partial sums: the largest need over the four projections at the largest eligible row count
    size_t sk = 0;
    if (g.dtype != SS_F32 && R > 128) {
        // S x M x N x 4 bytes is not monotone in M (the slice count S drops when another 128-row tile appears): take the
        // maximum over the upper end of every row-tile band that fits the engine, per projection
        const int64_t Mx = R < 512 ? (int64_t)R : 512;
        const int64_t shapes[4][2] = {{3 * (int64_t)H, (int64_t)H}, {(int64_t)H, (int64_t)H}, {2 * (int64_t)I, (int64_t)H}, {(int64_t)H, (int64_t)I}};
        for (auto& nk : shapes)
            for (int64_t m = 256; ; m += 128) {
                const int64_t mm = m < Mx ? m : Mx;
                const size_t b = gemm_splitk_workspace_bytes(mm, nk[0], nk[1]);
                if (b > sk) sk = b;
                if (mm == Mx) break;
            }
    }
    h->splitk_bytes = sk;
    h->splitk_ws = sk ? c.take(sk) : nullptr;
}

static int cfg_n_seq(const ss_llama_config* cfg) { return cfg->n_seq > 0 ? cfg->n_seq : 1; }

// one decode token (sample+forward) for the sequence slots [seq0, seq0+nb); eager or under stream
// capture.  `prof` (optional) receives an event before/after every launch class for profiling.
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

static int decode_token(ss_llama* h, hipStream_t s, ProfSink* prof, int seq0, int nb) {
    const ss_llama_config& g = h->cfg;
    const int dt = g.dtype;
    const int H = g.hidden, I = g.inter, hd = h->hd;
    const size_t e = h->esz;
    int32_t* st = h->state + (size_t)seq0 * 8;
    const int32_t* done = st + ST_DONE;
    const size_t plane = (size_t)g.n_heads * g.cache_cap * hd * e;
    const size_t seq_kv = h->seq_kv_bytes();
    const int64_t cache_stride = (int64_t)(seq_kv / e);
    char* logits = h->logits + (size_t)seq0 * g.vocab * e;
    int32_t* forced = h->forced + (size_t)seq0 * g.max_new;
    int32_t* gen_ids = h->gen_ids + (size_t)seq0 * g.max_new;
    char* hid_rows = h->hid_rows + (size_t)seq0 * g.max_new * H * e;
    float* attn_ws = (float*)((char*)h->attn_ws + (size_t)seq0 * ss_attn_decode_workspace_bytes(g.n_heads, hd));
#define MARK(c) do { if (prof) prof->mark(c); } while (0)
#define SAMPLE(T)                                                                                                  \
    hipLaunchKernelGGL(sample_embed_kernel<T>, dim3((unsigned)nb), dim3(1024), 0, s, (T*)logits, g.vocab, st,        \
                       h->img_ids, g.n_img_ids, forced, gen_ids, (const T*)h->w.embed, (T*)h->x, H, g.max_new)
#define FINAL(T)                                                                                                   \
    hipLaunchKernelGGL(final_norm_advance_kernel<T>, dim3((unsigned)nb), dim3(256), 0, s, (const T*)h->x,            \
                       (const T*)h->w.final_norm, (T*)h->xn, (T*)hid_rows, st, H, g.rms_eps, g.max_new)
    MARK(-1);
    if (dt == SS_BF16) SAMPLE(bf16_t);
    else if (dt == SS_F32) SAMPLE(float);
    else SAMPLE(f16_t);
    SS_LAUNCH_CHECK("sample_embed");
    MARK(3);
    for (int l = 0; l < g.n_layers; ++l) {
        const ss_llama_layer_weights& L = h->layers[l];
        char* kc = h->kc + (size_t)seq0 * seq_kv + (size_t)l * plane;
        char* vc = h->vc + (size_t)seq0 * seq_kv + (size_t)l * plane;
        int rc;
        MARK(-1);
        rc = gemv_batched_dev(L.wqkv, h->x, h->qkv, 3 * H, H, L.ln1, g.rms_eps, nullptr, nullptr, SS_EPI_NONE, done, 8,
                              nb, H, 3 * H, 0, dt, s);
        if (rc) return rc;
        MARK(0);
        // RoPE(q,k) + KV append + split-KV attention in one kernel (+ the split merge)
        rc = attn_decode_fused_dev(h->qkv, kc, vc, h->w.rope_cos, h->w.rope_sin, h->attn, attn_ws, st + ST_KV_LEN,
                                   st + ST_POS, done, g.n_heads, hd, g.cache_cap, nb, 8, cache_stride, dt, s);
        if (rc) return rc;
        MARK(1);
        rc = gemv_batched_dev(L.wo, h->attn, h->xn, H, H, nullptr, 0.f, nullptr, h->x, SS_EPI_RESIDUAL, done, 8, nb, H,
                              H, H, dt, s);
        if (rc) return rc;
        MARK(0);
        rc = gemv_batched_dev(L.wgu, h->xn, h->hm, I, H, L.ln2, g.rms_eps, nullptr, nullptr, SS_EPI_SILU_MUL, done, 8,
                              nb, H, I, 0, dt, s);
        if (rc) return rc;
        MARK(0);
        rc = gemv_batched_dev(L.wdown, h->hm, h->x, H, I, nullptr, 0.f, nullptr, h->xn, SS_EPI_RESIDUAL, done, 8, nb, I,
                              H, H, dt, s);
        if (rc) return rc;
        MARK(2);
    }
    if (dt == SS_BF16) FINAL(bf16_t);
    else if (dt == SS_F32) FINAL(float);
    else FINAL(f16_t);
    SS_LAUNCH_CHECK("final_norm_advance");
    MARK(3);
    int rc = gemv_batched_dev(h->w.lm_head, h->xn, logits, g.vocab, H, nullptr, 0.f, nullptr, nullptr, SS_EPI_NONE, done,
                              8, nb, H, g.vocab, 0, dt, s);
    if (rc) return rc;
    MARK(0);
#undef MARK
#undef SAMPLE
#undef FINAL
    return SS_OK;
}

static int read_state(ss_llama* h, hipStream_t s) {
    SS_HIP(hipMemcpyAsync(h->pinned, h->state, (size_t)h->n_seq * 8 * sizeof(int32_t), hipMemcpyDeviceToHost, s));
    SS_HIP(hipStreamSynchronize(s));
    return SS_OK;
}

// captured decode token for slots [seq0, seq0+nb), built on first use
static int graph_for(ss_llama* h, int seq0, int nb, hipGraphExec_t* out) {
    const int mode = tuning_get("gemm_f32_split", 0);
    for (const SeqGraph& sg : h->graphs)
        if (sg.seq0 == seq0 && sg.nb == nb && sg.mode == mode) { *out = sg.exec; return SS_OK; }
    SeqGraph sg;
    sg.seq0 = seq0; sg.nb = nb; sg.mode = mode; sg.graph = nullptr; sg.exec = nullptr;
    SS_HIP(hipStreamBeginCapture(h->cap_stream, hipStreamCaptureModeThreadLocal));
    int rc = decode_token(h, h->cap_stream, nullptr, seq0, nb);
    hipError_t ce = hipStreamEndCapture(h->cap_stream, &sg.graph);
    if (rc) return rc;
    SS_HIP(ce);
    SS_HIP(hipGraphInstantiate(&sg.exec, sg.graph, nullptr, nullptr, 0));
    h->graphs.push_back(sg);
    *out = sg.exec;
    return SS_OK;
}

// shared driver of ss_llama_generate / ss_llama_generate_batch: the state words of slots
// [seq0, seq0+nb) have been staged in h->pinned[upload area]; replays the decode graph until every
// slot reports done or `eff_limit` tokens were launched.
static int run_decode(ss_llama* h, int seq0, int nb, int64_t eff_limit, hipStream_t s) {
    int32_t* init = h->pinned + (size_t)h->n_seq * 8;
    SS_HIP(hipMemcpyAsync(h->state + (size_t)seq0 * 8, init + (size_t)seq0 * 8, (size_t)nb * 8 * sizeof(int32_t),
                          hipMemcpyHostToDevice, s));
    const bool use_graph = tuning_get("llama_graph", 1) != 0;
    hipGraphExec_t exec = nullptr;
    if (use_graph) { int rc = graph_for(h, seq0, nb, &exec); if (rc) return rc; }
    const int chunk = tuning_get("llama_done_poll", 8);
    int64_t launched = 0;
    while (launched < eff_limit) {
        const int64_t n = (eff_limit - launched) < chunk ? (eff_limit - launched) : chunk;
        for (int64_t i = 0; i < n; ++i) {
            if (use_graph) SS_HIP(hipGraphLaunch(exec, s));
            else { int rc = decode_token(h, s, nullptr, seq0, nb); if (rc) return rc; }
        }
        launched += n;
        int rc = read_state(h, s);
        if (rc) return rc;
        bool all = true;
        for (int b = seq0; b < seq0 + nb; ++b) all = all && h->pinned[b * 8 + ST_DONE];
        if (all) break;
    }
    if (launched == 0) { int rc = read_state(h, s); if (rc) return rc; }
    for (int b = seq0; b < seq0 + nb; ++b) {
        h->kv_len[b] = h->pinned[b * 8 + ST_KV_LEN];
        h->pos[b] = h->pinned[b * 8 + ST_POS];
    }
    return SS_OK;
}

extern "C" {

size_t ss_llama_workspace_bytes(const ss_llama_config* cfg, int64_t max_prefill_rows) {
    if (!cfg) return 0;
    ss_llama tmp;
    tmp.cfg = *cfg;
    tmp.hd = cfg->hidden / cfg->n_heads;
    tmp.n_seq = cfg_n_seq(cfg);
    tmp.max_rows = max_prefill_rows < tmp.n_seq ? tmp.n_seq : max_prefill_rows;
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
    SS_REQUIRE(cfg->n_seq >= 0 && cfg->n_seq <= 8, "llama_create: n_seq=%d unsupported (1..8)", cfg->n_seq);
    // the decode loop's per-slot stop word packs the EOS id in 16 bits (and the optional second stop id + 1 above it)
    SS_REQUIRE(cfg->vocab > 0 && cfg->vocab <= 0xFFFF && cfg->eos_id < 0xFFFF,
               "llama_create: vocab %d / eos_id %d do not fit the 16-bit stop word (vocab <= 65535)", (int)cfg->vocab,
               (int)cfg->eos_id);
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
    h->n_seq = cfg_n_seq(cfg);
    h->cur = 0;
    h->max_rows = max_prefill_rows < h->n_seq ? h->n_seq : max_prefill_rows;
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
    h->kv_len.assign(h->n_seq, 0);
    h->pos.assign(h->n_seq, 0);
    hipError_t e = hipStreamCreateWithFlags(&h->cap_stream, hipStreamNonBlocking);
    if (e == hipSuccess)
        e = hipHostMalloc((void**)&h->pinned, (size_t)h->n_seq * 16 * sizeof(int32_t) + 64, hipHostMallocDefault);
    // the whole workspace starts as zeros: a cache row a caller declares valid without having written it (set_lengths on a
    // fresh slot: profiling runs, KV mirrors) then holds zeros, not the allocator's residue — NaN bit patterns there turn
    // every logit into NaN
    if (e == hipSuccess) e = hipMemset(c.base, 0, c.off);
    if (e == hipSuccess && cfg->n_img_ids > 0)
        e = hipMemcpy(h->img_ids, host_img_ids, cfg->n_img_ids * sizeof(int32_t), hipMemcpyHostToDevice);
    if (e != hipSuccess) { rc = check_hip(e, "llama_create"); delete h; return rc; }
    *out = h;
    return SS_OK;
}

void ss_llama_destroy(ss_llama* h) {
    if (!h) return;
    for (SeqGraph& sg : h->graphs) {
        if (sg.exec) hipGraphExecDestroy(sg.exec);
        if (sg.graph) hipGraphDestroy(sg.graph);
    }
    if (h->cap_stream) hipStreamDestroy(h->cap_stream);
    if (h->pinned) hipHostFree(h->pinned);
    delete h;
}

int ss_llama_set_stop_id(ss_llama* h, int32_t token_id) {
    SS_REQUIRE(h && token_id >= -1 && token_id < 0x7FFE && token_id < h->cfg.vocab,
               "llama_set_stop_id: token id %d out of range (< min(vocab, 32766))", (int)token_id);
    h->stop2 = token_id;
    return SS_OK;
}

int ss_llama_select(ss_llama* h, int32_t seq) {
    SS_REQUIRE(h && seq >= 0 && seq < h->n_seq, "llama_select: sequence slot %d out of range", (int)seq);
    h->cur = seq;
    return SS_OK;
}

void* ss_llama_buffer(ss_llama* h, int which) {
    if (!h) return nullptr;
    const ss_llama_config& g = h->cfg;
    const size_t q = (size_t)h->cur;
    switch (which) {
        case 0: return h->kc + q * h->seq_kv_bytes();
        case 1: return h->vc + q * h->seq_kv_bytes();
        case 2: return h->gen_ids + q * g.max_new;
        case 3: return h->hid_rows + q * (size_t)g.max_new * g.hidden * h->esz;
        case 4: return h->logits + q * (size_t)g.vocab * h->esz;
        case 5: return h->state + q * 8;
        default: return nullptr;
    }
}

int ss_llama_set_lengths(ss_llama* h, int64_t kv_len, int64_t pos, void* stream) {
    SS_REQUIRE(h && kv_len >= 0 && kv_len <= h->cfg.cache_cap && pos >= 0 && pos <= h->cfg.max_pos,
               "llama_set_lengths: out of range (kv_len=%lld pos=%lld)", (long long)kv_len, (long long)pos);
    hipLaunchKernelGGL(set_state_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, h->state + (size_t)h->cur * 8,
                       (int)ST_KV_LEN, (int)kv_len, (int)ST_POS, (int)pos);
    SS_LAUNCH_CHECK("set_state");
    h->kv_len[h->cur] = kv_len;
    h->pos[h->cur] = pos;
    return SS_OK;
}

int ss_llama_get_lengths(ss_llama* h, int64_t* kv_len, int64_t* pos) {
    SS_REQUIRE(h, "llama_get_lengths: null handle");
    if (kv_len) *kv_len = h->kv_len[h->cur];
    if (pos) *pos = h->pos[h->cur];
    return SS_OK;
}

int ss_llama_kv_gather(ss_llama* h, const int32_t* keep_idx_dev, int64_t n_keep, void* stream) {
    SS_REQUIRE(h && keep_idx_dev && n_keep >= 0 && n_keep <= h->kv_len[h->cur], "llama_kv_gather: bad arguments");
    const ss_llama_config& g = h->cfg;
    const size_t e = h->esz;
    // scratch = qkv activation buffer: [max_rows][3*hidden] elements >= n_heads * n_keep * hd = n_keep * hidden
    SS_REQUIRE(n_keep <= 3 * h->max_rows, "llama_kv_gather: n_keep %lld exceeds scratch (3 x %lld rows)",
               (long long)n_keep, (long long)h->max_rows);
    hipStream_t s = (hipStream_t)stream;
    const size_t plane = (size_t)g.n_heads * g.cache_cap * h->hd * e;
    char* kbase = h->kc + (size_t)h->cur * h->seq_kv_bytes();
    char* vbase = h->vc + (size_t)h->cur * h->seq_kv_bytes();
    const int V = g.dtype == SS_F32 ? 4 : 8;
    dim3 grid((unsigned)cdiv(n_keep * (h->hd / V), 256) > 0 ? (unsigned)cdiv(n_keep * (h->hd / V), 256) : 1,
              (unsigned)g.n_heads);
    if (n_keep > 0) {
        for (int l = 0; l < g.n_layers; ++l) {
            for (int kv = 0; kv < 2; ++kv) {
                char* plane_p = (kv ? vbase : kbase) + (size_t)l * plane;
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
    return ss_llama_set_lengths(h, n_keep, h->pos[h->cur], stream);
}

int ss_llama_prefill(ss_llama* h, const void* embeds, int64_t M, const int32_t* pos_ids, void* hidden_out,
                     void* stream) {
    SS_REQUIRE(h && embeds && M > 0, "llama_prefill: bad arguments");
    SS_REQUIRE(M <= h->max_rows, "llama_prefill: M=%lld exceeds max_prefill_rows=%lld", (long long)M,
               (long long)h->max_rows);
    const int64_t cur_kv = h->kv_len[h->cur], cur_pos = h->pos[h->cur];
    SS_REQUIRE(cur_kv + M <= h->cfg.cache_cap, "llama_prefill: KV cache overflow (%lld + %lld > %d)",
               (long long)cur_kv, (long long)M, h->cfg.cache_cap);
    SS_REQUIRE(pos_ids || cur_pos + M <= h->cfg.max_pos, "llama_prefill: position overflow");
    const ss_llama_config& g = h->cfg;
    hipStream_t s = (hipStream_t)stream;
    const int dt = g.dtype;
    const int64_t H = g.hidden, I = g.inter;
    const int hd = h->hd;
    const size_t e = h->esz;
    const size_t plane = (size_t)g.n_heads * g.cache_cap * hd * e;
    char* kbase = h->kc + (size_t)h->cur * h->seq_kv_bytes();
    char* vbase = h->vc + (size_t)h->cur * h->seq_kv_bytes();
    const int64_t kv0 = cur_kv, kv1 = cur_kv + M;
    int rc;
    SS_HIP(hipMemcpyAsync(h->x, embeds, (size_t)M * H * e, hipMemcpyDeviceToDevice, s));
    for (int l = 0; l < g.n_layers; ++l) {
        const ss_llama_layer_weights& L = h->layers[l];
        char* kc = kbase + (size_t)l * plane;
        char* vc = vbase + (size_t)l * plane;
        if ((rc = rmsnorm_rows(h->x, L.ln1, h->xn, M, H, g.rms_eps, dt, s))) return rc;
        if ((rc = prefill_proj(h->splitk_ws, h->splitk_bytes, h->xn, L.wqkv, h->qkv, M, 3 * H, H, nullptr, dt, s))) return rc;
        if ((rc = ss_rope_kv_append(h->qkv, h->q, kc, vc, h->w.rope_cos, h->w.rope_sin, pos_ids, cur_pos, M, g.n_heads,
                                    hd, kv0, g.cache_cap, dt, stream)))
            return rc;
        if ((rc = ss_attention(h->q, kc, vc, h->attn, 1, g.n_heads, M, kv1, hd, 0, hd, H, 0, (int64_t)g.cache_cap * hd,
                               hd, 0, (int64_t)g.cache_cap * hd, hd, 0, hd, H, 1.0f / sqrtf((float)hd), 1, dt, stream)))
            return rc;
        if ((rc = prefill_proj(h->splitk_ws, h->splitk_bytes, h->attn, L.wo, h->x, M, H, H, h->x, dt, s))) return rc;
        if ((rc = rmsnorm_rows(h->x, L.ln2, h->xn, M, H, g.rms_eps, dt, s))) return rc;
        if ((rc = prefill_proj(h->splitk_ws, h->splitk_bytes, h->xn, L.wgu, h->gu, M, 2 * I, H, nullptr, dt, s))) return rc;
        if ((rc = ss_silu_mul(h->gu, h->hm, M, I, dt, stream))) return rc;
        if ((rc = prefill_proj(h->splitk_ws, h->splitk_bytes, h->hm, L.wdown, h->x, M, H, I, h->x, dt, s))) return rc;
    }
    // final norm (:652) for all rows, lm_head for the last row only (greedy consumes logits[:, -1])
    void* hid = hidden_out ? hidden_out : (void*)h->xn;
    if ((rc = rmsnorm_rows(h->x, h->w.final_norm, hid, M, H, g.rms_eps, dt, s))) return rc;
    const char* last = (const char*)hid + (size_t)(M - 1) * H * e;
    char* logits = h->logits + (size_t)h->cur * g.vocab * e;
    if ((rc = gemv_dev(h->w.lm_head, last, logits, g.vocab, H, nullptr, 0.f, nullptr, nullptr, SS_EPI_NONE, nullptr,
                       dt, s)))
        return rc;
    const int64_t new_pos = pos_ids ? cur_pos : cur_pos + M;  // explicit pos_ids: caller sets pos afterwards
    return ss_llama_set_lengths(h, kv1, new_pos, stream);
}

// The same forward for SEVERAL sequence slots in one sweep of the weights: the rows of all participating slots are
// stacked (slot-major) and every projection runs ONCE on the stack (M = sum of the slots' rows: the 13.2 GB of layer
// weights are streamed once per call instead of once per slot, and the GEMMs see 4x the rows); RoPE / KV append and the
// bottom-right causal attention stay per slot (each slot's rows against its own cache).  Used for the image-token block
// continuation of lock-step stories (4 x 66 rows) and for their prompt prefill (4 x S rows).
int ss_llama_prefill_batch(ss_llama* h, const void* embeds, const int64_t* host_rows, void* hidden_out, void* stream) {
    SS_REQUIRE(h && embeds && host_rows, "llama_prefill_batch: bad arguments");
    const ss_llama_config& g = h->cfg;
    int64_t M = 0;
    for (int b = 0; b < h->n_seq; ++b) {
        const int64_t r = host_rows[b];
        SS_REQUIRE(r >= 0, "llama_prefill_batch: negative row count for slot %d", b);
        SS_REQUIRE(h->kv_len[b] + r <= g.cache_cap, "llama_prefill_batch: KV cache overflow in slot %d (%lld + %lld > %d)", b,
                   (long long)h->kv_len[b], (long long)r, g.cache_cap);
        // (the bound of ss_llama_set_lengths, checked for EVERY slot before anything is launched: a slot that fails there
        // after the forward would leave host and device lengths inconsistent across the slots)
        // (`<=`: the last RoPE position used is pos + r - 1, so a prompt may exactly fill the position table — the same bound as
        // the single-slot ss_llama_prefill; decoding further is refused by the generate entry points)
        SS_REQUIRE(r == 0 || h->pos[b] + r <= g.max_pos, "llama_prefill_batch: position overflow in slot %d (%lld + %lld > %d)", b,
                   (long long)h->pos[b], (long long)r, g.max_pos);
        M += r;
    }
    SS_REQUIRE(M > 0 && M <= h->max_rows, "llama_prefill_batch: %lld stacked rows (max_prefill_rows=%lld)", (long long)M,
               (long long)h->max_rows);
    hipStream_t s = (hipStream_t)stream;
    const int dt = g.dtype;
    const int64_t H = g.hidden, I = g.inter;
    const int hd = h->hd;
    const size_t e = h->esz;
    const size_t plane = (size_t)g.n_heads * g.cache_cap * hd * e;
    int rc;
    bool uniform_rows = h->n_seq >= 2 && h->n_seq <= 8 && tuning_get("llama_batched_attn", 1) != 0;
    int32_t kv_lens[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int b = 0; b < h->n_seq && uniform_rows; ++b) {
        if (host_rows[b] != host_rows[0] || host_rows[b] <= 0) uniform_rows = false;
        else kv_lens[b] = (int32_t)(h->kv_len[b] + host_rows[b]);
    }
    SS_HIP(hipMemcpyAsync(h->x, embeds, (size_t)M * H * e, hipMemcpyDeviceToDevice, s));
    for (int l = 0; l < g.n_layers; ++l) {
        const ss_llama_layer_weights& L = h->layers[l];
        if ((rc = rmsnorm_rows(h->x, L.ln1, h->xn, M, H, g.rms_eps, dt, s))) return rc;
        if ((rc = prefill_proj(h->splitk_ws, h->splitk_bytes, h->xn, L.wqkv, h->qkv, M, 3 * H, H, nullptr, dt, s))) return rc;
        int64_t r0 = 0;
        for (int b = 0; b < h->n_seq; ++b) {
            const int64_t r = host_rows[b];
            if (!r) continue;
            char* kc = h->kc + (size_t)b * h->seq_kv_bytes() + (size_t)l * plane;
            char* vc = h->vc + (size_t)b * h->seq_kv_bytes() + (size_t)l * plane;
            const int64_t kv0 = h->kv_len[b], kv1 = kv0 + r;
            char* qkv_b = h->qkv + (size_t)r0 * 3 * H * e;
            char* q_b = h->q + (size_t)r0 * H * e;
            if ((rc = ss_rope_kv_append(qkv_b, q_b, kc, vc, h->w.rope_cos, h->w.rope_sin, nullptr, h->pos[b], r, g.n_heads,
                                        hd, kv0, g.cache_cap, dt, stream)))
                return rc;
            if (!uniform_rows &&
                (rc = ss_attention(q_b, kc, vc, h->attn + (size_t)r0 * H * e, 1, g.n_heads, r, kv1, hd, 0, hd, H, 0,
                                   (int64_t)g.cache_cap * hd, hd, 0, (int64_t)g.cache_cap * hd, hd, 0, hd, H,
                                   1.0f / sqrtf((float)hd), 1, dt, stream)))
                return rc;
            r0 += r;
        }
        if (uniform_rows) {
            // every slot feeds the same number of rows (the lock-step image-token block, equal-length prompts): ONE attention
            // launch over the slots, slot b attending to its own kv_len[b] + r keys (measured at 8 x 66 rows: 256 launches of
            // 24 us per pass = a third of the block's time, before)
            const int64_t r = host_rows[0];
            const int64_t seq_stride = (int64_t)(h->seq_kv_bytes() / e);
            if ((rc = ss_attention_ragged(h->q, h->kc + (size_t)l * plane, h->vc + (size_t)l * plane, h->attn, h->n_seq, g.n_heads,
                                          r, kv_lens, hd, r * H, hd, H, seq_stride, (int64_t)g.cache_cap * hd, hd, seq_stride,
                                          (int64_t)g.cache_cap * hd, hd, r * H, hd, H, 1.0f / sqrtf((float)hd), 1, dt, stream)))
                return rc;
        }
        if ((rc = prefill_proj(h->splitk_ws, h->splitk_bytes, h->attn, L.wo, h->x, M, H, H, h->x, dt, s))) return rc;
        if ((rc = rmsnorm_rows(h->x, L.ln2, h->xn, M, H, g.rms_eps, dt, s))) return rc;
        if ((rc = prefill_proj(h->splitk_ws, h->splitk_bytes, h->xn, L.wgu, h->gu, M, 2 * I, H, nullptr, dt, s))) return rc;
        if ((rc = ss_silu_mul(h->gu, h->hm, M, I, dt, stream))) return rc;
        if ((rc = prefill_proj(h->splitk_ws, h->splitk_bytes, h->hm, L.wdown, h->x, M, H, I, h->x, dt, s))) return rc;
    }
    void* hid = hidden_out ? hidden_out : (void*)h->xn;
    if ((rc = rmsnorm_rows(h->x, h->w.final_norm, hid, M, H, g.rms_eps, dt, s))) return rc;
    const int keep_cur = h->cur;
    int64_t r0 = 0;
    for (int b = 0; b < h->n_seq; ++b) {
        const int64_t r = host_rows[b];
        if (!r) continue;
        const char* last = (const char*)hid + (size_t)(r0 + r - 1) * H * e;
        char* logits = h->logits + (size_t)b * g.vocab * e;
        if ((rc = gemv_dev(h->w.lm_head, last, logits, g.vocab, H, nullptr, 0.f, nullptr, nullptr, SS_EPI_NONE, nullptr, dt, s)))
            return rc;
        h->cur = b;
        rc = ss_llama_set_lengths(h, h->kv_len[b] + r, h->pos[b] + r, stream);
        h->cur = keep_cur;
        if (rc) return rc;
        r0 += r;
    }
    return SS_OK;
}

// stage one slot's initial decode state in the pinned upload area; returns the launch bound
static int64_t stage_seq(ss_llama* h, int b, int64_t n_steps, int32_t last_id, const int32_t* forced, int64_t n_forced,
                         bool active) {
    const ss_llama_config& g = h->cfg;
    const int64_t limit = n_steps < g.max_new ? n_steps : g.max_new;
    int64_t eff = limit;
    for (int64_t i = 0; i < n_forced && i < eff; ++i)
        if (forced[i] == g.eos_id || forced[i] == h->stop2) { eff = i + 1; break; }  // the host already knows where it stops
    int32_t* init = h->pinned + (size_t)h->n_seq * 8 + (size_t)b * 8;
    init[ST_KV_LEN] = (int32_t)h->kv_len[b]; init[ST_POS] = (int32_t)h->pos[b]; init[ST_NGEN] = 0;
    init[ST_DONE] = active ? 0 : 1; init[ST_LAST] = last_id; init[ST_NFORCED] = (int32_t)n_forced;
    init[ST_LIMIT] = (int32_t)limit;
    init[ST_EOS] = (g.eos_id >= 0 ? (g.eos_id & 0xFFFF) : 0xFFFF) | ((h->stop2 + 1) << 16);
    return active ? eff : 0;
}

int ss_llama_generate(ss_llama* h, int64_t n_steps, int32_t last_prompt_id, const int32_t* host_forced,
                      int64_t n_forced, int64_t* host_n_generated, void* stream) {
    SS_REQUIRE(h && n_steps > 0, "llama_generate: bad arguments");
    const ss_llama_config& g = h->cfg;
    hipStream_t s = (hipStream_t)stream;
    const int q = h->cur;
    const int64_t limit = n_steps < g.max_new ? n_steps : g.max_new;
    SS_REQUIRE(n_forced >= 0 && n_forced <= g.max_new, "llama_generate: n_forced out of range");
    SS_REQUIRE(h->kv_len[q] + limit <= g.cache_cap, "llama_generate: KV cache overflow (%lld + %lld > %d)",
               (long long)h->kv_len[q], (long long)limit, g.cache_cap);
    SS_REQUIRE(h->pos[q] + limit <= g.max_pos, "llama_generate: position overflow (%lld + %lld > %d)", (long long)h->pos[q],
               (long long)limit, g.max_pos);
    if (n_forced > 0)
        SS_HIP(hipMemcpyAsync(h->forced + (size_t)q * g.max_new, host_forced, (size_t)n_forced * sizeof(int32_t),
                              hipMemcpyHostToDevice, s));
    // state upload staged in pinned memory; consumed before this call returns (read_state syncs)
    const int64_t eff_limit = stage_seq(h, q, n_steps, last_prompt_id, host_forced, n_forced, true);
    int rc = run_decode(h, q, 1, eff_limit, s);
    if (rc) return rc;
    if (host_n_generated) *host_n_generated = h->pinned[q * 8 + ST_NGEN];
    return SS_OK;
}

int ss_llama_generate_batch(ss_llama* h, int64_t n_steps, const int32_t* last_prompt_ids, const int32_t* host_forced,
                            int64_t forced_ld, const int64_t* n_forced, const int32_t* active,
                            int64_t* host_n_generated, void* stream) {
    SS_REQUIRE(h && n_steps > 0 && last_prompt_ids, "llama_generate_batch: bad arguments");
    const ss_llama_config& g = h->cfg;
    hipStream_t s = (hipStream_t)stream;
    const int64_t limit = n_steps < g.max_new ? n_steps : g.max_new;
    int64_t eff_limit = 0;
    for (int b = 0; b < h->n_seq; ++b) {
        const bool on = !active || active[b];
        const int64_t nf = (n_forced && host_forced) ? n_forced[b] : 0;
        SS_REQUIRE(nf >= 0 && nf <= g.max_new && nf <= forced_ld, "llama_generate_batch: n_forced[%d] out of range", b);
        SS_REQUIRE(!on || h->kv_len[b] + limit <= g.cache_cap,
                   "llama_generate_batch: KV cache overflow in slot %d (%lld + %lld > %d)", b, (long long)h->kv_len[b],
                   (long long)limit, g.cache_cap);
        SS_REQUIRE(!on || h->pos[b] + limit <= g.max_pos, "llama_generate_batch: position overflow in slot %d (%lld + %lld > %d)",
                   b, (long long)h->pos[b], (long long)limit, g.max_pos);
        const int32_t* f = host_forced ? host_forced + (size_t)b * forced_ld : nullptr;
        if (on && nf > 0)
            SS_HIP(hipMemcpyAsync(h->forced + (size_t)b * g.max_new, f, (size_t)nf * sizeof(int32_t),
                                  hipMemcpyHostToDevice, s));
        const int64_t eff = stage_seq(h, b, n_steps, last_prompt_ids[b], f, on ? nf : 0, on);
        if (eff > eff_limit) eff_limit = eff;
    }
    int rc = run_decode(h, 0, h->n_seq, eff_limit, s);
    if (rc) return rc;
    if (host_n_generated)
        for (int b = 0; b < h->n_seq; ++b) host_n_generated[b] = h->pinned[b * 8 + ST_NGEN];
    return SS_OK;
}

int ss_llama_profile_decode(ss_llama* h, int64_t n_tokens, float out_ms[8], double out_bytes[4], void* stream) {
    SS_REQUIRE(h && n_tokens > 0 && out_ms && out_bytes, "llama_profile_decode: bad arguments");
    const ss_llama_config& g = h->cfg;
    hipStream_t s = (hipStream_t)stream;
    int32_t* init = h->pinned + (size_t)h->n_seq * 8;
    for (int b = 0; b < h->n_seq; ++b) {
        SS_REQUIRE(h->kv_len[b] + n_tokens + 1 <= g.cache_cap, "llama_profile_decode: KV cache too full (slot %d)", b);
        int32_t* st = init + b * 8;
        st[ST_KV_LEN] = (int32_t)h->kv_len[b]; st[ST_POS] = (int32_t)h->pos[b]; st[ST_NGEN] = 0; st[ST_DONE] = 0;
        st[ST_LAST] = 0; st[ST_NFORCED] = 0; st[ST_LIMIT] = g.max_new; st[ST_EOS] = -1;
    }
    SS_HIP(hipMemcpyAsync(h->state, init, (size_t)h->n_seq * 8 * sizeof(int32_t), hipMemcpyHostToDevice, s));
    for (int i = 0; i < 8; ++i) out_ms[i] = 0.f;
    double cnt[4] = {0, 0, 0, 0};
    for (int64_t t = 0; t < n_tokens && t < g.max_new - 1; ++t) {
        ProfSink p;
        p.s = s;
        int rc = decode_token(h, s, &p, 0, h->n_seq);
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
    for (int b = 0; b < h->n_seq; ++b) {
        h->kv_len[b] = h->pinned[b * 8 + ST_KV_LEN];
        h->pos[b] = h->pinned[b * 8 + ST_POS];
    }
    return SS_OK;
}

}  // extern "C"
