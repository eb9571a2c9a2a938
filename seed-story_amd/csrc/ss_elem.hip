// Norms, RoPE + KV append and the small element-wise kernels of the SEED-Story hot path.
// All HBM-bound: 16-byte vectorised accesses, fp32 math, one rounding to the model dtype per
// value exactly where the reference's torch graph rounds.
#include "ss_common.h"

namespace ss {

// =====================================================================================
// RMSNorm — LlamaRMSNorm.forward, src/models_clm/modeling_llama_xformer.py:107-115
//   var = mean_fp32(x^2); h = x * rsqrt(var + eps) (fp32); h = round_T(h) when T is
//   half/bf16 (:111-113); y = w * h (rounded to T by the multiply in T, :115).
// One 256-thread block per row; the row stays in registers between the two passes.
// =====================================================================================
constexpr int kNormMaxPacks = 8;

template <typename T>
__global__ __launch_bounds__(256) void rmsnorm_kernel(const T* __restrict__ x, const T* __restrict__ w,
                                                      T* __restrict__ y, int cols, float eps) {
    constexpr int V = Tr<T>::kVec;
    __shared__ float red[16];
    const int64_t row = blockIdx.x;
    const int npack = cols / V;
    const T* xr = x + row * (int64_t)cols;
    T* yr = y + row * (int64_t)cols;
    uint4 px[kNormMaxPacks];
    float ss_ = 0.f;
#pragma unroll
    for (int i = 0; i < kNormMaxPacks; ++i) {
        const int p = threadIdx.x + i * 256;
        if (p < npack) {
            px[i] = ld16(xr + (int64_t)p * V);
            float f[V];
            unpack<T>(px[i], f);
#pragma unroll
            for (int j = 0; j < V; ++j) ss_ = fmaf(f[j], f[j], ss_);
        }
    }
    const float tot = block_sum(ss_, red);
    const float rstd = 1.0f / sqrtf(tot / (float)cols + eps);
#pragma unroll
    for (int i = 0; i < kNormMaxPacks; ++i) {
        const int p = threadIdx.x + i * 256;
        if (p < npack) {
            float f[V], g[V];
            unpack<T>(px[i], f);
            unpack<T>(ld16(w + (int64_t)p * V), g);
#pragma unroll
            for (int j = 0; j < V; ++j) f[j] = g[j] * Tr<T>::rnd(f[j] * rstd);
            st16(yr + (int64_t)p * V, pack<T>(f));
        }
    }
}

template <typename T>
int rmsnorm_launch(const void* x, const void* w, void* y, int64_t rows, int64_t cols, float eps, hipStream_t s) {
    constexpr int V = Tr<T>::kVec;
    SS_REQUIRE(cols % V == 0 && cols / V <= kNormMaxPacks * 256, "rmsnorm: cols=%lld unsupported", (long long)cols);
    if (rows == 0) return SS_OK;
    hipLaunchKernelGGL(rmsnorm_kernel<T>, dim3((unsigned)rows), dim3(256), 0, s, (const T*)x, (const T*)w, (T*)y,
                       (int)cols, eps);
    SS_LAUNCH_CHECK("rmsnorm");
    return SS_OK;
}

// =====================================================================================
// LayerNorm — nn.LayerNorm (qwen_visual.py:103,353; resampler.py:13,38-39,246): fp32 mean /
// biased variance, y = round_T((x - mean) * rstd * w + b).
// =====================================================================================
template <typename T>
__global__ __launch_bounds__(256) void layernorm_kernel(const T* __restrict__ x, const T* __restrict__ w,
                                                        const T* __restrict__ b, T* __restrict__ y, int cols,
                                                        float eps) {
    constexpr int V = Tr<T>::kVec;
    __shared__ float red[16];
    const int64_t row = blockIdx.x;
    const int npack = cols / V;
    const T* xr = x + row * (int64_t)cols;
    T* yr = y + row * (int64_t)cols;
    uint4 px[kNormMaxPacks];
    float s1 = 0.f;
#pragma unroll
    for (int i = 0; i < kNormMaxPacks; ++i) {
        const int p = threadIdx.x + i * 256;
        if (p < npack) {
            px[i] = ld16(xr + (int64_t)p * V);
            float f[V];
            unpack<T>(px[i], f);
#pragma unroll
            for (int j = 0; j < V; ++j) s1 += f[j];
        }
    }
    const float mean = block_sum(s1, red) / (float)cols;
    float s2 = 0.f;
#pragma unroll
    for (int i = 0; i < kNormMaxPacks; ++i) {
        const int p = threadIdx.x + i * 256;
        if (p < npack) {
            float f[V];
            unpack<T>(px[i], f);
#pragma unroll
            for (int j = 0; j < V; ++j) { const float d = f[j] - mean; s2 = fmaf(d, d, s2); }
        }
    }
    const float var = block_sum(s2, red) / (float)cols;
    const float rstd = 1.0f / sqrtf(var + eps);
#pragma unroll
    for (int i = 0; i < kNormMaxPacks; ++i) {
        const int p = threadIdx.x + i * 256;
        if (p < npack) {
            float f[V], g[V], h[V];
            unpack<T>(px[i], f);
            unpack<T>(ld16(w + (int64_t)p * V), g);
            unpack<T>(ld16(b + (int64_t)p * V), h);
#pragma unroll
            for (int j = 0; j < V; ++j) f[j] = (f[j] - mean) * rstd * g[j] + h[j];
            st16(yr + (int64_t)p * V, pack<T>(f));
        }
    }
}

// Narrow rows (<= 256 packs, e.g. the UNet's 640/1280-wide tokens): one WAVE per row — shuffle reductions only, no LDS, no
// barriers, every lane busy.  R rows per wave (knob layernorm_rows_per_wave, default 1): round 6 measured R = 2 / 4 (both rows'
// loads in flight before the first reduction; bit-identical results) and they are SLOWER — [16384, 1280]: 17.4 / 18.4 / 20.6 us for
// R = 1 / 2 / 4 (4.8 TB/s at R = 1), [65536, 640]: 42.8 / 43.0 / 41.1 (tools/ln_bench.py, profiles/round6_ln_bench.txt): with 32
// waves per CU the loads of OTHER waves already cover the latency, more rows per wave only lengthen the tail.
template <typename T, int R>
__global__ __launch_bounds__(256) void layernorm_wave_kernel(const T* __restrict__ x, const T* __restrict__ w,
                                                             const T* __restrict__ b, T* __restrict__ y, int64_t rows,
                                                             int cols, float eps) {
    constexpr int V = Tr<T>::kVec;
    constexpr int MAXP = 4;
    const int lane = threadIdx.x & 63;
    const int64_t row0 = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * R;
    if (row0 >= rows) return;
    const int npack = cols / V;
    uint4 px[R][MAXP];
    float s1[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int64_t row = row0 + r < rows ? row0 + r : rows - 1;      // a ragged last wave re-reads the last row (never stored twice)
        const T* xr = x + row * (int64_t)cols;
#pragma unroll
        for (int i = 0; i < MAXP; ++i) {
            const int p = lane + i * 64;
            if (p < npack) px[r][i] = ld16(xr + (int64_t)p * V);
        }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
        s1[r] = 0.f;
#pragma unroll
        for (int i = 0; i < MAXP; ++i) {
            const int p = lane + i * 64;
            if (p < npack) {
                float f[V];
                unpack<T>(px[r][i], f);
#pragma unroll
                for (int j = 0; j < V; ++j) s1[r] += f[j];
            }
        }
    }
    float mean[R], rstd[R];
#pragma unroll
    for (int r = 0; r < R; ++r) mean[r] = wave_sum(s1[r]) / (float)cols;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        float s2 = 0.f;
#pragma unroll
        for (int i = 0; i < MAXP; ++i) {
            const int p = lane + i * 64;
            if (p < npack) {
                float f[V];
                unpack<T>(px[r][i], f);
#pragma unroll
                for (int j = 0; j < V; ++j) { const float d = f[j] - mean[r]; s2 = fmaf(d, d, s2); }
            }
        }
        rstd[r] = 1.0f / sqrtf(wave_sum(s2) / (float)cols + eps);
    }
#pragma unroll
    for (int i = 0; i < MAXP; ++i) {
        const int p = lane + i * 64;
        if (p < npack) {
            float g[V], h[V];
            unpack<T>(ld16(w + (int64_t)p * V), g);
            unpack<T>(ld16(b + (int64_t)p * V), h);
#pragma unroll
            for (int r = 0; r < R; ++r) {
                if (row0 + r < rows) {
                    float f[V];
                    unpack<T>(px[r][i], f);
#pragma unroll
                    for (int j = 0; j < V; ++j) f[j] = (f[j] - mean[r]) * rstd[r] * g[j] + h[j];
                    st16(y + (row0 + r) * (int64_t)cols + (int64_t)p * V, pack<T>(f));
                }
            }
        }
    }
}

template <typename T>
int layernorm_launch(const void* x, const void* w, const void* b, void* y, int64_t rows, int64_t cols, float eps,
                     hipStream_t s) {
    constexpr int V = Tr<T>::kVec;
    SS_REQUIRE(cols % V == 0 && cols / V <= kNormMaxPacks * 256, "layernorm: cols=%lld unsupported", (long long)cols);
    if (rows == 0) return SS_OK;
    if (cols / V <= 256 && rows >= 256) {
        const int R = rows >= 8192 ? tuning_get("layernorm_rows_per_wave", 1) : 1;
        if (R == 2)
            hipLaunchKernelGGL((layernorm_wave_kernel<T, 2>), dim3((unsigned)cdiv(rows, 8)), dim3(256), 0, s, (const T*)x,
                               (const T*)w, (const T*)b, (T*)y, rows, (int)cols, eps);
        else if (R == 4)
            hipLaunchKernelGGL((layernorm_wave_kernel<T, 4>), dim3((unsigned)cdiv(rows, 16)), dim3(256), 0, s, (const T*)x,
                               (const T*)w, (const T*)b, (T*)y, rows, (int)cols, eps);
        else
            hipLaunchKernelGGL((layernorm_wave_kernel<T, 1>), dim3((unsigned)cdiv(rows, 4)), dim3(256), 0, s, (const T*)x,
                               (const T*)w, (const T*)b, (T*)y, rows, (int)cols, eps);
        SS_LAUNCH_CHECK("layernorm_wave");
        return SS_OK;
    }
    hipLaunchKernelGGL(layernorm_kernel<T>, dim3((unsigned)rows), dim3(256), 0, s, (const T*)x, (const T*)w,
                       (const T*)b, (T*)y, (int)cols, eps);
    SS_LAUNCH_CHECK("layernorm");
    return SS_OK;
}

// =====================================================================================
// RoPE + KV append — rotate_half / apply_rotary_pos_emb (:158-173) and the cache
// concatenation (:239-242).  Arithmetic in T like the reference: round(q*cos),
// round(rot*sin), round(sum).  Keys are cached after RoPE.
// grid (M, n_heads); hd/2 threads: thread d handles the pair (d, d + hd/2).
// =====================================================================================
template <typename T>
__global__ void rope_kv_append_kernel(const T* __restrict__ qkv, T* __restrict__ q_out, T* __restrict__ kc,
                                      T* __restrict__ vc, const T* __restrict__ cos_t, const T* __restrict__ sin_t,
                                      const int32_t* __restrict__ pos_ids, int pos_start, int n_heads, int hd,
                                      const int32_t* __restrict__ kv_start_dev, int kv_start, int cache_cap) {
    const int m = blockIdx.x, h = blockIdx.y, d = threadIdx.x, half = hd >> 1;
    const int pos = pos_ids ? pos_ids[m] : pos_start + m;
    const int slot = (kv_start_dev ? *kv_start_dev : kv_start) + m;
    const int64_t E = (int64_t)n_heads * hd;
    const T* row = qkv + (int64_t)m * 3 * E + (int64_t)h * hd;
    const float c0 = Tr<T>::ld(cos_t + (int64_t)pos * hd + d), c1 = Tr<T>::ld(cos_t + (int64_t)pos * hd + d + half);
    const float s0 = Tr<T>::ld(sin_t + (int64_t)pos * hd + d), s1 = Tr<T>::ld(sin_t + (int64_t)pos * hd + d + half);
    {
        const float a = Tr<T>::ld(row + d), b = Tr<T>::ld(row + d + half);
        const float lo = Tr<T>::rnd(Tr<T>::rnd(a * c0) + Tr<T>::rnd(-b * s0));
        const float hi = Tr<T>::rnd(Tr<T>::rnd(b * c1) + Tr<T>::rnd(a * s1));
        T* qo = q_out + (int64_t)m * E + (int64_t)h * hd;
        Tr<T>::st(qo + d, lo);
        Tr<T>::st(qo + d + half, hi);
    }
    if (slot < cache_cap) {
        const float a = Tr<T>::ld(row + E + d), b = Tr<T>::ld(row + E + d + half);
        const float lo = Tr<T>::rnd(Tr<T>::rnd(a * c0) + Tr<T>::rnd(-b * s0));
        const float hi = Tr<T>::rnd(Tr<T>::rnd(b * c1) + Tr<T>::rnd(a * s1));
        T* ko = kc + ((int64_t)h * cache_cap + slot) * hd;
        Tr<T>::st(ko + d, lo);
        Tr<T>::st(ko + d + half, hi);
        T* vo = vc + ((int64_t)h * cache_cap + slot) * hd;
        vo[d] = row[2 * E + d];
        vo[d + half] = row[2 * E + d + half];
    }
}

template <typename T>
int rope_kv_append_launch(const void* qkv, void* q_out, void* kc, void* vc, const void* cos_t, const void* sin_t,
                          const int32_t* pos_ids, int64_t pos_start, int64_t M, int64_t n_heads, int64_t hd,
                          const int32_t* kv_start_dev, int64_t kv_start, int64_t cache_cap, hipStream_t s) {
    SS_REQUIRE(hd % 2 == 0 && hd <= 256, "rope: hd=%lld unsupported", (long long)hd);
    SS_REQUIRE(kv_start_dev || kv_start + M <= cache_cap, "rope_kv_append: cache overflow (%lld + %lld > %lld)",
               (long long)kv_start, (long long)M, (long long)cache_cap);
    if (M == 0) return SS_OK;
    hipLaunchKernelGGL(rope_kv_append_kernel<T>, dim3((unsigned)M, (unsigned)n_heads), dim3((unsigned)(hd / 2)), 0, s,
                       (const T*)qkv, (T*)q_out, (T*)kc, (T*)vc, (const T*)cos_t, (const T*)sin_t, pos_ids,
                       (int)pos_start, (int)n_heads, (int)hd, kv_start_dev, (int)kv_start, (int)cache_cap);
    SS_LAUNCH_CHECK("rope_kv_append");
    return SS_OK;
}

// =====================================================================================
// small element-wise kernels (grid-stride over 16-byte packs)
// =====================================================================================
__device__ __forceinline__ float silu_f(float g) { return g / (1.0f + expf(-g)); }
__device__ __forceinline__ float gelu_erf_f(float v) { return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f)); }

// out[r,i] = round(round(silu(g)) * u)   — act_fn(gate) * up in T (LlamaMLP, :190-191)
template <typename T>
__global__ __launch_bounds__(256) void silu_mul_kernel(const T* __restrict__ gu, T* __restrict__ out, int64_t rows,
                                                       int inter) {
    constexpr int V = Tr<T>::kVec;
    const int ppr = inter / V;
    const int64_t total = rows * ppr;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / ppr;
        const int p = (int)(i % ppr);
        float g[V], u[V];
        unpack<T>(ld16(gu + r * 2 * inter + (int64_t)p * V), g);
        unpack<T>(ld16(gu + r * 2 * inter + inter + (int64_t)p * V), u);
#pragma unroll
        for (int j = 0; j < V; ++j) g[j] = Tr<T>::rnd(silu_f(g[j])) * u[j];
        st16(out + r * inter + (int64_t)p * V, pack<T>(g));
    }
}

// y[b,r,:] = x[b,r,:] + p[r,:]
template <typename T>
__global__ __launch_bounds__(256) void add_bcast_kernel(const T* __restrict__ x, const T* __restrict__ p,
                                                        T* __restrict__ y, int64_t batch, int64_t rows, int cols,
                                                        int64_t x_bs) {
    constexpr int V = Tr<T>::kVec;
    const int ppr = cols / V;
    const int64_t per = rows * ppr, total = batch * per;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = i / per, rp = i % per;
        float a[V], c[V];
        unpack<T>(ld16(x + b * x_bs + rp * V), a);
        unpack<T>(ld16(p + rp * V), c);
#pragma unroll
        for (int j = 0; j < V; ++j) a[j] += c[j];
        st16(y + (b * per + rp) * V, pack<T>(a));
    }
}

template <typename T>
__global__ __launch_bounds__(256) void gather_rows_kernel(const T* __restrict__ table, const int32_t* __restrict__ ids,
                                                          T* __restrict__ out, int64_t n, int cols) {
    constexpr int V = Tr<T>::kVec;
    const int ppr = cols / V;
    const int64_t total = n * ppr;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / ppr;
        const int p = (int)(i % ppr);
        st16(out + r * cols + (int64_t)p * V, ld16(table + (int64_t)ids[r] * cols + (int64_t)p * V));
    }
}

template <typename T>
__global__ __launch_bounds__(256) void scatter_rows_kernel(const T* __restrict__ src, const int32_t* __restrict__ idx,
                                                           T* __restrict__ dst, int64_t n, int cols) {
    constexpr int V = Tr<T>::kVec;
    const int ppr = cols / V;
    const int64_t total = n * ppr;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / ppr;
        const int p = (int)(i % ppr);
        st16(dst + (int64_t)idx[r] * cols + (int64_t)p * V, ld16(src + r * cols + (int64_t)p * V));
    }
}

// img [B,3,S,S] -> out [B*G*G, kpad], column (c*P + dy)*P + dx, zero beyond 3*P*P
template <typename T>
__global__ __launch_bounds__(256) void im2col_patch_kernel(const T* __restrict__ img, T* __restrict__ out,
                                                           int64_t batch, int S, int P, int kpad) {
    const int G = S / P;
    const int64_t total = batch * G * G * kpad;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int col = (int)(i % kpad);
        const int64_t tok = i / kpad;
        const int gx = (int)(tok % G), gy = (int)((tok / G) % G);
        const int64_t b = tok / ((int64_t)G * G);
        T v;
        if (col < 3 * P * P) {
            const int dx = col % P, dy = (col / P) % P, c = col / (P * P);
            v = img[((b * 3 + c) * S + (gy * P + dy)) * (int64_t)S + gx * P + dx];
        } else {
            Tr<T>::st(&v, 0.f);
        }
        out[i] = v;
    }
}

// F.normalize(x) over dim=1 of [B,L,C]: one thread per (b, c) column
template <typename T>
__global__ __launch_bounds__(256) void l2norm_dim1_kernel(const T* __restrict__ x, T* __restrict__ y, int64_t batch,
                                                          int L, int C) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= batch * C) return;
    const int64_t b = i / C;
    const int c = (int)(i % C);
    const T* xb = x + b * (int64_t)L * C + c;
    float s = 0.f;
    for (int l = 0; l < L; ++l) { const float v = Tr<T>::ld(xb + (int64_t)l * C); s = fmaf(v, v, s); }
    // torch: x / x.norm(2, dim, keepdim).clamp_min(eps); the norm is rounded to T first
    const float nrm = fmaxf(Tr<T>::rnd(sqrtf(s)), 1e-12f);
    T* yb = y + b * (int64_t)L * C + c;
    for (int l = 0; l < L; ++l) Tr<T>::st(yb + (int64_t)l * C, Tr<T>::ld(xb + (int64_t)l * C) / nrm);
}

static inline unsigned ew_grid(int64_t total) {
    int64_t g = (total + 255) / 256;
    if (g > 256 * 16) g = 256 * 16;
    if (g < 1) g = 1;
    return (unsigned)g;
}

template <typename T>
int silu_mul_launch(const void* gu, void* out, int64_t rows, int64_t inter, hipStream_t s) {
    SS_REQUIRE(inter % Tr<T>::kVec == 0, "silu_mul: inter %% %d != 0", Tr<T>::kVec);
    if (rows == 0) return SS_OK;
    hipLaunchKernelGGL(silu_mul_kernel<T>, dim3(ew_grid(rows * inter / Tr<T>::kVec)), dim3(256), 0, s, (const T*)gu,
                       (T*)out, rows, (int)inter);
    SS_LAUNCH_CHECK("silu_mul");
    return SS_OK;
}
template <typename T>
int add_bcast_launch(const void* x, const void* p, void* y, int64_t batch, int64_t rows, int64_t cols, int64_t x_bs,
                     hipStream_t s) {
    SS_REQUIRE(cols % Tr<T>::kVec == 0, "add_bcast: cols %% %d != 0", Tr<T>::kVec);
    if (batch * rows == 0) return SS_OK;
    hipLaunchKernelGGL(add_bcast_kernel<T>, dim3(ew_grid(batch * rows * cols / Tr<T>::kVec)), dim3(256), 0, s,
                       (const T*)x, (const T*)p, (T*)y, batch, rows, (int)cols, x_bs);
    SS_LAUNCH_CHECK("add_bcast");
    return SS_OK;
}
template <typename T>
int gather_rows_launch(const void* table, const int32_t* ids, void* out, int64_t n, int64_t cols, hipStream_t s) {
    SS_REQUIRE(cols % Tr<T>::kVec == 0, "gather_rows: cols %% %d != 0", Tr<T>::kVec);
    if (n == 0) return SS_OK;
    hipLaunchKernelGGL(gather_rows_kernel<T>, dim3(ew_grid(n * cols / Tr<T>::kVec)), dim3(256), 0, s, (const T*)table,
                       ids, (T*)out, n, (int)cols);
    SS_LAUNCH_CHECK("gather_rows");
    return SS_OK;
}
template <typename T>
int scatter_rows_launch(const void* src, const int32_t* idx, void* dst, int64_t n, int64_t cols, hipStream_t s) {
    SS_REQUIRE(cols % Tr<T>::kVec == 0, "scatter_rows: cols %% %d != 0", Tr<T>::kVec);
    if (n == 0) return SS_OK;
    hipLaunchKernelGGL(scatter_rows_kernel<T>, dim3(ew_grid(n * cols / Tr<T>::kVec)), dim3(256), 0, s, (const T*)src,
                       idx, (T*)dst, n, (int)cols);
    SS_LAUNCH_CHECK("scatter_rows");
    return SS_OK;
}
template <typename T>
int im2col_patch_launch(const void* img, void* out, int64_t batch, int64_t S, int64_t P, int64_t kpad, hipStream_t s) {
    SS_REQUIRE(S % P == 0 && kpad >= 3 * P * P, "im2col_patch: bad geometry");
    if (batch == 0) return SS_OK;
    hipLaunchKernelGGL(im2col_patch_kernel<T>, dim3(ew_grid(batch * (S / P) * (S / P) * kpad)), dim3(256), 0, s,
                       (const T*)img, (T*)out, batch, (int)S, (int)P, (int)kpad);
    SS_LAUNCH_CHECK("im2col_patch");
    return SS_OK;
}
template <typename T>
int l2norm_dim1_launch(const void* x, void* y, int64_t batch, int64_t L, int64_t C, hipStream_t s) {
    if (batch * C == 0) return SS_OK;
    hipLaunchKernelGGL(l2norm_dim1_kernel<T>, dim3((unsigned)cdiv(batch * C, 256)), dim3(256), 0, s, (const T*)x,
                       (T*)y, batch, (int)L, (int)C);
    SS_LAUNCH_CHECK("l2normalize_dim1");
    return SS_OK;
}

// internal entry used by the LLaMA engine (device-side kv_start)
int rope_kv_append_dev(const void* qkv, void* q_out, void* kc, void* vc, const void* cos_t, const void* sin_t,
                       const int32_t* pos_ids, int64_t M, int64_t n_heads, int64_t hd, const int32_t* kv_start_dev,
                       int64_t cache_cap, int dtype, hipStream_t s) {
    return SS_DISPATCH(dtype, rope_kv_append_launch, qkv, q_out, kc, vc, cos_t, sin_t, pos_ids, 0, M, n_heads, hd,
                       kv_start_dev, 0, cache_cap, s);
}

}  // namespace ss

using namespace ss;

extern "C" {

int ss_rmsnorm(const void* x, const void* w, void* y, int64_t rows, int64_t cols, float eps, int dtype, void* stream) {
    return SS_DISPATCH(dtype, rmsnorm_launch, x, w, y, rows, cols, eps, (hipStream_t)stream);
}
int ss_layernorm(const void* x, const void* w, const void* b, void* y, int64_t rows, int64_t cols, float eps,
                 int dtype, void* stream) {
    return SS_DISPATCH(dtype, layernorm_launch, x, w, b, y, rows, cols, eps, (hipStream_t)stream);
}
int ss_add_bcast(const void* x, const void* p, void* y, int64_t batch, int64_t rows, int64_t cols,
                 int64_t x_batch_stride, int dtype, void* stream) {
    return SS_DISPATCH(dtype, add_bcast_launch, x, p, y, batch, rows, cols, x_batch_stride, (hipStream_t)stream);
}
int ss_silu_mul(const void* gu, void* out, int64_t rows, int64_t inter, int dtype, void* stream) {
    return SS_DISPATCH(dtype, silu_mul_launch, gu, out, rows, inter, (hipStream_t)stream);
}
int ss_gather_rows(const void* table, const int32_t* ids, void* out, int64_t n, int64_t cols, int dtype,
                   void* stream) {
    return SS_DISPATCH(dtype, gather_rows_launch, table, ids, out, n, cols, (hipStream_t)stream);
}
int ss_scatter_rows(const void* src, const int32_t* idx, void* dst, int64_t n, int64_t cols, int dtype,
                    void* stream) {
    return SS_DISPATCH(dtype, scatter_rows_launch, src, idx, dst, n, cols, (hipStream_t)stream);
}
int ss_im2col_patch(const void* img, void* out, int64_t batch, int64_t size, int64_t patch, int64_t kpad, int dtype,
                    void* stream) {
    return SS_DISPATCH(dtype, im2col_patch_launch, img, out, batch, size, patch, kpad, (hipStream_t)stream);
}
int ss_l2normalize_dim1(const void* x, void* y, int64_t batch, int64_t len, int64_t cols, int dtype, void* stream) {
    return SS_DISPATCH(dtype, l2norm_dim1_launch, x, y, batch, len, cols, (hipStream_t)stream);
}
int ss_rope_kv_append(const void* qkv, void* q_out, void* kcache, void* vcache, const void* cos, const void* sin,
                      const int32_t* pos_ids, int64_t pos_start, int64_t M, int64_t n_heads, int64_t hd,
                      int64_t kv_start, int64_t cache_cap, int dtype, void* stream) {
    return SS_DISPATCH(dtype, rope_kv_append_launch, qkv, q_out, kcache, vcache, cos, sin, pos_ids, pos_start, M,
                       n_heads, hd, (const int32_t*)nullptr, kv_start, cache_cap, (hipStream_t)stream);
}

}  // extern "C"
