// Batch-1 decode projection  y[N] = W[N,K] · x[K]  — the HBM-bound hot kernel of the MLLM
// decode loop (13.22 GB of bf16 weights per generated token).
//
// Design (MI355X): wave-autonomous weight streaming.
//   * every wave keeps its slice of x in registers for the whole launch (x is 8-22 KB and
//     L2/L1 resident; lane l owns elements {it*64*V + l*V .. +V} for it < NIT), so the
//     inner loop is nothing but 16-byte non-temporal weight loads + v_dot2c_f32_bf16:
//     no LDS, no barriers, no cross-wave traffic;
//   * a wave owns ROWS consecutive output rows per trip and issues all their loads
//     (ROWS x min(NIT,8) x 16 B per lane) before the first use, so ≥8-16 KB per wave are
//     in flight; rows are dealt to waves round-robin so the chip sweeps W sequentially;
//   * optional fused prologue: RMSNorm of x (each wave recomputes the 4096-element
//     statistic redundantly from its registers: cheaper than a kernel boundary);
//   * epilogues: +bias, +residual (in T, like `residual + hidden`, :352,:359),
//     SiLU(gate)·up for the fused [gate; up] projection (LlamaMLP.forward, :190-191).
// Reference call sites: modeling_llama_xformer.py:228-230 (q/k/v), :297 (o), :191 (MLP),
// :759 (lm_head); LlamaRMSNorm :107-115 for the prologue.
#include "ss_common.h"

namespace ss {

struct GemvArgs {
    const void* W;
    const void* x;
    void* y;
    const void* norm_w;
    const void* bias;
    const void* residual;
    const int32_t* done_flag;  // optional device flag: skip all work when *done_flag != 0
    int N, K, epi;
    float eps;
    int use_nt;
};

__device__ __forceinline__ float silu_g(float g) { return g / (1.0f + expf(-g)); }

template <typename T, int NIT, int ROWS, bool PIPE>
__global__ __launch_bounds__(256) void gemv_kernel(const GemvArgs a) {
    constexpr int V = Tr<T>::kVec;
    if (a.done_flag && *a.done_flag) return;
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int nwaves = gridDim.x * (blockDim.x >> 6);
    const T* __restrict__ W = (const T*)a.W;
    const int K = a.K, N = a.N;
    const bool silu = (a.epi & SS_EPI_SILU_MUL) != 0;

    // logical rows: silu -> N rows, each the pair (n, n+N) of W;  else ROWS consecutive rows.
    const int ngroups = silu ? N : (N + ROWS - 1) / ROWS;
    auto row_of = [&](int g, int r) -> int64_t { return silu ? (int64_t)g + (int64_t)r * N : (int64_t)g * ROWS + r; };
    auto load_group = [&](int g, uint4 (&wv)[ROWS][NIT]) {
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
            const int64_t row = row_of(g, r);
            const bool ok = silu || row < N;
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int k = (it * 64 + lane) * V;
                if (ok && k < K) {
                    const T* p = W + row * K + k;
                    wv[r][it] = a.use_nt ? ld_nt16(p) : ld16(p);
                } else {
                    wv[r][it] = make_uint4(0, 0, 0, 0);
                }
            }
        }
    };

    // The first group's weight loads go out BEFORE the x prologue: the HBM round trip overlaps the
    // (L2-served) x fetch and the RMSNorm arithmetic.
    uint4 wv[ROWS][NIT];
    int g = wave;
    if (PIPE && g < ngroups) load_group(g, wv);

    // ---- x slice into registers (+ fused RMSNorm) ------------------------------------------
    uint4 xr[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int k = (it * 64 + lane) * V;
        xr[it] = (k < K) ? ld16((const T*)a.x + k) : make_uint4(0, 0, 0, 0);
    }
    if (a.norm_w) {
        float ssq = 0.f;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            float f[V];
            unpack<T>(xr[it], f);
#pragma unroll
            for (int j = 0; j < V; ++j) ssq = fmaf(f[j], f[j], ssq);
        }
        ssq = wave_sum(ssq);
        const float rstd = 1.0f / sqrtf(ssq / (float)K + a.eps);
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int k = (it * 64 + lane) * V;
            if (k < K) {
                float f[V], gw[V];
                unpack<T>(xr[it], f);
                unpack<T>(ld16((const T*)a.norm_w + k), gw);
#pragma unroll
                for (int j = 0; j < V; ++j) f[j] = gw[j] * Tr<T>::rnd(f[j] * rstd);
                xr[it] = pack<T>(f);
            }
        }
    }

    // ---- stream the rows: dot the resident group, immediately re-issue the registers for the next
    // group, then reduce/store while those loads fly ---------------------------------------------
    while (g < ngroups) {
        // PIPE=0 (default): plain load -> dot -> reduce per group: 125 VGPRs, 4 waves/SIMD; measured
        // faster on MI355X than both software-pipelined forms (177-198 VGPRs, 2 waves/SIMD).
        if (!PIPE) load_group(g, wv);
        float acc[ROWS];
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
            acc[r] = 0.f;
#pragma unroll
            for (int it = 0; it < NIT; ++it) acc[r] = dot_pack<T>(wv[r][it], xr[it], acc[r]);
        }
        const int gn = g + nwaves;
        if (PIPE && gn < ngroups) load_group(gn, wv);
#pragma unroll
        for (int r = 0; r < ROWS; ++r) acc[r] = wave_sum(acc[r]);
        if (lane == 0) {
            if (silu) {
                // gate = round(acc0), up = round(acc1); y = round(round(silu(gate)) * up)
                const float gt = Tr<T>::rnd(acc[0]), up = Tr<T>::rnd(acc[ROWS > 1 ? 1 : 0]);
                Tr<T>::st((T*)a.y + g, Tr<T>::rnd(silu_g(gt)) * up);
            } else {
#pragma unroll
                for (int r = 0; r < ROWS; ++r) {
                    const int64_t row = row_of(g, r);
                    if (row >= N) continue;
                    float v = acc[r];
                    if (a.epi & SS_EPI_BIAS) v += Tr<T>::ld((const T*)a.bias + row);
                    v = Tr<T>::rnd(v);
                    if (a.epi & SS_EPI_RESIDUAL) v += Tr<T>::ld((const T*)a.residual + row);
                    Tr<T>::st((T*)a.y + row, v);
                }
            }
        }
        g = gn;
    }
}

// Long-K variant (K > 64*V*8, e.g. the 11008-wide down projection): the x slice would cost
// 88+ VGPRs per lane, so x (optionally RMS-normalised) is staged once per block in LDS and read
// back with conflict-free ds_read_b128 (lanes read consecutive 16-byte slots); the k loop runs in
// chunks of 8 x 64 packs with ROWS x 8 weight loads in flight.
template <typename T, int ROWS, bool PIPE>
__global__ __launch_bounds__(256) void gemv_ldsx_kernel(const GemvArgs a) {
    constexpr int V = Tr<T>::kVec;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    uint4* xs = reinterpret_cast<uint4*>(smem_raw);  // [npack_pad]
    __shared__ float red[16];
    if (a.done_flag && *a.done_flag) return;
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int nwaves = gridDim.x * 4;
    const T* __restrict__ W = (const T*)a.W;
    const int K = a.K, N = a.N;
    const bool silu = (a.epi & SS_EPI_SILU_MUL) != 0;
    const int npack = K / V;
    const int nchunk = (npack + 511) / 512;  // chunks of 8 wave-iterations
    const int npack_pad = nchunk * 512;

    const int ngroups = silu ? N : (N + ROWS - 1) / ROWS;
    auto row_of = [&](int g, int r) -> int64_t { return silu ? (int64_t)g + (int64_t)r * N : (int64_t)g * ROWS + r; };
    auto load_tile = [&](int g, int c, uint4 (&wv)[ROWS][8]) {
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
            const int64_t row = row_of(g, r);
            const bool ok = silu || row < N;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int p = (c * 8 + i) * 64 + lane;
                if (ok && p < npack) {
                    const T* ptr = W + row * K + (int64_t)p * V;
                    wv[r][i] = a.use_nt ? ld_nt16(ptr) : ld16(ptr);
                } else {
                    wv[r][i] = make_uint4(0, 0, 0, 0);
                }
            }
        }
    };
    int g = wave, c = 0;
    uint4 wv[ROWS][8];
    float acc[ROWS];
#pragma unroll
    for (int r = 0; r < ROWS; ++r) acc[r] = 0.f;
    if (PIPE && g < ngroups) load_tile(g, 0, wv);  // first tile's HBM round trip overlaps the x staging

    float ssq = 0.f;
    for (int p = threadIdx.x; p < npack_pad; p += 256) {
        uint4 v = make_uint4(0, 0, 0, 0);
        if (p < npack) {
            v = ld16((const T*)a.x + (int64_t)p * V);
            if (a.norm_w) {
                float f[V];
                unpack<T>(v, f);
#pragma unroll
                for (int j = 0; j < V; ++j) ssq = fmaf(f[j], f[j], ssq);
            }
        }
        xs[p] = v;
    }
    if (a.norm_w) {
        const float rstd = 1.0f / sqrtf(block_sum(ssq, red) / (float)K + a.eps);
        for (int p = threadIdx.x; p < npack; p += 256) {
            float f[V], gw[V];
            unpack<T>(xs[p], f);
            unpack<T>(ld16((const T*)a.norm_w + (int64_t)p * V), gw);
#pragma unroll
            for (int j = 0; j < V; ++j) f[j] = gw[j] * Tr<T>::rnd(f[j] * rstd);
            xs[p] = pack<T>(f);
        }
    }
    __syncthreads();

    while (g < ngroups) {
        if (!PIPE) load_tile(g, c, wv);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const uint4 xv = xs[(c * 8 + i) * 64 + lane];
#pragma unroll
            for (int r = 0; r < ROWS; ++r) acc[r] = dot_pack<T>(wv[r][i], xv, acc[r]);
        }
        // next (group, chunk) tile goes out before the reduction of the finished row
        int gn = g, cn = c + 1;
        if (cn == nchunk) { cn = 0; gn = g + nwaves; }
        if (PIPE && gn < ngroups) load_tile(gn, cn, wv);
        if (cn == 0) {
#pragma unroll
            for (int r = 0; r < ROWS; ++r) acc[r] = wave_sum(acc[r]);
            if (lane == 0) {
                if (silu) {
                    const float gt = Tr<T>::rnd(acc[0]), up = Tr<T>::rnd(acc[ROWS > 1 ? 1 : 0]);
                    Tr<T>::st((T*)a.y + g, Tr<T>::rnd(silu_g(gt)) * up);
                } else {
#pragma unroll
                    for (int r = 0; r < ROWS; ++r) {
                        const int64_t row = row_of(g, r);
                        if (row >= N) continue;
                        float v = acc[r];
                        if (a.epi & SS_EPI_BIAS) v += Tr<T>::ld((const T*)a.bias + row);
                        v = Tr<T>::rnd(v);
                        if (a.epi & SS_EPI_RESIDUAL) v += Tr<T>::ld((const T*)a.residual + row);
                        Tr<T>::st((T*)a.y + row, v);
                    }
                }
            }
#pragma unroll
            for (int r = 0; r < ROWS; ++r) acc[r] = 0.f;
        }
        g = gn;
        c = cn;
    }
}

template <typename T, int NIT>
static int gemv_launch_nit(const GemvArgs& a, int blocks, hipStream_t s) {
    // ROWS=2 keeps 16 x 16 B per lane in flight at NIT=8 (and SiLU pairs need exactly 2 rows)
    if (tuning_get("gemv_pipe", 0))
        hipLaunchKernelGGL((gemv_kernel<T, NIT, 2, true>), dim3((unsigned)blocks), dim3(256), 0, s, a);
    else
        hipLaunchKernelGGL((gemv_kernel<T, NIT, 2, false>), dim3((unsigned)blocks), dim3(256), 0, s, a);
    SS_LAUNCH_CHECK("gemv");
    return SS_OK;
}

template <typename T>
int gemv_launch(const void* W, const void* x, void* y, int64_t N, int64_t K, const void* norm_w, float eps,
                const void* bias, const void* residual, int epi, const int32_t* done_flag, hipStream_t s) {
    constexpr int V = Tr<T>::kVec;
    SS_REQUIRE(K % V == 0 && K > 0 && N > 0, "gemv: K=%lld must be a positive multiple of %d", (long long)K, V);
    SS_REQUIRE(!(epi & SS_EPI_SILU_MUL) || !(epi & (SS_EPI_BIAS | SS_EPI_RESIDUAL | SS_EPI_GELU)),
               "gemv: SILU_MUL cannot be combined with other epilogues");
    SS_REQUIRE(!(epi & SS_EPI_GELU), "gemv: GELU epilogue not supported");
    GemvArgs a;
    a.W = W; a.x = x; a.y = y; a.norm_w = norm_w; a.bias = bias; a.residual = residual; a.done_flag = done_flag;
    a.N = (int)N; a.K = (int)K; a.epi = epi; a.eps = eps;
    a.use_nt = tuning_get("gemv_nt", 1);
    const int nit = cdiv(K, 64 * V);
    const int64_t groups = (epi & SS_EPI_SILU_MUL) ? N : (N + 1) / 2;
    // waves: enough to fill the chip, but several row-groups per wave so the x prologue amortises
    // auto: ~2.5k waves (8-12 per CU) measured best on MI355X for every LLaMA-7B projection
    int gpw = tuning_get("gemv_groups_per_wave", 0);
    if (gpw <= 0) { gpw = (int)((groups + 1280) / 2560); if (gpw < 1) gpw = 1; }
    int64_t waves = (groups + gpw - 1) / gpw;
    int blocks = (int)((waves + 3) / 4);
    const int max_blocks = tuning_get("gemv_max_blocks", 256 * 8);
    if (blocks > max_blocks) blocks = max_blocks;
    if (blocks < 1) blocks = 1;
    if (nit <= 1) return gemv_launch_nit<T, 1>(a, blocks, s);
    if (nit <= 2) return gemv_launch_nit<T, 2>(a, blocks, s);
    if (nit <= 4) return gemv_launch_nit<T, 4>(a, blocks, s);
    if (nit <= 8 && !tuning_get("gemv_force_lds", 0)) return gemv_launch_nit<T, 8>(a, blocks, s);
    const size_t lds = (size_t)cdiv(K / V, 512) * 512 * 16;
    SS_REQUIRE(lds <= 128 * 1024, "gemv: K=%lld too large", (long long)K);
    if (tuning_get("gemv_pipe", 0))
        hipLaunchKernelGGL((gemv_ldsx_kernel<T, 2, true>), dim3((unsigned)blocks), dim3(256), lds, s, a);
    else
        hipLaunchKernelGGL((gemv_ldsx_kernel<T, 2, false>), dim3((unsigned)blocks), dim3(256), lds, s, a);
    SS_LAUNCH_CHECK("gemv_ldsx");
    return SS_OK;
}

int gemv_dev(const void* W, const void* x, void* y, int64_t N, int64_t K, const void* norm_w, float eps,
             const void* bias, const void* residual, int epi, const int32_t* done_flag, int dtype, hipStream_t s) {
    return SS_DISPATCH(dtype, gemv_launch, W, x, y, N, K, norm_w, eps, bias, residual, epi, done_flag, s);
}

}  // namespace ss

extern "C" int ss_gemv(const void* W, const void* x, void* y, int64_t N, int64_t K, const void* norm_w, float eps,
                       const void* bias, const void* residual, int epilogue, int dtype, void* stream) {
    return ss::gemv_dev(W, x, y, N, K, norm_w, eps, bias, residual, epilogue, nullptr, dtype, (hipStream_t)stream);
}
