// Decode projection  y[b][N] = W[N,K] · x[b][K], b < NB <= 4  — the HBM-bound hot kernel of the MLLM
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
    const void* x;         // [nb][x_ld]
    void* y;               // [nb][y_ld]
    const void* norm_w;
    const void* bias;
    const void* residual;  // [nb][res_ld]
    const int32_t* done_flag;  // optional device flags (one per sequence, done_stride ints apart):
                               // the launch is skipped when every sequence's flag is set
    int N, K, epi;
    float eps;
    int use_nt;
    int nb, done_stride;
    int64_t x_ld, y_ld, res_ld;
};

__device__ __forceinline__ float silu_g(float g) { return g / (1.0f + expf(-g)); }

__device__ __forceinline__ bool gemv_all_done(const GemvArgs& a) {
    if (!a.done_flag) return false;
    for (int b = 0; b < a.nb; ++b)
        if (!a.done_flag[(int64_t)b * a.done_stride]) return false;
    return true;
}

// epilogue of one finished row-group for sequence b (lane 0 of the wave)
template <typename T, int ROWS>
__device__ __forceinline__ void gemv_store(const GemvArgs& a, int b, int g, const float (&acc)[ROWS], bool silu) {
    const int N = a.N;
    T* y = (T*)a.y + (int64_t)b * a.y_ld;
    if (silu) {
        // gate = round(acc0), up = round(acc1); y = round(round(silu(gate)) * up)
        const float gt = Tr<T>::rnd(acc[0]), up = Tr<T>::rnd(acc[ROWS > 1 ? 1 : 0]);
        Tr<T>::st(y + g, Tr<T>::rnd(silu_g(gt)) * up);
    } else {
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
            const int64_t row = (int64_t)g * ROWS + r;
            if (row >= N) continue;
            float v = acc[r];
            if (a.epi & SS_EPI_BIAS) v += Tr<T>::ld((const T*)a.bias + row);
            v = Tr<T>::rnd(v);
            if (a.epi & SS_EPI_RESIDUAL) v += Tr<T>::ld((const T*)a.residual + (int64_t)b * a.res_ld + row);
            Tr<T>::st(y + row, v);
        }
    }
}

// NB = sequences sharing one sweep of W (the decode batch: every weight pack is dotted with NB
// resident x slices, so the HBM traffic per generated token falls as 1/NB).
template <typename T, int NIT, int ROWS, int NB>
__global__ __launch_bounds__(256) void gemv_kernel(const GemvArgs a) {
    constexpr int V = Tr<T>::kVec;
    if (gemv_all_done(a)) return;
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

    // ---- x slices into registers (+ fused RMSNorm) -----------------------------------------
    // The RMS statistic is recomputed by every wave from its registers (cheap, and no cross-wave
    // traffic), but the element-wise normalise/round/scale (~6 VALU ops per element in bf16) is shared:
    // each of the block's 4 waves normalises a quarter of the row into LDS and all read it back.
    constexpr bool COOP = (NIT % 4 == 0);
    constexpr int NQ = COOP ? NIT / 4 : 1;
    __shared__ uint4 xn_s[COOP ? NB * NIT * 64 : 1];
    uint4 xr[NB][NIT];
    uint4 xq[NB][NQ], gq[NQ];
    const int wid = threadIdx.x >> 6;
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const T* xb = (const T*)a.x + (int64_t)b * a.x_ld;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int k = (it * 64 + lane) * V;
            xr[b][it] = (k < K) ? ld16(xb + k) : make_uint4(0, 0, 0, 0);
        }
        if (COOP && a.norm_w) {
#pragma unroll
            for (int j4 = 0; j4 < NQ; ++j4) {
                const int k = ((wid * NQ + j4) * 64 + lane) * V;
                xq[b][j4] = (k < K) ? ld16(xb + k) : make_uint4(0, 0, 0, 0);
            }
        }
    }
    if (a.norm_w) {
        if constexpr (COOP) {
#pragma unroll
            for (int j4 = 0; j4 < NQ; ++j4) {
                const int k = ((wid * NQ + j4) * 64 + lane) * V;
                gq[j4] = (k < K) ? ld16((const T*)a.norm_w + k) : make_uint4(0, 0, 0, 0);
            }
        }
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            float ssq = 0.f;
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                float f[V];
                unpack<T>(xr[b][it], f);
#pragma unroll
                for (int j = 0; j < V; ++j) ssq = fmaf(f[j], f[j], ssq);
            }
            ssq = wave_sum(ssq);
            const float rstd = 1.0f / sqrtf(ssq / (float)K + a.eps);
            if constexpr (COOP) {
#pragma unroll
                for (int j4 = 0; j4 < NQ; ++j4) {
                    float f[V], gw[V];
                    unpack<T>(xq[b][j4], f);
                    unpack<T>(gq[j4], gw);
#pragma unroll
                    for (int j = 0; j < V; ++j) f[j] = gw[j] * Tr<T>::rnd(f[j] * rstd);
                    xn_s[(b * NIT + wid * NQ + j4) * 64 + lane] = pack<T>(f);   // k >= K packs are zeros already
                }
            } else {
#pragma unroll
                for (int it = 0; it < NIT; ++it) {
                    const int k = (it * 64 + lane) * V;
                    if (k < K) {
                        float f[V], gw[V];
                        unpack<T>(xr[b][it], f);
                        unpack<T>(ld16((const T*)a.norm_w + k), gw);
#pragma unroll
                        for (int j = 0; j < V; ++j) f[j] = gw[j] * Tr<T>::rnd(f[j] * rstd);
                        xr[b][it] = pack<T>(f);
                    }
                }
            }
        }
        if constexpr (COOP) {
            __syncthreads();
#pragma unroll
            for (int b = 0; b < NB; ++b)
#pragma unroll
                for (int it = 0; it < NIT; ++it) xr[b][it] = xn_s[(b * NIT + it) * 64 + lane];
        }
    }

    // ---- stream the rows: plain load -> dot -> reduce per group (125 VGPRs at NB=1, 4 waves/SIMD;
    // measured faster on MI355X than software-pipelined forms that cost 177-198 VGPRs) ------------
    // (requesting the first group's weights before the prologue was measured: the 64 extra live VGPRs
    // cost two waves/SIMD of occupancy and more than the overlap gains)
    uint4 wv[ROWS][NIT];
    for (int g = wave; g < ngroups; g += nwaves) {
        load_group(g, wv);
        float acc[NB][ROWS];
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int r = 0; r < ROWS; ++r) acc[b][r] = 0.f;
#pragma unroll
        for (int it = 0; it < NIT; ++it)
#pragma unroll
            for (int r = 0; r < ROWS; ++r)
#pragma unroll
                for (int b = 0; b < NB; ++b) acc[b][r] = dot_pack<T>(wv[r][it], xr[b][it], acc[b][r]);
#pragma unroll
        for (int b = 0; b < NB; ++b) {
#pragma unroll
            for (int r = 0; r < ROWS; ++r) acc[b][r] = wave_sum(acc[b][r]);
            if (lane == 0) gemv_store<T, ROWS>(a, b, g, acc[b], silu);
        }
    }
}

// Long-K / wide-batch variant (K > 64*V*8, e.g. the 11008-wide down projection, or NB x slices that
// do not fit the register file): x (optionally RMS-normalised) is staged once per block in LDS and
// read back with conflict-free ds_read_b128 (lanes read consecutive 16-byte slots); the k loop runs
// in chunks of 8 x 64 packs with ROWS x 8 weight loads in flight.
template <typename T, int ROWS, int NB>
__global__ __launch_bounds__(1024) void gemv_ldsx_kernel(const GemvArgs a) {
    constexpr int V = Tr<T>::kVec;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    uint4* xs = reinterpret_cast<uint4*>(smem_raw);  // [NB][npack_pad]
    if (gemv_all_done(a)) return;
    const int lane = threadIdx.x & 63;
    const int wpb = blockDim.x >> 6;
    const int wave = blockIdx.x * wpb + (threadIdx.x >> 6);
    const int nwaves = gridDim.x * wpb;
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

    // ---- stage the NB activation rows in LDS: ONE global read of x per block ------------------------------------
    // (every block needs all of x; at NB = 4 the old per-wave statistic + second read pulled more bytes of x through
    // L2 than the launch streams weights from HBM)
    float ssq[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) ssq[b] = 0.f;
#pragma unroll 2
    for (int p = threadIdx.x; p < npack_pad; p += blockDim.x) {
        const bool ok = p < npack;
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const uint4 v = ok ? ld16((const T*)a.x + (int64_t)b * a.x_ld + (int64_t)p * V) : make_uint4(0, 0, 0, 0);
            if (a.norm_w) {
                float f[V];
                unpack<T>(v, f);
#pragma unroll
                for (int j = 0; j < V; ++j) ssq[b] = fmaf(f[j], f[j], ssq[b]);
            }
            xs[(size_t)b * npack_pad + p] = v;
        }
    }
    if (a.norm_w) {
        // fused RMSNorm: block-wide statistic in a fixed order (lane tree, then waves 0..n-1), then each thread
        // normalises its own packs in place
        __shared__ float red[NB][16];
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const float w = wave_sum(ssq[b]);
            if (lane == 0) red[b][threadIdx.x >> 6] = w;
        }
        __syncthreads();
        float rstd[NB];
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            float t = 0.f;
            for (int w = 0; w < wpb; ++w) t += red[b][w];
            rstd[b] = 1.0f / sqrtf(t / (float)K + a.eps);
        }
        for (int p = threadIdx.x; p < npack; p += blockDim.x) {
            float gw[V];
            unpack<T>(ld16((const T*)a.norm_w + (int64_t)p * V), gw);
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                float f[V];
                unpack<T>(xs[(size_t)b * npack_pad + p], f);
#pragma unroll
                for (int j = 0; j < V; ++j) f[j] = gw[j] * Tr<T>::rnd(f[j] * rstd[b]);
                xs[(size_t)b * npack_pad + p] = pack<T>(f);
            }
        }
    }
    __syncthreads();

    uint4 wv[ROWS][8];
    for (int g = wave; g < ngroups; g += nwaves) {
        float acc[NB][ROWS];
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int r = 0; r < ROWS; ++r) acc[b][r] = 0.f;
        for (int c = 0; c < nchunk; ++c) {
            load_tile(g, c, wv);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
#pragma unroll
                for (int b = 0; b < NB; ++b) {
                    const uint4 xv = xs[(size_t)b * npack_pad + (c * 8 + i) * 64 + lane];
#pragma unroll
                    for (int r = 0; r < ROWS; ++r) acc[b][r] = dot_pack<T>(wv[r][i], xv, acc[b][r]);
                }
            }
        }
#pragma unroll
        for (int b = 0; b < NB; ++b) {
#pragma unroll
            for (int r = 0; r < ROWS; ++r) acc[b][r] = wave_sum(acc[b][r]);
            if (lane == 0) gemv_store<T, ROWS>(a, b, g, acc[b], silu);
        }
    }
}

template <typename T, int NIT, int NB>
static int gemv_launch_reg(const GemvArgs& a, int blocks, hipStream_t s) {
    // ROWS=2 keeps 16 x 16 B per lane in flight at NIT=8 (and SiLU pairs need exactly 2 rows)
    hipLaunchKernelGGL((gemv_kernel<T, NIT, 2, NB>), dim3((unsigned)blocks), dim3(256), 0, s, a);
    SS_LAUNCH_CHECK("gemv");
    return SS_OK;
}

template <typename T, int NB>
static int gemv_launch_nb(const GemvArgs& a, int nit, int64_t waves, hipStream_t s) {
    constexpr int V = Tr<T>::kVec;
    const int max_blocks = tuning_get("gemv_max_blocks", 256 * 8);
    auto blocks_for = [&](int wpb) {
        int64_t b = (waves + wpb - 1) / wpb;
        if (b > max_blocks) b = max_blocks;
        return (int)(b < 1 ? 1 : b);
    };
    // x slices live in registers while NB * NIT <= reg budget (16 packs = 64 VGPRs at NB=2, NIT=8)
    const int reg_budget = tuning_get("gemv_x_reg_packs", 16);
    if (nit <= 8 && NB * nit <= reg_budget && !tuning_get("gemv_force_lds", 0)) {
        const int blocks = blocks_for(4);
        if (nit <= 1) return gemv_launch_reg<T, 1, NB>(a, blocks, s);
        if (nit <= 2) return gemv_launch_reg<T, 2, NB>(a, blocks, s);
        if (nit <= 4) return gemv_launch_reg<T, 4, NB>(a, blocks, s);
        return gemv_launch_reg<T, 8, NB>(a, blocks, s);
    }
    const size_t lds = (size_t)NB * cdiv(a.K / V, 512) * 512 * 16;
    SS_REQUIRE(lds <= 152 * 1024, "gemv: K=%d x batch %d too large for LDS staging", a.K, NB);
    int threads, blocks;
    if (NB == 1) {
        // batch 1 (the reference configuration): several 256-thread blocks per CU, ~2.5k waves (measured optimum)
        threads = lds <= 36 * 1024 ? 256 : lds <= 72 * 1024 ? 512 : 1024;
        blocks = blocks_for(threads / 64);
    } else {
        // every block stages all NB rows of x, so blocks are fat and their count is a whole number per CU
        // (no CU ends up with one block more than its neighbour): 2 x 512 threads per CU, or 1 x 1024 when the
        // staged activations exceed half the LDS
        const int per_cu = lds <= 76 * 1024 ? 2 : 1;
        threads = per_cu == 2 ? 512 : 1024;
        blocks = 256 * per_cu;
        const int64_t max_b = (waves * 2 + threads / 64 - 1) / (threads / 64);   // keep >= ~2 row groups per wave
        if (blocks > max_b) blocks = (int)(max_b < 1 ? 1 : max_b);
    }
    if (lds > 64 * 1024)
        hipFuncSetAttribute((const void*)gemv_ldsx_kernel<T, 2, NB>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds);
    hipLaunchKernelGGL((gemv_ldsx_kernel<T, 2, NB>), dim3((unsigned)blocks), dim3((unsigned)threads), lds, s, a);
    SS_LAUNCH_CHECK("gemv_ldsx");
    return SS_OK;
}

template <typename T>
int gemv_launch(const GemvArgs& a0, hipStream_t s) {
    constexpr int V = Tr<T>::kVec;
    GemvArgs a = a0;
    const int64_t N = a.N, K = a.K;
    const int epi = a.epi;
    SS_REQUIRE(K % V == 0 && K > 0 && N > 0, "gemv: K=%lld must be a positive multiple of %d", (long long)K, V);
    SS_REQUIRE(!(epi & SS_EPI_SILU_MUL) || !(epi & (SS_EPI_BIAS | SS_EPI_RESIDUAL | SS_EPI_GELU)),
               "gemv: SILU_MUL cannot be combined with other epilogues");
    SS_REQUIRE(!(epi & SS_EPI_GELU), "gemv: GELU epilogue not supported");
    SS_REQUIRE(a.nb >= 1 && a.nb <= 4, "gemv: batch %d unsupported (1..4)", a.nb);
    a.use_nt = tuning_get("gemv_nt", 1);
    const int nit = cdiv(K, 64 * V);
    const int64_t groups = (epi & SS_EPI_SILU_MUL) ? N : (N + 1) / 2;
    // waves: enough to fill the chip, but several row-groups per wave so the x prologue amortises
    // auto: ~2.5k waves (8-12 per CU) measured best on MI355X for every LLaMA-7B projection
    int gpw = tuning_get("gemv_groups_per_wave", 0);
    if (gpw <= 0) { gpw = (int)((groups + 1280) / 2560); if (gpw < 1) gpw = 1; }
    const int64_t waves = (groups + gpw - 1) / gpw;
    // a batch whose LDS-staged activations would not fit one CU's LDS is swept in two halves
    const bool in_regs = nit <= 8 && a.nb * nit <= tuning_get("gemv_x_reg_packs", 16) && !tuning_get("gemv_force_lds", 0);
    if (!in_regs && a.nb > 1 && (size_t)a.nb * cdiv(K / V, 512) * 512 * 16 > 152 * 1024) {
        const int h1 = a.nb / 2;
        GemvArgs lo = a, hi = a;
        lo.nb = h1;
        hi.nb = a.nb - h1;
        hi.x = (const T*)a.x + (int64_t)h1 * a.x_ld;
        hi.y = (T*)a.y + (int64_t)h1 * a.y_ld;
        if (a.residual) hi.residual = (const T*)a.residual + (int64_t)h1 * a.res_ld;
        if (a.done_flag) hi.done_flag = a.done_flag + (int64_t)h1 * a.done_stride;
        const int rc = gemv_launch<T>(lo, s);
        return rc ? rc : gemv_launch<T>(hi, s);
    }
    switch (a.nb) {
        case 1: return gemv_launch_nb<T, 1>(a, nit, waves, s);
        case 2: return gemv_launch_nb<T, 2>(a, nit, waves, s);
        case 3: return gemv_launch_nb<T, 3>(a, nit, waves, s);
        default: return gemv_launch_nb<T, 4>(a, nit, waves, s);
    }
}

int gemv_batched_dev(const void* W, const void* x, void* y, int64_t N, int64_t K, const void* norm_w, float eps,
                     const void* bias, const void* residual, int epi, const int32_t* done_flag, int done_stride,
                     int nb, int64_t x_ld, int64_t y_ld, int64_t res_ld, int dtype, hipStream_t s) {
    GemvArgs a;
    a.W = W; a.x = x; a.y = y; a.norm_w = norm_w; a.bias = bias; a.residual = residual; a.done_flag = done_flag;
    a.N = (int)N; a.K = (int)K; a.epi = epi; a.eps = eps; a.use_nt = 1;
    a.nb = nb; a.done_stride = done_stride; a.x_ld = x_ld; a.y_ld = y_ld; a.res_ld = res_ld;
    return SS_DISPATCH(dtype, gemv_launch, a, s);
}

int gemv_dev(const void* W, const void* x, void* y, int64_t N, int64_t K, const void* norm_w, float eps,
             const void* bias, const void* residual, int epi, const int32_t* done_flag, int dtype, hipStream_t s) {
    return gemv_batched_dev(W, x, y, N, K, norm_w, eps, bias, residual, epi, done_flag, 0, 1, K, N, N, dtype, s);
}

}  // namespace ss

extern "C" int ss_gemv(const void* W, const void* x, void* y, int64_t N, int64_t K, const void* norm_w, float eps,
                       const void* bias, const void* residual, int epilogue, int dtype, void* stream) {
    return ss::gemv_dev(W, x, y, N, K, norm_w, eps, bias, residual, epilogue, nullptr, dtype, (hipStream_t)stream);
}

extern "C" int ss_gemv_batched(const void* W, const void* x, void* y, int64_t N, int64_t K, int64_t nb,
                               const void* norm_w, float eps, const void* bias, const void* residual, int epilogue,
                               int dtype, void* stream) {
    return ss::gemv_batched_dev(W, x, y, N, K, norm_w, eps, bias, residual, epilogue, nullptr, 0, (int)nb, K, N, N,
                                dtype, (hipStream_t)stream);
}
