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

#include <utility>

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

// ---- MFMA form: 3..16 sequences per sweep (16-bit types, K <= 4096 or K = 11008) ----------------------------------------------------
// Past 4 sequences the dot-product kernels above run out of VALU / LDS rate (8 sequences = 8 v_dot2 per 4 weight bytes
// plus 8 LDS reads per pack), while one v_mfma_f32_16x16x32 consumes 1 KB of weights for up to SIXTEEN sequences in 8
// passes: the matrix core turns the batched decode projection back into a pure weight stream.
//   * operands: A = 16 weight rows x 32 k (lane l: row l & 15, k-chunk l >> 4: ONE 16-byte load per lane per step, 64
//     contiguous bytes per row per instruction, the two halves of a 128-B line in consecutive instructions),
//     B = the activations (lane l: sequence l & 15, same k-chunk), D[row][sequence] in 4 VGPRs;
//   * a workgroup is 8 waves = 8 K slices of the same 16-row tile (K = 4096: 16 steps of 32 per wave); every wave keeps
//     ITS slice of the activations in 64 VGPRs for the whole launch (loaded — and RMS-normalised, statistic summed
//     across the 8 waves in a fixed order — once), so the stream loop issues nothing but weight loads;
//   * persistent: one workgroup per CU walks row tiles blockIdx, +grid, ...; the weights of the NEXT half-tile are
//     requested before the MFMAs of the current one (two register buffers of 8 steps), also across tile boundaries
//     and across the barrier: 8 - 16 KB per wave stay in flight the whole launch;
//   * per tile the 8 partial accumulators meet in LDS (double-buffered: one barrier per tile) and wave 0 applies the
//     epilogue of gemv_store (bias, rounding, residual | SiLU(gate) * up with the [gate; up] rows as two MFMA chains).
template <int N, typename F, int... I>
__device__ __forceinline__ void gv_static_for_impl(F&& f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void gv_static_for(F&& f) { gv_static_for_impl<N>(f, std::make_integer_sequence<int, N>{}); }

template <typename T> __device__ __forceinline__ f32x4_t gv_mfma(const uint4& a, const uint4& b, f32x4_t c);
template <> __device__ __forceinline__ f32x4_t gv_mfma<bf16_t>(const uint4& a, const uint4& b, f32x4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}
template <> __device__ __forceinline__ f32x4_t gv_mfma<f16_t>(const uint4& a, const uint4& b, f32x4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
}

constexpr int kGvWaves = 8, kGvSteps = 16, kGvChunk = 8;

// the 8 waves' partial accumulators of one row tile -> LDS -> (barrier) -> wave 0 folds them in wave order and applies the
// epilogue of gemv_store.  `part` = [8 waves][M][256] floats of this tile's parity.
// The epilogue's residual / bias values of wave 0 are requested at the TOP of the tile (gv_epi_prefetch), ahead of the weight
// loads: a load issued in the epilogue itself would have to be waited for with every younger weight prefetch in front of
// it (vmcnt is in-order), which stalls wave 0 — and through the next barrier the whole workgroup — once per tile.
struct GvEpi { uint2 res, bias; bool vec; };
template <typename T>
__device__ __forceinline__ GvEpi gv_epi_prefetch(const GemvArgs& a, int tile, int lane, bool vec_ok) {
    GvEpi e;
    e.vec = vec_ok;
    e.res = make_uint2(0, 0);
    e.bias = make_uint2(0, 0);
    if (vec_ok) {       // kernel-uniform: N, y_ld, res_ld multiples of 4 -> 8-byte accesses of 4 consecutive rows
        const int i = lane & 15, q = lane >> 4;
        int row0 = tile * 16 + q * 4;
        if (row0 > a.N - 4) row0 = a.N - 4;
        const int seq = i < a.nb ? i : 0;
        // (absent operands read the output row instead: in range, never used)
        const T* rp = (a.epi & SS_EPI_RESIDUAL) ? (const T*)a.residual + (int64_t)seq * a.res_ld : (const T*)a.y + (int64_t)seq * a.y_ld;
        const T* bp = (a.epi & SS_EPI_BIAS) ? (const T*)a.bias : (const T*)a.y + (int64_t)seq * a.y_ld;
        e.res = *reinterpret_cast<const uint2*>(rp + row0);
        e.bias = *reinterpret_cast<const uint2*>(bp + row0);
    }
    return e;
}
template <typename T> __device__ __forceinline__ float gv_elem(const uint2& v, int r) {
    const uint32_t w = r < 2 ? v.x : v.y;
    T t;
    t.v = (uint16_t)((r & 1) ? (w >> 16) : (w & 0xffffu));
    return Tr<T>::ld(&t);
}

template <typename T, bool SILU>
__device__ __forceinline__ void gv_fold_store(const GemvArgs& a, float* part, const f32x4_t (&acc)[SILU ? 2 : 1], int tile,
                                              int wave, int lane, const GvEpi& epi) {
    constexpr int M = SILU ? 2 : 1;
    const int i = lane & 15, q = lane >> 4;
    const int N = a.N;
#pragma unroll
    for (int m = 0; m < M; ++m) *reinterpret_cast<f32x4_t*>(part + (wave * M + m) * 256 + lane * 4) = acc[m];
    __syncthreads();
    if (wave != 0) return;
    f32x4_t v[M];
#pragma unroll
    for (int m = 0; m < M; ++m) {
        v[m] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int w = 0; w < kGvWaves; ++w) v[m] += *reinterpret_cast<const f32x4_t*>(part + (w * M + m) * 256 + lane * 4);
    }
    const int row0 = tile * 16 + q * 4;                       // D[row 4q + r][sequence i]
    if (i >= a.nb || row0 >= N) return;
    T* y = (T*)a.y + (int64_t)i * a.y_ld;
    float o[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = row0 + r < N ? row0 + r : N - 1;
        if constexpr (SILU) {
            const float gt = Tr<T>::rnd(v[0][r]), up = Tr<T>::rnd(v[M - 1][r]);
            o[r] = Tr<T>::rnd(silu_g(gt)) * up;
        } else {
            float t = v[0][r];
            if (epi.vec) {
                if (a.epi & SS_EPI_BIAS) t += gv_elem<T>(epi.bias, r);
                t = Tr<T>::rnd(t);
                if (a.epi & SS_EPI_RESIDUAL) t += gv_elem<T>(epi.res, r);
            } else {
                if (a.epi & SS_EPI_BIAS) t += Tr<T>::ld((const T*)a.bias + row);
                t = Tr<T>::rnd(t);
                if (a.epi & SS_EPI_RESIDUAL) t += Tr<T>::ld((const T*)a.residual + (int64_t)i * a.res_ld + row);
            }
            o[r] = t;
        }
    }
    if (row0 + 3 < N && ((a.y_ld | row0) & 3) == 0) {
        T tmp[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) Tr<T>::st(tmp + r, o[r]);
        *reinterpret_cast<uint2*>(y + row0) = *reinterpret_cast<const uint2*>(tmp);
    } else {
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (row0 + r < N) Tr<T>::st(y + row0 + r, o[r]);
    }
}

// any K <= 4096 (multiple of 8), any N: predicated loads — correct for every shape, with the prefetch serialised by the
// predicates' branches (the tiny test models and odd widths; the LLaMA-7B depths take gemv_mfma_exact_kernel below)
template <typename T, bool SILU>
__global__ __launch_bounds__(512) void gemv_mfma_kernel(const GemvArgs a, const int spw, const int ntiles) {
    constexpr int NS = kGvSteps, CH = kGvChunk, M = SILU ? 2 : 1;
    __shared__ __attribute__((aligned(16))) float part[2][kGvWaves][M][256];
    __shared__ float red[kGvWaves][16];
    if (gemv_all_done(a)) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 15, q = lane >> 4;
    const int K = a.K, N = a.N, nb = a.nb;
    const T* __restrict__ W = (const T*)a.W;
    const int k0 = wave * spw * 32 + q * 8;                         // this lane's k at the wave's step 0
    int nvalid = k0 < K ? (K - k0 + 31) / 32 : 0;                   // steps of this lane that lie inside K
    if (nvalid > spw) nvalid = spw;

    // ---- the wave's slice of the activations, in B-operand order (+ fused RMSNorm) -----------------------------------
    // (lanes of sequences >= nb read sequence 0: their columns of D are never stored)
    uint4 xf[NS];
    {
        const T* xr = (const T*)a.x + (int64_t)(i < nb ? i : 0) * a.x_ld + k0;
        gv_static_for<NS>([&](auto s_) {
            constexpr int s = decltype(s_)::value;
            xf[s] = s < nvalid ? ld16(xr + s * 32) : make_uint4(0, 0, 0, 0);
        });
        if (a.norm_w) {
            float ssq = 0.f;
            gv_static_for<NS>([&](auto s_) {
                constexpr int s = decltype(s_)::value;
                float f[8];
                unpack<T>(xf[s], f);
#pragma unroll
                for (int j = 0; j < 8; ++j) ssq = fmaf(f[j], f[j], ssq);
            });
            ssq += __shfl_xor(ssq, 16, 64);
            ssq += __shfl_xor(ssq, 32, 64);
            if (lane < 16) red[wave][lane] = ssq;
            __syncthreads();
            float tot = 0.f;
#pragma unroll
            for (int w = 0; w < kGvWaves; ++w) tot += red[w][i];
            const float rstd = 1.0f / sqrtf(tot / (float)K + a.eps);
            const T* gw = (const T*)a.norm_w + k0;
            gv_static_for<NS>([&](auto s_) {
                constexpr int s = decltype(s_)::value;
                if (s < nvalid) {
                    float f[8], g[8];
                    unpack<T>(xf[s], f);
                    unpack<T>(ld16(gw + s * 32), g);
#pragma unroll
                    for (int j = 0; j < 8; ++j) f[j] = g[j] * Tr<T>::rnd(f[j] * rstd);
                    xf[s] = pack<T>(f);
                }
            });
        }
    }
    uint4 wa[M][NS];
    int it = 0;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x, ++it) {
        int row = tile * 16 + i;
        if (row >= N) row = N - 1;
        f32x4_t acc[M];
#pragma unroll
        for (int m = 0; m < M; ++m) {
            acc[m] = f32x4_t{0.f, 0.f, 0.f, 0.f};
            const T* p = W + ((int64_t)row + (int64_t)m * N) * K + k0;
            gv_static_for<NS>([&](auto s_) {
                constexpr int s = decltype(s_)::value;
                wa[m][s] = s < nvalid ? ld16(p + s * 32) : make_uint4(0, 0, 0, 0);
            });
        }
        gv_static_for<NS>([&](auto s_) {
            constexpr int s = decltype(s_)::value;
#pragma unroll
            for (int m = 0; m < M; ++m) acc[m] = gv_mfma<T>(wa[m][s], xf[s], acc[m]);
        });
        GvEpi epi;
        epi.vec = false;
        gv_fold_store<T, SILU>(a, &part[it & 1][0][0][0], acc, tile, wave, lane, epi);
    }
}

// ---- the predicate-free stream loop for the two LLaMA-7B depths ---------------------------------------------------------------
// K == 8 waves x SPW steps x 32 exactly (SPW = 16: K = 4096, hidden; SPW = 43: K = 11008, the MLP width): every load of
// every lane is in range, so the stream loop carries no predicate and no branch (a predicated load is a branch to hipcc, and
// every branch join waits for vmcnt(0): the prefetch would be serialised).  The K slice of a wave is walked in chunks of 8
// steps through two register buffers; the loads of chunk c + 1 (or of the next tile's chunk 0) are issued before the MFMAs
// of chunk c, across the barrier too.
// SPW = 43 (PACK): 43 activation fragments per lane would take 172 VGPRs; only sequences 0..7 exist (nb <= 8), so lanes
// 8..15 of every 16-lane row carry steps 22..43 of sequences 0..7 and a row_shl:8 DPP move hands them to lanes 0..7 when
// their step comes up: 88 VGPRs.  (Columns 8..15 of D then hold garbage that is never stored.)  No RMSNorm in this form.
__device__ __forceinline__ void gv_fence(uint4& v) { asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w)); }

__device__ __forceinline__ uint4 gv_row_shift8(const uint4& v) {
    constexpr int ctrl = 0x108;                    // row_shl:8: lane l of a 16-lane row reads lane l + 8
    uint4 r;
    r.x = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v.x, ctrl, 0xf, 0xf, true);
    r.y = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v.y, ctrl, 0xf, 0xf, true);
    r.z = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v.z, ctrl, 0xf, 0xf, true);
    r.w = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v.w, ctrl, 0xf, 0xf, true);
    return r;
}

template <typename T, bool SILU, int SPW, bool NT>
__global__ __launch_bounds__(512) void gemv_mfma_exact_kernel(const GemvArgs a, const int ntiles) {
    constexpr int CH = kGvChunk, M = SILU ? 2 : 1;
    constexpr int NCH = (SPW + CH - 1) / CH;
    constexpr bool PACK = SPW > kGvSteps;
    constexpr int HALF = PACK ? (SPW + 1) / 2 : SPW;        // fragments a lane holds
    static_assert(NCH % 2 == 0, "an even number of chunks keeps the buffer parity across tiles");
    __shared__ __attribute__((aligned(16))) float part[2][kGvWaves][M][256];
    __shared__ float red[kGvWaves][16];
    if (gemv_all_done(a)) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 15, q = lane >> 4;
    const int K = a.K, N = a.N, nb = a.nb;
    const T* __restrict__ W = (const T*)a.W;
    const int k0 = wave * SPW * 32 + q * 8;                         // this lane's k at the wave's step 0
    int tile = blockIdx.x, it = 0;
    if (tile >= ntiles) return;

    uint4 wa[2][M][CH];
    auto load_chunk = [&](auto c_, int tl) {                        // chunk c of row tile tl -> buffer c & 1
        constexpr int c = decltype(c_)::value;
        int row = tl * 16 + i;
        if (row >= N) row = N - 1;
#pragma unroll
        for (int m = 0; m < M; ++m) {
            const T* p = W + ((int64_t)row + (int64_t)m * N) * K + k0 + c * CH * 32;
            gv_static_for<CH>([&](auto e_) {
                constexpr int e = decltype(e_)::value;
                if constexpr (c * CH + e < SPW) wa[c & 1][m][e] = NT ? ld_nt16(p + e * 32) : ld16(p + e * 32);
            });
        }
    };

    // ---- the wave's slice of the activations, in B-operand order (+ fused RMSNorm) -----------------------------------
    uint4 xf[HALF];
    if constexpr (!PACK) {
        // (lanes of sequences >= nb read sequence 0: their columns of D are never stored)
        const T* xr = (const T*)a.x + (int64_t)(i < nb ? i : 0) * a.x_ld + k0;
        gv_static_for<HALF>([&](auto s_) { constexpr int s = decltype(s_)::value; xf[s] = ld16(xr + s * 32); });
        load_chunk(std::integral_constant<int, 0>{}, tile);         // the first weights travel while the statistic is formed
        __builtin_amdgcn_sched_barrier(0);
        if (a.norm_w) {
            float ssq = 0.f;
            gv_static_for<HALF>([&](auto s_) {
                constexpr int s = decltype(s_)::value;
                float f[8];
                unpack<T>(xf[s], f);
#pragma unroll
                for (int j = 0; j < 8; ++j) ssq = fmaf(f[j], f[j], ssq);
            });
            // (the fragments are unpacked AGAIN below: without this fence hipcc keeps the 128 unpacked floats alive across
            // the barrier and spills)
            gv_static_for<HALF>([&](auto s_) {
                constexpr int s = decltype(s_)::value;
                gv_fence(xf[s]);
            });
            ssq += __shfl_xor(ssq, 16, 64);
            ssq += __shfl_xor(ssq, 32, 64);
            if (lane < 16) red[wave][lane] = ssq;
            __syncthreads();
            float tot = 0.f;
#pragma unroll
            for (int w = 0; w < kGvWaves; ++w) tot += red[w][i];
            const float rstd = 1.0f / sqrtf(tot / (float)K + a.eps);
            const T* gw = (const T*)a.norm_w + k0;
            // the gain in two batches of 8 fragments (32 VGPRs in flight, not 64: with the activations and the first weights
            // resident the single batch spills)
            gv_static_for<2>([&](auto h_) {
                constexpr int h = decltype(h_)::value;
                uint4 gf[HALF / 2];
                gv_static_for<HALF / 2>([&](auto s_) { constexpr int s = decltype(s_)::value; gf[s] = ld16(gw + (h * (HALF / 2) + s) * 32); });
                gv_static_for<HALF / 2>([&](auto s_) {
                    constexpr int s = decltype(s_)::value, t = h * (HALF / 2) + s;
                    float f[8], g[8];
                    unpack<T>(xf[t], f);
                    unpack<T>(gf[s], g);
#pragma unroll
                    for (int j = 0; j < 8; ++j) f[j] = g[j] * Tr<T>::rnd(f[j] * rstd);
                    xf[t] = pack<T>(f);
                });
                __builtin_amdgcn_sched_barrier(0);
            });
        }
    } else {
        const int seq = i & 7, upper = i >> 3;
        const T* xr = (const T*)a.x + (int64_t)(seq < nb ? seq : 0) * a.x_ld + k0 + upper * HALF * 32;
        gv_static_for<HALF>([&](auto s_) {
            constexpr int s = decltype(s_)::value;
            if constexpr (s + HALF < SPW) xf[s] = ld16(xr + s * 32);
            else {                                                  // step s + HALF does not exist: the upper lanes hold zeros
                const uint4 v = ld16(xr + (upper ? s - 1 : s) * 32);
                xf[s] = upper ? make_uint4(0, 0, 0, 0) : v;
            }
        });
        load_chunk(std::integral_constant<int, 0>{}, tile);
    }
    // 8-byte epilogue accesses need 8-byte aligned BASES too (a bias / residual view at an odd element offset)
    const bool vec_ok = ((a.y_ld | a.res_ld | N) & 3) == 0 && N >= 4 && (a.epi & (SS_EPI_RESIDUAL | SS_EPI_BIAS)) &&
                        ((((size_t)a.y | (size_t)a.residual | (size_t)a.bias) & 7) == 0);

    // one row tile: per chunk — next chunk requested (the next TILE's first chunk behind the last one), this chunk consumed;
    // then partial sums to LDS, barrier, wave 0 folds the 8 slices and stores
    auto tile_body = [&](int tl, int itn, auto has_next_) {
        constexpr bool has_next = decltype(has_next_)::value;
        f32x4_t acc[M];
#pragma unroll
        for (int m = 0; m < M; ++m) acc[m] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        GvEpi epi;
        epi.vec = false;
        if constexpr (!SILU) epi = gv_epi_prefetch<T>(a, tl, lane, vec_ok);
        gv_static_for<NCH>([&](auto c_) {
            constexpr int c = decltype(c_)::value;
            if constexpr (c + 1 < NCH) load_chunk(std::integral_constant<int, c + 1>{}, tl);
            else if constexpr (has_next) load_chunk(std::integral_constant<int, 0>{}, tl + (int)gridDim.x);
            __builtin_amdgcn_sched_barrier(0);      // the scheduler otherwise sinks these loads below the MFMAs that free
            gv_static_for<CH>([&](auto e_) {        // their registers: one buffer in flight instead of two
                constexpr int e = decltype(e_)::value, s = c * CH + e;
                if constexpr (s < SPW) {
                    uint4 b;
                    if constexpr (!PACK) b = xf[s];
                    else if constexpr (s < HALF) b = xf[s];
                    else b = gv_row_shift8(xf[s - HALF]);
#pragma unroll
                    for (int m = 0; m < M; ++m) acc[m] = gv_mfma<T>(wa[c & 1][m][e], b, acc[m]);
                }
            });
            __builtin_amdgcn_sched_barrier(0);
        });
        gv_fold_store<T, SILU>(a, &part[itn & 1][0][0][0], acc, tl, wave, lane, epi);
    };
    for (; tile + (int)gridDim.x < ntiles; tile += gridDim.x, ++it) tile_body(tile, it, std::true_type{});
    tile_body(tile, it, std::false_type{});
}

// ---- split-bf16 MFMA form for fp32 tensors (the gate mode, tuning knob gemm_f32_split; round 5) ---------------------------------
// The exact fp32 decode has no MFMA form: 8 lock-step sequences run as TWO 4-sequence sweeps of the dot-product kernels, i.e. the
// 26.4 GB of fp32 LLaMA weights are streamed twice per token (measured: ~700 of the 1093 ms the gate-mode MLLM half of a round of
// 8 stories takes).  Here one sweep serves 3..8 sequences through the matrix core with the arithmetic of the split GEMM
// (ss_gemm.hip SPLIT): w = hi + lo, x = hi + lo in bf16, y += Whi.xhi + Whi.xlo + Wlo.xhi with fp32 accumulation, ~4.5e-6 per
// product against the exact chain.
//   * the activations of the launch (<= 8 sequences x <= 4096 k) are split ONCE per workgroup into two bf16 planes in LDS
//     (wave w stages sequence w, RMSNorm fused: statistic over the whole row in a fixed lane order), sequence stride
//     2 k + 32 bytes so that the 16 lanes of a ds_read_b128 group (8 sequences x 2 k-chunks) fall on 16 different 16-byte slots;
//   * a lane streams 32 bytes of fp32 weights per step (row l & 15, k-chunk l >> 4), splits them in registers (8 conversions
//     + 8 subtractions, far below the HBM time of those bytes) and feeds three MFMAs; two register buffers of 4 steps keep
//     8-16 KB per wave in flight across tiles; out-of-range steps read a valid address and are zeroed by a select (no branch:
//     a predicated load would serialise the prefetch);
//   * K > 4096 (the 11008-deep down projection: N = 4096 = one row tile per workgroup) walks K slices of 4096 INSIDE the tile: the
//     planes are re-staged per slice (two barriers), the weight stream and the accumulators carry on — one launch, no read-modify-
//     write of y (the first version ran three launches: 70 us against ~48 for the same bytes);
//   * persistent workgroups of 8 waves = 8 K ranges of a 16-row tile; partial sums meet in LDS, wave 0 applies the epilogue.
constexpr int kGsSlice = 4096;
template <bool SILU>
__global__ __launch_bounds__(512) void gemv_split_f32_kernel(const GemvArgs a, const int ntiles) {
    constexpr int CH = 4, NCH = kGvSteps / CH, M = SILU ? 2 : 1;
    extern __shared__ __attribute__((aligned(16))) char gs_smem[];
    if (gemv_all_done(a)) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 15, q = lane >> 4;
    const int K = a.K, N = a.N, nb = a.nb;
    const int kmax = K < kGsSlice ? K : kGsSlice;
    const int SS = kmax * 2 + 32;                            // bytes between sequences of a plane
    const int nslices = (K + kGsSlice - 1) / kGsSlice;       // K > 4096 (the 11008-deep down projection): slices walked INSIDE the tile
    char* xh = gs_smem;
    char* xl = gs_smem + 8 * SS;
    float* part = reinterpret_cast<float*>(gs_smem + 16 * SS);   // [8 waves][M][256]
    const float* __restrict__ W = (const float*)a.W;

    // ---- the activations of K slice [kbase, kbase + kslice) -> the two bf16 planes: wave w <-> sequence w -----------------
    float rstd = 1.f;
    const bool norm = a.norm_w != nullptr;                   // (single slice only: the launcher guarantees K <= 4096 with a norm)
    {
        const bool live = wave < nb;
        const float* xr = (const float*)a.x + (int64_t)(live ? wave : 0) * a.x_ld;
        if (norm) {
            float ssq = 0.f;
            for (int k = lane * 4; k < K; k += 256) {
                const float4 v = *reinterpret_cast<const float4*>(xr + k);
                ssq = fmaf(v.x, v.x, ssq); ssq = fmaf(v.y, v.y, ssq); ssq = fmaf(v.z, v.z, ssq); ssq = fmaf(v.w, v.w, ssq);
            }
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) ssq += __shfl_xor(ssq, off, 64);
            rstd = 1.0f / sqrtf(ssq / (float)K + a.eps);
        }
    }
    auto stage = [&](int kbase, int kslice) {
        const int seq = wave;
        const bool live = seq < nb;
        const float* xr = (const float*)a.x + (int64_t)(live ? seq : 0) * a.x_ld;
        for (int k = lane * 4; k < kslice; k += 256) {
            float4 v = live ? *reinterpret_cast<const float4*>(xr + kbase + k) : make_float4(0.f, 0.f, 0.f, 0.f);
            if (norm) {
                const float4 gw = *reinterpret_cast<const float4*>((const float*)a.norm_w + kbase + k);
                v.x = gw.x * (v.x * rstd); v.y = gw.y * (v.y * rstd); v.z = gw.z * (v.z * rstd); v.w = gw.w * (v.w * rstd);
            }
            const uint32_t h01 = f32x2_to_bf16x2_bits(v.x, v.y), h23 = f32x2_to_bf16x2_bits(v.z, v.w);
            const float r0 = v.x - __uint_as_float(h01 << 16), r1 = v.y - __uint_as_float(h01 & 0xffff0000u);
            const float r2 = v.z - __uint_as_float(h23 << 16), r3 = v.w - __uint_as_float(h23 & 0xffff0000u);
            *reinterpret_cast<uint2*>(xh + seq * SS + k * 2) = make_uint2(h01, h23);
            *reinterpret_cast<uint2*>(xl + seq * SS + k * 2) = make_uint2(f32x2_to_bf16x2_bits(r0, r1), f32x2_to_bf16x2_bits(r2, r3));
        }
    };

    // B-fragment base of this lane: sequence i & 7 (columns 8..15 of D duplicate 0..7 and are never stored)
    const char* bh0 = xh + (i & 7) * SS + q * 16;
    const char* bl0 = xl + (i & 7) * SS + q * 16;
    int tile = blockIdx.x;
    if (tile >= ntiles) return;

    // geometry of a slice: steps of 32 k dealt to the 8 waves
    auto slice_len = [&](int sl) { const int r = K - sl * kGsSlice; return r < kGsSlice ? r : kGsSlice; };
    auto slice_spw = [&](int kslice) { return ((kslice + 31) / 32 + kGvWaves - 1) / kGvWaves; };   // <= 16

    uint4 wb[2][M][CH][2];
    auto load_chunk = [&](auto c_, int tl, int sl) {         // chunk c (4 steps) of slice sl of row tile tl -> buffer c & 1
        constexpr int c = decltype(c_)::value;
        const int kbase = sl * kGsSlice, kslice = slice_len(sl), spw = slice_spw(kslice), s_w0 = wave * spw;
        int row = tl * 16 + i;
        if (row >= N) row = N - 1;
#pragma unroll
        for (int m = 0; m < M; ++m) {
            const float* p = W + ((int64_t)row + (int64_t)m * N) * K + kbase;
            gv_static_for<CH>([&](auto e_) {
                constexpr int e = decltype(e_)::value;
                const int st = s_w0 + c * CH + e;
                const bool ok = (c * CH + e) < spw && st * 32 + q * 8 < kslice;
                const float* pp = p + (ok ? st * 32 + q * 8 : 0);   // an invalid step re-reads the slice's first 32 bytes (in range) ...
                wb[c & 1][m][e][0] = ld16(pp);
                wb[c & 1][m][e][1] = ld16(pp + 4);
            });
        }
    };
    auto split8 = [&](const uint4& w0, const uint4& w1, bool ok, uint4& hi, uint4& lo) {
        const float x0 = __uint_as_float(w0.x), x1 = __uint_as_float(w0.y), x2 = __uint_as_float(w0.z), x3 = __uint_as_float(w0.w);
        const float x4 = __uint_as_float(w1.x), x5 = __uint_as_float(w1.y), x6 = __uint_as_float(w1.z), x7 = __uint_as_float(w1.w);
        const uint32_t h0 = f32x2_to_bf16x2_bits(x0, x1), h1 = f32x2_to_bf16x2_bits(x2, x3);
        const uint32_t h2 = f32x2_to_bf16x2_bits(x4, x5), h3 = f32x2_to_bf16x2_bits(x6, x7);
        const uint32_t l0 = f32x2_to_bf16x2_bits(x0 - __uint_as_float(h0 << 16), x1 - __uint_as_float(h0 & 0xffff0000u));
        const uint32_t l1 = f32x2_to_bf16x2_bits(x2 - __uint_as_float(h1 << 16), x3 - __uint_as_float(h1 & 0xffff0000u));
        const uint32_t l2 = f32x2_to_bf16x2_bits(x4 - __uint_as_float(h2 << 16), x5 - __uint_as_float(h2 & 0xffff0000u));
        const uint32_t l3 = f32x2_to_bf16x2_bits(x6 - __uint_as_float(h3 << 16), x7 - __uint_as_float(h3 & 0xffff0000u));
        hi = ok ? make_uint4(h0, h1, h2, h3) : make_uint4(0, 0, 0, 0);     // ... and contributes zeros
        lo = ok ? make_uint4(l0, l1, l2, l3) : make_uint4(0, 0, 0, 0);
    };

    load_chunk(std::integral_constant<int, 0>{}, tile, 0);  // the first weights travel while the activations are staged
    stage(0, slice_len(0));
    __syncthreads();
    for (;;) {
        const int tnext = tile + (int)gridDim.x;
        const bool has_next = tnext < ntiles;                 // workgroup-uniform
        f32x4_t acc[M];
#pragma unroll
        for (int m = 0; m < M; ++m) acc[m] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        for (int sl = 0; sl < nslices; ++sl) {
            const int kslice = slice_len(sl), spw = slice_spw(kslice), s_w0 = wave * spw;
            const bool last_slice = sl + 1 == nslices;
            gv_static_for<NCH>([&](auto c_) {
                constexpr int c = decltype(c_)::value;
                if constexpr (c + 1 < NCH) load_chunk(std::integral_constant<int, c + 1>{}, tile, sl);
                else {                                         // NCH is even: the next slice's / tile's chunk 0 -> buffer 0
                    if (!last_slice) load_chunk(std::integral_constant<int, 0>{}, tile, sl + 1);
                    else if (has_next) load_chunk(std::integral_constant<int, 0>{}, tnext, 0);
                }
                gv_static_for<CH>([&](auto e_) {
                    constexpr int e = decltype(e_)::value;
                    const int st = s_w0 + c * CH + e;
                    const bool ok = (c * CH + e) < spw && st * 32 + q * 8 < kslice;
                    const int sb = ok ? st * 64 : -q * 16;    // 32 k x 2 bytes per step (invalid: the plane's first 16 bytes)
                    uint4 bh = *reinterpret_cast<const uint4*>(bh0 + sb);
                    uint4 bl = *reinterpret_cast<const uint4*>(bl0 + sb);
                    // (zero operands on BOTH sides: beyond the slice the planes hold whatever the LDS held, and 0 x NaN is NaN)
                    if (!ok) { bh = make_uint4(0, 0, 0, 0); bl = make_uint4(0, 0, 0, 0); }
#pragma unroll
                    for (int m = 0; m < M; ++m) {
                        uint4 ah, al;
                        split8(wb[c & 1][m][e][0], wb[c & 1][m][e][1], ok, ah, al);
                        acc[m] = gv_mfma<bf16_t>(al, bh, acc[m]);
                        acc[m] = gv_mfma<bf16_t>(ah, bl, acc[m]);
                        acc[m] = gv_mfma<bf16_t>(ah, bh, acc[m]);
                    }
                });
            });
            if (nslices > 1) {                                 // the planes of the next slice (of this tile, or slice 0 of the next tile)
                const int nsl = last_slice ? 0 : sl + 1;
                if (!last_slice || has_next) {
                    __syncthreads();                           // every wave has read its fragments of the current slice
                    stage(nsl * kGsSlice, slice_len(nsl));
                    __syncthreads();
                }
            }
        }
        // ---- fold the 8 K ranges and store ----------------------------------------------------------------------------
        __syncthreads();                                      // the previous tile's fold has finished reading `part`
#pragma unroll
        for (int m = 0; m < M; ++m) *reinterpret_cast<f32x4_t*>(part + (wave * M + m) * 256 + lane * 4) = acc[m];
        __syncthreads();
        if (wave == 0) {
            f32x4_t v[M];
#pragma unroll
            for (int m = 0; m < M; ++m) {
                v[m] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int w = 0; w < kGvWaves; ++w) v[m] += *reinterpret_cast<const f32x4_t*>(part + (w * M + m) * 256 + lane * 4);
            }
            const int row0 = tile * 16 + q * 4;               // D[row 4q + r][sequence i]
            if (i < nb && row0 < N) {
                float* y = (float*)a.y + (int64_t)i * a.y_ld;
                float o[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = row0 + r < N ? row0 + r : N - 1;
                    if constexpr (SILU) {
                        o[r] = silu_g(v[0][r]) * v[M - 1][r];
                    } else {
                        float t = v[0][r];
                        if (a.epi & SS_EPI_BIAS) t += ((const float*)a.bias)[row];
                        if (a.epi & SS_EPI_RESIDUAL) t += ((const float*)a.residual)[(int64_t)i * a.res_ld + row];
                        o[r] = t;
                    }
                }
                if (row0 + 3 < N && ((a.y_ld | row0) & 3) == 0 && (((size_t)a.y) & 15) == 0) {
                    *reinterpret_cast<float4*>(y + row0) = make_float4(o[0], o[1], o[2], o[3]);
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (row0 + r < N) y[row0 + r] = o[r];
                }
            }
        }
        if (!has_next) break;
        tile = tnext;
    }
}

static int gemv_launch_split_f32(const GemvArgs& a, hipStream_t s) {
    const bool silu = (a.epi & SS_EPI_SILU_MUL) != 0;
    const int ntiles = cdiv(a.N, 16);
    const int cus = tuning_get("gemv_mfma_blocks", 256);
    const int rounds = cdiv(ntiles, cus);
    const int blocks = cdiv(ntiles, rounds);
    const int kmax = a.K < kGsSlice ? a.K : kGsSlice;
    const size_t lds = (size_t)16 * (kmax * 2 + 32) + (size_t)kGvWaves * (silu ? 2 : 1) * 256 * sizeof(float);
    auto kern = silu ? gemv_split_f32_kernel<true> : gemv_split_f32_kernel<false>;
    // (the largest request of either instantiation: a full 4096-wide slice, SiLU pair)
    const size_t lds_max = (size_t)16 * (kGsSlice * 2 + 32) + (size_t)kGvWaves * 2 * 256 * sizeof(float);
    if (silu) SS_DYN_LDS(gemv_split_f32_kernel<true>, lds_max); else SS_DYN_LDS(gemv_split_f32_kernel<false>, lds_max);
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(512), lds, s, a, ntiles);
    SS_LAUNCH_CHECK("gemv_split_f32");
    return SS_OK;
}

template <typename T>
static bool gemv_mfma_eligible(const GemvArgs& a) {
    if (a.K <= kGvWaves * kGvSteps * 32) return true;
    // the packed 43-step form: the LLaMA-7B MLP width only, <= 8 sequences, no RMSNorm prologue
    return a.K == kGvWaves * 43 * 32 && a.nb <= 8 && !a.norm_w && !tuning_get("gemv_mfma_generic", 0) &&
           tuning_get("gemv_mfma_long", 1);
}

template <typename T>
static int gemv_launch_mfma(const GemvArgs& a, hipStream_t s) {
    const bool silu = (a.epi & SS_EPI_SILU_MUL) != 0;
    const int ksteps = cdiv(a.K, 32);
    const int spw = cdiv(ksteps, kGvWaves);
    const int ntiles = cdiv(a.N, 16);
    // whole rounds: ceil(tiles / 256) tiles per workgroup on as few workgroups as that takes (688 tiles -> 230 x 3, not 256
    // workgroups of which 80 run a third round alone)
    const int cus = tuning_get("gemv_mfma_blocks", 256);
    const int rounds = cdiv(ntiles, cus);
    const int blocks = cdiv(ntiles, rounds);
    const bool generic = tuning_get("gemv_mfma_generic", 0) != 0;
    // plain loads, not non-temporal ones: a fragment load touches HALF a 128-byte line per row and the next instruction the
    // other half; measured on the five LLaMA-7B shapes at 8 sequences, nt is 3 - 13 % slower (lm_head 52.0 vs 46.2 us)
    const bool nt = tuning_get("gemv_mfma_nt", 0) != 0;
    const dim3 g((unsigned)blocks), b(512);
#define SS_GV_EXACT(SPW)                                                                                                  \
    do {                                                                                                                  \
        if (silu) { if (nt) hipLaunchKernelGGL((gemv_mfma_exact_kernel<T, true, SPW, true>), g, b, 0, s, a, ntiles);      \
                    else hipLaunchKernelGGL((gemv_mfma_exact_kernel<T, true, SPW, false>), g, b, 0, s, a, ntiles); }       \
        else { if (nt) hipLaunchKernelGGL((gemv_mfma_exact_kernel<T, false, SPW, true>), g, b, 0, s, a, ntiles);          \
               else hipLaunchKernelGGL((gemv_mfma_exact_kernel<T, false, SPW, false>), g, b, 0, s, a, ntiles); }           \
    } while (0)
    if (!generic && a.K == kGvWaves * kGvSteps * 32) SS_GV_EXACT(16);
    else if (!generic && a.K == kGvWaves * 43 * 32) {
        SS_GV_EXACT(43);
    }
    else if (silu) hipLaunchKernelGGL((gemv_mfma_kernel<T, true>), g, b, 0, s, a, spw, ntiles);
    else hipLaunchKernelGGL((gemv_mfma_kernel<T, false>), g, b, 0, s, a, spw, ntiles);
#undef SS_GV_EXACT
    SS_LAUNCH_CHECK("gemv_mfma");
    return SS_OK;
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
    if (lds > 64 * 1024)     // (the request depends on K: set per launch, return code checked)
        SS_HIP(hipFuncSetAttribute((const void*)gemv_ldsx_kernel<T, 2, NB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
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
    SS_REQUIRE(a.nb >= 1 && a.nb <= 16, "gemv: batch %d unsupported (1..16)", a.nb);
    a.use_nt = tuning_get("gemv_nt", 1);
    // 3+ sequences (knob): the MFMA form when the shape allows it (16-bit, 8 waves x 16 steps of 32 cover K, or the packed
    // 11008-deep form).  Measured on the five LLaMA-7B projections (tools/gemv_bench.py, rotating weights, us per launch):
    // dot-product kernels at 1 | 4 sequences 20.7 9.9 32.3 19.3 46.1 | 24.4 11.0 39.0 25.7 47.8; MFMA form at 8 sequences
    // 24.2 9.9 37.3 21.9 46.2 — one sequence stays with the dot-product kernel, 3 and more go through the matrix core.
    if constexpr (V == 8) {
        if (a.nb >= tuning_get("gemv_mfma_min_nb", 3) && gemv_mfma_eligible<T>(a)) return gemv_launch_mfma<T>(a, s);
    } else {
        // fp32 tensors in the gate mode: 3..8 sequences through the split-bf16 MFMA form (one sweep of the weights instead of two
        // 4-sequence sweeps of the exact dot-product kernels); the RMSNorm prologue needs the whole row in one K slice; SiLU pairs too
        if (a.nb >= tuning_get("gemv_mfma_min_nb", 3) && a.nb <= 8 && tuning_get("gemm_f32_split", 0) && K % 8 == 0 &&
            ((!a.norm_w && !(epi & SS_EPI_SILU_MUL)) || K <= kGsSlice) && (((size_t)a.x | (size_t)a.W | (size_t)a.norm_w) & 15) == 0 &&
            (a.x_ld & 3) == 0)
            return gemv_launch_split_f32(a, s);
        // the gate mode was asked for and this launch cannot take the split form (RMSNorm / SiLU row wider than one 4096 slice,
        // or 16-byte misalignment): it runs as two EXACT sweeps — correct, slower, and a different arithmetic than the rest of
        // the run.  Counted (tuning key gemv_split_refused, readable through ss_get_tuning) and reported once.
        if (a.nb >= tuning_get("gemv_mfma_min_nb", 3) && a.nb <= 8 && tuning_get("gemm_f32_split", 0)) {
            static bool told = false;
            tuning_set("gemv_split_refused", tuning_get("gemv_split_refused", 0) + 1);
            if (!told) {
                told = true;
                fprintf(stderr, "[ss] gate mode: GEMV [N=%d, K=%d, %d slots]%s%s runs as exact 4-slot sweeps (split form needs norm / SiLU rows <= %d "
                                "wide and 16-byte alignment)\n", N, K, a.nb, a.norm_w ? " +RMSNorm" : "", (epi & SS_EPI_SILU_MUL) ? " +SiLU" : "", kGsSlice);
            }
        }
    }
    if (a.nb > 4) {      // no MFMA form for this shape / type: two sweeps of half the sequences each
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
