// HBM-bound kernels of the SDXL de-tokenizer half (UNet ResBlocks / Transformer2D, VAE decoder,
// Euler + classifier-free guidance) — the reference reaches these through diffusers
// (src/models_ipa/adapter_modules.py:455-466).
// Activations are NHWC ([B, H*W, C] row-major) so every convolution is an implicit GEMM over
// channels (ss_gemm.hip) and the transformer blocks need no permutes.
#include "ss_common.h"

namespace ss {

__device__ __forceinline__ float silu_d(float g) { return g / (1.0f + expf(-g)); }
__device__ __forceinline__ float gelu_d(float v) { return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f)); }

// =====================================================================================
// GroupNorm over NHWC: statistics per (batch, group) of H*W x (C/G) elements, DETERMINISTIC (round 4: no atomics).
//   pass 1: every block reduces its rows to per-channel (sum, sum of squares) in a fixed order (thread-sequential over its
//           rows, then a fixed-order fold over the row lanes through LDS), folds the channels of a group in channel order
//           in fp64 and writes its partial to part[b][block][g][2]
//   pass 1b: one block per image sums the block partials in block order (fixed-size slices, slices folded in order)
//           -> stats[b][g][2] (fp64: E[x^2] - mean^2 without cancellation)
//   pass 2: y = (x - mean) * rstd * gamma[c] + beta[c]  (+ SiLU), rounded once to T
// Two runs on the same input give the same bits whatever the dispatch order (rounds 2-3 used fp64 atomics: order-dependent in
// the last bit, which a 70-block UNet amplified to a 2e-2 difference between two runs when the sums were fp32).
// (torch GroupNorm computes in fp32 and rounds the result; SiLU then rounds again.)
// =====================================================================================
// fast sigmoid-linear unit on the hardware transcendentals: x * rcp(1 + exp2(-x * log2 e))
__device__ __forceinline__ float silu_fast(float g) {
    return g * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * g));
}

// Both passes give every thread ONE fixed 16-byte channel pack and a strided set of pixel rows of one image,
// so everything that depends on the channel (group index, gamma/beta, mean/rstd) is hoisted out of the row loop
// and the loop body is a 16-byte load + 2 (stats) or ~4 (apply) VALU ops per element: both passes run at the
// HBM rate instead of being bound by per-element integer divisions and LDS atomics.
//   thread t of a block: pack p = pc + t % cw, first row r0 + t / cw, row stride 256 / cw   (cw = packs per chunk)
template <typename T>
__global__ __launch_bounds__(256) void groupnorm_stats_kernel(const T* __restrict__ x, double* __restrict__ part,
                                                              int HW, int C, int G, int rows_per_block) {
    constexpr int V = Tr<T>::kVec;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* red = reinterpret_cast<float*>(smem_raw);   // [256][2 V]: per-thread channel sums of the current pack chunk
    float* chs = red + 256 * 2 * V;                     // [C][2]: this block's per-channel (sum, sum of squares)
    const int b = blockIdx.y;
    const int cg = C / G, ppr = C / V;
    const int r0 = blockIdx.x * rows_per_block;
    const int r1 = min(HW, r0 + rows_per_block);
    const T* xb = x + ((int64_t)b * HW) * C;
    for (int pc = 0; pc < ppr; pc += 256) {
        const int cw = min(256, ppr - pc);
        const int rip = 256 / cw;                      // rows in flight per block pass
        const int ry = threadIdx.x / cw, p = pc + threadIdx.x % cw;
        const bool active = ry < rip;
        float s1[V], s2[V];                            // per-channel partial sums of this thread's pack
#pragma unroll
        for (int j = 0; j < V; ++j) { s1[j] = 0.f; s2[j] = 0.f; }
        if (active) {
            const T* xp = xb + (int64_t)p * V;
            int r = r0 + ry;
            for (; r + 3 * rip < r1; r += 4 * rip) {       // four independent loads in flight (round 5: 2.0 -> HBM rate; same add order)
                uint4 u[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) u[q] = ld16(xp + (int64_t)(r + q * rip) * C);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float f[V];
                    unpack<T>(u[q], f);
#pragma unroll
                    for (int j = 0; j < V; ++j) { s1[j] += f[j]; s2[j] = fmaf(f[j], f[j], s2[j]); }
                }
            }
            for (; r + rip < r1; r += 2 * rip) {           // two independent loads in flight
                float f[V], h[V];
                const uint4 u0 = ld16(xp + (int64_t)r * C), u1 = ld16(xp + (int64_t)(r + rip) * C);
                unpack<T>(u0, f);
                unpack<T>(u1, h);
#pragma unroll
                for (int j = 0; j < V; ++j) {
                    s1[j] += f[j]; s2[j] = fmaf(f[j], f[j], s2[j]);
                    s1[j] += h[j]; s2[j] = fmaf(h[j], h[j], s2[j]);
                }
            }
            if (r < r1) {
                float f[V];
                unpack<T>(ld16(xp + (int64_t)r * C), f);
#pragma unroll
                for (int j = 0; j < V; ++j) { s1[j] += f[j]; s2[j] = fmaf(f[j], f[j], s2[j]); }
            }
#pragma unroll
            for (int j = 0; j < V; ++j) { red[threadIdx.x * 2 * V + 2 * j] = s1[j]; red[threadIdx.x * 2 * V + 2 * j + 1] = s2[j]; }
        }
        __syncthreads();
        if (active && ry == 0) {       // fold the row lanes of this pack in lane order
            for (int k = 1; k < rip; ++k) {
                const float* o = red + (k * cw + (int)threadIdx.x) * 2 * V;
#pragma unroll
                for (int j = 0; j < V; ++j) { s1[j] += o[2 * j]; s2[j] += o[2 * j + 1]; }
            }
#pragma unroll
            for (int j = 0; j < V; ++j) { chs[(p * V + j) * 2] = s1[j]; chs[(p * V + j) * 2 + 1] = s2[j]; }
        }
        __syncthreads();
    }
    double* out = part + (((int64_t)b * gridDim.x + blockIdx.x) * G) * 2;
    for (int g = threadIdx.x; g < G; g += 256) {       // channels of a group in channel order, fp64
        double a1 = 0.0, a2 = 0.0;
        for (int c = g * cg; c < (g + 1) * cg; ++c) { a1 += (double)chs[2 * c]; a2 += (double)chs[2 * c + 1]; }
        out[2 * g] = a1;
        out[2 * g + 1] = a2;
    }
}

// stats[b][g] = sum over the blocks of image b, in block order: thread (slice, g) sums a contiguous slice of blocks, the slices
// are folded in slice order
__global__ __launch_bounds__(256) void groupnorm_finalize_kernel(const double* __restrict__ part, double* __restrict__ stats,
                                                                 int nblk, int G) {
    __shared__ double sl[256 * 2];
    const int b = blockIdx.x;
    const int gw = G < 256 ? G : 256;                   // groups handled per sweep
    const int ns = 256 / gw;                            // slices
    for (int g0 = 0; g0 < G; g0 += gw) {
        const int g = g0 + (int)threadIdx.x % gw, sidx = (int)threadIdx.x / gw;
        double a1 = 0.0, a2 = 0.0;
        if (sidx < ns && g < G) {
            const int per = (nblk + ns - 1) / ns;
            const int k0 = sidx * per, k1 = min(nblk, k0 + per);
            const double2* pp = reinterpret_cast<const double2*>(part) + (int64_t)b * nblk * G + g;
            for (int kb = k0; kb < k1; kb += 8) {       // 8 independent loads in flight, then the adds in block order
                double2 t[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) t[k] = (kb + k < k1) ? pp[(int64_t)(kb + k) * G] : make_double2(0.0, 0.0);
#pragma unroll
                for (int k = 0; k < 8; ++k) { a1 += t[k].x; a2 += t[k].y; }
            }
        }
        sl[threadIdx.x * 2] = a1;
        sl[threadIdx.x * 2 + 1] = a2;
        __syncthreads();
        if (sidx == 0 && g < G) {
            for (int k = 1; k < ns; ++k) { a1 += sl[(k * gw + (int)threadIdx.x) * 2]; a2 += sl[(k * gw + (int)threadIdx.x) * 2 + 1]; }
            stats[((int64_t)b * G + g) * 2] = a1;
            stats[((int64_t)b * G + g) * 2 + 1] = a2;
        }
        __syncthreads();
    }
}

template <typename T>
__global__ __launch_bounds__(256) void groupnorm_apply_kernel(const T* __restrict__ x, const double* __restrict__ stats,
                                                              const T* __restrict__ gamma, const T* __restrict__ beta,
                                                              T* __restrict__ y, int HW, int C, int G, float eps,
                                                              int silu, int rows_per_block) {
    constexpr int V = Tr<T>::kVec;
    const int b = blockIdx.y;
    const int cg = C / G, ppr = C / V;
    const int r0 = blockIdx.x * rows_per_block;
    const int r1 = min(HW, r0 + rows_per_block);
    const float inv_n = 1.0f / ((float)HW * (float)cg);
    const T* xb = x + ((int64_t)b * HW) * C;
    T* yb = y + ((int64_t)b * HW) * C;
    for (int pc = 0; pc < ppr; pc += 256) {
        const int cw = min(256, ppr - pc);
        const int rip = 256 / cw;
        const int ry = threadIdx.x / cw, p = pc + threadIdx.x % cw;
        if (ry >= rip) continue;
        // y = (x - mean) * rstd * gamma + beta  ==  x * sc + sh  with per-channel constants hoisted out of the loop
        float sc[V], sh[V], ga[V], be[V];
        unpack<T>(ld16(gamma + p * V), ga);
        unpack<T>(ld16(beta + p * V), be);
#pragma unroll
        for (int j = 0; j < V; ++j) {
            const int g = (p * V + j) / cg;
            const double md = stats[((int64_t)b * G + g) * 2] * (double)inv_n;      // E[x^2] - mean^2 in fp64: no cancellation
            const float mean = (float)md;
            const float var = (float)fmax(stats[((int64_t)b * G + g) * 2 + 1] * (double)inv_n - md * md, 0.0);
            const float rstd = 1.0f / sqrtf(var + eps);
            sc[j] = rstd * ga[j];
            sh[j] = be[j] - mean * sc[j];
        }
        const T* xp = xb + (int64_t)p * V;
        T* yp = yb + (int64_t)p * V;
        auto one = [&](const uint4& u, int r) {
            float f[V];
            unpack<T>(u, f);
#pragma unroll
            for (int j = 0; j < V; ++j) {
                float v = fmaf(f[j], sc[j], sh[j]);
                if (silu) v = silu_fast(Tr<T>::rnd(v));
                f[j] = v;
            }
            st16(yp + (int64_t)r * C, pack<T>(f));
        };
        int r = r0 + ry;
        for (; r + 3 * rip < r1; r += 4 * rip) {           // four rows in flight per thread
            uint4 u[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) u[q] = ld16(xp + (int64_t)(r + q * rip) * C);
#pragma unroll
            for (int q = 0; q < 4; ++q) one(u[q], r + q * rip);
        }
        for (; r + rip < r1; r += 2 * rip) {
            const uint4 u0 = ld16(xp + (int64_t)r * C), u1 = ld16(xp + (int64_t)(r + rip) * C);
            one(u0, r);
            one(u1, r + rip);
        }
        if (r < r1) one(ld16(xp + (int64_t)r * C), r);
    }
}

static int groupnorm_rows_per_block(int64_t B, int64_t HW, int64_t C, size_t elem) {
    // ~128 KiB of activations per block, at least enough blocks to fill the chip several times
    int rows_per_block = (int)((128 * 1024) / (C * elem));
    if (rows_per_block < 16) rows_per_block = 16;
    while (rows_per_block > 16 && B * cdiv(HW, rows_per_block) < 2048) rows_per_block /= 2;
    if (rows_per_block > HW) rows_per_block = (int)HW;
    return rows_per_block;
}

// workspace = [B][G][2] final sums + [B][blocks][G][2] block partials, doubles
size_t groupnorm_workspace_bytes(int64_t B, int64_t HW, int64_t C, int64_t G, int dtype) {
    if (B * HW == 0) return 0;
    const int rpb = groupnorm_rows_per_block(B, HW, C, dtype_size(dtype));
    return (size_t)B * G * 2 * sizeof(double) * (1 + (size_t)cdiv(HW, rpb));
}

template <typename T>
int groupnorm_launch(const void* x, const void* gamma, const void* beta, void* y, double* ws, int64_t B, int64_t HW,
                     int64_t C, int64_t G, float eps, int silu, hipStream_t s) {
    constexpr int V = Tr<T>::kVec;
    SS_REQUIRE(C % G == 0 && C % V == 0, "groupnorm: C=%lld G=%lld unsupported", (long long)C, (long long)G);
    if (B * HW == 0) return SS_OK;
    const int rows_per_block = groupnorm_rows_per_block(B, HW, C, sizeof(T));
    dim3 grid((unsigned)cdiv(HW, rows_per_block), (unsigned)B);
    double* stats = ws;
    double* part = ws + (size_t)B * G * 2;
    const size_t lds = (size_t)256 * 2 * V * sizeof(float) + (size_t)C * 2 * sizeof(float);
    SS_REQUIRE(lds <= 64 * 1024, "groupnorm: C=%lld too wide", (long long)C);
    hipLaunchKernelGGL(groupnorm_stats_kernel<T>, grid, dim3(256), lds, s, (const T*)x, part, (int)HW, (int)C, (int)G,
                       rows_per_block);
    SS_LAUNCH_CHECK("groupnorm_stats");
    hipLaunchKernelGGL(groupnorm_finalize_kernel, dim3((unsigned)B), dim3(256), 0, s, (const double*)part, stats, (int)grid.x, (int)G);
    SS_LAUNCH_CHECK("groupnorm_finalize");
    hipLaunchKernelGGL(groupnorm_apply_kernel<T>, grid, dim3(256), 0, s, (const T*)x, stats, (const T*)gamma,
                       (const T*)beta, (T*)y, (int)HW, (int)C, (int)G, eps, silu, rows_per_block);
    SS_LAUNCH_CHECK("groupnorm_apply");
    return SS_OK;
}

// =====================================================================================
// small element-wise / layout kernels
// =====================================================================================
// GEGLU: out[r, i] = val[r, i] * gelu(gate[r, i]); in [rows, 2D] = [val | gate]  (diffusers GEGLU)
template <typename T>
__global__ __launch_bounds__(256) void geglu_kernel(const T* __restrict__ in, T* __restrict__ out, int64_t rows, int D) {
    constexpr int V = Tr<T>::kVec;
    const int ppr = D / V;
    const int64_t total = rows * ppr;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / ppr;
        const int p = (int)(i % ppr);
        float a[V], g[V];
        unpack<T>(ld16(in + r * 2 * D + (int64_t)p * V), a);
        unpack<T>(ld16(in + r * 2 * D + D + (int64_t)p * V), g);
#pragma unroll
        for (int j = 0; j < V; ++j) a[j] = a[j] * Tr<T>::rnd(gelu_d(g[j]));
        st16(out + r * D + (int64_t)p * V, pack<T>(a));
    }
}

// in-place row softmax of s[rows, cols] * scale (fp32 math, rounded to T)
template <typename T>
__global__ __launch_bounds__(256) void softmax_rows_kernel(T* __restrict__ s, int cols, float scale) {
    constexpr int V = Tr<T>::kVec;
    __shared__ float red[16];
    T* row = s + (int64_t)blockIdx.x * cols;
    const int npack = cols / V;
    float mx = -1e30f;
    for (int p = threadIdx.x; p < npack; p += 256) {
        float f[V];
        unpack<T>(ld16(row + (int64_t)p * V), f);
#pragma unroll
        for (int j = 0; j < V; ++j) mx = fmaxf(mx, f[j] * scale);
    }
    mx = block_max(mx, red);
    float sum = 0.f;
    for (int p = threadIdx.x; p < npack; p += 256) {
        float f[V];
        unpack<T>(ld16(row + (int64_t)p * V), f);
#pragma unroll
        for (int j = 0; j < V; ++j) sum += expf(f[j] * scale - mx);
    }
    sum = block_sum(sum, red);
    const float inv = 1.0f / sum;
    for (int p = threadIdx.x; p < npack; p += 256) {
        float f[V];
        unpack<T>(ld16(row + (int64_t)p * V), f);
#pragma unroll
        for (int j = 0; j < V; ++j) f[j] = expf(f[j] * scale - mx) * inv;
        st16(row + (int64_t)p * V, pack<T>(f));
    }
}

// out[c, r] = in[r, c]  (32x32 LDS tile transpose)
template <typename T>
__global__ __launch_bounds__(256) void transpose_kernel(const T* __restrict__ in, T* __restrict__ out, int R, int Cc) {
    __shared__ T tile[32][33];
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    for (int j = ty; j < 32; j += 8)
        if (r0 + j < R && c0 + tx < Cc) tile[j][tx] = in[(int64_t)(r0 + j) * Cc + c0 + tx];
    __syncthreads();
    for (int j = ty; j < 32; j += 8)
        if (c0 + j < Cc && r0 + tx < R) out[(int64_t)(c0 + j) * R + r0 + tx] = tile[tx][j];
}

// out[r, :C1] = a[r, :], out[r, C1:] = b[r, :]   (channel concat of two NHWC tensors: torch.cat(dim=1) in NCHW)
template <typename T>
__global__ __launch_bounds__(256) void concat_channels_kernel(const T* __restrict__ a, const T* __restrict__ b,
                                                              T* __restrict__ out, int64_t rows, int C1, int C2) {
    constexpr int V = Tr<T>::kVec;
    const int ppr = (C1 + C2) / V, p1 = C1 / V;
    const int64_t total = rows * ppr;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / ppr;
        const int p = (int)(i % ppr);
        const uint4 v = p < p1 ? ld16(a + r * C1 + (int64_t)p * V) : ld16(b + r * C2 + (int64_t)(p - p1) * V);
        st16(out + r * (C1 + C2) + (int64_t)p * V, v);
    }
}

// NCHW [B, C, HW] <-> NHWC [B, HW, Cpad] for the tiny latent tensors (C = 4 -> Cpad = 8, zero padded)
template <typename T>
__global__ void nchw_to_nhwc_kernel(const T* __restrict__ in, T* __restrict__ out, int64_t B, int C, int HW, int Cpad) {
    const int64_t total = B * HW * Cpad;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % Cpad);
        const int64_t r = i / Cpad;
        const int64_t b = r / HW, px = r % HW;
        T v;
        if (c < C) v = in[(b * C + c) * HW + px]; else Tr<T>::st(&v, 0.f);
        out[i] = v;
    }
}
template <typename T>
__global__ void nhwc_to_nchw_kernel(const T* __restrict__ in, T* __restrict__ out, int64_t B, int C, int HW, int Cpad) {
    const int64_t total = B * C * HW;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t px = i % HW;
        const int64_t bc = i / HW;
        const int c = (int)(bc % C);
        const int64_t b = bc / C;
        out[i] = in[(b * HW + px) * Cpad + c];
    }
}

// EulerDiscreteScheduler.scale_model_input + CFG duplication: xin[0] = xin[1] = x / sqrt(sigma^2 + 1)
// classifier-free guidance + Euler step:  e = eu + g (ec - eu);  x += e * (sigma_next - sigma)
// (fp32 latents like diffusers keeps them when the pipeline dtype is fp32; T otherwise)
template <typename T>
__global__ void euler_scale_dup_kernel(const T* __restrict__ x, T* __restrict__ xin, int64_t n, float inv) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float v = Tr<T>::ld(x + i) * inv;
        Tr<T>::st(xin + i, v);
        Tr<T>::st(xin + n + i, v);
    }
}
template <typename T>
__global__ void euler_cfg_step_kernel(T* __restrict__ x, const T* __restrict__ eps, int64_t n, float guidance, float dsigma) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float eu = Tr<T>::ld(eps + i), ec = Tr<T>::ld(eps + n + i);
        const float e = Tr<T>::rnd(eu + Tr<T>::rnd(guidance * Tr<T>::rnd(ec - eu)));
        Tr<T>::st(x + i, Tr<T>::ld(x + i) + Tr<T>::rnd(e * dsigma));
    }
}

// VAE output NHWC [B, HW, Cpad] (first 3 channels) -> uint8 HWC: round(clamp(x/2 + 0.5, 0, 1) * 255)
template <typename T>
__global__ void image_to_u8_kernel(const T* __restrict__ in, uint8_t* __restrict__ out, int64_t pixels, int Cpad) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < pixels * 3; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % 3);
        const int64_t px = i / 3;
        float v = Tr<T>::rnd(Tr<T>::ld(in + px * Cpad + c) * 0.5f + 0.5f);
        v = fminf(fmaxf(v, 0.f), 1.f);
        out[i] = (uint8_t)rintf(v * 255.0f);
    }
}

// y = silu(x) | gelu(x), rounded to T  (time-embedding activations: F.silu(emb))
template <typename T>
__global__ void unary_kernel(const T* __restrict__ x, T* __restrict__ y, int64_t n, int op) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float v = Tr<T>::ld(x + i);
        Tr<T>::st(y + i, op == 0 ? silu_d(v) : gelu_d(v));
    }
}

static inline unsigned egrid(int64_t total) {
    int64_t g = (total + 255) / 256;
    if (g > 256 * 16) g = 256 * 16;
    if (g < 1) g = 1;
    return (unsigned)g;
}

template <typename T>
int geglu_launch(const void* in, void* out, int64_t rows, int64_t D, hipStream_t s) {
    SS_REQUIRE(D % Tr<T>::kVec == 0, "geglu: D %% %d != 0", Tr<T>::kVec);
    if (rows == 0) return SS_OK;
    hipLaunchKernelGGL(geglu_kernel<T>, dim3(egrid(rows * D / Tr<T>::kVec)), dim3(256), 0, s, (const T*)in, (T*)out, rows, (int)D);
    SS_LAUNCH_CHECK("geglu");
    return SS_OK;
}
template <typename T>
int unary_launch(const void* x, void* y, int64_t n, int op, hipStream_t s) {
    if (n == 0) return SS_OK;
    hipLaunchKernelGGL(unary_kernel<T>, dim3(egrid(n)), dim3(256), 0, s, (const T*)x, (T*)y, n, op);
    SS_LAUNCH_CHECK("unary");
    return SS_OK;
}
template <typename T>
int softmax_rows_launch(void* sc, int64_t rows, int64_t cols, float scale, hipStream_t s) {
    SS_REQUIRE(cols % Tr<T>::kVec == 0, "softmax_rows: cols %% %d != 0", Tr<T>::kVec);
    if (rows == 0) return SS_OK;
    hipLaunchKernelGGL(softmax_rows_kernel<T>, dim3((unsigned)rows), dim3(256), 0, s, (T*)sc, (int)cols, scale);
    SS_LAUNCH_CHECK("softmax_rows");
    return SS_OK;
}
template <typename T>
int transpose_launch(const void* in, void* out, int64_t R, int64_t Cc, hipStream_t s) {
    if (R * Cc == 0) return SS_OK;
    hipLaunchKernelGGL(transpose_kernel<T>, dim3((unsigned)cdiv(Cc, 32), (unsigned)cdiv(R, 32)), dim3(256), 0, s,
                       (const T*)in, (T*)out, (int)R, (int)Cc);
    SS_LAUNCH_CHECK("transpose");
    return SS_OK;
}
template <typename T>
int concat_channels_launch(const void* a, const void* b, void* out, int64_t rows, int64_t C1, int64_t C2, hipStream_t s) {
    SS_REQUIRE(C1 % Tr<T>::kVec == 0 && C2 % Tr<T>::kVec == 0, "concat_channels: channels %% %d != 0", Tr<T>::kVec);
    if (rows == 0) return SS_OK;
    hipLaunchKernelGGL(concat_channels_kernel<T>, dim3(egrid(rows * (C1 + C2) / Tr<T>::kVec)), dim3(256), 0, s,
                       (const T*)a, (const T*)b, (T*)out, rows, (int)C1, (int)C2);
    SS_LAUNCH_CHECK("concat_channels");
    return SS_OK;
}
template <typename T>
int layout_launch(const void* in, void* out, int64_t B, int64_t C, int64_t HW, int64_t Cpad, int to_nhwc, hipStream_t s) {
    if (B * HW == 0) return SS_OK;
    if (to_nhwc)
        hipLaunchKernelGGL(nchw_to_nhwc_kernel<T>, dim3(egrid(B * HW * Cpad)), dim3(256), 0, s, (const T*)in, (T*)out, B,
                           (int)C, (int)HW, (int)Cpad);
    else
        hipLaunchKernelGGL(nhwc_to_nchw_kernel<T>, dim3(egrid(B * C * HW)), dim3(256), 0, s, (const T*)in, (T*)out, B,
                           (int)C, (int)HW, (int)Cpad);
    SS_LAUNCH_CHECK("layout");
    return SS_OK;
}
template <typename T>
int euler_scale_dup_launch(const void* x, void* xin, int64_t n, float sigma, hipStream_t s) {
    hipLaunchKernelGGL(euler_scale_dup_kernel<T>, dim3(egrid(n)), dim3(256), 0, s, (const T*)x, (T*)xin, n,
                       1.0f / sqrtf(sigma * sigma + 1.0f));
    SS_LAUNCH_CHECK("euler_scale_dup");
    return SS_OK;
}
template <typename T>
int euler_cfg_step_launch(void* x, const void* eps, int64_t n, float guidance, float sigma, float sigma_next, hipStream_t s) {
    hipLaunchKernelGGL(euler_cfg_step_kernel<T>, dim3(egrid(n)), dim3(256), 0, s, (T*)x, (const T*)eps, n, guidance,
                       sigma_next - sigma);
    SS_LAUNCH_CHECK("euler_cfg_step");
    return SS_OK;
}
template <typename T>
int image_to_u8_launch(const void* in, void* out, int64_t pixels, int64_t Cpad, hipStream_t s) {
    hipLaunchKernelGGL(image_to_u8_kernel<T>, dim3(egrid(pixels * 3)), dim3(256), 0, s, (const T*)in, (uint8_t*)out, pixels,
                       (int)Cpad);
    SS_LAUNCH_CHECK("image_to_u8");
    return SS_OK;
}

}  // namespace ss

using namespace ss;

extern "C" {

size_t ss_groupnorm_workspace_bytes(int64_t batch, int64_t hw, int64_t channels, int64_t groups, int dtype) {
    return ss::groupnorm_workspace_bytes(batch, hw, channels, groups, dtype);
}

int ss_groupnorm(const void* x, const void* gamma, const void* beta, void* y, void* stats_ws, int64_t batch, int64_t hw,
                 int64_t channels, int64_t groups, float eps, int fuse_silu, int dtype, void* stream) {
    return SS_DISPATCH(dtype, groupnorm_launch, x, gamma, beta, y, (double*)stats_ws, batch, hw, channels, groups, eps,
                       fuse_silu, (hipStream_t)stream);
}
int ss_geglu(const void* in, void* out, int64_t rows, int64_t d, int dtype, void* stream) {
    return SS_DISPATCH(dtype, geglu_launch, in, out, rows, d, (hipStream_t)stream);
}
int ss_unary(const void* x, void* y, int64_t n, int op, int dtype, void* stream) {
    return SS_DISPATCH(dtype, unary_launch, x, y, n, op, (hipStream_t)stream);
}
int ss_softmax_rows(void* scores, int64_t rows, int64_t cols, float scale, int dtype, void* stream) {
    return SS_DISPATCH(dtype, softmax_rows_launch, scores, rows, cols, scale, (hipStream_t)stream);
}
int ss_transpose(const void* in, void* out, int64_t rows, int64_t cols, int dtype, void* stream) {
    return SS_DISPATCH(dtype, transpose_launch, in, out, rows, cols, (hipStream_t)stream);
}
int ss_concat_channels(const void* a, const void* b, void* out, int64_t rows, int64_t c1, int64_t c2, int dtype,
                       void* stream) {
    return SS_DISPATCH(dtype, concat_channels_launch, a, b, out, rows, c1, c2, (hipStream_t)stream);
}
int ss_layout_nchw_nhwc(const void* in, void* out, int64_t batch, int64_t channels, int64_t hw, int64_t cpad,
                        int to_nhwc, int dtype, void* stream) {
    return SS_DISPATCH(dtype, layout_launch, in, out, batch, channels, hw, cpad, to_nhwc, (hipStream_t)stream);
}
int ss_euler_scale_dup(const void* x, void* xin, int64_t n, float sigma, int dtype, void* stream) {
    return SS_DISPATCH(dtype, euler_scale_dup_launch, x, xin, n, sigma, (hipStream_t)stream);
}
int ss_euler_cfg_step(void* x, const void* eps, int64_t n, float guidance, float sigma, float sigma_next, int dtype,
                      void* stream) {
    return SS_DISPATCH(dtype, euler_cfg_step_launch, x, eps, n, guidance, sigma, sigma_next, (hipStream_t)stream);
}
int ss_image_to_u8(const void* in, void* out_u8, int64_t pixels, int64_t cpad, int dtype, void* stream) {
    return SS_DISPATCH(dtype, image_to_u8_launch, in, out_u8, pixels, cpad, (hipStream_t)stream);
}

}  // extern "C"

// ---- debug probe: semantics of ds_read_b64_tr_b16 (used when developing the attention kernel) -----------
namespace ss {
typedef short v4s_t __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) v4s_t lds_v4s_t;
__global__ void tr_probe_kernel(const uint16_t* src, uint16_t* dst, const int* lane_off) {
    __shared__ __attribute__((aligned(16))) uint16_t sm[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) sm[i] = src[i];
    __syncthreads();
    const int off = lane_off[threadIdx.x];
    v4s_t r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s_t*)(sm + off));
    for (int j = 0; j < 4; ++j) dst[threadIdx.x * 4 + j] = (uint16_t)r[j];
}
}  // namespace ss
extern "C" int ss_debug_tr_probe(const void* src, void* dst, const void* lane_off, void* stream) {
    hipLaunchKernelGGL(ss::tr_probe_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (const uint16_t*)src,
                       (uint16_t*)dst, (const int*)lane_off);
    SS_LAUNCH_CHECK("tr_probe");
    return SS_OK;
}
