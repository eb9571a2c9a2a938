// C[M,N] = A[M,K] · W[N,K]^T (+bias)(+GELU)(+residual) on the CDNA4 matrix cores.
//
// Every nn.Linear on the hot path has this shape with BOTH operands K-contiguous
// (modeling_llama_xformer.py:228-230,297,191; qwen_visual.py:191,196,259; resampler.py),
// which is exactly the MFMA fragment shape: lane l of a wave supplies 8 consecutive k of
// row (l & 15) for k-group (l >> 4).
//
// Operand roles are swapped w.r.t. the math so that stores vectorise: the MFMA "A" operand
// is the WEIGHT tile (rows -> n) and the "B" operand the ACTIVATION tile (cols -> m), so a
// lane ends up with C[m = l&15][n = 4*(l>>4) .. +3]: four consecutive n per row = one 8-byte
// (bf16) / 16-byte (fp32) store, bias is a per-lane 4-vector and the residual one load.
//
//   bf16 / fp16 : v_mfma_f32_16x16x32_{bf16,f16}, BK = 64
//   fp32        : v_mfma_f32_16x16x4_f32 (exact fp32 FMA chain; the CPU-parity mode), BK = 32
//
// Tiles: <BM, BN, WM, WN> = block tile (activation rows x weight rows) and the wave grid.
// Global -> register -> LDS staging with the next tile's loads in flight during the MFMAs
// (guide T14 write-late form), padded LDS rows (+1 pack) to break the 128-byte stride.
#include <map>
#include <mutex>
#include <tuple>

#include "ss_common.h"

namespace ss {

typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_t;
template <typename T> struct Mma;
template <> struct Mma<bf16_t> {
    static constexpr int kK = 32;
    static __device__ __forceinline__ f32x4_t run(const uint4& a, const uint4& b, f32x4_t c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a),
                                                       __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
    }
    // in-place form with the accumulator pinned to its VGPRs (see gemm_sp_kernel<.., NH = 2>)
    static __device__ __forceinline__ void run_inplace(const uint4& a, const uint4& b, f32x4_t& c) {
        asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0"
                     : "+v"(c)
                     : "v"(__builtin_bit_cast(u32x4_t, a)), "v"(__builtin_bit_cast(u32x4_t, b)));
    }
};
template <> struct Mma<f16_t> {
    static constexpr int kK = 32;
    static __device__ __forceinline__ f32x4_t run(const uint4& a, const uint4& b, f32x4_t c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a),
                                                      __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
    }
    static __device__ __forceinline__ void run_inplace(const uint4& a, const uint4& b, f32x4_t& c) {
        asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0"
                     : "+v"(c)
                     : "v"(__builtin_bit_cast(u32x4_t, a)), "v"(__builtin_bit_cast(u32x4_t, b)));
    }
};

__device__ __forceinline__ float gelu_erf(float v) { return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f)); }
// erf-GELU for 16-bit outputs: Abramowitz-Stegun 7.1.26 (|erf error| < 1.5e-7, three orders below a bf16 ulp) on the
// hardware rcp / exp2 — about a third of the VALU work of libm's branchy erff.  The GEGLU epilogue of the UNet's ff1
// evaluates 42 M of these per launch, serially after the K loop.
__device__ __forceinline__ float gelu_erf16(float v) {
    const float x = fabsf(v) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, x, 1.0f));
    float poly = fmaf(1.061405429f, t, -1.453152027f);
    poly = fmaf(poly, t, 1.421413741f);
    poly = fmaf(poly, t, -0.284496736f);
    poly = fmaf(poly, t, 0.254829592f);
    const float e = __builtin_amdgcn_exp2f(-1.4426950408889634f * x * x);
    const float erf_abs = fmaf(-poly * t, e, 1.0f);          // erf(|v|/sqrt2)
    return 0.5f * v + 0.5f * fabsf(v) * erf_abs;              // 0.5 v (1 + sign(v) erf(|v|/sqrt2))
}
template <typename T> __device__ __forceinline__ float gelu_for(float v) {
    if constexpr (Tr<T>::kVec == 8) return gelu_erf16(v);
    else return gelu_erf(v);
}

struct GemmArgs {
    const void* A; const void* W; void* C; const void* bias; const void* residual;
    int M, N, K;
    int64_t lda, ldw, ldc, ldr;
    int epi;
    // per-(batch, n) additive vector (the ResBlock's projected time embedding): rowvec[m / rows_per_batch][n]
    const void* rowvec; int rows_per_batch; int64_t rowvec_ld;
    // implicit-GEMM 3x3 convolution over an NHWC tensor (A = [B, H, W, Cin]); K = 9 * Cin
    int conv_H, conv_W, conv_Cin, conv_stride, conv_up, conv_Ho, conv_Wo;
    int swz;   // XCD-aware tile order (0 = row-major block ids)
};

// blockIdx -> output tile.  MI355X deals workgroups to its 8 XCDs round-robin by linear workgroup id and every XCD
// has a private 4 MB L2, so with row-major tile ids the blocks that share an A row-tile (or a W column-tile) land on
// eight different L2s and nothing is reused below the Infinity Cache: a K=5120 GEMM then pulls > 5 TB/s through
// MALL/HBM and is memory-bound, not MFMA-bound.  Remap: (1) XCD k owns a CONTIGUOUS range of logical tile ids,
// (2) logical ids walk groups of GM m-tiles n-major, so the ~32-64 blocks resident on one XCD form a GM x (32/GM)
// patch of the output that shares GM A-tiles and a few W-tiles through that XCD's L2.
__device__ __forceinline__ void xcd_tile(int swz, int MT, int NT, int& mt, int& nt) {
    if (!swz) { mt = blockIdx.y; nt = blockIdx.x; return; }
    const int total = MT * NT;
    const int id = blockIdx.y * gridDim.x + blockIdx.x;
    const int xcd = id & 7, j = id >> 3;
    const int base = total >> 3, rem = total & 7;
    const int L = xcd * base + (xcd < rem ? xcd : rem) + j;
    const int GM = swz;
    const int per_group = GM * NT;
    const int gidx = L / per_group, r = L - gidx * per_group;
    const int m0 = gidx * GM;
    const int gm = (MT - m0) < GM ? (MT - m0) : GM;
    nt = r / gm;
    mt = m0 + r - nt * gm;
}

template <typename T, int BM, int BN, int WM, int WN, int KT, bool CONV>
__global__ __launch_bounds__(64 * WM * WN) void gemm_kernel(const GemmArgs g) {
    constexpr int V = Tr<T>::kVec;
    constexpr int NT = 64 * WM * WN;     // threads per block
    constexpr int PPR = 8 * KT;          // packs per tile row
    constexpr int BK = PPR * V;          // k extent of one LDS tile
    constexpr int LS = BK + V;           // padded LDS row stride (elements)
    constexpr int TM = BM / WM, TN = BN / WN;
    constexpr int FM = TM / 16, FN = TN / 16;
    constexpr int PA = (BM * PPR + NT - 1) / NT, PW = (BN * PPR + NT - 1) / NT;  // packs per thread per tile
    static_assert(TM % 16 == 0 && TN % 16 == 0, "wave tile must be MFMA-shaped");

    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    T* As = reinterpret_cast<T*>(smem_raw);
    T* Ws = As + BM * LS;

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid / WN, wn = wid % WN;
    const int l15 = lane & 15, grp = lane >> 4;
    const int m_blk = blockIdx.y * BM, n_blk = blockIdx.x * BN;
    const T* __restrict__ A = (const T*)g.A;
    const T* __restrict__ W = (const T*)g.W;
    const int K = g.K, M = g.M, N = g.N;
    const int ntiles = (K + BK - 1) / BK;

    f32x4_t acc[FN][FM];
#pragma unroll
    for (int i = 0; i < FN; ++i)
#pragma unroll
        for (int j = 0; j < FM; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    uint4 ra[PA], rw[PW];
    // conv mode: this thread's PA tile rows are fixed for the whole K loop -> decode (b, oy, ox) once
    int cb[PA], cy[PA], cx[PA];
    if constexpr (CONV) {
#pragma unroll
        for (int i = 0; i < PA; ++i) {
            const int m = m_blk + ((tid + i * NT) / PPR);
            const int hw = g.conv_Ho * g.conv_Wo;
            cb[i] = m / hw;
            const int rem = m - cb[i] * hw;
            cy[i] = rem / g.conv_Wo;
            cx[i] = rem - cy[i] * g.conv_Wo;
        }
    }
    auto load_tile = [&](int t) {
        const int k0 = t * BK;
#pragma unroll
        for (int i = 0; i < PA; ++i) {
            const int p = tid + i * NT;
            const int r = p / PPR, c = p % PPR;
            const int m = m_blk + r, k = k0 + c * V;
            if constexpr (CONV) {
                // k -> (tap, ci); tap -> (dy, dx); zero padding 1; optional nearest 2x upsample of the input
                const int tap = k / g.conv_Cin, ci = k - tap * g.conv_Cin;
                const int dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;
                int iy = cy[i] * g.conv_stride + dy, ix = cx[i] * g.conv_stride + dx;
                const int Hin = g.conv_up ? 2 * g.conv_H : g.conv_H, Win = g.conv_up ? 2 * g.conv_W : g.conv_W;
                const bool ok = p < BM * PPR && m < M && k < K && iy >= 0 && iy < Hin && ix >= 0 && ix < Win;
                if (g.conv_up) { iy >>= 1; ix >>= 1; }
                ra[i] = ok ? ld16(A + (((int64_t)cb[i] * g.conv_H + iy) * g.conv_W + ix) * g.conv_Cin + ci)
                           : make_uint4(0, 0, 0, 0);
            } else {
                ra[i] = (p < BM * PPR && m < M && k < K) ? ld16(A + (int64_t)m * g.lda + k) : make_uint4(0, 0, 0, 0);
            }
        }
#pragma unroll
        for (int i = 0; i < PW; ++i) {
            const int p = tid + i * NT;
            const int r = p / PPR, c = p % PPR;
            const int n = n_blk + r, k = k0 + c * V;
            rw[i] = (p < BN * PPR && n < N && k < K) ? ld16(W + (int64_t)n * g.ldw + k) : make_uint4(0, 0, 0, 0);
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int i = 0; i < PA; ++i) {
            const int p = tid + i * NT;
            if (p < BM * PPR) st16(As + (p / PPR) * LS + (p % PPR) * V, ra[i]);
        }
#pragma unroll
        for (int i = 0; i < PW; ++i) {
            const int p = tid + i * NT;
            if (p < BN * PPR) st16(Ws + (p / PPR) * LS + (p % PPR) * V, rw[i]);
        }
    };

    load_tile(0);
    for (int t = 0; t < ntiles; ++t) {
        store_tile();
        __syncthreads();
        if (t + 1 < ntiles) load_tile(t + 1);  // in flight while the MFMAs run
        if constexpr (V == 8) {
#pragma unroll
            for (int ks = 0; ks < BK / 32; ++ks) {
                uint4 fw[FN], fa[FM];
#pragma unroll
                for (int i = 0; i < FN; ++i) fw[i] = ld16(Ws + (wn * TN + i * 16 + l15) * LS + ks * 32 + grp * 8);
#pragma unroll
                for (int j = 0; j < FM; ++j) fa[j] = ld16(As + (wm * TM + j * 16 + l15) * LS + ks * 32 + grp * 8);
#pragma unroll
                for (int i = 0; i < FN; ++i)
#pragma unroll
                    for (int j = 0; j < FM; ++j) acc[i][j] = Mma<T>::run(fw[i], fa[j], acc[i][j]);
            }
        } else {
#pragma unroll
            for (int kk = 0; kk < BK / 4; ++kk) {
                float fw[FN], fa[FM];
#pragma unroll
                for (int i = 0; i < FN; ++i) fw[i] = ((const float*)Ws)[(wn * TN + i * 16 + l15) * LS + kk * 4 + grp];
#pragma unroll
                for (int j = 0; j < FM; ++j) fa[j] = ((const float*)As)[(wm * TM + j * 16 + l15) * LS + kk * 4 + grp];
#pragma unroll
                for (int i = 0; i < FN; ++i)
#pragma unroll
                    for (int j = 0; j < FM; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fw[i], fa[j], acc[i][j], 0, 0, 0);
            }
        }
        __syncthreads();
    }

    // ---- epilogue: lane holds C[m = .. + l15][n = .. + grp*4 + r] -------------------------------
    T* __restrict__ C = (T*)g.C;
    const T* bias = (const T*)g.bias;
    const T* res = (const T*)g.residual;
#pragma unroll
    for (int i = 0; i < FN; ++i) {
        const int n0 = n_blk + wn * TN + i * 16 + grp * 4;
        float bv[4] = {0.f, 0.f, 0.f, 0.f};
        if (g.epi & SS_EPI_BIAS) {
#pragma unroll
            for (int r = 0; r < 4; ++r) if (n0 + r < N) bv[r] = Tr<T>::ld(bias + n0 + r);
        }
#pragma unroll
        for (int j = 0; j < FM; ++j) {
            const int m = m_blk + wm * TM + j * 16 + l15;
            if (m >= M) continue;
            float v[4];
            float rv[4] = {0.f, 0.f, 0.f, 0.f};
            if (g.rowvec) {
                const T* rp = (const T*)g.rowvec + (int64_t)(m / g.rows_per_batch) * g.rowvec_ld;
#pragma unroll
                for (int r = 0; r < 4; ++r) if (n0 + r < N) rv[r] = Tr<T>::ld(rp + n0 + r);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float t = acc[i][j][r] + bv[r];
                if (g.epi & SS_EPI_GELU) t = gelu_erf(Tr<T>::rnd(t));  // Linear output is rounded, then GELU
                v[r] = Tr<T>::rnd(t);
                if (g.rowvec) v[r] = Tr<T>::rnd(v[r] + rv[r]);   // h = conv(x) + temb[:, :, None, None]
            }
            if (g.epi & SS_EPI_RESIDUAL) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (n0 + r < N) v[r] += Tr<T>::ld(res + (int64_t)m * g.ldr + n0 + r);
            }
            if (g.epi & SS_EPI_GEGLU_PAIR) {   // (value, gate) interleaved columns -> out[m][n/2] = value * gelu(gate)
#pragma unroll
                for (int r = 0; r < 4; r += 2)
                    if (n0 + r + 1 < N) Tr<T>::st(C + (int64_t)m * g.ldc + ((n0 + r) >> 1), v[r] * Tr<T>::rnd(gelu_erf(v[r + 1])));
                continue;
            }
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (n0 + r < N) Tr<T>::st(C + (int64_t)m * g.ldc + n0 + r, v[r]);
        }
    }
}

template <typename T, int BM, int BN, int WM, int WN, int KT = 1>
static int gemm_launch_cfg(const GemmArgs& g, hipStream_t s) {
    constexpr int V = Tr<T>::kVec;
    constexpr int LS = 8 * KT * V + V;
    const size_t lds = (size_t)(BM + BN) * LS * sizeof(T);
    dim3 grid((unsigned)cdiv(g.N, BN), (unsigned)cdiv(g.M, BM));
    if (g.conv_Cin > 0)
        hipLaunchKernelGGL((gemm_kernel<T, BM, BN, WM, WN, KT, true>), grid, dim3(64 * WM * WN), lds, s, g);
    else
        hipLaunchKernelGGL((gemm_kernel<T, BM, BN, WM, WN, KT, false>), grid, dim3(64 * WM * WN), lds, s, g);
    SS_LAUNCH_CHECK("gemm");
    return SS_OK;
}


// =====================================================================================
// v2 main loop for bf16/f16: global -> LDS by DMA (global_load_lds_dwordx4: no VGPR round trip, no
// ds_write issue cost), double-buffered, ONE barrier per k-tile, XOR-swizzled LDS image.
//   LDS image of a tile: rows of BK = 64 elements = 8 chunks of 16 B, row stride 128 B (linear, as
//   the DMA requires: destination = wave-uniform base + lane * 16); chunk c of row r is stored at
//   physical chunk c ^ (r & 7).  The permutation is applied on the SOURCE address of each lane (a
//   permutation inside one 128-byte line: coalescing unchanged) and on the fragment read address,
//   which makes every ds_read_b128 lane group hit 16 distinct 16-byte slots (conflict-free).
//   Out-of-range rows / k read from a zero page (the DMA cannot synthesise zeros).
// =====================================================================================
__device__ __attribute__((aligned(16))) unsigned int g_zero_page[64];

typedef __attribute__((address_space(3))) void lds_void_t;

// One LDS-DMA piece: 64 lanes x 16 B from per-lane global addresses to LDS [lds_base .. +1 KiB) (lane-linear).
// Issued from inline asm on purpose: hipcc models the builtin form as an LDS store and drains it
// (s_waitcnt vmcnt(0)) in front of the next ds_read of the OTHER buffer, serialising the pipeline; the asm
// form is invisible to that bookkeeping, and the explicit vmcnt(0)+barrier below is the only wait.
// M0 carries the LDS base and is compiler-reserved: saved/restored inside the same statement.
__device__ __forceinline__ void dma16(const void* gsrc, uint32_t lds_base) {
    uint32_t keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %2\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, off\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(gsrc), "s"(lds_base)
        : "memory");
}

template <int N> __device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// Epilogue shared by the LDS-DMA kernels: acc[i][j] is the 16x16 fragment at rows m_base + j*16.., columns
// n_base + i*16.. (lane l15 -> row, lane group grp -> 4 consecutive columns); same contract as gemm_kernel.
// A lane owns 4 consecutive columns of one row per fragment, so bias / rowvec / residual / C move as one 8-byte
// access each when the 4 columns are in range and the operands are 8-byte aligned (every shape on the path).
// Fragment columns are processed one i at a time (sched_barrier): the FM fragments of a column block have their
// loads in flight together, but live ranges do not span the whole tile (32 fragments at 256x256).
template <typename T>
__device__ __forceinline__ void ld4(const T* p, float (&v)[4]) {
    if constexpr (Tr<T>::kVec == 8) {
        const uint2 u = *reinterpret_cast<const uint2*>(p);
        float f[8];
        unpack<T>(make_uint4(u.x, u.y, 0, 0), f);
        v[0] = f[0]; v[1] = f[1]; v[2] = f[2]; v[3] = f[3];
    } else {
        const float4 u = *reinterpret_cast<const float4*>(p);
        v[0] = u.x; v[1] = u.y; v[2] = u.z; v[3] = u.w;
    }
}

template <typename T, int FM, int FN>
__device__ __forceinline__ void gemm_epilogue(const GemmArgs& g, f32x4_t (&acc)[FN][FM], int m_base, int n_base,
                                              int l15, int grp) {
    const int M = g.M, N = g.N;
    T* __restrict__ C = (T*)g.C;
    const T* bias = (const T*)g.bias;
    const T* res = (const T*)g.residual;
    constexpr size_t AL = 4 * sizeof(T) - 1;   // alignment mask of a 4-element access
    const bool vec_ok = (((size_t)g.bias | (size_t)g.residual | (size_t)g.rowvec) & AL) == 0 &&
                        ((g.ldr | g.rowvec_ld) & 3) == 0;
    const bool vec_c = ((size_t)g.C & AL) == 0 && (g.ldc & 3) == 0;
#pragma unroll
    for (int i = 0; i < FN; ++i) {
        const int n0 = n_base + i * 16 + grp * 4;
        const bool full = n0 + 3 < N;
        const bool fast = full && vec_ok;
        float bv[4] = {0.f, 0.f, 0.f, 0.f};
        if (g.epi & SS_EPI_BIAS) {
            if (fast) ld4<T>(bias + n0, bv);
            else {
#pragma unroll
                for (int r = 0; r < 4; ++r) if (n0 + r < N) bv[r] = Tr<T>::ld(bias + n0 + r);
            }
        }
#pragma unroll
        for (int j = 0; j < FM; ++j) {
            const int m = m_base + j * 16 + l15;
            if (m >= M) continue;
            float v[4];
            float rv[4] = {0.f, 0.f, 0.f, 0.f};
            if (g.rowvec) {
                const T* rp = (const T*)g.rowvec + (int64_t)(m / g.rows_per_batch) * g.rowvec_ld;
                if (fast) ld4<T>(rp + n0, rv);
                else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) if (n0 + r < N) rv[r] = Tr<T>::ld(rp + n0 + r);
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float t = acc[i][j][r] + bv[r];
                if (g.epi & SS_EPI_GELU) t = gelu_for<T>(Tr<T>::rnd(t));
                v[r] = Tr<T>::rnd(t);
                if (g.rowvec) v[r] = Tr<T>::rnd(v[r] + rv[r]);   // h = conv(x) + temb[:, :, None, None]
            }
            if (g.epi & SS_EPI_RESIDUAL) {
                if (fast) {
                    float rr[4];
                    ld4<T>(res + (int64_t)m * g.ldr + n0, rr);
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] += rr[r];
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (n0 + r < N) v[r] += Tr<T>::ld(res + (int64_t)m * g.ldr + n0 + r);
                }
            }
            if (g.epi & SS_EPI_GEGLU_PAIR) {   // (value, gate) interleaved columns -> out[m][n/2] = value * gelu(gate)
                const float o0 = v[0] * Tr<T>::rnd(gelu_for<T>(v[1])), o1 = v[2] * Tr<T>::rnd(gelu_for<T>(v[3]));
                if (full && ((g.ldc & 1) == 0) && (((size_t)g.C & 3) == 0) && Tr<T>::kVec == 8) {
                    float pk[8] = {o0, o1, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                    *reinterpret_cast<uint32_t*>(C + (int64_t)m * g.ldc + (n0 >> 1)) = pack<T>(pk).x;
                } else {
                    if (n0 + 1 < N) Tr<T>::st(C + (int64_t)m * g.ldc + (n0 >> 1), o0);
                    if (n0 + 3 < N) Tr<T>::st(C + (int64_t)m * g.ldc + (n0 >> 1) + 1, o1);
                }
                continue;
            }
            if (full && vec_c && Tr<T>::kVec == 8) {
                float pk[8] = {v[0], v[1], v[2], v[3], 0.f, 0.f, 0.f, 0.f};
                const uint4 u = pack<T>(pk);
                *reinterpret_cast<uint2*>(C + (int64_t)m * g.ldc + n0) = make_uint2(u.x, u.y);
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (n0 + r < N) Tr<T>::st(C + (int64_t)m * g.ldc + n0 + r, v[r]);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

template <typename T, int BM, int BN, int WM, int WN, int NS, bool CONV>
__global__ __launch_bounds__(64 * WM * WN) void gemm_glds_kernel(const GemmArgs g) {
    constexpr int V = 8;
    constexpr int NW = WM * WN;
    constexpr int BK = 64;
    constexpr int TM = BM / WM, TN = BN / WN;
    constexpr int FM = TM / 16, FN = TN / 16;
    constexpr int IA = BM / 8 / NW, IW = BN / 8 / NW;   // DMA instructions (8 rows each) per wave per tile
    static_assert(BM % (8 * NW) == 0 && BN % (8 * NW) == 0, "tile rows must split evenly over the waves");
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    // [buf][A rows | W rows][128 B]
    constexpr int TILE_BYTES = (BM + BN) * 128;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);   // provably wave-uniform (SGPR) for the DMA bases
    const int wm = wid / WN, wn = wid % WN;
    const int l15 = lane & 15, grp = lane >> 4;
    int mt_, nt_;
    xcd_tile(g.swz, (g.M + BM - 1) / BM, (g.N + BN - 1) / BN, mt_, nt_);
    const int m_blk = mt_ * BM, n_blk = nt_ * BN;
    const T* __restrict__ A = (const T*)g.A;
    const T* __restrict__ W = (const T*)g.W;
    const int K = g.K, M = g.M, N = g.N;
    const int ntiles = (K + BK - 1) / BK;
    const T* zero = reinterpret_cast<const T*>(g_zero_page);
    // wave-uniform LDS byte address of the dynamic segment (SGPR)
    const uint32_t lds0 = __builtin_amdgcn_readfirstlane((uint32_t)(size_t)(lds_void_t*)smem_raw);

    // per-lane staging coordinates: instruction i covers tile rows (wid*I + i)*8 .. +8; lane -> (row, phys chunk)
    const int srow = lane >> 3;                 // row within the 8-row group
    const int schunk = (lane & 7) ^ srow;       // logical chunk this lane must fetch ((row & 7) == srow)
    int cb[IA], cy[IA], cx[IA];
    if constexpr (CONV) {
#pragma unroll
        for (int i = 0; i < IA; ++i) {
            const int m = m_blk + (wid * IA + i) * 8 + srow;
            const int hw = g.conv_Ho * g.conv_Wo;
            cb[i] = m / hw;
            const int rem = m - cb[i] * hw;
            cy[i] = rem / g.conv_Wo;
            cx[i] = rem - cy[i] * g.conv_Wo;
        }
    }
    auto issue_tile = [&](int t, int buf) {
        const int k = t * BK + schunk * V;
        const uint32_t base = lds0 + (uint32_t)(buf * TILE_BYTES);
#pragma unroll
        for (int i = 0; i < IA; ++i) {
            const int r8 = (wid * IA + i) * 8;
            const int m = m_blk + r8 + srow;
            const T* src = zero;
            if constexpr (CONV) {
                // Cin % 64 == 0 (every UNet / VAE conv but conv_in): the whole 64-wide k-tile lies inside one
                // filter tap, so the tap is a wave-uniform scalar division per tile instead of one per lane
                int tap, ci;
                if ((g.conv_Cin & 63) == 0) {
                    const int k0 = t * BK;
                    tap = k0 / g.conv_Cin;
                    ci = k0 - tap * g.conv_Cin + schunk * V;
                } else {
                    tap = k / g.conv_Cin;
                    ci = k - tap * g.conv_Cin;
                }
                const int dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;
                int iy = cy[i] * g.conv_stride + dy, ix = cx[i] * g.conv_stride + dx;
                const int Hin = g.conv_up ? 2 * g.conv_H : g.conv_H, Win = g.conv_up ? 2 * g.conv_W : g.conv_W;
                const bool ok = m < M && k < K && iy >= 0 && iy < Hin && ix >= 0 && ix < Win;
                if (g.conv_up) { iy >>= 1; ix >>= 1; }
                if (ok) src = A + (((int64_t)cb[i] * g.conv_H + iy) * g.conv_W + ix) * g.conv_Cin + ci;
            } else {
                if (m < M && k < K) src = A + (int64_t)m * g.lda + k;
            }
            dma16(src, __builtin_amdgcn_readfirstlane(base + (uint32_t)(r8 * 128)));
        }
#pragma unroll
        for (int i = 0; i < IW; ++i) {
            const int r8 = (wid * IW + i) * 8;
            const int n = n_blk + r8 + srow;
            const T* src = (n < N && k < K) ? W + (int64_t)n * g.ldw + k : zero;
            dma16(src, __builtin_amdgcn_readfirstlane(base + (uint32_t)((BM + r8) * 128)));
        }
    };

    f32x4_t acc[FN][FM];
#pragma unroll
    for (int i = 0; i < FN; ++i)
#pragma unroll
        for (int j = 0; j < FM; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    // NS-deep ring: tiles t+1 .. t+NS-2 stay in flight across the barrier (counted vmcnt, never drained
    // in steady state); small tiles have too little MFMA work per tile to hide a DMA round trip otherwise.
    constexpr int IPT = IA + IW;   // DMA instructions per tile per wave
#pragma unroll
    for (int st = 0; st < NS - 1; ++st)
        if (st < ntiles) issue_tile(st, st);
    int buf = 0;
    for (int t = 0; t < ntiles; ++t) {
        if (t + NS - 2 < ntiles) wait_vmcnt<(NS - 2) * IPT>();   // tile t landed; later tiles may still fly
        else wait_vmcnt<0>();
        __syncthreads();                                    // ... everyone's pieces too; buf[(t-1)%NS] is free again
        if (t + NS - 1 < ntiles) {
            int nb = buf + NS - 1;
            if (nb >= NS) nb -= NS;
            issue_tile(t + NS - 1, nb);
        }
        const char* abuf = smem_raw + buf * TILE_BYTES;
        if (++buf == NS) buf = 0;
        const char* wbuf = abuf + BM * 128;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            uint4 fw[FN], fa[FM];
#pragma unroll
            for (int i = 0; i < FN; ++i) {
                const int r = wn * TN + i * 16 + l15;
                fw[i] = *reinterpret_cast<const uint4*>(wbuf + r * 128 + (((ks * 4 + grp) ^ (r & 7)) << 4));
            }
#pragma unroll
            for (int j = 0; j < FM; ++j) {
                const int r = wm * TM + j * 16 + l15;
                fa[j] = *reinterpret_cast<const uint4*>(abuf + r * 128 + (((ks * 4 + grp) ^ (r & 7)) << 4));
            }
#pragma unroll
            for (int i = 0; i < FN; ++i)
#pragma unroll
                for (int j = 0; j < FM; ++j) acc[i][j] = Mma<T>::run(fw[i], fa[j], acc[i][j]);
        }
    }

    gemm_epilogue<T, FM, FN>(g, acc, m_blk + wm * TM, n_blk + wn * TN, l15, grp);
}

template <typename T, int BM, int BN, int WM, int WN, int NS>
static int gemm_glds_launch_cfg(const GemmArgs& g, hipStream_t s);

// ---- software-pipelined LDS-DMA GEMM (K % 64 == 0; conv: Cin % 64 == 0) ---------------------------------
// Same tile format as gemm_glds_kernel (64-wide K tiles, 128-byte rows, 16-byte chunks XOR-swizzled by row),
// restructured around what rocprofv3 showed limits that kernel (MFMA busy 39 %, a third of the time parked in
// s_waitcnt/s_barrier): (1) the fragment reads of the NEXT half-tile are in flight while the MFMAs of the
// current one run (two register sets), so no MFMA ever waits on LDS latency; (2) the one barrier per K tile sits
// in the MIDDLE of the tile's MFMA work, between the two half-tiles; (3) DMA sources are `SGPR base + 32-bit
// lane offset`: the lane offsets are loop-invariant, the base advances by one scalar add per tile, so issuing a
// tile costs no vector ALU work (the old form rebuilt a 64-bit address and two bounds tests per DMA).
// Rows beyond M / N are clamped to the last valid row at setup (their results are never stored).
__device__ __forceinline__ uint32_t m0_save() {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0" : "=s"(keep)::"memory");
    return keep;
}
__device__ __forceinline__ void m0_restore(uint32_t keep) { asm volatile("s_mov_b32 m0, %0" ::"s"(keep) : "memory"); }
// one 16-byte-per-lane DMA: global (sbase + voff) -> LDS (lds_base + lane*16).  M0 is left modified.
__device__ __forceinline__ void dma16s(uint32_t voff, const void* sbase, uint32_t lds_base) {
    asm volatile(
        "s_mov_b32 m0, %2\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %0, %1"
        :
        : "v"(voff), "s"(sbase), "s"(lds_base)
        : "memory");
}

template <typename T, int BM, int BN, int WM, int WN, int NH, bool CONV>
__global__ __launch_bounds__(64 * WM * WN) void gemm_sp_kernel(const GemmArgs g) {
    constexpr int NW = WM * WN;
    constexpr int TM = BM / WM, TN = BN / WN;
    constexpr int FM = TM / 16, FN = TN / 16;
    constexpr int IA = BM / 8 / NW, IW = BN / 8 / NW;   // DMA instructions (8 rows each) per wave per tile
    static_assert(BM % (8 * NW) == 0 && BN % (8 * NW) == 0, "tile rows must split evenly over the waves");
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    constexpr int TILE_BYTES = (BM + BN) * 128;          // [A rows | W rows] x 128 B, two buffers

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid / WN, wn = wid % WN;
    const int l15 = lane & 15, grp = lane >> 4;
    int mt_, nt_;
    xcd_tile(g.swz, (g.M + BM - 1) / BM, (g.N + BN - 1) / BN, mt_, nt_);
    const int m_blk = mt_ * BM, n_blk = nt_ * BN;
    const int M = g.M, N = g.N;
    const int ntiles = g.K / 64;
    const uint32_t lds0 = __builtin_amdgcn_readfirstlane((uint32_t)(size_t)(lds_void_t*)smem_raw);

    // ---- staging coordinates (loop invariant) ------------------------------------------------------------
    const int srow = lane >> 3;                 // row within an 8-row DMA group
    const int schunk = (lane & 7) ^ srow;       // logical 16-byte chunk this lane fetches (source-side swizzle)
    uint32_t voffW[IW];
#pragma unroll
    for (int i = 0; i < IW; ++i) {
        int n = n_blk + (wid * IW + i) * 8 + srow;
        n = n < N ? n : N - 1;
        voffW[i] = (uint32_t)(((int64_t)(n - n_blk) * g.ldw + schunk * 8) * 2);
    }
    const char* sbaseW = (const char*)g.W + (int64_t)n_blk * g.ldw * 2;
    uint32_t voffA[IA];
    int cbH[IA], cyS[IA], cxS[IA];
    const char* sbaseA = (const char*)g.A;
    if constexpr (CONV) {
        const int hw = g.conv_Ho * g.conv_Wo;
#pragma unroll
        for (int i = 0; i < IA; ++i) {
            int m = m_blk + (wid * IA + i) * 8 + srow;
            m = m < M ? m : M - 1;
            const int b = m / hw;
            const int rem = m - b * hw;
            const int y = rem / g.conv_Wo;
            cbH[i] = b * g.conv_H;
            cyS[i] = y * g.conv_stride;
            cxS[i] = (rem - y * g.conv_Wo) * g.conv_stride;
        }
    } else {
#pragma unroll
        for (int i = 0; i < IA; ++i) {
            int m = m_blk + (wid * IA + i) * 8 + srow;
            m = m < M ? m : M - 1;
            voffA[i] = (uint32_t)(((int64_t)(m - m_blk) * g.lda + schunk * 8) * 2);
        }
        sbaseA += (int64_t)m_blk * g.lda * 2;
    }

    auto issue_tile = [&](int t, int buf) {
        const uint32_t base = lds0 + (uint32_t)(buf * TILE_BYTES);
        const uint32_t keep = m0_save();
        if constexpr (CONV) {
            const int k0 = t * 64;
            const int tap = k0 / g.conv_Cin;                       // wave-uniform: the whole tile is one filter tap
            const int ci = k0 - tap * g.conv_Cin + schunk * 8;
            const int dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;
            const int Hin = g.conv_up ? 2 * g.conv_H : g.conv_H, Win = g.conv_up ? 2 * g.conv_W : g.conv_W;
#pragma unroll
            for (int i = 0; i < IA; ++i) {
                const int r8 = (wid * IA + i) * 8;
                int iy = cyS[i] + dy, ix = cxS[i] + dx;
                const bool ok = (unsigned)iy < (unsigned)Hin && (unsigned)ix < (unsigned)Win;
                if (g.conv_up) { iy >>= 1; ix >>= 1; }
                const uint32_t off = (uint32_t)((((cbH[i] + iy) * g.conv_W + ix) * g.conv_Cin + ci) * 2);
                if (ok) {
                    dma16s(off, sbaseA, __builtin_amdgcn_readfirstlane(base + (uint32_t)(r8 * 128)));
                } else {   // zero padding: this lane's 16-byte slot is written directly (the DMA skips masked lanes)
                    *reinterpret_cast<uint4*>(smem_raw + buf * TILE_BYTES + r8 * 128 + lane * 16) = make_uint4(0, 0, 0, 0);
                }
            }
        } else {
            const char* sb = sbaseA + (int64_t)t * 128;
#pragma unroll
            for (int i = 0; i < IA; ++i)
                dma16s(voffA[i], sb, __builtin_amdgcn_readfirstlane(base + (uint32_t)((wid * IA + i) * 8 * 128)));
        }
        const char* sw = sbaseW + (int64_t)t * 128;
#pragma unroll
        for (int i = 0; i < IW; ++i)
            dma16s(voffW[i], sw, __builtin_amdgcn_readfirstlane(base + (uint32_t)((BM + (wid * IW + i) * 8) * 128)));
        m0_restore(keep);
    };

    // ---- fragment addressing: row r of a tile lives at r*128, chunk c at ((c ^ (r & 7)) << 4); r & 7 == l15 & 7
    // for every fragment, so ks = 1 is the ks = 0 address XOR 64 and fragments are 2048 B apart ------------------
    const uint32_t fa0 = (uint32_t)((wm * TM + l15) * 128 + ((grp ^ (l15 & 7)) << 4));
    const uint32_t fw0 = (uint32_t)((BM + wn * TN + l15) * 128 + ((grp ^ (l15 & 7)) << 4));
    // A K tile is consumed in P = 2*NH phases (ks, h): k-step ks of the tile, h-th 1/NH of the wave's A rows.
    // Phase p multiplies A set (p & 1) with W set (ks & 1); the operands of phase p+1 are read while the MFMAs
    // of phase p run.  NH = 2 halves the fragment registers of a 128-row wave tile (64 instead of 96 VGPRs).
    constexpr int FMH = FM / NH, P = 2 * NH;
    constexpr bool PIN = true;   // accumulate-in-place asm MFMAs: the allocator otherwise shuttles accumulators through copies
    static_assert(FM % NH == 0, "A fragments must split evenly");
    uint4 fa[2][FMH], fw[2][FN];
    auto read_phase = [&](int buf, int p) {   // operands of phase p of the tile in buffer `buf`
        const int ks = p / NH, h = p % NH;
        const char* b = smem_raw + buf * TILE_BYTES;
        const uint32_t xa = ks ? (fa0 ^ 64u) : fa0, xw = ks ? (fw0 ^ 64u) : fw0;
        if (h == 0) {
#pragma unroll
            for (int i = 0; i < FN; ++i) fw[ks & 1][i] = *reinterpret_cast<const uint4*>(b + xw + i * 2048);
        }
#pragma unroll
        for (int j = 0; j < FMH; ++j) fa[p & 1][j] = *reinterpret_cast<const uint4*>(b + xa + (h * FMH + j) * 2048);
    };

    f32x4_t acc[FN][FM];
#pragma unroll
    for (int i = 0; i < FN; ++i)
#pragma unroll
        for (int j = 0; j < FM; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    auto mma_phase = [&](int p) {
        const int ks = p / NH, h = p % NH;
#pragma unroll
        for (int i = 0; i < FN; ++i)
#pragma unroll
            for (int j = 0; j < FMH; ++j) {
                // 128-row wave tiles (NH = 2): the allocator otherwise shuttles the 128 accumulator registers
                // through copies around every MFMA; the asm form pins accumulate-in-place
                if constexpr (PIN) Mma<T>::run_inplace(fw[ks & 1][i], fa[p & 1][j], acc[i][h * FMH + j]);
                else acc[i][h * FMH + j] = Mma<T>::run(fw[ks & 1][i], fa[p & 1][j], acc[i][h * FMH + j]);
            }
    };

    issue_tile(0, 0);
    if (ntiles > 1) issue_tile(1, 1);
    // tile 0 landed (tile 1 may still fly).  CONV: a wave whose 64 lanes are all padding skips that DMA entirely, so
    // the number of outstanding loads per tile is not a constant there and only a full drain is safe
    if (ntiles > 1 && !CONV) wait_vmcnt<IA + IW>(); else wait_vmcnt<0>();
    __syncthreads();
    read_phase(0, 0);
    if constexpr (PIN) asm volatile("s_nop 7" ::: "memory");   // asm MFMAs are opaque to the hazard recognizer
    // one K tile: phases 0 .. P-2 each prefetch their successor; before the LAST phase's MFMAs comes the tile's one
    // barrier (tile t+1 landed, buffer of tile t drained), the DMA of tile t+2 and the prefetch of tile t+1's phase 0.
    // (sched_barrier pins the phase order: left alone, the scheduler sinks the MFMAs behind the barrier and the
    // fragment reads in front of their consumers, which re-exposes the LDS latency this layout exists to hide)
    auto step = [&](int t, int buf) {
#pragma unroll
        for (int p = 0; p < P; ++p) {
            if (p < P - 1) {
                read_phase(buf, p + 1);
            } else if (t + 1 < ntiles) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // all my fragments of tile t are in registers
                wait_vmcnt<0>();                                       // my share of tile t+1 has landed
                __syncthreads();
                if (t + 2 < ntiles) issue_tile(t + 2, buf);
                read_phase(buf ^ 1, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            mma_phase(p);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    int t = 0;
    for (; t + 1 < ntiles; t += 2) {
        step(t, 0);
        step(t + 1, 1);
    }
    if (t < ntiles) step(t, 0);
    if constexpr (PIN) asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");   // MFMA results settle before VALU reads them

    gemm_epilogue<T, FM, FN>(g, acc, m_blk + wm * TM, n_blk + wn * TN, l15, grp);
}

template <typename T, int BM, int BN, int WM, int WN, int NH = 1>
static int gemm_sp_launch_cfg(const GemmArgs& g, hipStream_t s) {
    if constexpr (Tr<T>::kVec == 8) {
        const bool conv = g.conv_Cin > 0;
        // the pipelined kernel needs whole 64-wide K tiles (conv: one filter tap per tile) and 32-bit byte offsets
        const bool ok = (g.K % 64 == 0) && (!conv || g.conv_Cin % 64 == 0) && g.K >= 64 &&
                        (conv ? (int64_t)g.M * g.conv_stride * g.conv_stride * g.conv_Cin * 2 < (1ll << 31)
                              : ((int64_t)BM * g.lda * 2 < (1ll << 31)) ) && (int64_t)BN * g.ldw * 2 < (1ll << 31);
        if (!ok) return gemm_glds_launch_cfg<T, BM, BN, WM, WN, 2>(g, s);
        const size_t lds = (size_t)2 * (BM + BN) * 128;
        dim3 grid((unsigned)cdiv(g.N, BN), (unsigned)cdiv(g.M, BM));
        if (lds > 64 * 1024) {
            if (conv) hipFuncSetAttribute((const void*)gemm_sp_kernel<T, BM, BN, WM, WN, NH, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            else hipFuncSetAttribute((const void*)gemm_sp_kernel<T, BM, BN, WM, WN, NH, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        }
        if (conv)
            hipLaunchKernelGGL((gemm_sp_kernel<T, BM, BN, WM, WN, NH, true>), grid, dim3(64 * WM * WN), lds, s, g);
        else
            hipLaunchKernelGGL((gemm_sp_kernel<T, BM, BN, WM, WN, NH, false>), grid, dim3(64 * WM * WN), lds, s, g);
        SS_LAUNCH_CHECK("gemm_sp");
        return SS_OK;
    } else {
        return gemm_launch_cfg<T, 128, 128, 2, 2>(g, s);
    }
}

template <typename T, int BM, int BN, int WM, int WN, int NS>
static int gemm_glds_launch_cfg(const GemmArgs& g, hipStream_t s) {
    if constexpr (Tr<T>::kVec == 8) {
        const size_t lds = (size_t)NS * (BM + BN) * 128;
        dim3 grid((unsigned)cdiv(g.N, BN), (unsigned)cdiv(g.M, BM));
        if (g.conv_Cin > 0)
            hipLaunchKernelGGL((gemm_glds_kernel<T, BM, BN, WM, WN, NS, true>), grid, dim3(64 * WM * WN), lds, s, g);
        else
            hipLaunchKernelGGL((gemm_glds_kernel<T, BM, BN, WM, WN, NS, false>), grid, dim3(64 * WM * WN), lds, s, g);
        SS_LAUNCH_CHECK("gemm_glds");
        return SS_OK;
    } else {
        return gemm_launch_cfg<T, 128, 128, 2, 2>(g, s);   // fp32 (CPU-parity mode) keeps the register-staged kernel
    }
}

template <typename T>
static int gemm_dispatch_cfg(int cfg, const GemmArgs& g, hipStream_t s) {
    switch (cfg) {
        case 1: return gemm_launch_cfg<T, 128, 128, 2, 2>(g, s);
        case 2: return gemm_launch_cfg<T, 64, 64, 2, 2>(g, s);
        case 4: return gemm_launch_cfg<T, 128, 128, 2, 2, 2>(g, s);   // BK = 128
        case 5: return gemm_launch_cfg<T, 256, 128, 4, 2>(g, s);      // 8 waves
        case 6: return gemm_launch_cfg<T, 128, 256, 2, 4>(g, s);      // 8 waves
        case 7: return gemm_launch_cfg<T, 256, 256, 4, 2>(g, s);      // 8 waves, 64x128 per wave
        case 8: return gemm_glds_launch_cfg<T, 128, 128, 2, 2, 2>(g, s); // DMA staging, swizzled, double-buffered
        case 9: return gemm_glds_launch_cfg<T, 256, 128, 4, 2, 2>(g, s);
        case 10: return gemm_glds_launch_cfg<T, 64, 64, 2, 2, 2>(g, s);
        case 11: return gemm_glds_launch_cfg<T, 128, 256, 2, 4, 2>(g, s);
        case 12: return gemm_glds_launch_cfg<T, 64, 64, 2, 2, 3>(g, s);
        case 13: return gemm_glds_launch_cfg<T, 64, 64, 2, 2, 4>(g, s);
        case 14: return gemm_glds_launch_cfg<T, 128, 64, 2, 2, 3>(g, s);
        case 15: return gemm_glds_launch_cfg<T, 128, 64, 2, 2, 2>(g, s);
        case 16: return gemm_glds_launch_cfg<T, 128, 128, 2, 2, 3>(g, s);
        case 17: return gemm_glds_launch_cfg<T, 64, 128, 2, 2, 3>(g, s);
        case 20: return gemm_sp_launch_cfg<T, 128, 128, 2, 2>(g, s);   // software-pipelined DMA kernels
        case 21: return gemm_sp_launch_cfg<T, 128, 64, 2, 2>(g, s);
        case 22: return gemm_sp_launch_cfg<T, 64, 64, 2, 2>(g, s);
        case 23: return gemm_sp_launch_cfg<T, 256, 128, 4, 2>(g, s);
        case 24: return gemm_sp_launch_cfg<T, 256, 256, 2, 4, 2>(g, s);   // 8 waves, 128x64 per wave, A in halves
        case 25: return gemm_sp_launch_cfg<T, 256, 128, 2, 2, 2>(g, s);   // 4 waves, 128x64 per wave
        // 160-wide tiles: every SDXL channel count (640 ... 10240) is a multiple of 160, and [8192 x 1280] outputs
        // are exactly 512 tiles of 128x160 = one full wave of 2 blocks per CU (128x128 leaves the second wave 3/4 empty)
        case 26: return gemm_sp_launch_cfg<T, 128, 160, 2, 2>(g, s);
        case 28: return gemm_sp_launch_cfg<T, 128, 320, 2, 4>(g, s);   // 8 waves, 64x80 per wave, one block per CU
        case 29: return gemm_sp_launch_cfg<T, 64, 160, 2, 2>(g, s);    // 32x80 per wave: small-M shapes
        default: return gemm_launch_cfg<T, 128, 32, 4, 1>(g, s);
    }
}

// Tile choice (measured on MI355X, tools/gpu_diag.py gemm_unet): the DMA-staged kernels (8 = 128x128,
// 10 = 64x64) beat the register-staged ones by 30-45 % on every shape with M > 128; 128x128 needs
// >= ~256 blocks to fill 256 CUs, otherwise 64x64 tiles win.  M <= 128 is weight streaming (narrow-N tiles).
static int pick_cfg(int64_t M, int64_t N) {
    const int force = tuning_get("gemm_cfg", 0);
    if (force) return force;
    const int64_t big_blocks = (int64_t)cdiv(M, 128) * cdiv(N, 128);
    if (M <= 128) return 3;
    if (big_blocks >= 256) return 8;
    return 10;
}


// ---- per-shape autotuning of the tile configuration -------------------------------------------------
// The best tile (128x128 / 128x64 / 64x64 DMA-staged kernels) depends on M, N, K in ways a closed-form rule
// misses by 10-40 % (tools/gpu_diag.py gemm_unet).  The first call with a new (dtype, M, N, K, conv geometry)
// times each candidate with HIP events on the caller's stream — into a scratch output, residual disabled, so
// in-place residual updates are not applied twice — and caches the winner.  ~40 distinct shapes per pipeline.
typedef std::tuple<int, int, int, int, int, int, int, int> TuneKey;
static std::mutex g_tune_mutex;
static std::map<TuneKey, int>& tune_cache() {
    static std::map<TuneKey, int> m;
    return m;
}

// A tuned entry is (tile config, XCD group size): the tile order's effect is as shape-dependent as the tile's
// (tools/gpu_diag.py gemm_unet: -13 % ... +28 %).  Candidates are timed COLD — a 320 MB scratch fill between runs
// evicts the operands from L2 / Infinity Cache — because in the pipeline every weight is touched once per forward.
template <typename T>
static int autotuned_cfg(GemmArgs& g0, hipStream_t s) {
    const int fallback = pick_cfg(g0.M, g0.N);
    if (Tr<T>::kVec != 8 || g0.M <= 128 || tuning_get("gemm_cfg", 0) || !tuning_get("gemm_autotune", 1)) return fallback;
    const TuneKey key(Tr<T>::kDtype, g0.M, g0.N, g0.K, g0.conv_Cin, g0.conv_stride * 2 + g0.conv_up, g0.conv_H, g0.conv_W);
    {
        std::lock_guard<std::mutex> lk(g_tune_mutex);
        auto it = tune_cache().find(key);
        if (it != tune_cache().end()) { g0.swz = it->second / 100; return it->second % 100; }
    }
    // an untuned shape met under stream capture (the UNet forward graph) cannot be timed: closed-form choice
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) return fallback;
    static void* flush = nullptr;
    const size_t flush_bytes = (size_t)320 << 20;
    if (!flush && hipMalloc(&flush, flush_bytes) != hipSuccess) { flush = nullptr; return fallback; }
    void* scratch = nullptr;
    if (hipMalloc(&scratch, (size_t)g0.M * g0.N * sizeof(T)) != hipSuccess) return fallback;
    GemmArgs g = g0;
    g.C = scratch; g.ldc = g0.N; g.residual = nullptr; g.epi &= ~SS_EPI_RESIDUAL;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    // 8/15/10 = double-buffered DMA kernels (128x128, 128x64, 64x64); 20/21/23 = software-pipelined DMA kernels
    // (128x128, 128x64, 256x128); 24 = 256x256 with 128x64 wave tiles (accumulators pinned); 26 = 128x160
    const int cands[10] = {8, 15, 10, 20, 21, 23, 24, 26, 28, 29};
    const int swzs[3] = {0, 4, 8};
    int best = fallback, best_swz = g0.swz;
    float best_ms = 1e30f;
    for (int c : cands) {
        for (int z : swzs) {
            g.swz = z;
            if (gemm_dispatch_cfg<T>(c, g, s) != SS_OK) continue;   // warm-up (also faults pages in)
            float tmin = 1e30f;              // best of 3 cold runs: robust against a stray slow run
            bool ok = true;
            for (int r = 0; r < 3 && ok; ++r) {
                hipMemsetAsync(flush, r, flush_bytes, s);
                hipEventRecord(e0, s);
                gemm_dispatch_cfg<T>(c, g, s);
                hipEventRecord(e1, s);
                ok = hipEventSynchronize(e1) == hipSuccess;
                float ms = 0.f;
                if (ok) { hipEventElapsedTime(&ms, e0, e1); if (ms < tmin) tmin = ms; }
            }
            if (ok && tmin < best_ms) { best_ms = tmin; best = c; best_swz = z; }
        }
    }
    if (tuning_get("gemm_autotune_log", 0))
        fprintf(stderr, "[ss autotune] M=%d N=%d K=%d conv=%d -> cfg %d swz %d (%.1f us)\n", g0.M, g0.N, g0.K, g0.conv_Cin,
                best, best_swz, best_ms * 1e3f);
    hipEventDestroy(e0); hipEventDestroy(e1);
    hipFree(scratch);
    std::lock_guard<std::mutex> lk(g_tune_mutex);
    tune_cache()[key] = best + 100 * best_swz;
    g0.swz = best_swz;
    return best;
}

template <typename T>
int gemm_launch(const void* A, const void* W, void* C, int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldw,
                int64_t ldc, const void* bias, const void* residual, int64_t ldr, int epi, hipStream_t s) {
    constexpr int V = Tr<T>::kVec;
    SS_REQUIRE(K % V == 0 && lda % V == 0 && ldw % V == 0, "gemm: K/lda/ldw must be multiples of %d (K=%lld)", V,
               (long long)K);
    SS_REQUIRE(!(epi & SS_EPI_SILU_MUL), "gemm: SILU_MUL is a GEMV epilogue (use ss_silu_mul)");
    SS_REQUIRE(!(epi & SS_EPI_GEGLU_PAIR) || (!(epi & (SS_EPI_GELU | SS_EPI_RESIDUAL)) && N % 2 == 0),
               "gemm: GEGLU_PAIR cannot be combined with GELU/RESIDUAL and needs an even N");
    SS_REQUIRE(!(epi & SS_EPI_BIAS) || bias, "gemm: bias epilogue without bias");
    SS_REQUIRE(!(epi & SS_EPI_RESIDUAL) || residual, "gemm: residual epilogue without residual");
    if (M == 0 || N == 0) return SS_OK;
    GemmArgs g;
    g.A = A; g.W = W; g.C = C; g.bias = bias; g.residual = residual;
    g.M = (int)M; g.N = (int)N; g.K = (int)K; g.lda = lda; g.ldw = ldw; g.ldc = ldc; g.ldr = ldr; g.epi = epi;
    g.rowvec = nullptr; g.rows_per_batch = 1; g.rowvec_ld = 0;
    g.conv_H = g.conv_W = g.conv_Cin = g.conv_stride = g.conv_up = g.conv_Ho = g.conv_Wo = 0;
    g.swz = tuning_get("gemm_xcd_swizzle", 8);
    return gemm_dispatch_cfg<T>(autotuned_cfg<T>(g, s), g, s);
}

// 3x3 convolution, padding 1, stride 1|2, optional fused nearest-2x upsample of the input, NHWC:
// x [B, H, W, Cin] -> y [B, Ho, Wo, Cout];  w [Cout, 9*Cin] with k = (ky*3 + kx)*Cin + ci.
template <typename T>
int conv3x3_launch(const void* x, const void* w, void* y, int64_t B, int64_t H, int64_t Wd, int64_t Cin, int64_t Cout,
                   int64_t stride, int64_t up, const void* bias, const void* rowvec, int64_t rowvec_ld,
                   const void* residual, int epi, hipStream_t s) {
    constexpr int V = Tr<T>::kVec;
    SS_REQUIRE(Cin % V == 0, "conv3x3: Cin=%lld must be a multiple of %d (pad the channels)", (long long)Cin, V);
    SS_REQUIRE((stride == 1 || stride == 2) && !(up && stride != 1), "conv3x3: unsupported stride/upsample");
    const int64_t Hin = up ? 2 * H : H, Win = up ? 2 * Wd : Wd;
    const int64_t Ho = (Hin + 2 - 3) / stride + 1, Wo = (Win + 2 - 3) / stride + 1;
    GemmArgs g;
    g.A = x; g.W = w; g.C = y; g.bias = bias; g.residual = residual;
    g.M = (int)(B * Ho * Wo); g.N = (int)Cout; g.K = (int)(9 * Cin);
    g.lda = 0; g.ldw = 9 * Cin; g.ldc = Cout; g.ldr = Cout; g.epi = epi;
    g.rowvec = rowvec; g.rows_per_batch = (int)(Ho * Wo); g.rowvec_ld = rowvec_ld > 0 ? rowvec_ld : Cout;
    g.swz = tuning_get("gemm_xcd_swizzle", 8);
    g.conv_H = (int)H; g.conv_W = (int)Wd; g.conv_Cin = (int)Cin; g.conv_stride = (int)stride; g.conv_up = (int)up;
    g.conv_Ho = (int)Ho; g.conv_Wo = (int)Wo;
    if (g.M == 0) return SS_OK;
    return gemm_dispatch_cfg<T>(autotuned_cfg<T>(g, s), g, s);
}

int gemm_dev(const void* A, const void* W, void* C, int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldw,
             int64_t ldc, const void* bias, const void* residual, int64_t ldr, int epi, int dtype, hipStream_t s) {
    return SS_DISPATCH(dtype, gemm_launch, A, W, C, M, N, K, lda, ldw, ldc, bias, residual, ldr, epi, s);
}

}  // namespace ss

extern "C" int ss_conv3x3(const void* x, const void* w, void* y, int64_t batch, int64_t H, int64_t W_, int64_t Cin,
                          int64_t Cout, int64_t stride, int64_t upsample2x, const void* bias, const void* rowvec,
                          int64_t rowvec_stride, const void* residual, int dtype, void* stream) {
    const int epi = (bias ? SS_EPI_BIAS : 0) | (residual ? SS_EPI_RESIDUAL : 0);
    return SS_DISPATCH(dtype, ss::conv3x3_launch, x, w, y, batch, H, W_, Cin, Cout, stride, upsample2x, bias, rowvec,
                       rowvec_stride, residual, epi, (hipStream_t)stream);
}

extern "C" int ss_gemm(const void* A, const void* W, void* C, int64_t M, int64_t N, int64_t K, int64_t lda,
                       int64_t ldw, int64_t ldc, const void* bias, const void* residual, int64_t ldr, int epilogue,
                       int dtype, void* stream) {
    return ss::gemm_dev(A, W, C, M, N, K, lda, ldw, ldc, bias, residual, ldr, epilogue, dtype, (hipStream_t)stream);
}
