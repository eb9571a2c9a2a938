// C[M,N] = A[M,K] · W[N,K]^T (+bias)(+GELU)(+residual) on the CDNA4 matrix cores: host dispatch, the register-staged
// kernel (fp32 CPU-parity mode and ragged shapes), the double-buffered LDS-DMA kernel, and the tile-configuration
// table.  The software-pipelined LDS-DMA kernels live in ss_gemm_sp.inc (see ss_gemm_common.h for the operand
// conventions shared by all of them).
//
//   bf16 / fp16 : v_mfma_f32_16x16x32_{bf16,f16}, BK = 64
//   fp32        : v_mfma_f32_16x16x4_f32 (exact fp32 FMA chain; the CPU-parity mode), BK = 32
//
// Tiles: <BM, BN, WM, WN> = block tile (activation rows x weight rows) and the wave grid.
#include <stdio.h>

#include <map>
#include <mutex>

#include "ss_gemm_common.h"

namespace ss {

__device__ __attribute__((aligned(16))) unsigned int g_zero_page[64];

// Register-staged kernel: global -> register -> LDS staging with the next tile's loads in flight during the MFMAs
// (guide T14 write-late form), padded LDS rows (+1 pack) to break the 128-byte stride.
//
// SPLIT (fp32 tensors only; tuning knob `gemm_f32_split`, the "gate mode" of DESIGN §5): the fp32 operands are split on their
// way into LDS, x = hi + lo with hi = bf16(x) and lo = bf16(x - hi) (x - hi is exact in fp32), and the product is formed on the
// bf16 matrix pipe as  A·W^T ~= Ahi·Whi^T + Ahi·Wlo^T + Alo·Whi^T  with fp32 accumulation — three v_mfma_f32_16x16x32_bf16 per
// fragment pair and 32 k instead of eight v_mfma_f32_16x16x4_f32 at 1/16 of the rate.  The dropped Alo·Wlo term and the
// 16-bit operand mantissas leave a relative error of ~2^-17 per product (random sign): measured 4.5e-6 per GEMM against the
// fp64 product (exact fp32 chain: 1.4e-7 .. 1.6e-6), two orders inside the 1e-3 gate on the regressed image features.  Same
// loads, same epilogue, same LDS bytes per row (BK/2 bf16 of hi, then BK/2 bf16 of lo, then the 16-byte pad — where the exact
// mode keeps BK fp32 values).  Launched with 64-wide K tiles (KT = 2): with 32-wide tiles the three MFMAs per fragment pair are
// done in ~770 cycles, less than the L2 round trip of the NEXT tile's register-staged loads, and the kernel ran only 1.25x
// faster than the exact chain (round 5, first measurement: MLLM half 1100 vs 1283 ms per round).
template <typename T, int BM, int BN, int WM, int WN, int KT, bool CONV, bool SPLIT = false>
__global__ __launch_bounds__(64 * WM * WN) void gemm_kernel(const GemmArgs g) {
    static_assert(!SPLIT || Tr<T>::kVec == 4, "SPLIT: fp32 operands");
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
    // Block -> tile.  Default: x walks N (consecutive workgroups share an A row panel).  SPLIT with g.swz < 0 (knob gemm_f32_split_order):
    // M-fastest in groups of -g.swz N tiles — consecutive workgroups share a W panel and a band of |swz| W panels stays hot while the
    // M tiles are walked (the counters put the N-fastest order of [7304, 12288, 4096] at 7.4 x its algorithmic bytes from the fabric)
    int bx = blockIdx.x, by = blockIdx.y;
    if constexpr (SPLIT) {
        if (g.swz <= -1000) {
            // XCD-aware order (round 6; knob gemm_f32_split_order = 1000 + GM): the dispatcher deals linear workgroup ids round-robin
            // over the 8 XCDs, so with any id -> tile map that is monotone in the id every XCD's private L2 sees ALL of A and ALL of W
            // (counters: 10-11 x the algorithmic bytes from the fabric on the stacked-prefill products, bands or not).  xcd_tile_id gives
            // XCD k a contiguous range of tiles walked in groups of GM m-tiles, as the 16-bit kernels do.
            int mt_, nt_;
            xcd_tile_id(-g.swz - 1000, (int)gridDim.y, (int)gridDim.x, (int)(by * gridDim.x + bx), mt_, nt_);
            by = mt_; bx = nt_;
        } else if (g.swz < 0) {
            const int G = -g.swz, NT = gridDim.x, MT = gridDim.y;
            const int lin = by * NT + bx;
            const int band = lin / (G * MT), rem = lin - band * (G * MT);
            const int gw = (band + 1) * G <= NT ? G : NT - band * G;      // width of this band of N tiles
            by = rem / gw;
            bx = band * G + (rem - by * gw);
        }
    }
    const int m_blk = by * BM, n_blk = bx * BN;
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
    // SPLIT: pack c of a row (k = 4c .. 4c+3) -> 8 bytes of hi at byte 8c and 8 bytes of lo at byte 2*BK + 8c of the row
    auto split_store = [&](char* row, int c, const uint4& u) {
        const float x0 = __uint_as_float(u.x), x1 = __uint_as_float(u.y), x2 = __uint_as_float(u.z), x3 = __uint_as_float(u.w);
        const uint32_t h01 = f32x2_to_bf16x2_bits(x0, x1), h23 = f32x2_to_bf16x2_bits(x2, x3);
        const float r0 = x0 - __uint_as_float(h01 << 16), r1 = x1 - __uint_as_float(h01 & 0xffff0000u);
        const float r2 = x2 - __uint_as_float(h23 << 16), r3 = x3 - __uint_as_float(h23 & 0xffff0000u);
        *reinterpret_cast<uint2*>(row + c * 8) = make_uint2(h01, h23);
        *reinterpret_cast<uint2*>(row + 2 * BK + c * 8) = make_uint2(f32x2_to_bf16x2_bits(r0, r1), f32x2_to_bf16x2_bits(r2, r3));
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int i = 0; i < PA; ++i) {
            const int p = tid + i * NT;
            if (p < BM * PPR) {
                if constexpr (SPLIT) split_store(reinterpret_cast<char*>(As + (p / PPR) * LS), p % PPR, ra[i]);
                else st16(As + (p / PPR) * LS + (p % PPR) * V, ra[i]);
            }
        }
#pragma unroll
        for (int i = 0; i < PW; ++i) {
            const int p = tid + i * NT;
            if (p < BN * PPR) {
                if constexpr (SPLIT) split_store(reinterpret_cast<char*>(Ws + (p / PPR) * LS), p % PPR, rw[i]);
                else st16(Ws + (p / PPR) * LS + (p % PPR) * V, rw[i]);
            }
        }
    };

    load_tile(0);
    for (int t = 0; t < ntiles; ++t) {
        store_tile();
        __syncthreads();
        if (t + 1 < ntiles) load_tile(t + 1);  // in flight while the MFMAs run
        if constexpr (SPLIT) {
#pragma unroll
            for (int ks = 0; ks < BK / 32; ++ks) {
                uint4 fwh[FN], fwl[FN], fah[FM], fal[FM];
#pragma unroll
                for (int i = 0; i < FN; ++i) {
                    const char* r = reinterpret_cast<const char*>(Ws + (wn * TN + i * 16 + l15) * LS) + ks * 64 + grp * 16;
                    fwh[i] = *reinterpret_cast<const uint4*>(r);
                    fwl[i] = *reinterpret_cast<const uint4*>(r + 2 * BK);
                }
#pragma unroll
                for (int j = 0; j < FM; ++j) {
                    const char* r = reinterpret_cast<const char*>(As + (wm * TM + j * 16 + l15) * LS) + ks * 64 + grp * 16;
                    fah[j] = *reinterpret_cast<const uint4*>(r);
                    fal[j] = *reinterpret_cast<const uint4*>(r + 2 * BK);
                }
#pragma unroll
                for (int i = 0; i < FN; ++i)
#pragma unroll
                    for (int j = 0; j < FM; ++j) {       // the two cross terms first, the leading term last
                        acc[i][j] = Mma<bf16_t>::run(fwl[i], fah[j], acc[i][j]);
                        acc[i][j] = Mma<bf16_t>::run(fwh[i], fal[j], acc[i][j]);
                        acc[i][j] = Mma<bf16_t>::run(fwh[i], fah[j], acc[i][j]);
                    }
            }
        } else if constexpr (V == 8) {
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
    if constexpr (V == 4 && KT == 1) {      // fp32 tensors: the split-bf16 "gate mode" instead of the exact fp32 FMA chain
        if (tuning_get("gemm_f32_split", 0)) {
            constexpr int SKT = 2;                                  // 64-wide K tiles (see the kernel's header comment)
            constexpr int SLS = 8 * SKT * V + V;
            const size_t slds = (size_t)(BM + BN) * SLS * sizeof(T);
            auto kern = g.conv_Cin > 0 ? gemm_kernel<T, BM, BN, WM, WN, SKT, true, true> : gemm_kernel<T, BM, BN, WM, WN, SKT, false, true>;
            GemmArgs g2 = g;
            // bands of G N tiles walked M-fastest (default 16: [7304, 12288, 4096] 2266 -> 2199 us, [7304, 22016, 4096] 4333 -> 4059 us,
            // the other MLLM shapes unchanged, profiles/round5_split_gemm_bench.json); 0 = the plain N-fastest grid
            g2.swz = -tuning_get("gemm_f32_split_order", 16);
            if (slds > 64 * 1024) {
                if (g.conv_Cin > 0) SS_DYN_LDS((gemm_kernel<T, BM, BN, WM, WN, SKT, true, true>), slds);
                else SS_DYN_LDS((gemm_kernel<T, BM, BN, WM, WN, SKT, false, true>), slds);
            }
            hipLaunchKernelGGL(kern, grid, dim3(64 * WM * WN), slds, s, g2);
            SS_LAUNCH_CHECK("gemm_split");
            return SS_OK;
        }
    }
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

template <typename T> int gemm_sp_dispatch_conv(int cfg, const GemmArgs& g, hipStream_t s);
template <> int gemm_sp_dispatch<float>(int, const GemmArgs&, hipStream_t) { return 1; }
template <> int gemm_sp_dispatch_conv<float>(int, const GemmArgs&, hipStream_t) { return 1; }
// ping-pong 8-phase 256x256 tiles (ss_gemm_pp.inc): cfg 50-59; same contract (1 = not eligible)
template <typename T> int gemm_pp_dispatch(int cfg, const GemmArgs& g, hipStream_t s);
template <typename T> int gemm_pp_dispatch_conv(int cfg, const GemmArgs& g, hipStream_t s);
template <> int gemm_pp_dispatch<float>(int, const GemmArgs&, hipStream_t) { return 1; }
template <> int gemm_pp_dispatch_conv<float>(int, const GemmArgs&, hipStream_t) { return 1; }
// one-stream-per-SIMD kernels (ss_gemm_w4.inc): cfg 90-99; same contract (1 = not eligible)
template <typename T> int gemm_w4_dispatch(int cfg, const GemmArgs& g, hipStream_t s);
template <typename T> int gemm_w4_dispatch_conv(int cfg, const GemmArgs& g, hipStream_t s);
template <> int gemm_w4_dispatch<float>(int, const GemmArgs&, hipStream_t) { return 1; }
template <> int gemm_w4_dispatch_conv<float>(int, const GemmArgs&, hipStream_t) { return 1; }

// cfg ids: 1-3 register-staged, 8/10/15 double-buffered LDS-DMA (any K % 8 == 0, ragged tiles through a zero page),
// 20-52 software-pipelined / role-split LDS-DMA (ss_gemm_sp.inc; K % 64 == 0) with the double-buffered kernel of the nearest tile
// as their fallback for ineligible shapes.
template <typename T>
static int gemm_dispatch_cfg(int cfg, const GemmArgs& g, hipStream_t s) {
    if (cfg >= 90 && cfg < 100 && Tr<T>::kVec == 8) {   // 4-wave / AGPR-accumulator tiles; ineligible shapes take the 8-wave 256x256 tile
        const int rc = g.conv_Cin > 0 ? gemm_w4_dispatch_conv<T>(cfg, g, s) : gemm_w4_dispatch<T>(cfg, g, s);
        if (rc <= 0) return rc;
        cfg = g.conv_Cin > 0 ? 69 : 60;
    }
    if (cfg >= 50 && cfg < 60 && Tr<T>::kVec == 8) {   // ping-pong tiles; ineligible shapes (conv, K % 64) take the one-barrier 256x256 tile
        const int rc = g.conv_Cin > 0 ? gemm_pp_dispatch_conv<T>(cfg, g, s) : gemm_pp_dispatch<T>(cfg, g, s);
        if (rc <= 0) return rc;
        cfg = g.conv_Cin > 0 ? 69 : 60;
    }
    if (cfg >= 20 && Tr<T>::kVec == 8) {
        const int rc = g.conv_Cin > 0 ? gemm_sp_dispatch_conv<T>(cfg, g, s) : gemm_sp_dispatch<T>(cfg, g, s);
        if (rc <= 0) return rc;
        cfg = (cfg == 21 || cfg == 68) ? 15 : (cfg == 22 || cfg == 29 || cfg == 67 || cfg == 70) ? 10 : 8;
    }
    switch (cfg) {
        case 1:
            if constexpr (Tr<T>::kVec == 4) {
                // split-bf16 gate mode: 256x128 with 8 waves for the big stacked-prefill products (measured, tools/split_bench.py:
                // [7304, 12288, 4096] 2447 -> 2290 us, [7304, 22016, 4096] 4883 -> 4362 us; N = 4096 equal; the 528-row block is
                // 40-70 % SLOWER on it, 128x256 never wins) — knob gemm_f32_split_tile: 0 = this rule, 1 / 2 = force 256x128 / 128x256,
                // 3 = force 128x128
                if (tuning_get("gemm_f32_split", 0)) {
                    const int st = tuning_get("gemm_f32_split_tile", 0);
                    if (st == 1 || (st == 0 && g.conv_Cin == 0 && g.M >= 2048 && g.N >= 8192)) return gemm_launch_cfg<T, 256, 128, 4, 2>(g, s);
                    if (st == 2) return gemm_launch_cfg<T, 128, 256, 2, 4>(g, s);
                }
            }
            return gemm_launch_cfg<T, 128, 128, 2, 2>(g, s);
        case 2: return gemm_launch_cfg<T, 64, 64, 2, 2>(g, s);
        case 8: return gemm_glds_launch_cfg<T, 128, 128, 2, 2, 2>(g, s);   // DMA staging, swizzled, double-buffered
        case 10: return gemm_glds_launch_cfg<T, 64, 64, 2, 2, 2>(g, s);
        case 15: return gemm_glds_launch_cfg<T, 128, 64, 2, 2, 2>(g, s);
        default:
            if (cfg >= 20) return gemm_launch_cfg<T, 128, 128, 2, 2>(g, s);   // fp32 asked for a 16-bit-only tile
            return gemm_launch_cfg<T, 128, 32, 4, 1>(g, s);                   // cfg 3: weight streaming (M <= 128)
    }
}

// ---- tile-configuration table ------------------------------------------------------------------------------
// The best (tile, XCD group) of a shape depends on M, N, K in ways a closed-form rule misses by 10-40 %.  The choice
// is DATA, not a side effect of the first call: ss_gemm / ss_conv3x3 only look the shape up (no timing, no
// allocation, no synchronisation — they stay asynchronous and capturable); entries come from ss_gemm_tune /
// ss_conv3x3_tune (explicit, caller-provided workspace, the one synchronising pair of entry points) or from
// ss_tune_import (a table measured earlier, e.g. seedstory/tune_gfx950.json).  Shapes without an entry use the
// closed-form rule below.  GEMM keys bucket M to a multiple of 128 (LLaMA prefill M is the prompt length).
struct TuneKey {
    int32_t v[8];   // dtype, M', N, K, conv_Cin, conv_stride*2+up, conv_H, conv_W
    bool operator<(const TuneKey& o) const {
        for (int i = 0; i < 8; ++i) if (v[i] != o.v[i]) return v[i] < o.v[i];
        return false;
    }
};
static std::mutex g_tune_mutex;
static std::map<TuneKey, int>& tune_cache() {
    static std::map<TuneKey, int> m;
    return m;
}
static TuneKey make_key(int dtype, const GemmArgs& g) {
    const bool conv = g.conv_Cin > 0;
    const int Mk = conv ? g.M : (g.M + 127) / 128 * 128;
    return TuneKey{{dtype, Mk, g.N, g.K, g.conv_Cin, conv ? g.conv_stride * 2 + g.conv_up : 0, conv ? g.conv_H : 0,
                    conv ? g.conv_W : 0}};
}

// Closed-form choice (measured on MI355X, profiles/round2_gemm_tiles.json): M <= 128 is weight streaming
// (narrow-N register-staged tiles); otherwise the largest tile that still yields ~one block per CU, 160-wide
// where N is a multiple of 160 (every SDXL width), 256-row tiles once there are >= 256 of them.
static int pick_cfg(const GemmArgs& g, bool f32) {
    const int force = tuning_get("gemm_cfg", 0);
    if (force) return force;
    const int64_t M = g.M, N = g.N;
    if (M <= 128) return 3;
    if (f32) return 1;
    const bool sp_ok = (g.K % 64 == 0) && (g.conv_Cin == 0 || g.conv_Cin % 64 == 0);
    const int64_t t256 = (int64_t)cdiv(M, 256), t128 = (int64_t)cdiv(M, 128);
    if (sp_ok) {   // LDS-staged-epilogue configurations (60-72); sustained-mode measurements, profiles/round2_gemm_insitu.json
        if (N % 160 == 0 && N < 2560) {
            if (t256 * (N / 160) >= 200) return 62;                 // 256x160, 3-slot ring
            if (t128 * (N / 160) >= 200) return 61;                 // 128x160, two workgroups per CU
            return 67;
        }
        if (t256 * cdiv(N, 256) >= 200) return g.conv_Cin > 0 ? 69 : 60;
        if (N % 160 == 0 && t128 * (N / 160) >= 200) return 61;
        if (t128 * cdiv(N, 128) >= 256) return 65;
        return t128 * cdiv(N, 64) >= 256 ? 68 : 70;
    }
    return t128 * cdiv(N, 128) >= 256 ? 8 : 10;
}

template <typename T>
static int lookup_cfg(GemmArgs& g) {
    const int fallback = pick_cfg(g, Tr<T>::kVec != 8);
    if (Tr<T>::kVec != 8 || g.M <= 128 || tuning_get("gemm_cfg", 0) || !tuning_get("gemm_table", 1)) return fallback;
    const TuneKey key = make_key(Tr<T>::kDtype, g);
    std::lock_guard<std::mutex> lk(g_tune_mutex);
    auto it = tune_cache().find(key);
    if (it == tune_cache().end()) return fallback;
    g.swz = it->second / 100;
    return it->second % 100;
}

// pseudo-random fill of the tuning operands: the chip clocks by its power budget, so candidates must be timed on
// data that toggles like real activations (zero-filled operands run ~20 % faster and mis-rank tiles)
__global__ void tune_fill_kernel(uint16_t* p, size_t n, uint32_t seed, int is_bf16, float scale) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        uint32_t x = (uint32_t)i * 2654435761u + seed;
        x ^= x >> 15; x *= 2246822519u; x ^= x >> 13; x *= 3266489917u; x ^= x >> 16;
        const float f = ((float)(x >> 8) * (1.0f / 8388608.0f) - 1.0f) * scale;   // [-scale, scale)
        p[i] = is_bf16 ? (uint16_t)f32_to_bf16_bits(f) : (uint16_t)f32_to_f16_bits(f);
    }
}

// rotating weight copies of the sustained-mode tuner: enough to cycle through MORE than the 256 MB Infinity Cache
// (up to 128 copies / 352 MB), never fewer than 2
static size_t tune_rot_bytes(size_t w_bytes) {
    const size_t wb = (w_bytes + 255) / 256 * 256;
    size_t n = ((size_t)352 << 20) / wb;
    if (n > 128) n = 128;
    if (n < 2) n = 2;
    return n * wb;
}

static const int kTuneCands[] = {8, 15, 10, 22, 54, 55, 56, 57, 58, 60, 61, 62, 63, 64, 65, 66, 67, 68, 69, 70, 71, 72};

// Candidates are timed in SUSTAINED mode: back-to-back launches over rotating weight copies (every UNet weight is
// touched once per forward: the copies cycle through more than the 256 MB Infinity Cache when the weight allows it), one
// HIP-event pair around the whole run.  A single cold launch after an idle gap runs at boost clocks and with an empty
// memory system and mis-ranks tiles by up to 20 % against what the same kernel does inside a forward
// (tools/gemm_insitu.py); the activation operand stays warm, as it is behind the kernel that produced it.
template <typename T>
static int tune_shape(GemmArgs g, void* ws, size_t ws_bytes, size_t a_elems, hipStream_t s, float* best_us) {
    if constexpr (Tr<T>::kVec != 8) {
        return SS_OK;   // fp32 mode has one kernel
    } else {
        const size_t e = sizeof(T);
        auto up = [](size_t b) { return (b + 255) / 256 * 256; };
        const size_t a_b = up(a_elems * e), w_b = up((size_t)g.N * g.K * e), c_b = up((size_t)g.M * g.N * e);
        SS_REQUIRE(ws && ws_bytes >= a_b + 2 * w_b + c_b, "tune: workspace too small (%zu bytes)", ws_bytes);
        int nW = (int)((ws_bytes - a_b - c_b) / w_b);
        if (nW > 128) nW = 128;
        char* p = (char*)ws;
        g.A = p; g.C = p + a_b;
        char* w0 = p + a_b + c_b;
        g.bias = nullptr; g.residual = nullptr; g.rowvec = nullptr;
        g.epi &= SS_EPI_GEGLU_PAIR | SS_EPI_GELU;
        if (g.epi & SS_EPI_GEGLU_PAIR) g.ldc = g.N / 2;
        hipLaunchKernelGGL(tune_fill_kernel, dim3(2048), dim3(256), 0, s, (uint16_t*)g.A, a_elems, 0x1234u,
                           Tr<T>::kDtype == SS_BF16, 1.0f);
        hipLaunchKernelGGL(tune_fill_kernel, dim3(2048), dim3(256), 0, s, (uint16_t*)w0, (size_t)nW * (w_b / e), 0x9876u,
                           Tr<T>::kDtype == SS_BF16, 0.05f);
        SS_LAUNCH_CHECK("tune_fill");
        hipEvent_t e0, e1;
        SS_HIP(hipEventCreate(&e0));
        SS_HIP(hipEventCreate(&e1));
        const int swzs[3] = {0, 4, 8};
        int best = -1, best_swz = 0;
        float best_ms = 1e30f;
        const bool log = tuning_get("gemm_autotune_log", 0) != 0;
        const int R = nW + nW / 2 > 12 ? nW + nW / 2 : 12;
        for (int c : kTuneCands) {
            for (int z : swzs) {
                g.swz = z;
                g.W = w0;
                if (c >= 50 && c < 60) {   // ping-pong tiles take whole-tile / stride-1 shapes only: time the tile itself, never its fallback
                    const int rc = g.conv_Cin > 0 ? gemm_pp_dispatch_conv<T>(c, g, s) : gemm_pp_dispatch<T>(c, g, s);
                    if (rc != SS_OK) break;
                }
                if (gemm_dispatch_cfg<T>(c, g, s) != SS_OK) continue;   // warm-up (also faults pages in)
                g.W = w0 + w_b; gemm_dispatch_cfg<T>(c, g, s);
                hipEventRecord(e0, s);
                for (int r = 0; r < R; ++r) {
                    g.W = w0 + (size_t)(r % nW) * w_b;
                    gemm_dispatch_cfg<T>(c, g, s);
                }
                hipEventRecord(e1, s);
                float ms = 0.f;
                if (hipEventSynchronize(e1) != hipSuccess) continue;
                hipEventElapsedTime(&ms, e0, e1);
                ms /= R;
                if (log) fprintf(stderr, "[ss tune]   cfg %2d swz %d: %.1f us\n", c, z, ms * 1e3f);
                if (ms < best_ms) { best_ms = ms; best = c; best_swz = z; }
            }
        }
        hipEventDestroy(e0);
        hipEventDestroy(e1);
        SS_REQUIRE(best >= 0, "tune: no candidate ran");
        if (log)
            fprintf(stderr, "[ss tune] M=%d N=%d K=%d conv=%d(%dx%d s%d u%d) epi=%d -> cfg %d swz %d (%.1f us, %.0f TFLOP/s, %d weight copies)\n",
                    g.M, g.N, g.K, g.conv_Cin, g.conv_H, g.conv_W, g.conv_stride, g.conv_up, g.epi, best, best_swz,
                    best_ms * 1e3f, 2.0 * g.M * g.N * g.K / (best_ms * 1e-3) / 1e12, nW);
        if (best_us) *best_us = best_ms * 1e3f;
        std::lock_guard<std::mutex> lk(g_tune_mutex);
        tune_cache()[make_key(Tr<T>::kDtype, g)] = best + 100 * best_swz;
        return SS_OK;
    }
}

static void conv_geometry(GemmArgs& g, int64_t B, int64_t H, int64_t Wd, int64_t Cin, int64_t Cout, int64_t stride,
                          int64_t up) {
    const int64_t Hin = up ? 2 * H : H, Win = up ? 2 * Wd : Wd;
    const int64_t Ho = (Hin + 2 - 3) / stride + 1, Wo = (Win + 2 - 3) / stride + 1;
    g.M = (int)(B * Ho * Wo); g.N = (int)Cout; g.K = (int)(9 * Cin);
    g.lda = 0; g.ldw = 9 * Cin; g.ldc = Cout; g.ldr = Cout;
    g.rows_per_batch = (int)(Ho * Wo);
    g.conv_H = (int)H; g.conv_W = (int)Wd; g.conv_Cin = (int)Cin; g.conv_stride = (int)stride; g.conv_up = (int)up;
    g.conv_Ho = (int)Ho; g.conv_Wo = (int)Wo;
}

// tile of a statistics-producing GEMM (ss_gemm_rowstat / ss_gemm_rowpart): the shape's own tile when it has a statistics
// instantiation, else the 160- / 128-wide staged tile; *bn = tile width, *tn = one wave's column strip
template <typename T>
static int rowstat_cfg(GemmArgs& g, int* bn, int* tn) {
    int cfg = lookup_cfg<T>(g);
    const bool has = cfg == 61 || cfg == 62 || cfg == 63 || cfg == 64 || cfg == 65 || cfg == 67 || cfg == 71 || cfg == 72;
    if (!has) cfg = g.N % 160 == 0 ? (g.M >= 2048 ? 62 : 61) : 65;
    *bn = (cfg == 63 || cfg == 72) ? 320 : cfg == 65 ? 128 : 160;
    *tn = cfg == 65 ? 64 : 80;
    return cfg;
}

// Fallback producer of the RSTAT / rowpart statistics for shapes no statistics tile takes: one wave per stored row of C.
//   stat != null:  stat[m] += (sum, sum of squares) of the row            (the accumulator form, fp64)
//   part != null:  part[(m * part_ld + strip) * 2 ..] = the strip's sums    (strips of `tn` columns, fp32 like the epilogue's)
template <typename T>
__global__ __launch_bounds__(256) void rowstat_fallback_kernel(const T* __restrict__ C, int64_t ldc, int M, int N, double* __restrict__ stat,
                                                               float* __restrict__ part, int part_ld, int tn) {
    const int m = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (m >= M) return;
    const T* row = C + (int64_t)m * ldc;
    double t1 = 0.0, t2 = 0.0;
    for (int n0 = 0, strip = 0; n0 < N; n0 += tn, ++strip) {
        float s1 = 0.f, s2 = 0.f;
        const int n1 = n0 + tn < N ? n0 + tn : N;
        for (int n = n0 + lane; n < n1; n += 64) {
            const float v = Tr<T>::ld(row + n);
            s1 += v;
            s2 = fmaf(v, v, s2);
        }
        s1 = wave_sum(s1);
        s2 = wave_sum(s2);
        if (part && lane == 0 && strip < part_ld) {
            part[((int64_t)m * part_ld + strip) * 2] = s1;
            part[((int64_t)m * part_ld + strip) * 2 + 1] = s2;
        }
        t1 += (double)s1;
        t2 += (double)s2;
    }
    if (stat && lane == 0) {
        stat[(int64_t)m * 2] += t1;
        stat[(int64_t)m * 2 + 1] += t2;
    }
}

template <typename T>
int gemm_launch(const void* A, const void* W, void* C, int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldw,
                int64_t ldc, const void* bias, const void* residual, int64_t ldr, int epi, hipStream_t s,
                double* rowstat_out = nullptr, float* rowpart = nullptr) {
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
    if (tuning_get("gemm_epi_generic", 0)) g.epi |= SS_EPI_INTERNAL_GENERIC;
    if (rowstat_out || rowpart) {
        // statistics epilogue: only the staged software-pipelined tiles 61..72 have it (ids + 200); 256x256 tiles (60 / 69)
        // and non-staged choices fall to the 160- / 128-wide staged tile of the shape
        if constexpr (Tr<T>::kVec != 8) {
            set_error("ss_gemm_rowstat: 16-bit dtypes only");
            return SS_EINVAL;
        } else {
            SS_REQUIRE(K % 64 == 0 && N % 8 == 0 && M > 128, "ss_gemm_rowstat: needs K %% 64 == 0, N %% 8 == 0, M > 128 (M=%lld N=%lld K=%lld)",
                       (long long)M, (long long)N, (long long)K);
            g.rowstat_out = rowstat_out;
            int bn = 0, tn = 0;
            const int cfg = rowstat_cfg<T>(g, &bn, &tn);
            if (rowpart) {      // partial sums per wave column strip: needs the staged epilogue on every wave
                constexpr size_t A16 = 15;
                SS_REQUIRE(N % bn == 0 && (((size_t)C | (size_t)residual | (size_t)bias) & A16) == 0 && ((ldc | ldr) & 7) == 0,
                           "ss_gemm_rowpart: N %% %d != 0 or operands not 16-byte aligned (N=%lld)", bn, (long long)N);
                g.rowpart = rowpart;
                g.rowpart_ld = (int)(N / tn);
            }
            const bool force_fb = tuning_get("gemm_rowstat_fallback", 0) != 0;      // (tests)
            int rc = force_fb ? 1 : gemm_sp_dispatch<T>(cfg + 200, g, s);
            if (rc == 1 && !rowpart && !force_fb) rc = gemm_sp_dispatch<T>(265, g, s);       // the 128x128 staged tile takes more shapes
            if (rc == 1) {
                // no statistics tile for this shape / alignment (a choice ss_gemm itself would have run on the double-buffered
                // kernel): the plain GEMM, then one pass over the stored rows that produces the same sums
                g.rowstat_out = nullptr;
                g.rowpart = nullptr;
                rc = gemm_dispatch_cfg<T>(lookup_cfg<T>(g), g, s);
                if (rc) return rc;
                hipLaunchKernelGGL(rowstat_fallback_kernel<T>, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, s, (const T*)C, ldc, (int)M,
                                   (int)N, rowstat_out, rowpart, rowpart ? (int)(N / tn) : 0, tn > 0 ? tn : (int)N);
                SS_LAUNCH_CHECK("rowstat_fallback");
                return SS_OK;
            }
            return rc;
        }
    }
    return gemm_dispatch_cfg<T>(lookup_cfg<T>(g), g, s);
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
    GemmArgs g;
    conv_geometry(g, B, H, Wd, Cin, Cout, stride, up);
    g.A = x; g.W = w; g.C = y; g.bias = bias; g.residual = residual; g.epi = epi;
    g.rowvec = rowvec; g.rowvec_ld = rowvec_ld > 0 ? rowvec_ld : Cout;
    g.swz = tuning_get("gemm_xcd_swizzle", 8);
    if (g.M == 0) return SS_OK;
    return gemm_dispatch_cfg<T>(lookup_cfg<T>(g), g, s);
}

template <typename T>
int gemm_tune_launch(int64_t M, int64_t N, int64_t K, int epi, void* ws, size_t ws_bytes, hipStream_t s, float* us) {
    GemmArgs g;
    g.M = (int)((M + 127) / 128 * 128); g.N = (int)N; g.K = (int)K; g.lda = K; g.ldw = K; g.ldc = N; g.ldr = N; g.epi = epi;
    g.rowvec = nullptr; g.rows_per_batch = 1; g.rowvec_ld = 0;
    g.conv_H = g.conv_W = g.conv_Cin = g.conv_stride = g.conv_up = g.conv_Ho = g.conv_Wo = 0;
    return tune_shape<T>(g, ws, ws_bytes, (size_t)g.M * K, s, us);
}
template <typename T>
int conv_tune_launch(int64_t B, int64_t H, int64_t Wd, int64_t Cin, int64_t Cout, int64_t stride, int64_t up, void* ws,
                     size_t ws_bytes, hipStream_t s, float* us) {
    GemmArgs g;
    conv_geometry(g, B, H, Wd, Cin, Cout, stride, up);
    g.epi = 0; g.rowvec = nullptr; g.rowvec_ld = 0;
    return tune_shape<T>(g, ws, ws_bytes, (size_t)B * H * Wd * Cin, s, us);
}

// GEMM with a LayerNorm folded into it (see ss_gemm_lnfold in the header).  Runs the shape's own tile (table / rule)
// in its folded-epilogue instantiation (id + 100: the staged family 60..72).
template <typename T>
int gemm_lnfold_launch(const void* A, const void* Wg, void* C, int64_t M, int64_t N, int64_t K, int64_t ldc, const float* rstd,
                       const float* shift, const float* colsum, const void* bias, int epi, hipStream_t s,
                       const float* ln_part = nullptr, int64_t ln_nstrip = 0, int64_t ln_width = 0, float ln_eps = 0.f) {
    if constexpr (Tr<T>::kVec != 8) {
        set_error("ss_gemm_lnfold: 16-bit dtypes only");
        return SS_EINVAL;
    } else {
        SS_REQUIRE(K % 64 == 0 && N % 16 == 0, "ss_gemm_lnfold: K %% 64 and N %% 16 must be 0 (K=%lld N=%lld)", (long long)K, (long long)N);
        SS_REQUIRE(!(epi & ~(SS_EPI_BIAS | SS_EPI_GELU | SS_EPI_GEGLU_PAIR)), "ss_gemm_lnfold: unsupported epilogue %d", epi);
        SS_REQUIRE(!(epi & SS_EPI_BIAS) || bias, "ss_gemm_lnfold: bias epilogue without bias");
        SS_REQUIRE(((rstd && shift) || (ln_part && ln_nstrip > 0 && ln_width > 0)) && colsum && (((size_t)colsum) & 15) == 0 &&
                   (((size_t)ln_part) & 7) == 0, "ss_gemm_lnfold: row statistics / column vector missing or misaligned");
        if (M == 0 || N == 0) return SS_OK;
        GemmArgs g;
        g.A = A; g.W = Wg; g.C = C; g.bias = bias; g.residual = nullptr;
        g.M = (int)M; g.N = (int)N; g.K = (int)K; g.lda = K; g.ldw = K; g.ldc = ldc; g.ldr = 0; g.epi = epi;
        g.rowvec = nullptr; g.rows_per_batch = 1; g.rowvec_ld = 0;
        g.conv_H = g.conv_W = g.conv_Cin = g.conv_stride = g.conv_up = g.conv_Ho = g.conv_Wo = 0;
        g.swz = tuning_get("gemm_xcd_swizzle", 8);
        g.scale_a = rstd; g.shift_a = shift; g.scale_w = colsum;
        g.ln_part = ln_part; g.ln_nstrip = (int)ln_nstrip; g.ln_inv_width = ln_width > 0 ? 1.0f / (float)ln_width : 0.f; g.ln_eps = ln_eps;
        int cfg = lookup_cfg<T>(g);
        if (cfg < 60 || cfg > 72) cfg = N % 160 == 0 ? (M >= 2048 ? 62 : 61) : (M * N >= 128 * 128 * 256 ? 65 : 70);
        // The folded epilogue runs on the 8-wave tiles only.  On the 4-wave tiles (61 / 65 / 67 / 70: two workgroups per CU) it
        // sporadically returns ONE wrong element per 16-row strip of the last fragment column of a wave.  Round 5 re-ran
        // tools/dbg_lnfold_vec.py WITH the accumulator fences behind the K loop (ss_gemm_sp.inc, "MFMA results settle before
        // VALU reads them") and `lnfold_w4` = 1 (profiles/round5_lnfold_w4_recheck.txt): still 80 / 96 / 96 / 304 bad rows per
        // 4 launches at [32768, 640, 640] on cfg 61 / 65 / 67 / 70, none on 62 / 68 / 60 — the fences do NOT fix it, the cause is
        // still open, the reroute stays.  The same run cleared the PRODUCER side: the rowpart statistics epilogue on the 4-wave
        // tiles 261 / 265 / 267 (and 262) gave 0 bad statistics rows and 0 bad value rows, and the plain epilogue on cfg 61 none.
        if (!tuning_get("lnfold_w4", 0)) {
            if (cfg == 61 || cfg == 67) cfg = 62;
            else if (cfg == 65 || cfg == 68 || cfg == 70) cfg = N % 160 == 0 ? 62 : 66;
        }
        const int rc = gemm_sp_dispatch<T>(cfg + 100, g, s);
        if (rc == 1) {
            set_error("ss_gemm_lnfold: no kernel for cfg %d / shape [%lld, %lld, %lld]", cfg + 100, (long long)M, (long long)N, (long long)K);
            return SS_EINVAL;
        }
        return rc;
    }
}

// per-row LayerNorm statistics in the form the folded epilogue consumes: rstd[m] and -mean[m] * rstd[m]; one wave per row,
// two-pass variance on the register-resident row (the arithmetic of layernorm_wave_kernel)
template <typename T>
__global__ __launch_bounds__(256) void rowstats_kernel(const T* __restrict__ x, int64_t ld, int rows, int cols, float eps,
                                                       float* __restrict__ rstd, float* __restrict__ shift) {
    constexpr int V = Tr<T>::kVec;
    constexpr int MAXP = 4;
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int npack = cols / V;
    const T* xr = x + (int64_t)row * ld;
    uint4 px[MAXP];
    float s1 = 0.f;
#pragma unroll
    for (int i = 0; i < MAXP; ++i) {
        const int p = lane + i * 64;
        if (p < npack) {
            px[i] = *reinterpret_cast<const uint4*>(xr + (int64_t)p * V);
            float f[V];
            unpack<T>(px[i], f);
#pragma unroll
            for (int j = 0; j < V; ++j) s1 += f[j];
        }
    }
    const float mean = wave_sum(s1) / (float)cols;
    float s2 = 0.f;
#pragma unroll
    for (int i = 0; i < MAXP; ++i) {
        const int p = lane + i * 64;
        if (p < npack) {
            float f[V];
            unpack<T>(px[i], f);
#pragma unroll
            for (int j = 0; j < V; ++j) { const float d = f[j] - mean; s2 = fmaf(d, d, s2); }
        }
    }
    const float r = 1.0f / sqrtf(wave_sum(s2) / (float)cols + eps);
    if (lane == 0) { rstd[row] = r; shift[row] = -mean * r; }
}

// (sum, sum of squares) accumulated by a producer GEMM (RSTAT epilogue) -> the folded LayerNorm's row vectors; the
// accumulator is re-zeroed for its next producer.  mean / variance in fp64: E[x^2] - mean^2 does not cancel there.
__global__ __launch_bounds__(256) void rowstat_finalize_kernel(double* __restrict__ stat, int rows, double inv_width, float eps,
                                                               float* __restrict__ rstd, float* __restrict__ shift) {
    const int m = blockIdx.x * 256 + threadIdx.x;
    if (m >= rows) return;
    double2* p = reinterpret_cast<double2*>(stat) + m;
    const double2 st = *p;
    *p = make_double2(0.0, 0.0);
    const double mean = st.x * inv_width;
    const double var = fmax(st.y * inv_width - mean * mean, 0.0);
    const float r = (float)(1.0 / sqrt(var + (double)eps));
    rstd[m] = r;
    shift[m] = -(float)mean * r;
}

template <typename T>
int rowstats_launch(const void* x, int64_t ld, int64_t M, int64_t K, float eps, float* rstd, float* shift, hipStream_t s) {
    if constexpr (Tr<T>::kVec != 8) {
        set_error("ss_rowstats: 16-bit dtypes only");
        return SS_EINVAL;
    } else {
        SS_REQUIRE(K % 8 == 0 && K / 8 <= 256 && ld % 8 == 0, "ss_rowstats: K=%lld unsupported (multiple of 8, <= 2048)", (long long)K);
        hipLaunchKernelGGL(rowstats_kernel<T>, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, s, (const T*)x, ld, (int)M, (int)K, eps, rstd, shift);
        SS_LAUNCH_CHECK("rowstats");
        return SS_OK;
    }
}

// ---- small-M weight-streaming GEMM: split-K over the pipelined tiles --------------------------------------------------
// 128 < M <= 512 against a LLaMA projection (N x K = 4096..22016 x 4096..11008) is neither a GEMV nor a full GEMM: three
// or four 128-row tiles cover M, so a [M, 4096, 11008] product is 96 workgroups walking 172 K tiles each (138 us, a third
// of the CUs idle, latency-bound K loop).  Splitting K S ways gives every CU two workgroups with short K loops; the fp32
// partial sums (S x M x N, a few MB) are folded by splitk_reduce_kernel, which applies the usual epilogue
// (bias, rounding to T, residual) in the order of gemm_epilogue.
template <typename T>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ P, int S, int M, int N, T* __restrict__ C,
                                                            int64_t ldc, const T* __restrict__ bias, const T* __restrict__ res,
                                                            int64_t ldr) {
    const int64_t idx = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (idx >= (int64_t)M * N) return;
    const int m = (int)(idx / N), n = (int)(idx - (int64_t)m * N);     // N % 4 == 0: the 4 elements share a row
    float4 a = *reinterpret_cast<const float4*>(P + idx);
    for (int k = 1; k < S; ++k) {
        const float4 b = *reinterpret_cast<const float4*>(P + (int64_t)k * M * N + idx);
        a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    }
    float v[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        if (bias) v[r] += Tr<T>::ld(bias + n + r);
        v[r] = Tr<T>::rnd(v[r]);
        if (res) v[r] += Tr<T>::ld(res + (int64_t)m * ldr + n + r);
        Tr<T>::st(C + (int64_t)m * ldc + n + r, v[r]);
    }
}

// (tile, split count) of the split-K path for [M, N, K]; 0 splits = not eligible
static void splitk_plan(int64_t M, int64_t N, int64_t K, int* cfg, int* S) {
    *cfg = 0; *S = 0;
    if (M <= 128 || M > 512 || N % 64 || K % 64 || K < 1024) return;
    const int mt = cdiv(M, 128);
    // measured (profiles/round3_splitk.txt): with >= 256 workgroups of 128x128 already ([M, 12288, 4096]: 288 - 384) splitting
    // K gains nothing or loses (61 vs 63 us at 264 rows, 74 vs 64 at 460); it pays where the column count leaves most CUs
    // idle (N = 4096: 96 - 128 workgroups): 128x64 tiles x 2 - 3 K ranges
    if (mt * cdiv(N, 128) >= 256) return;
    int c = 301, base = mt * cdiv(N, 64);
    int s = cdiv(2 * 256, base);          // two workgroups per CU of the 256
    const int kt = (int)(K / 64);
    if (s > kt / 8) s = kt / 8;            // >= 8 K tiles per split: the pipeline needs a few tiles to fill
    if (s > 8) s = 8;
    if (s < 2) return;                     // enough workgroups without splitting: the regular path
    while (s > 2 && (s - 1) * cdiv(kt, s) >= kt) --s;     // no empty last range
    *cfg = c; *S = s;
}

size_t gemm_splitk_workspace_bytes(int64_t M, int64_t N, int64_t K) {
    int cfg, S;
    splitk_plan(M, N, K, &cfg, &S);
    return S ? (size_t)S * M * N * sizeof(float) : 0;
}

// C = A W^T (+bias)(+residual) through the split-K tiles when the shape is eligible and the workspace suffices, else ss_gemm
template <typename T>
int gemm_splitk_launch(const void* A, const void* W, void* C, int64_t M, int64_t N, int64_t K, const void* bias,
                       const void* residual, void* ws, size_t ws_bytes, hipStream_t s) {
    int cfg = 0, S = 0;
    if constexpr (Tr<T>::kVec == 8) {
        if (tuning_get("gemm_splitk", 1)) splitk_plan(M, N, K, &cfg, &S);
    }
    {
        const int force = tuning_get("gemm_splitk_s", 0);
        if (S && force >= 2 && force <= 8 && (size_t)force * M * N * sizeof(float) > ws_bytes) S = 0;   // forced count needs the room
    }
    if (!S || !ws || ws_bytes < (size_t)S * M * N * sizeof(float))
        return gemm_launch<T>(A, W, C, M, N, K, K, K, N, bias, residual, N, (bias ? SS_EPI_BIAS : 0) | (residual ? SS_EPI_RESIDUAL : 0), s);
    if constexpr (Tr<T>::kVec == 8) {
        GemmArgs g;
        g.A = A; g.W = W; g.C = ws; g.bias = nullptr; g.residual = nullptr;
        g.M = (int)M; g.N = (int)N; g.K = (int)K; g.lda = K; g.ldw = K; g.ldc = N; g.ldr = N; g.epi = 0;
        g.rowvec = nullptr; g.rows_per_batch = 1; g.rowvec_ld = 0;
        g.conv_H = g.conv_W = g.conv_Cin = g.conv_stride = g.conv_up = g.conv_Ho = g.conv_Wo = 0;
        // row-major tile ids (measured, profiles/round3_splitk_microbench.json: the XCD-grouped order that puts the 3 - 4 row
        // tiles of a W column block on one XCD is 5 - 20 % SLOWER here — the row tiles then run back to back on few CUs
        // while the W stream of the other column blocks waits; with row-major ids the repeats hit the Infinity Cache)
        g.swz = tuning_get("gemm_splitk_swz", 0);
        g.ksplit = S;
        const int force = tuning_get("gemm_splitk_s", 0);
        if (force >= 2 && force <= 8 && force <= (int)(K / 64) / 2) g.ksplit = S = force;
        const int rc = gemm_sp_dispatch<T>(cfg, g, s);
        if (rc) {
            if (rc == 1) set_error("gemm_splitk: no kernel for cfg %d", cfg);
            return rc == 1 ? SS_EINVAL : rc;
        }
        const int64_t quads = M * N / 4;
        hipLaunchKernelGGL(splitk_reduce_kernel<T>, dim3((unsigned)((quads + 255) / 256)), dim3(256), 0, s, (const float*)ws, S, (int)M,
                           (int)N, (T*)C, N, (const T*)bias, (const T*)residual, N);
        SS_LAUNCH_CHECK("splitk_reduce");
    }
    return SS_OK;
}

int gemm_splitk_dev(const void* A, const void* W, void* C, int64_t M, int64_t N, int64_t K, const void* bias,
                    const void* residual, void* ws, size_t ws_bytes, int dtype, hipStream_t s) {
    return SS_DISPATCH(dtype, gemm_splitk_launch, A, W, C, M, N, K, bias, residual, ws, ws_bytes, s);
}

int gemm_dev(const void* A, const void* W, void* C, int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldw,
             int64_t ldc, const void* bias, const void* residual, int64_t ldr, int epi, int dtype, hipStream_t s) {
    return SS_DISPATCH(dtype, gemm_launch, A, W, C, M, N, K, lda, ldw, ldc, bias, residual, ldr, epi, s);
}

}  // namespace ss

extern "C" {

int ss_conv3x3(const void* x, const void* w, void* y, int64_t batch, int64_t H, int64_t W_, int64_t Cin,
               int64_t Cout, int64_t stride, int64_t upsample2x, const void* bias, const void* rowvec,
               int64_t rowvec_stride, const void* residual, int dtype, void* stream) {
    const int epi = (bias ? SS_EPI_BIAS : 0) | (residual ? SS_EPI_RESIDUAL : 0);
    return SS_DISPATCH(dtype, ss::conv3x3_launch, x, w, y, batch, H, W_, Cin, Cout, stride, upsample2x, bias, rowvec,
                       rowvec_stride, residual, epi, (hipStream_t)stream);
}

int ss_gemm(const void* A, const void* W, void* C, int64_t M, int64_t N, int64_t K, int64_t lda,
            int64_t ldw, int64_t ldc, const void* bias, const void* residual, int64_t ldr, int epilogue,
            int dtype, void* stream) {
    return ss::gemm_dev(A, W, C, M, N, K, lda, ldw, ldc, bias, residual, ldr, epilogue, dtype, (hipStream_t)stream);
}

int ss_gemm_rowstat(const void* A, const void* W, void* C, int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldw,
                    int64_t ldc, const void* bias, const void* residual, int64_t ldr, int epilogue, double* rowstat_accum,
                    int dtype, void* stream) {
    SS_REQUIRE(rowstat_accum && (((size_t)rowstat_accum) & 15) == 0, "ss_gemm_rowstat: rowstat_accum missing or not 16-byte aligned");
    SS_REQUIRE(!(epilogue & SS_EPI_GEGLU_PAIR), "ss_gemm_rowstat: not defined for the GEGLU epilogue");
    return SS_DISPATCH(dtype, ss::gemm_launch, A, W, C, M, N, K, lda, ldw, ldc, bias, residual, ldr, epilogue,
                       (hipStream_t)stream, rowstat_accum);
}

size_t ss_gemm_splitk_workspace_bytes(int64_t M, int64_t N, int64_t K) { return ss::gemm_splitk_workspace_bytes(M, N, K); }

int ss_gemm_splitk(const void* A, const void* W, void* C, int64_t M, int64_t N, int64_t K, const void* bias, const void* residual,
                   void* workspace, size_t workspace_bytes, int dtype, void* stream) {
    SS_REQUIRE(A && W && C && M > 0 && N > 0 && K > 0, "ss_gemm_splitk: bad arguments");
    return ss::gemm_splitk_dev(A, W, C, M, N, K, bias, residual, workspace, workspace_bytes, dtype, (hipStream_t)stream);
}

int ss_rowstat_finalize(double* rowstat, int64_t M, int64_t width, float eps, float* rstd_out, float* shift_out, void* stream) {
    SS_REQUIRE(rowstat && rstd_out && shift_out && M > 0 && width > 0 && (((size_t)rowstat) & 15) == 0, "ss_rowstat_finalize: bad arguments");
    hipLaunchKernelGGL(ss::rowstat_finalize_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, (hipStream_t)stream, rowstat,
                       (int)M, 1.0 / (double)width, eps, rstd_out, shift_out);
    SS_LAUNCH_CHECK("rowstat_finalize");
    return SS_OK;
}

int ss_rowstats(const void* x, int64_t ld, int64_t M, int64_t K, float eps, float* rstd_out, float* shift_out, int dtype, void* stream) {
    SS_REQUIRE(x && rstd_out && shift_out && M > 0 && K > 0 && ld >= K, "ss_rowstats: bad arguments");
    return SS_DISPATCH(dtype, ss::rowstats_launch, x, ld, M, K, eps, rstd_out, shift_out, (hipStream_t)stream);
}

int64_t ss_gemm_rowpart_strips(int64_t M, int64_t N, int64_t K, int dtype) {
    if (dtype != SS_BF16 && dtype != SS_F16) return 0;
    ss::GemmArgs g;
    g.M = (int)M; g.N = (int)N; g.K = (int)K; g.conv_Cin = 0; g.epi = 0;
    g.conv_H = g.conv_W = g.conv_stride = g.conv_up = g.conv_Ho = g.conv_Wo = 0;
    int bn = 0, tn = 0;
    if (dtype == SS_BF16) ss::rowstat_cfg<ss::bf16_t>(g, &bn, &tn); else ss::rowstat_cfg<ss::f16_t>(g, &bn, &tn);
    return (N % bn == 0) ? N / tn : 0;
}

int ss_gemm_rowpart(const void* A, const void* W, void* C, int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldw,
                    int64_t ldc, const void* bias, const void* residual, int64_t ldr, int epilogue, float* rowpart,
                    int dtype, void* stream) {
    SS_REQUIRE(rowpart && (((size_t)rowpart) & 7) == 0, "ss_gemm_rowpart: rowpart missing or not 8-byte aligned");
    SS_REQUIRE(!(epilogue & SS_EPI_GEGLU_PAIR), "ss_gemm_rowpart: not defined for the GEGLU epilogue");
    return SS_DISPATCH(dtype, ss::gemm_launch, A, W, C, M, N, K, lda, ldw, ldc, bias, residual, ldr, epilogue,
                       (hipStream_t)stream, nullptr, rowpart);
}

int ss_gemm_lnfold_part(const void* A, const void* Wg, void* C, int64_t M, int64_t N, int64_t K, int64_t ldc, const float* rowpart,
                        int64_t strips, int64_t width, float eps, const float* colsum, const void* bias, int epilogue, int dtype,
                        void* stream) {
    SS_REQUIRE(A && Wg && C && rowpart && strips > 0 && width > 0, "ss_gemm_lnfold_part: bad arguments");
    return SS_DISPATCH(dtype, ss::gemm_lnfold_launch, A, Wg, C, M, N, K, ldc, nullptr, nullptr, colsum, bias, epilogue, (hipStream_t)stream,
                       rowpart, strips, width, eps);
}

int ss_gemm_lnfold(const void* A, const void* Wg, void* C, int64_t M, int64_t N, int64_t K, int64_t ldc, const float* rstd,
                   const float* shift, const float* colsum, const void* bias, int epilogue, int dtype, void* stream) {
    SS_REQUIRE(A && Wg && C, "ss_gemm_lnfold: NULL argument");
    return SS_DISPATCH(dtype, ss::gemm_lnfold_launch, A, Wg, C, M, N, K, ldc, rstd, shift, colsum, bias, epilogue, (hipStream_t)stream);
}

size_t ss_gemm_tune_workspace_bytes(int64_t M, int64_t N, int64_t K, int dtype) {
    const size_t e = ss::dtype_size(dtype);
    const size_t Mk = (size_t)((M + 127) / 128 * 128);
    return (Mk * K + Mk * N) * e + ss::tune_rot_bytes((size_t)N * K * e) + 1024;
}

int ss_gemm_tune(int64_t M, int64_t N, int64_t K, int epilogue, int dtype, void* workspace, size_t workspace_bytes,
                 void* stream, float* best_us_host) {
    SS_REQUIRE(M > 0 && N > 0 && K > 0 && K % 8 == 0, "gemm_tune: bad shape");
    return SS_DISPATCH(dtype, ss::gemm_tune_launch, M, N, K, epilogue, workspace, workspace_bytes, (hipStream_t)stream,
                       best_us_host);
}

size_t ss_conv3x3_tune_workspace_bytes(int64_t batch, int64_t H, int64_t W_, int64_t Cin, int64_t Cout, int64_t stride,
                                       int64_t upsample2x, int dtype) {
    const size_t e = ss::dtype_size(dtype);
    const int64_t Hin = upsample2x ? 2 * H : H, Win = upsample2x ? 2 * W_ : W_;
    const int64_t Ho = (Hin + 2 - 3) / stride + 1, Wo = (Win + 2 - 3) / stride + 1;
    return ((size_t)batch * H * W_ * Cin + (size_t)batch * Ho * Wo * Cout) * e + ss::tune_rot_bytes((size_t)Cout * 9 * Cin * e) + 1024;
}

int ss_conv3x3_tune(int64_t batch, int64_t H, int64_t W_, int64_t Cin, int64_t Cout, int64_t stride, int64_t upsample2x,
                    int dtype, void* workspace, size_t workspace_bytes, void* stream, float* best_us_host) {
    SS_REQUIRE(batch > 0 && H > 0 && W_ > 0 && Cin % 8 == 0 && Cout > 0 && (stride == 1 || stride == 2),
               "conv3x3_tune: bad shape");
    return SS_DISPATCH(dtype, ss::conv_tune_launch, batch, H, W_, Cin, Cout, stride, upsample2x, workspace,
                       workspace_bytes, (hipStream_t)stream, best_us_host);
}

int ss_tune_lookup(int64_t M, int64_t N, int64_t K, int64_t conv_Cin, int64_t conv_H, int64_t conv_W, int64_t stride,
                   int64_t upsample2x, int dtype, int32_t out[2]) {
    SS_REQUIRE(out, "tune_lookup: out == NULL");
    ss::GemmArgs g;
    g.M = (int)M; g.N = (int)N; g.K = (int)K; g.conv_Cin = (int)conv_Cin; g.conv_H = (int)conv_H; g.conv_W = (int)conv_W;
    g.conv_stride = (int)stride; g.conv_up = (int)upsample2x;
    const ss::TuneKey key = ss::make_key(dtype, g);
    std::lock_guard<std::mutex> lk(ss::g_tune_mutex);
    auto it = ss::tune_cache().find(key);
    if (it == ss::tune_cache().end()) { out[0] = -1; out[1] = 0; return 1; }
    out[0] = it->second % 100; out[1] = it->second / 100;
    return SS_OK;
}

int64_t ss_tune_export(int32_t* out, int64_t cap_entries) {
    std::lock_guard<std::mutex> lk(ss::g_tune_mutex);
    int64_t n = 0;
    for (auto& kv : ss::tune_cache()) {
        if (out && n < cap_entries) {
            for (int i = 0; i < 8; ++i) out[n * 10 + i] = kv.first.v[i];
            out[n * 10 + 8] = kv.second % 100;
            out[n * 10 + 9] = kv.second / 100;
        }
        ++n;
    }
    return n;   // number of entries in the table (may exceed cap_entries)
}

int ss_tune_import(const int32_t* in, int64_t n_entries) {
    SS_REQUIRE(in || n_entries == 0, "tune_import: NULL table");
    std::lock_guard<std::mutex> lk(ss::g_tune_mutex);
    for (int64_t n = 0; n < n_entries; ++n) {
        ss::TuneKey k;
        for (int i = 0; i < 8; ++i) k.v[i] = in[n * 10 + i];
        const int cfg = in[n * 10 + 8], swz = in[n * 10 + 9];
        if (cfg <= 0 || cfg >= 100 || swz < 0 || swz > 64) continue;
        ss::tune_cache()[k] = cfg + 100 * swz;
    }
    return SS_OK;
}

int ss_tune_clear(void) {
    std::lock_guard<std::mutex> lk(ss::g_tune_mutex);
    ss::tune_cache().clear();
    return SS_OK;
}

}  // extern "C"
