// Shared pieces of the MFMA GEMM / implicit-GEMM conv kernels (ss_gemm.hip, ss_gemm_sp_*.hip):
// argument block, MFMA wrappers, the XCD-aware tile order, LDS-DMA helpers and the fused epilogue.
//
// C[M,N] = A[M,K] · W[N,K]^T (+bias)(+GELU)(+residual): every nn.Linear on the hot path has this shape with
// BOTH operands K-contiguous (modeling_llama_xformer.py:228-230,297,191; qwen_visual.py:191,196,259;
// resampler.py; the diffusers UNet's Linear / Conv2d layers), which is exactly the MFMA fragment shape:
// lane l of a wave supplies 8 consecutive k of row (l & 15) for k-group (l >> 4).
// Operand roles are swapped w.r.t. the math so that stores vectorise: the MFMA "A" operand is the WEIGHT
// tile (rows -> n) and the "B" operand the ACTIVATION tile (cols -> m), so a lane ends up with
// C[m = l&15][n = 4*(l>>4) .. +3]: four consecutive n per row.
#pragma once
#include <type_traits>
#include <utility>

#include "ss_common.h"

namespace ss {

typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_t;
typedef __attribute__((ext_vector_type(8))) uint32_t u32x8_t;
// One lane's MFMA operand: NV 16-byte pieces (1 for the 16-bit 16x16x32 forms, 2 for the 128-deep fp8 form)
template <int NV> struct Frag { uint4 v[NV]; };
// the type results are written in (fp8 is an operand format only: activations stay bf16 between kernels)
template <typename T> struct OutT { using type = T; };
template <> struct OutT<fp8_t> { using type = bf16_t; };
template <typename T> struct Mma;
template <> struct Mma<bf16_t> {
    static constexpr int kK = 32;
    static constexpr int kEB = 2;   // bytes per element
    static constexpr int kNV = 1;
    static __device__ __forceinline__ f32x4_t run(const uint4& a, const uint4& b, f32x4_t c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a),
                                                       __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
    }
    // in-place form with the accumulator pinned to its VGPRs (see gemm_sp_kernel)
    static __device__ __forceinline__ void run_inplace(const uint4& a, const uint4& b, f32x4_t& c) {
        asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0"
                     : "+v"(c)
                     : "v"(__builtin_bit_cast(u32x4_t, a)), "v"(__builtin_bit_cast(u32x4_t, b)));
    }
    static __device__ __forceinline__ uint32_t make_aux() { return 0u; }
    static __device__ __forceinline__ void run_inplace(const Frag<1>& a, const Frag<1>& b, f32x4_t& c, uint32_t) { run_inplace(a.v[0], b.v[0], c); }
};
template <> struct Mma<f16_t> {
    static constexpr int kK = 32;
    static constexpr int kEB = 2;
    static constexpr int kNV = 1;
    static __device__ __forceinline__ f32x4_t run(const uint4& a, const uint4& b, f32x4_t c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a),
                                                      __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
    }
    static __device__ __forceinline__ void run_inplace(const uint4& a, const uint4& b, f32x4_t& c) {
        asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0"
                     : "+v"(c)
                     : "v"(__builtin_bit_cast(u32x4_t, a)), "v"(__builtin_bit_cast(u32x4_t, b)));
    }
    static __device__ __forceinline__ uint32_t make_aux() { return 0u; }
    static __device__ __forceinline__ void run_inplace(const Frag<1>& a, const Frag<1>& b, f32x4_t& c, uint32_t) { run_inplace(a.v[0], b.v[0], c); }
};

// OCP e4m3 x e4m3 -> fp32 through the block-scaled 128-deep instruction (the only fp8 MFMA that runs at twice the bf16
// rate on gfx950; the 16x16x32 fp8 form runs at the bf16 rate).  Lane l supplies 32 consecutive k (bytes) of row l & 15
// for k-block l >> 4; the hardware block scales (one E8M0 per lane = per 32 k) are held at 2^0 — tensors are scaled per
// row / per output channel in fp32 in the epilogue instead, which keeps the quantisation grid independent of K.
template <> struct Mma<fp8_t> {
    static constexpr int kK = 128;
    static constexpr int kEB = 1;
    static constexpr int kNV = 2;
    // The block-scale operand (E8M0 1.0 = 127 in byte 0, op_sel 0) must sit in a VGPR written LONG before the MFMA:
    // handed to the asm as a plain constant, the compiler re-materialises it (v_mov) right in front of an asm MFMA,
    // which is opaque to the hazard recognizer — the instruction then reads the scale register before the v_mov has
    // landed and multiplies by whatever 2^x the register held (seen as a rare, run-dependent corruption of whole tiles).
    // make_aux() produces it through an asm the compiler cannot look into, once per kernel.
    static __device__ __forceinline__ uint32_t make_aux() {
        uint32_t one;
        asm volatile("v_mov_b32 %0, 0x7f\n\ts_nop 4" : "=v"(one));
        return one;
    }
    static __device__ __forceinline__ void run_inplace(const Frag<2>& a, const Frag<2>& b, f32x4_t& c, uint32_t one) {
        u32x8_t av = {a.v[0].x, a.v[0].y, a.v[0].z, a.v[0].w, a.v[1].x, a.v[1].y, a.v[1].z, a.v[1].w};
        u32x8_t bv = {b.v[0].x, b.v[0].y, b.v[0].z, b.v[0].w, b.v[1].x, b.v[1].y, b.v[1].z, b.v[1].w};
        asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %3 op_sel_hi:[0,0,0]"
                     : "+v"(c)
                     : "v"(av), "v"(bv), "v"(one));
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

// internal epilogue bit (never part of the C ABI's SS_EPI_* set): take the run-time-flag epilogue variant instead of the
// compile-time one — tuning knob "gemm_epi_generic", for A/B measurements of the specialisation
constexpr int SS_EPI_INTERNAL_GENERIC = 1 << 30;

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
    // fp8 operands: fp32 de-quantisation scales, one per A row (token) and one per W row (output channel)
    const float* scale_a = nullptr; const float* scale_w = nullptr;
    // folded LayerNorm (16-bit operands): scale_a = rstd[m], shift_a = -mean[m] * rstd[m], scale_w = c[n] (see EM below)
    const float* shift_a = nullptr;
    // row statistics for the folded LayerNorm of the CONSUMER, accumulated by the PRODUCER of the tensor (kernels
    // instantiated with RSTAT): the epilogue adds  sum_n out[m][n]  and  sum_n out[m][n]^2  of the values it stores
    // (rounded to T, residual included) into rowstat_out[2m], rowstat_out[2m+1] — fp64 atomics, one pair per row per
    // column strip of a wave; ss_rowstat_finalize turns them into (rstd, shift) and re-zeroes the array.
    double* rowstat_out = nullptr;
    // round 4: the same statistics WITHOUT atomics — every wave column strip of the producer writes its (sum, sum of
    // squares) of row m to rowpart[(m * rowpart_ld + strip) * 2 ..+1] (strip = first column of the wave's sub-tile / its width:
    // every (row, strip) entry is written exactly once per launch — nothing to zero, nothing order-dependent), and the
    // CONSUMER's folded-LayerNorm epilogue sums the ln_nstrip partials of its rows itself (fp64 mean / variance), so there
    // is no finalize launch between the two GEMMs.
    float* rowpart = nullptr; int rowpart_ld = 0;
    const float* ln_part = nullptr; int ln_nstrip = 0; float ln_inv_width = 0.f, ln_eps = 0.f;
    // split-K (kernels instantiated with SPLITK, grid.y = ksplit): workgroup (x, y) multiplies K tiles
    // [y * ktiles_per_split, ...) and stores its raw fp32 accumulators to ((float*)C)[y][m][n] (ldc = N);
    // splitk_reduce_kernel sums the slices and applies the epilogue
    int ksplit = 1;
};

// Linear workgroup id -> output tile.  MI355X deals workgroups to its 8 XCDs round-robin by linear workgroup id and
// every XCD has a private 4 MB L2, so with row-major tile ids the blocks that share an A row-tile (or a W column-tile)
// land on eight different L2s and nothing is reused below the Infinity Cache: a K=5120 GEMM then pulls > 5 TB/s through
// MALL/HBM and is memory-bound, not MFMA-bound.  Remap: (1) XCD k owns a CONTIGUOUS range of logical tile ids,
// (2) logical ids walk groups of GM m-tiles n-major, so the ~32-64 blocks resident on one XCD form a GM x (32/GM)
// patch of the output that shares GM A-tiles and a few W-tiles through that XCD's L2.
__device__ __forceinline__ void xcd_tile_id(int swz, int MT, int NT, int id, int& mt, int& nt) {
    if (!swz) { mt = id / NT; nt = id - mt * NT; return; }
    const int total = MT * NT;
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
__device__ __forceinline__ void xcd_tile(int swz, int MT, int NT, int& mt, int& nt) {
    xcd_tile_id(swz, MT, NT, blockIdx.y * gridDim.x + blockIdx.x, mt, nt);
}


typedef __attribute__((address_space(3))) void lds_void_t;

// One LDS-DMA piece: 64 lanes x 16 B from per-lane global addresses to LDS [lds_base .. +1 KiB) (lane-linear).
// Issued from inline asm on purpose: hipcc models the builtin form as an LDS store and drains it
// (s_waitcnt vmcnt(0)) in front of the next ds_read of the OTHER buffer, serialising the pipeline; the asm
// form is invisible to that bookkeeping, and the explicit vmcnt+barrier of the caller is the only wait.
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

// a wave-uniform pointer, made PROVABLY uniform (the "s" operand of dma16s): block-id arithmetic that passes through the
// persistent tile loop is sometimes classified divergent, and the asm then gets a VGPR pair (assembler error)
__device__ __forceinline__ const char* uniform_ptr(const char* p) {
    const uint64_t v = (uint64_t)(size_t)p;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return (const char*)(size_t)(((uint64_t)hi << 32) | lo);
}

// the same for data pointers that are DEREFERENCED afterwards: rebuilt from integers a pointer is "flat" to the compiler
// (flat_load / flat_store: both counters, no scalar-base form), so the result is typed as a global-memory pointer
typedef __attribute__((address_space(1))) char gmem_char_t;
typedef __attribute__((address_space(1))) u32x4_t gmem_u32x4_t;
__device__ __forceinline__ uint4 gmem_ld16(const gmem_char_t* p) {
    const u32x4_t v = *reinterpret_cast<const gmem_u32x4_t*>(p);
    return make_uint4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ void gmem_st16(gmem_char_t* p, const uint4& u) {
    const u32x4_t v = {u.x, u.y, u.z, u.w};
    *reinterpret_cast<gmem_u32x4_t*>(p) = v;
}
__device__ __forceinline__ gmem_char_t* uniform_gptr(const char* p) {
    const uint64_t v = (uint64_t)(size_t)p;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return (gmem_char_t*)(size_t)(((uint64_t)hi << 32) | lo);
}

template <int N> __device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// Epilogue shared by the LDS-DMA kernels: acc[i][j] is the 16x16 fragment at rows m_base + j*16.., columns
// n_base + i*16.. (lane l15 -> row, lane group grp -> 4 consecutive columns).
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

// EM (epilogue scaling mode): 0 none | 1 fp8 de-quantisation  t = acc * scale_a[m] * scale_w[n]  | 2 folded LayerNorm
//   t = acc * scale_a[m] + shift_a[m] * scale_w[n]   (scale_a = rstd, shift_a = -mean * rstd, scale_w = column sums of the
//   gamma-scaled weight; beta's contribution rides in the bias) — see ss_gemm_lnfold
template <typename T, int FM, int FN, int EM = 0, bool RSTAT = false>
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
        float sw[4] = {1.f, 1.f, 1.f, 1.f};
        if constexpr (EM != 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r) if (n0 + r < N) sw[r] = g.scale_w[n0 + r];
        }
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
            float sa = 1.f, sh = 0.f;
            if constexpr (EM != 0) sa = g.scale_a[m];
            if constexpr (EM == 2) sh = g.shift_a[m];
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
                float t = acc[i][j][r];
                if constexpr (EM == 1) t *= sa * sw[r];
                if constexpr (EM == 2) t = fmaf(t, sa, sh * sw[r]);
                t += bv[r];
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
            if constexpr (RSTAT) {   // statistics of the values as stored; the 4 lane groups of a row fold into group 0.
                // (rows beyond M took `continue` above: their lanes sit out the shuffle, which only pairs lanes of ONE row)
                if (g.rowstat_out) {
                    float a = 0.f, b = 0.f;
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (n0 + r < N) { const float o = Tr<T>::rnd(v[r]); a += o; b = fmaf(o, o, b); }
                    a += __shfl_xor(a, 16, 64); b += __shfl_xor(b, 16, 64);
                    a += __shfl_xor(a, 32, 64); b += __shfl_xor(b, 32, 64);
                    if (grp == 0) {
                        unsafeAtomicAdd(g.rowstat_out + 2 * (int64_t)m, (double)a);
                        unsafeAtomicAdd(g.rowstat_out + 2 * (int64_t)m + 1, (double)b);
                    }
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

// split-K partial result: the raw fp32 accumulators of this workgroup's K range -> P[split][m][n] (row stride N)
template <int FM, int FN>
__device__ __forceinline__ void gemm_epilogue_partial(const GemmArgs& g, f32x4_t (&acc)[FN][FM], int m_base, int n_base,
                                                      int l15, int grp, int split) {
    float* __restrict__ P = (float*)g.C + (int64_t)split * g.M * g.N;
#pragma unroll
    for (int i = 0; i < FN; ++i) {
        const int n0 = n_base + i * 16 + grp * 4;
#pragma unroll
        for (int j = 0; j < FM; ++j) {
            const int m = m_base + j * 16 + l15;
            if (m >= g.M) continue;
            float* o = P + (int64_t)m * g.N + n0;
            if (n0 + 3 < g.N) {
                *reinterpret_cast<float4*>(o) = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) if (n0 + r < g.N) o[r] = acc[i][j][r];
            }
        }
    }
}

// LDS-staged epilogue for 16-bit outputs.  The direct epilogue above stores 8 bytes per lane with the 16 lanes of a
// group on 16 different rows: every store instruction touches 16 cache lines with a 32-byte piece each (4-byte pieces
// with GEGLU), and the residual is read the same way — store-ISSUE-bound: ~15 us of fixed cost per 256x256 tile against
// 1.5 us per K tile, i.e. a third of a K = 1280 GEMM.  Here a wave transposes its fragments through a PRIVATE LDS
// strip (CR rows at a time; no barrier: one wave's LDS operations execute in order) and then moves whole rows: every
// lane stores 16 contiguous bytes and a store instruction covers full lines; bias / GELU / time-embedding are applied
// before staging (value rounded to T exactly where the direct path rounds it), the residual is added on the coalesced
// read-back.  Requirements (checked by the caller): the wave's TM x TN sub-tile lies inside [M, N) in N (rows are
// masked), C / residual 16-byte aligned with ldc / ldr % 8 == 0.
// EV (epilogue variant): the per-value options as COMPILE-TIME constants — 1 plain (bias / residual only), 2 +GELU, 3 +rowvec,
// 4 GEGLU pair; 0 = every option tested at run time per value (any combination).  With run-time flags the per-value stream
// carries a scalar branch per GELU test and a round/add/round/select per rowvec test even when the option is off: 28 vector
// instructions per accumulator value on the ff1 GEGLU tile, ~40 % of them for options that are not in use.
template <typename T, int FM, int FN, int CR, int EM = 0, bool RSTAT = false, int EV = 0>
__device__ __forceinline__ void gemm_epilogue_staged(const GemmArgs& g, f32x4_t (&acc)[FN][FM], int m_base, int n_base,
                                                     int lane, char* stg) {
    static_assert(Tr<T>::kVec == 8, "16-bit outputs only");
    constexpr int TN = FN * 16;
    constexpr int FPC = CR / 16;                       // M fragments per chunk
    static_assert(CR % 16 == 0 && FM % FPC == 0, "chunk rows");
    const int l15 = lane & 15, grp = lane >> 4;
    const bool geglu = EV == 0 ? (g.epi & SS_EPI_GEGLU_PAIR) != 0 : EV == 4;
    const bool do_gelu = EV == 0 ? (g.epi & SS_EPI_GELU) != 0 : EV == 2;
    const bool do_rowvec = EV == 0 ? g.rowvec != nullptr : EV == 3;
    const int M = g.M;
    T* __restrict__ C = (T*)g.C;
    const T* bias = (const T*)g.bias;
    const T* res = (const T*)g.residual;
    constexpr int RS = TN * 2 + 16;                    // strip row stride (bytes): +16 breaks the power-of-two stride
    // per-lane bias of its 4 columns per fragment column block
    float bv[FN][4];
    float swv[EM != 0 ? FN : 1][4];
#pragma unroll
    for (int i = 0; i < FN; ++i) {
        bv[i][0] = bv[i][1] = bv[i][2] = bv[i][3] = 0.f;
        if (g.epi & SS_EPI_BIAS) ld4<T>(bias + n_base + i * 16 + grp * 4, bv[i]);
        if constexpr (EM != 0) ld4<float>(g.scale_w + n_base + i * 16 + grp * 4, swv[i]);
    }
    // folded LayerNorm fed by producer PARTIALS (g.ln_part): lane l forms (rstd, -mean * rstd) of rows l and l + 64 of this
    // wave's sub-tile from their ln_nstrip (sum, sum of squares) pairs — fp64, E[x^2] - mean^2 does not cancel there — and the
    // staging loop below fetches a fragment row's pair with two lane shuffles
    constexpr int NSET = EM == 2 ? (FM * 16 + 63) / 64 : 1;
    float st_sa[NSET], st_sh[NSET];
    const bool from_part = EM == 2 && g.ln_part != nullptr;
    if constexpr (EM == 2) {
        if (from_part) {
#pragma unroll
            for (int h = 0; h < NSET; ++h) {
                int m = m_base + h * 64 + lane;
                m = m < M ? m : M - 1;
                const float2* pp = reinterpret_cast<const float2*>(g.ln_part) + (int64_t)m * g.ln_nstrip;
                double a = 0.0, b = 0.0;
                // 16 partials at a time, all loads issued before the first add (a dependent load per strip costs one L2
                // round trip each: 16 x 2 x ~500 cycles per tile, measured +6 ms per forward)
                for (int k0 = 0; k0 < g.ln_nstrip; k0 += 16) {
                    float2 t[16];
#pragma unroll
                    for (int k = 0; k < 16; ++k) t[k] = (k0 + k < g.ln_nstrip) ? pp[k0 + k] : make_float2(0.f, 0.f);
#pragma unroll
                    for (int k = 0; k < 16; ++k) { a += (double)t[k].x; b += (double)t[k].y; }
                }
                const double mean = a * (double)g.ln_inv_width;
                const double var = fmax(b * (double)g.ln_inv_width - mean * mean, 0.0);
                const float r = (float)(1.0 / sqrt(var + (double)g.ln_eps));
                st_sa[h] = r;
                st_sh[h] = -(float)mean * r;
            }
        }
    }
#pragma unroll
    for (int c = 0; c < FM / FPC; ++c) {
        // ---- stage CR rows: lane (l15, grp) holds rows j*16 + l15, columns i*16 + grp*4 .. +3 ----
#pragma unroll
        for (int jj = 0; jj < FPC; ++jj) {
            const int j = c * FPC + jj;
            const int m = m_base + j * 16 + l15;
            float rv[FN][4];
            if (do_rowvec) {
                const int mm = m < M ? m : M - 1;
                const T* rp = (const T*)g.rowvec + (int64_t)(mm / g.rows_per_batch) * g.rowvec_ld + n_base + grp * 4;
#pragma unroll
                for (int i = 0; i < FN; ++i) ld4<T>(rp + i * 16, rv[i]);
            }
            char* rowp = stg + (jj * 16 + l15) * RS;
            float sa = 1.f, sh = 0.f;
            if constexpr (EM == 2) {
                if (from_part) {
                    sa = __shfl(st_sa[(j * 16) / 64], (j * 16) % 64 + l15, 64);
                    sh = __shfl(st_sh[(j * 16) / 64], (j * 16) % 64 + l15, 64);
                } else {
                    sa = g.scale_a[m < M ? m : M - 1];
                    sh = g.shift_a[m < M ? m : M - 1];
                }
            } else if constexpr (EM != 0) {
                sa = g.scale_a[m < M ? m : M - 1];
            }
#pragma unroll
            for (int i = 0; i < FN; ++i) {
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float t = acc[i][j][r];
                    if constexpr (EM == 1) t *= sa * swv[i][r];
                    if constexpr (EM == 2) t = fmaf(t, sa, sh * swv[i][r]);
                    t += bv[i][r];
                    if (do_gelu) t = gelu_for<T>(Tr<T>::rnd(t));
                    v[r] = Tr<T>::rnd(t);
                    if (do_rowvec) v[r] = Tr<T>::rnd(v[r] + rv[i][r]);
                }
                if (geglu) {
                    const float o0 = v[0] * Tr<T>::rnd(gelu_for<T>(v[1])), o1 = v[2] * Tr<T>::rnd(gelu_for<T>(v[3]));
                    float pk[8] = {o0, o1, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                    *reinterpret_cast<uint32_t*>(rowp + (i * 8 + grp * 2) * 2) = pack<T>(pk).x;
                } else {
                    float pk[8] = {v[0], v[1], v[2], v[3], 0.f, 0.f, 0.f, 0.f};
                    const uint4 u = pack<T>(pk);
                    *reinterpret_cast<uint2*>(rowp + (i * 16 + grp * 4) * 2) = make_uint2(u.x, u.y);
                }
            }
        }
        // ---- read back whole rows (16 B per lane) and store; GEGLU rows are TN/2 outputs wide ----
        auto readback = [&](auto lpr_tag) {
            constexpr int LPR = decltype(lpr_tag)::value;   // lanes (16-byte pieces) per output row of the strip
            const int n_out0 = geglu ? (n_base >> 1) : n_base;
#pragma unroll
            for (int k = 0; k < (CR * LPR + 63) / 64; ++k) {
                const int idx = lane + k * 64;
                const int row = idx / LPR, c16 = idx - row * LPR;
                const int m = m_base + c * CR + row;
                if (idx < CR * LPR && m < M) {
                    uint4 u = *reinterpret_cast<const uint4*>(stg + row * RS + c16 * 16);
                    if (g.epi & SS_EPI_RESIDUAL) {
                        const uint4 rr = *reinterpret_cast<const uint4*>(res + (int64_t)m * g.ldr + n_base + c16 * 8);
                        float a[8], b[8];
                        unpack<T>(u, a);
                        unpack<T>(rr, b);
#pragma unroll
                        for (int e = 0; e < 8; ++e) a[e] += b[e];
                        u = pack<T>(a);
                    }
                    *reinterpret_cast<uint4*>(C + (int64_t)m * g.ldc + n_out0 + c16 * 8) = u;
                    if constexpr (RSTAT) {
                        if (g.rowstat_out || g.rowpart) {   // (sum, sum of squares) of the 8 stored values, parked in the piece just consumed
                            float a[8];
                            unpack<T>(u, a);
                            float sv = 0.f, qv = 0.f;
#pragma unroll
                            for (int e = 0; e < 8; ++e) { sv += a[e]; qv = fmaf(a[e], a[e], qv); }
                            *reinterpret_cast<float2*>(stg + row * RS + c16 * 16) = make_float2(sv, qv);
                        }
                    }
                }
            }
            if constexpr (RSTAT) {
                if (g.rowstat_out || g.rowpart) {   // one wave's LDS operations execute in order: lane r < CR folds row r's LPR partials
                    const int m = m_base + c * CR + lane;
                    if (lane < CR && m < M) {
                        float sv = 0.f, qv = 0.f;
#pragma unroll
                        for (int q = 0; q < LPR; ++q) {
                            const float2 t = *reinterpret_cast<const float2*>(stg + lane * RS + q * 16);
                            sv += t.x; qv += t.y;
                        }
                        if (g.rowpart) {      // this wave's column strip of row m: written once, no atomics
                            *reinterpret_cast<float2*>(g.rowpart + ((int64_t)m * g.rowpart_ld + n_base / TN) * 2) = make_float2(sv, qv);
                        } else {
                            unsafeAtomicAdd(g.rowstat_out + 2 * (int64_t)m, (double)sv);
                            unsafeAtomicAdd(g.rowstat_out + 2 * (int64_t)m + 1, (double)qv);
                        }
                    }
                }
            }
        };
        if (geglu) readback(std::integral_constant<int, TN / 16>{});
        else readback(std::integral_constant<int, TN / 8>{});
    }
}

// Round 6: the same staged epilogue as a SOFTWARE PIPELINE over the 16-row chunks (ping-pong tiles; EM = 0, no row statistics).
// What the ISA of gemm_epilogue_staged showed on the 256x320 tile (hipcc -S): per chunk ~65 VALU of staging (every value rounded,
// unpacked and packed AGAIN: round-then-pack is two conversions), then three strictly serial [exec mask -> ds_read_b128 ->
// lgkmcnt(0) -> two quarter-rate 32-bit multiplies + a 64-bit mad for the row address -> store] pieces, and with a residual
// three dependent global round trips (load -> vmcnt(0) -> add -> store) — 8 chunks x 3 latency chains per tile with nothing under
// them: ~5.4 us per tile whether the tile is 128 KB or 160 KB (profiles/round6_gemm_epilogue_cost.txt), i.e. latency, not bytes.
// Here, per wave:   [residual loads(0, 1)] stage(0) | read(0) | for c: [residual loads(c+2)] stage(c+1) | wait | add, store(c) | read(c+1)
//   * a wave's LDS operations execute in order, so stage(c+1) may overwrite the strip right behind read(c)'s ISSUE: the
//     read-back latency (and the residual's) runs under the next chunk's conversion work, one strip is enough;
//   * the pieces of a chunk are read / loaded / stored as a batch (no per-piece exec regions: invalid lanes read strip byte 0);
//   * the row address is (wave-uniform chunk base) + (per-lane byte offset formed once per tile);
//   * values are converted once (pack of the unrounded sum == pack of the rounded one).
// Same arithmetic per value, same rounding points as gemm_epilogue_staged: results are bit-identical (ubench equality screen
// against the one-barrier tiles, which keep gemm_epilogue_staged).  WHOLE: the wave's rows are all inside M (the 320-wide and conv tiles take whole tiles only).
template <typename T> __device__ __forceinline__ uint32_t pack2(float a, float b) {
    if constexpr (std::is_same<T, bf16_t>::value) return f32x2_to_bf16x2_bits(a, b);
    else return f32_to_f16_bits(a) | (f32_to_f16_bits(b) << 16);
}
template <typename T> __device__ __forceinline__ void unpack2(uint32_t w, float& lo, float& hi) {
    if constexpr (std::is_same<T, bf16_t>::value) { lo = __uint_as_float(w << 16); hi = __uint_as_float(w & 0xffff0000u); }
    else { lo = f16_bits_to_f32(w & 0xffffu); hi = f16_bits_to_f32(w >> 16); }
}
template <typename T, int FM, int FN, int EV, bool RES, bool WHOLE>
__device__ __forceinline__ void gemm_epilogue_staged_pipe(const GemmArgs& g, f32x4_t (&acc)[FN][FM], int m_base, int n_base,
                                                          int lane, char* stg) {
    static_assert(Tr<T>::kVec == 8, "16-bit outputs only");
    static_assert(EV >= 1 && EV <= 4, "compile-time variants only");
    constexpr int TN = FN * 16, RS = TN * 2 + 16;
    constexpr bool GEGLU = EV == 4;
    constexpr int LPR = GEGLU ? TN / 16 : TN / 8;      // 16-byte pieces per output row of the strip
    constexpr int NPC = 16 * LPR;                      // ... per chunk
    constexpr int NP = (NPC + 63) / 64;
    const int l15 = lane & 15, grp = lane >> 4;
    const int M = g.M;
    constexpr bool has_res = RES;                      // the caller dispatches on SS_EPI_RESIDUAL
    const T* bias = (const T*)g.bias;

    // bias of the lane's 4 columns per fragment: all FN loads in flight together (a test of the flag per fragment put a full
    // global round trip between any two of them)
    float bv[FN][4];
    {
        uint2 braw[FN];
        const bool hb = (g.epi & SS_EPI_BIAS) != 0;
#pragma unroll
        for (int i = 0; i < FN; ++i) braw[i] = hb ? *reinterpret_cast<const uint2*>(bias + n_base + i * 16 + grp * 4) : make_uint2(0u, 0u);
        // (measured and dropped: the tile's bias segment DMA'd next to the ring by the prologue and read from LDS here — no
        // difference on any shape, profiles/round6_epilogue_pipe2_ubench.txt: the five loads in flight together were already enough)
#pragma unroll
        for (int i = 0; i < FN; ++i) {
            float f[8];
            unpack<T>(make_uint4(braw[i].x, braw[i].y, 0u, 0u), f);
            bv[i][0] = f[0]; bv[i][1] = f[1]; bv[i][2] = f[2]; bv[i][3] = f[3];
        }
    }
    // time-embedding vector (EV 3; conv tiles): rows of one 16-row chunk share the batch index (rows_per_batch is a power of
    // two >= 64 there), kept packed (2 registers per fragment) and re-loaded only when the chunk's batch index changes
    uint2 rvp[EV == 3 ? FN : 1];
    int rv_b = -1;
    auto load_rv = [&](int c) {
        if constexpr (EV == 3) {
            const int bidx = (m_base + c * 16) / g.rows_per_batch;      // wave-uniform
            if (bidx != rv_b) {
                rv_b = bidx;
                const T* rp = (const T*)g.rowvec + (int64_t)bidx * g.rowvec_ld + n_base + grp * 4;
#pragma unroll
                for (int i = 0; i < FN; ++i) rvp[i] = *reinterpret_cast<const uint2*>(rp + i * 16);
            }
        }
    };

    // per-lane piece coordinates, once per tile
    uint32_t loff[NP], goff[NP], roff[NP];
    bool pv[NP];
    int prow[NP];
#pragma unroll
    for (int k = 0; k < NP; ++k) {
        const int idx = lane + k * 64;
        const int row = idx / LPR, c16 = idx - row * LPR;
        pv[k] = (NPC % 64 == 0 || k + 1 < NP) ? true : idx < NPC;
        prow[k] = row;
        loff[k] = pv[k] ? (uint32_t)(row * RS + c16 * 16) : 0u;
        goff[k] = (uint32_t)row * (uint32_t)(g.ldc * 2) + (uint32_t)(c16 * 16);
        roff[k] = pv[k] ? (uint32_t)row * (uint32_t)(g.ldr * 2) + (uint32_t)(c16 * 16) : 0u;   // lanes without a piece re-read piece 0 (no exec region around the loads)
    }
    const int n_out0 = GEGLU ? (n_base >> 1) : n_base;
    char* Cb = reinterpret_cast<char*>((T*)g.C + (int64_t)m_base * g.ldc + n_out0);                          // wave-uniform
    const char* Rb = reinterpret_cast<const char*>((const T*)g.residual + (int64_t)m_base * g.ldr + n_base);
    const int64_t cstep = 32 * g.ldc, rstep = 32 * g.ldr;      // bytes per 16-row chunk
    char* wrow = stg + l15 * RS + (GEGLU ? grp * 4 : grp * 8);

    auto stage = [&](int c) {
        load_rv(c);
#pragma unroll
        for (int i = 0; i < FN; ++i) {
            float t[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) t[r] = acc[i][c][r] + bv[i][r];
            if constexpr (EV == 2) {
#pragma unroll
                for (int r = 0; r < 4; ++r) t[r] = gelu_for<T>(Tr<T>::rnd(t[r]));
            }
            if constexpr (EV == 3) {
                float rv[8];
                asm volatile("" : "+v"(rvp[i].x), "+v"(rvp[i].y));     // stays packed: unpacked once per chunk, not hoisted into 4 registers per fragment
                unpack<T>(make_uint4(rvp[i].x, rvp[i].y, 0u, 0u), rv);
#pragma unroll
                for (int r = 0; r < 4; ++r) t[r] = Tr<T>::rnd(t[r]) + rv[r];
            }
            if constexpr (GEGLU) {
                // (value, gate) sit in ADJACENT accumulator registers: rounding them as the pairs they come in and taking the two
                // gates out of the two packed words leaves the gates in a register pair of the compiler's choice — written as four
                // separate roundings it gathers (v1, v3) and (v0, v2) into pairs with four v_mov per fragment first
                float v0, v1, v2, v3;
                unpack2<T>(pack2<T>(t[0], t[1]), v0, v1);
                unpack2<T>(pack2<T>(t[2], t[3]), v2, v3);
                const float o0 = v0 * Tr<T>::rnd(gelu_for<T>(v1)), o1 = v2 * Tr<T>::rnd(gelu_for<T>(v3));
                *reinterpret_cast<uint32_t*>(wrow + i * 16) = pack2<T>(o0, o1);
            } else {
                *reinterpret_cast<uint2*>(wrow + i * 32) = make_uint2(pack2<T>(t[0], t[1]), pack2<T>(t[2], t[3]));
            }
        }
    };
    uint4 u[NP];
    auto readback = [&]() {
#pragma unroll
        for (int k = 0; k < NP; ++k) u[k] = *reinterpret_cast<const uint4*>(stg + loff[k]);
    };

    // residual pieces: requested RD chunks ahead of their use (12 registers per chunk in flight; the accumulators of the chunks
    // already staged are free by then)
    constexpr int RD = 2;
    uint4 rr[FM][NP];
    auto row_ok = [&](int c, int k) { return pv[k] && (WHOLE || m_base + c * 16 + prow[k] < M); };
    auto load_res = [&](int c) {
        const gmem_char_t* Rc = uniform_gptr(Rb + c * rstep);  // scalar base + 32-bit lane offset (left alone the compiler forms
                                                               // 64-bit per-lane addresses for the later chunks and spills them)
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            if constexpr (WHOLE) rr[c][k] = gmem_ld16(Rc + roff[k]);
            else rr[c][k] = row_ok(c, k) ? gmem_ld16(Rc + roff[k]) : make_uint4(0u, 0u, 0u, 0u);
        }
    };
    load_rv(0);
    if (has_res) {
#pragma unroll
        for (int c = 0; c < RD && c < FM; ++c) load_res(c);
    }
    __builtin_amdgcn_sched_barrier(0);
    stage(0);
    __builtin_amdgcn_sched_barrier(0);
    readback();
#pragma unroll
    for (int c = 0; c < FM; ++c) {
        if (has_res && c + RD < FM) load_res(c + RD);
        __builtin_amdgcn_sched_barrier(0);
        if (c + 1 < FM) stage(c + 1);      // behind read(c) in the wave's LDS order; its VALU covers the read-back latency
        __builtin_amdgcn_sched_barrier(0);
        gmem_char_t* Cc = uniform_gptr(Cb + c * cstep);
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            uint4 o = u[k];
            if (has_res) {
                float a[8], b[8];
                unpack<T>(o, a);
                unpack<T>(rr[c][k], b);
#pragma unroll
                for (int e = 0; e < 8; ++e) a[e] += b[e];
                o = pack<T>(a);
            }
            if (row_ok(c, k)) gmem_st16(Cc + goff[k], o);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (c + 1 < FM) readback();
    }
}

// Software-pipelined LDS-DMA kernels (ss_gemm_sp.inc, instantiated per dtype in ss_gemm_sp_{bf16,f16}.hip).
// Returns SS_OK, or 1 when `cfg` is not a pipelined configuration / the shape is not eligible (the caller
// then falls back to the double-buffered kernel).
template <typename T> int gemm_sp_dispatch(int cfg, const GemmArgs& g, hipStream_t s);
// fp8 (e4m3) operands, bf16 results: ss_gemm_sp_fp8.hip
int gemm_sp_dispatch_fp8(int cfg, const GemmArgs& g, hipStream_t s);
// fp8 on the 4-wave / AGPR-accumulator tiles (cfg 95, 96): ss_gemm_w4_fp8.hip
int gemm_w4_dispatch_fp8(int cfg, const GemmArgs& g, hipStream_t s);

}  // namespace ss
