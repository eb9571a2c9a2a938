// Attention kernels of the SEED-Story hot path.
//
//  (1) flash_attn_kernel — fused QK^T -> online softmax -> PV on the matrix cores, replacing
//      xops.memory_efficient_attention(q, k, v, LowerTriangularFromBottomRightMask)
//      (src/models_clm/modeling_llama_xformer.py:289-295; query i sees keys j <= i + kv - q),
//      the materialised bmm/softmax/bmm of the ViT (src/models/qwen_visual.py:207-220,
//      head_dim 104), nn.MultiheadAttention in the Resamplers (:147-149) and
//      PerceiverAttention (src/models_ipa/resampler.py:69-72).
//      Layout trick: compute S^T = K·Q^T so that a lane owns 16 scores of ONE query
//      (q = lane & 15): the row max/sum are 15 in-lane ops + 2 shuffles, and the
//      probabilities are already the B-fragment of O^T = V^T·P^T — P never moves between
//      lanes; O^T's accumulator layout gives each lane 4 consecutive d of its query.
//  (2) attn_decode_kernel + attn_combine_kernel — q_len = 1 split-KV decode attention over the
//      KV cache (HBM-bound: streams K and V of one head once), lengths read on the device so
//      the launch replays from a hipGraph.
#include "ss_common.h"

namespace ss {

// --------------------------------------------------------------------------------------------
// (1) flash attention
// --------------------------------------------------------------------------------------------
struct AttnArgs {
    const void *q, *k, *v;
    void* out;
    int q_len, kv_len, hd, n_heads;
    int64_t q_sb, q_sh, q_ss, k_sb, k_sh, k_ss, v_sb, v_sh, v_ss, o_sb, o_sh, o_ss;
    float scale;
    int causal_br;
    int xcd_heads;   // flash_attn3: > 0 = XCD-aware 1-D grid, value = query blocks per head
    int ragged;      // 1 = batch element b attends to kv_len_b[b] keys (kv_len = their maximum); batch <= 8
    int kv_len_b[8];
};

template <typename T> struct AttnMma;
template <> struct AttnMma<bf16_t> {
    static __device__ __forceinline__ f32x4_t run(const uint4& a, const uint4& b, f32x4_t c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a),
                                                       __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
    }
};
template <> struct AttnMma<f16_t> {
    static __device__ __forceinline__ f32x4_t run(const uint4& a, const uint4& b, f32x4_t c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a),
                                                      __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
    }
};

constexpr int kBQ = 64;   // query rows per block (4 waves x 16)
constexpr int kBKV = 64;  // keys per tile

template <typename T, int HD>
__global__ __launch_bounds__(256) void flash_attn_kernel(const AttnArgs a) {
    constexpr int V = Tr<T>::kVec;
    constexpr int KS = HD + V;    // K tile row stride (elements), padded
    constexpr int VS = kBKV + V;  // V^T tile row stride
    constexpr int NDB = HD / 16;  // 16-wide d blocks of the output
    constexpr int NKB = kBKV / 16;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    T* Ks = reinterpret_cast<T*>(smem_raw);  // [kBKV][KS]
    T* Vt = Ks + kBKV * KS;                  // [HD][VS]

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int l15 = lane & 15, grp = lane >> 4;
    const int bh = blockIdx.y, b = bh / a.n_heads, h = bh % a.n_heads;
    const int kvl = a.ragged ? a.kv_len_b[b] : a.kv_len;      // per-batch key count (stacked story slots)
    const int q0 = blockIdx.x * kBQ;
    const int qi = q0 + wid * 16 + l15;  // this lane's query row
    const int hd = a.hd;
    const T* __restrict__ qp = (const T*)a.q + (int64_t)b * a.q_sb + (int64_t)h * a.q_sh;
    const T* __restrict__ kp = (const T*)a.k + (int64_t)b * a.k_sb + (int64_t)h * a.k_sh;
    const T* __restrict__ vp = (const T*)a.v + (int64_t)b * a.v_sb + (int64_t)h * a.v_sh;
    const int shift = kvl - a.q_len;  // bottom-right alignment

    // ---- Q fragments (B operand of S^T = K Q^T), resident for the whole kernel ----------------
    // bf16/f16: qf[ks] = Q[qi][ks*32 + grp*8 .. +8];  f32: qs[kk] = Q[qi][kk*4 + grp]
    uint4 qf[V == 8 ? HD / 32 : 1];
    float qs[V == 4 ? HD / 4 : 1];
    if constexpr (V == 8) {
#pragma unroll
        for (int ks = 0; ks < HD / 32; ++ks) {
            const int d = ks * 32 + grp * 8;
            qf[ks] = (qi < a.q_len && d < hd) ? ld16(qp + (int64_t)qi * a.q_ss + d) : make_uint4(0, 0, 0, 0);
        }
    } else {
#pragma unroll
        for (int kk = 0; kk < HD / 4; ++kk) {
            const int d = kk * 4 + grp;
            qs[kk] = (qi < a.q_len && d < hd) ? Tr<T>::ld(qp + (int64_t)qi * a.q_ss + d) : 0.f;
        }
    }

    f32x4_t o[NDB];
#pragma unroll
    for (int i = 0; i < NDB; ++i) o[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    float m_run = -1e30f, l_run = 0.f;

    int kv_end = kvl;
    if (a.causal_br) {
        const int lim = q0 + kBQ - 1 + shift + 1;  // one past the last key any row of this block sees
        if (lim < kv_end) kv_end = lim;
        if (kv_end < 0) kv_end = 0;
    }

    for (int t0 = 0; t0 < kv_end; t0 += kBKV) {
        __syncthreads();  // previous tile fully consumed
        // ---- stage K tile [key][d] and V tile transposed [d][key] ------------------------------
        constexpr int PPR = HD / V;  // packs per key row
        for (int p = tid; p < kBKV * PPR; p += 256) {
            const int key = p / PPR, c = p % PPR, d = c * V;
            const int kg = t0 + key;
            const bool ok = kg < kvl && d < hd;
            const uint4 kk = ok ? ld16(kp + (int64_t)kg * a.k_ss + d) : make_uint4(0, 0, 0, 0);
            st16(Ks + key * KS + d, kk);
            const uint4 vv = ok ? ld16(vp + (int64_t)kg * a.v_ss + d) : make_uint4(0, 0, 0, 0);
            const T* ve = reinterpret_cast<const T*>(&vv);
#pragma unroll
            for (int j = 0; j < V; ++j) Vt[(d + j) * VS + key] = ve[j];
        }
        __syncthreads();

        // ---- S^T tile: 4 key blocks x (16 keys x 16 queries) --------------------------------------
        f32x4_t s[NKB];
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {
            s[kb] = f32x4_t{0.f, 0.f, 0.f, 0.f};
            if constexpr (V == 8) {
#pragma unroll
                for (int ks = 0; ks < HD / 32; ++ks) {
                    const uint4 kf = ld16(Ks + (kb * 16 + l15) * KS + ks * 32 + grp * 8);
                    s[kb] = AttnMma<T>::run(kf, qf[ks], s[kb]);
                }
            } else {
#pragma unroll
                for (int kk = 0; kk < HD / 4; ++kk) {
                    const float kf = ((const float*)Ks)[(kb * 16 + l15) * KS + kk * 4 + grp];
                    s[kb] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf, qs[kk], s[kb], 0, 0, 0);
                }
            }
        }
        // lane now holds S[qi][key = t0 + kb*16 + grp*4 + r]
        float tmax = -1e30f;
        bool msk[NKB][4];
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int kg = t0 + kb * 16 + grp * 4 + r;
                const bool ok = kg < kvl && (!a.causal_br || kg <= qi + shift);
                msk[kb][r] = ok;
                const float sv = ok ? s[kb][r] * a.scale : -1e30f;
                s[kb][r] = sv;
                tmax = fmaxf(tmax, sv);
            }
        tmax = fmaxf(tmax, __shfl_xor(tmax, 16, 64));
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
        const float m_new = fmaxf(m_run, tmax);
        const float alpha = expf(m_run - m_new);
        float lsum = 0.f;
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float p = msk[kb][r] ? expf(s[kb][r] - m_new) : 0.f;
                s[kb][r] = p;
                lsum += p;
            }
        lsum += __shfl_xor(lsum, 16, 64);
        lsum += __shfl_xor(lsum, 32, 64);
        l_run = l_run * alpha + lsum;
        m_run = m_new;
#pragma unroll
        for (int i = 0; i < NDB; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) o[i][r] *= alpha;

        // ---- O^T += V^T · P^T ------------------------------------------------------------------------
        if constexpr (V == 8) {
#pragma unroll
            for (int kp2 = 0; kp2 < NKB / 2; ++kp2) {
                // k-slots of lane group g: j<4 -> key kb0*16+g*4+j ; j>=4 -> key kb1*16+g*4+(j-4)
                float pf[8];
#pragma unroll
                for (int r = 0; r < 4; ++r) { pf[r] = s[2 * kp2][r]; pf[4 + r] = s[2 * kp2 + 1][r]; }
                const uint4 pfrag = pack<T>(pf);
#pragma unroll
                for (int db = 0; db < NDB; ++db) {
                    const T* vrow = Vt + (db * 16 + l15) * VS + grp * 4;
                    const uint2 v0 = *reinterpret_cast<const uint2*>(vrow + (2 * kp2) * 16);
                    const uint2 v1 = *reinterpret_cast<const uint2*>(vrow + (2 * kp2 + 1) * 16);
                    const uint4 vfrag = make_uint4(v0.x, v0.y, v1.x, v1.y);
                    o[db] = AttnMma<T>::run(vfrag, pfrag, o[db]);
                }
            }
        } else {
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int db = 0; db < NDB; ++db) {
                        const float vf = ((const float*)Vt)[(db * 16 + l15) * VS + kb * 16 + grp * 4 + r];
                        o[db] = __builtin_amdgcn_mfma_f32_16x16x4f32(vf, s[kb][r], o[db], 0, 0, 0);
                    }
        }
    }

    // ---- epilogue: lane holds O[qi][d = db*16 + grp*4 + r] -------------------------------------------
    if (qi < a.q_len) {
        const float inv = l_run > 0.f ? 1.0f / l_run : 0.f;
        T* op = (T*)a.out + (int64_t)b * a.o_sb + (int64_t)h * a.o_sh + (int64_t)qi * a.o_ss;
#pragma unroll
        for (int db = 0; db < NDB; ++db) {
            const int d = db * 16 + grp * 4;
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (d + r < hd) Tr<T>::st(op + d + r, o[db][r] * inv);
        }
    }
}

// --------------------------------------------------------------------------------------------
// (1b) flash attention v2 (bf16 / f16): same S^T = K Q^T / O^T = V^T P^T lane algebra, but
//   * 32 query rows per wave (128 per block): every K / V fragment read from LDS feeds two MFMAs;
//   * K and V tiles arrive by LDS-DMA (global_load_lds_dwordx4, inline asm so hipcc does not drain it),
//     double-buffered, one barrier per 64-key tile;
//   * K image XOR-swizzled (conflict-free ds_read_b128, as in the GEMM); V image row-major and read with
//     ds_read_b64_tr_b16: inside a 16-lane group lane i points at key i/4, d (i%4)*4 and receives
//     [4 keys][d = i] (probed on MI355X, tools/tr_probe.py) = exactly the V^T fragment — no transposed
//     scalar LDS writes;
//   * exp2 with the scale folded in (log2 domain running max).
// --------------------------------------------------------------------------------------------
typedef short v4s_t __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) v4s_t lds_v4s_t;
typedef __attribute__((address_space(3))) void lds_void2_t;
__device__ __attribute__((aligned(16))) unsigned int g_attn_zero_page[64];

__device__ __forceinline__ void attn_dma16(const void* gsrc, uint32_t lds_base) {
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

#if SS_EXPERIMENTAL   // v2 (round 2; superseded by v3 / v3p, selected by no default): `make EXPERIMENTAL=1` keeps it for A/B runs
template <typename T, int HD>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(HD <= 64 ? 3 : 2)))
void flash_attn2_kernel(const AttnArgs a) {
    constexpr int V = 8;
    constexpr int BQ2 = 128;             // query rows per block: 4 waves x 32
    constexpr int CPR = HD / V;          // 16-byte chunks per key row
    constexpr int KPI = 64 / CPR;        // keys per DMA instruction
    constexpr int NI = kBKV / KPI;       // DMA instructions per tile per operand
    constexpr int IPW = NI / 4;          // ... per wave
    constexpr int ROWB = HD * 2;         // bytes per key row
    constexpr int TILE_B = kBKV * ROWB;  // bytes of one K (or V) tile
    constexpr int NDB = HD / 16, NKB = kBKV / 16, NKS = HD / 32;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    // [buf][K tile | V tile]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, grp = lane >> 4;
    const int bh = blockIdx.y, b = bh / a.n_heads, h = bh % a.n_heads;
    const int kvl = a.ragged ? a.kv_len_b[b] : a.kv_len;      // per-batch key count (stacked story slots)
    const int q0 = blockIdx.x * BQ2;
    const int hd = a.hd;
    const T* __restrict__ qp = (const T*)a.q + (int64_t)b * a.q_sb + (int64_t)h * a.q_sh;
    const T* __restrict__ kp = (const T*)a.k + (int64_t)b * a.k_sb + (int64_t)h * a.k_sh;
    const T* __restrict__ vp = (const T*)a.v + (int64_t)b * a.v_sb + (int64_t)h * a.v_sh;
    const int shift = kvl - a.q_len;
    const T* zero = reinterpret_cast<const T*>(g_attn_zero_page);
    const uint32_t lds0 = __builtin_amdgcn_readfirstlane((uint32_t)(size_t)(lds_void2_t*)smem_raw);
    const float sl2 = a.scale * 1.4426950408889634f;   // scores kept in the log2 domain

    int qi[2];
    uint4 qf[2][NKS];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        qi[qb] = q0 + wid * 32 + qb * 16 + l15;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            const int d = ks * 32 + grp * 8;
            qf[qb][ks] = (qi[qb] < a.q_len && d < hd) ? ld16(qp + (int64_t)qi[qb] * a.q_ss + d) : make_uint4(0, 0, 0, 0);
        }
    }
    f32x4_t o[2][NDB];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb)
#pragma unroll
        for (int i = 0; i < NDB; ++i) o[qb][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    float m_run[2] = {-1e30f, -1e30f}, l_run[2] = {0.f, 0.f};

    int kv_end = kvl;
    if (a.causal_br) {
        const int lim = q0 + BQ2 - 1 + shift + 1;
        if (lim < kv_end) kv_end = lim;
        if (kv_end < 0) kv_end = 0;
    }
    const int ntiles = (kv_end + kBKV - 1) / kBKV;

    // DMA coordinates of this lane: key row within the instruction and physical chunk
    const int skey = lane / CPR, spc = lane % CPR;
    auto issue_tile = [&](int t, int buf) {
        const uint32_t kbase = lds0 + (uint32_t)(buf * 2 * TILE_B);
        const uint32_t vbase = kbase + TILE_B;
#pragma unroll
        for (int i = 0; i < IPW; ++i) {
            const int key = (wid * IPW + i) * KPI + skey;       // key row inside the tile
            const int kg = t * kBKV + key;
            const int lc = spc ^ (key & (CPR - 1));               // K: swizzled on the source side
            const bool okk = kg < kvl && lc * V < hd;
            const bool okv = kg < kvl && spc * V < hd;
            const T* ks = okk ? kp + (int64_t)kg * a.k_ss + lc * V : zero;
            const T* vs = okv ? vp + (int64_t)kg * a.v_ss + spc * V : zero;
            const uint32_t roff = (uint32_t)((wid * IPW + i) * KPI * ROWB);
            attn_dma16(ks, __builtin_amdgcn_readfirstlane(kbase + roff));
            attn_dma16(vs, __builtin_amdgcn_readfirstlane(vbase + roff));
        }
    };

    // 3-deep K/V ring: tiles t+1 and t+2 are in flight while tile t is consumed.  One tile of prefetch distance
    // (~0.6 us of work at head_dim 64) is shorter than a DMA round trip, so a 2-deep ring stalled at every barrier.
    constexpr int NSTAGE = HD <= 64 ? 3 : 2;   // head_dim 128 tiles are twice as large: 2 stages keep 2 blocks per CU
    constexpr int DPT = 2 * IPW;               // DMA instructions per tile per wave (K and V)
#pragma unroll
    for (int st = 0; st < NSTAGE - 1; ++st)
        if (st < ntiles) issue_tile(st, st);
    int stage = 0;
    for (int t = 0; t < ntiles; ++t) {
        if (NSTAGE == 3 && t + 1 < ntiles) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DPT) : "memory");   // tile t landed, t+1 may fly
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                     // ... for every wave; the stage of tile t-1 is drained
        if (t + NSTAGE - 1 < ntiles) {
            int ns = stage + NSTAGE - 1;
            if (ns >= NSTAGE) ns -= NSTAGE;
            issue_tile(t + NSTAGE - 1, ns);
        }
        const char* Kb = smem_raw + stage * 2 * TILE_B;
        const char* Vb = Kb + TILE_B;
        if (++stage == NSTAGE) stage = 0;
        const int t0 = t * kBKV;

        // ---- S^T = K Q^T for both 16-row query blocks ---------------------------------------------
        f32x4_t s[2][NKB];
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {
            s[0][kb] = f32x4_t{0.f, 0.f, 0.f, 0.f};
            s[1][kb] = f32x4_t{0.f, 0.f, 0.f, 0.f};
            const int r = kb * 16 + l15;
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {
                const uint4 kf = *reinterpret_cast<const uint4*>(Kb + r * ROWB + (((ks * 4 + grp) ^ (r & (CPR - 1))) << 4));
                s[0][kb] = AttnMma<T>::run(kf, qf[0][ks], s[0][kb]);
                s[1][kb] = AttnMma<T>::run(kf, qf[1][ks], s[1][kb]);
            }
        }
        // ---- online softmax (log2 domain) --------------------------------------------------------------
        // The softmax is the VALU-bound part of the kernel (32 scores per lane per tile against 32 MFMAs), so
        // full tiles take a mask-free path, exp2 is the raw v_exp_f32 (arguments <= 0: no range fix-up
        // needed).
        uint4 pfrag[2][NKB / 2];
        const bool need_mask = a.causal_br || (t0 + kBKV > kvl);   // wave-uniform
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            // running max m_run is kept in RAW score units; p = exp2(s*sl2 - m*sl2) is one FMA + v_exp_f32
            float tmax = -1e30f;
            if (need_mask) {
#pragma unroll
                for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int kg = t0 + kb * 16 + grp * 4 + r;
                        const bool ok = kg < kvl && (!a.causal_br || kg <= qi[qb] + shift);
                        const float sv = ok ? s[qb][kb][r] : -1e30f;
                        s[qb][kb][r] = sv;
                        tmax = fmaxf(tmax, sv);
                    }
            } else {
#pragma unroll
                for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) tmax = fmaxf(tmax, s[qb][kb][r]);
            }
            tmax = fmaxf(tmax, __shfl_xor(tmax, 16, 64));
            tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
            const float m_new = fmaxf(m_run[qb], tmax);
            const float alpha = __builtin_amdgcn_exp2f((m_run[qb] - m_new) * sl2);
            // masked scores sit at -1e30, so exp2((s - m)*sl2) is exactly 0 for them — unless the whole row is
            // masked (m = -1e30 too); subtracting at least -1e29 keeps that case at exp2(-huge) = 0 as well
            const float nm = -fmaxf(m_new, -1e29f) * sl2;
            float lsum = 0.f;
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float p = __builtin_amdgcn_exp2f(fmaf(s[qb][kb][r], sl2, nm));
                    s[qb][kb][r] = p;
                    lsum += p;
                }
            lsum += __shfl_xor(lsum, 16, 64);
            lsum += __shfl_xor(lsum, 32, 64);
            l_run[qb] = l_run[qb] * alpha + lsum;
            m_run[qb] = m_new;
#pragma unroll
            for (int i = 0; i < NDB; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) o[qb][i][r] *= alpha;
#pragma unroll
            for (int kp2 = 0; kp2 < NKB / 2; ++kp2) {
                float pf[8];
#pragma unroll
                for (int r = 0; r < 4; ++r) { pf[r] = s[qb][2 * kp2][r]; pf[4 + r] = s[qb][2 * kp2 + 1][r]; }
                pfrag[qb][kp2] = pack<T>(pf);
            }
        }
        // ---- O^T += V^T P^T: V^T fragments by transposed LDS reads, shared by both query blocks --------
#pragma unroll
        for (int kp2 = 0; kp2 < NKB / 2; ++kp2) {
#pragma unroll
            for (int db = 0; db < NDB; ++db) {
                const int kr0 = (2 * kp2) * 16 + grp * 4 + (l15 >> 2);
                const int dcol = db * 16 + (l15 & 3) * 4;
                const v4s_t v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s_t*)(Vb + kr0 * ROWB + dcol * 2));
                const v4s_t v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s_t*)(Vb + (kr0 + 16) * ROWB + dcol * 2));
                const uint2 w0 = __builtin_bit_cast(uint2, v0), w1 = __builtin_bit_cast(uint2, v1);
                const uint4 vfrag = make_uint4(w0.x, w0.y, w1.x, w1.y);
                o[0][db] = AttnMma<T>::run(vfrag, pfrag[0][kp2], o[0][db]);
                o[1][db] = AttnMma<T>::run(vfrag, pfrag[1][kp2], o[1][db]);
            }
        }
    }

#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        if (qi[qb] >= a.q_len) continue;
        const float inv = l_run[qb] > 0.f ? 1.0f / l_run[qb] : 0.f;
        T* op = (T*)a.out + (int64_t)b * a.o_sb + (int64_t)h * a.o_sh + (int64_t)qi[qb] * a.o_ss;
#pragma unroll
        for (int db = 0; db < NDB; ++db) {
            const int d = db * 16 + grp * 4;
            if (d + 3 < hd && ((a.o_ss | a.o_sh | a.o_sb) & 3) == 0) {
                float pk[8] = {o[qb][db][0] * inv, o[qb][db][1] * inv, o[qb][db][2] * inv, o[qb][db][3] * inv, 0, 0, 0, 0};
                const uint4 u = pack<T>(pk);
                *reinterpret_cast<uint2*>(op + d) = make_uint2(u.x, u.y);
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (d + r < hd) Tr<T>::st(op + d + r, o[qb][db][r] * inv);
            }
        }
    }
}
#endif  // SS_EXPERIMENTAL

// --------------------------------------------------------------------------------------------
// (1c) flash attention v3 (bf16 / f16): the v2 data path (32 query rows per wave, K/V tiles by LDS-DMA in a ring,
//   swizzled K, V^T fragments by ds_read_b64_tr_b16) with the softmax re-cut around what the counters showed limits
//   v2 at head_dim 64 (rocprofv3: 288 VALU instructions per 32 MFMAs, VALU busy 82 % of the wave's issue time, MFMA
//   29 %, half of the LDS cycles bank conflicts):
//   * deferred rescale (guide T13): the running max is only raised — and O rescaled — when some row's tile maximum
//     exceeds it by more than 2^8; the check is a lane-local max + one wave vote, the cross-lane max reduction and
//     the 32 accumulator multiplies happen only on that (rare after the first tiles) path.  P <= 2^8 is exact
//     enough in bf16/f16 (relative precision unchanged) and the final 1/l normalisation removes the common factor;
//   * the row sum l is accumulated by the matrix cores: one extra MFMA per P fragment against an all-ones operand
//     (every row of that 16x16 product is sum_k P[q][k]) instead of 32 VALU adds + 4 cross-lane shuffles per tile;
//     l therefore sums the ROUNDED probabilities, i.e. exactly the weights the PV product uses;
//   * V image XOR-swizzled in 32-byte units by key row (source-side permutation in the DMA, same XOR on the
//     transposed read), so the 8 key rows a half-wave touches per ds_read_b64_tr_b16 land on 8 different bank octets.
// --------------------------------------------------------------------------------------------
__device__ __forceinline__ float max3_asm(float a, float b, float c) {
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
template <typename T> struct OnesFrag;
template <> struct OnesFrag<bf16_t> { static constexpr uint32_t kPair = 0x3F803F80u; };
template <> struct OnesFrag<f16_t> { static constexpr uint32_t kPair = 0x3C003C00u; };

template <typename T, int HD, bool VSWZ>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(HD <= 64 ? 3 : 2)))
void flash_attn3_kernel(const AttnArgs a) {
    constexpr int V = 8;
    constexpr int BQ2 = 128;             // query rows per block: 4 waves x 32
    constexpr int CPR = HD / V;          // 16-byte chunks per key row
    constexpr int KPI = 64 / CPR;        // keys per DMA instruction
    constexpr int NI = kBKV / KPI;       // DMA instructions per tile per operand
    constexpr int IPW = NI / 4;          // ... per wave
    constexpr int ROWB = HD * 2;         // bytes per key row
    constexpr int TILE_B = kBKV * ROWB;  // bytes of one K (or V) tile
    constexpr int NDB = HD / 16, NKB = kBKV / 16, NKS = HD / 32;
    constexpr float THR_L2 = 8.0f;       // deferred-rescale threshold, log2 units
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, grp = lane >> 4;
    // Workgroup -> (batch*head, query block).  Workgroups are dealt to the 8 XCDs round-robin by linear id and every XCD
    // has its own 4 MB L2: with the plain (x = query block, y = head) order the 32 query blocks that share one head's
    // K/V (1 MB at 4096 x 64) land on all eight L2s and every XCD streams every head (PMC: 4.5x the algorithmic bytes
    // from the fabric, L2 hit 74 %).  With a 1-D grid, XCD k owns heads k, k+8, ... and walks each head's query blocks
    // back to back, so a head's K/V is fetched into ONE L2 once.
    int bh, qblk;
    if (gridDim.y == 1 && a.xcd_heads > 0) {
        const int id = blockIdx.x, xcd = id & 7, j = id >> 3;
        const int nqb = a.xcd_heads;                 // query blocks per head
        bh = (j / nqb) * 8 + xcd;
        qblk = j - (j / nqb) * nqb;
    } else {
        bh = blockIdx.y; qblk = blockIdx.x;
    }
    const int b = bh / a.n_heads, h = bh % a.n_heads;
    const int kvl = a.ragged ? a.kv_len_b[b] : a.kv_len;      // per-batch key count (stacked story slots)
    const int q0 = qblk * BQ2;
    const int hd = a.hd;
    const T* __restrict__ qp = (const T*)a.q + (int64_t)b * a.q_sb + (int64_t)h * a.q_sh;
    const T* __restrict__ kp = (const T*)a.k + (int64_t)b * a.k_sb + (int64_t)h * a.k_sh;
    const T* __restrict__ vp = (const T*)a.v + (int64_t)b * a.v_sb + (int64_t)h * a.v_sh;
    const int shift = kvl - a.q_len;
    const T* zero = reinterpret_cast<const T*>(g_attn_zero_page);
    const uint32_t lds0 = __builtin_amdgcn_readfirstlane((uint32_t)(size_t)(lds_void2_t*)smem_raw);
    const float sl2 = a.scale * 1.4426950408889634f;   // scores kept in the log2 domain
    const float thr_raw = THR_L2 / fmaxf(sl2, 1e-30f);

    int qi[2];
    uint4 qf[2][NKS];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        qi[qb] = q0 + wid * 32 + qb * 16 + l15;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            const int d = ks * 32 + grp * 8;
            qf[qb][ks] = (qi[qb] < a.q_len && d < hd) ? ld16(qp + (int64_t)qi[qb] * a.q_ss + d) : make_uint4(0, 0, 0, 0);
        }
    }
    f32x4_t o[2][NDB], osum[2];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        osum[qb] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < NDB; ++i) o[qb][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    }
    float m_run[2] = {-1e30f, -1e30f};
    const uint4 ones = make_uint4(OnesFrag<T>::kPair, OnesFrag<T>::kPair, OnesFrag<T>::kPair, OnesFrag<T>::kPair);

    int kv_end = kvl;
    if (a.causal_br) {
        const int lim = q0 + BQ2 - 1 + shift + 1;
        if (lim < kv_end) kv_end = lim;
        if (kv_end < 0) kv_end = 0;
    }
    const int ntiles = (kv_end + kBKV - 1) / kBKV;

    // V swizzle: 32-byte chunk c of key row r is stored at chunk c ^ vsw(r); rows 128 B apart (head_dim 64) alias
    // in pairs, rows 256 B apart (head_dim 128) all alias
    auto vsw = [](int r) { return VSWZ ? (HD <= 64 ? ((r >> 1) & 3) : (r & 7)) : 0; };

    // DMA coordinates of this lane: key row within the instruction and physical chunk
    const int skey = lane / CPR, spc = lane % CPR;
    auto issue_tile = [&](int t, int buf) {
        const uint32_t kbase = lds0 + (uint32_t)(buf * 2 * TILE_B);
        const uint32_t vbase = kbase + TILE_B;
#pragma unroll
        for (int i = 0; i < IPW; ++i) {
            const int key = (wid * IPW + i) * KPI + skey;       // key row inside the tile
            const int kg = t * kBKV + key;
            const int lc = spc ^ (key & (CPR - 1));               // K: 16-byte chunks swizzled on the source side
            const int lv = ((((spc >> 1) ^ vsw(key)) << 1) | (spc & 1));   // V: 32-byte chunks
            const bool okk = kg < kvl && lc * V < hd;
            const bool okv = kg < kvl && lv * V < hd;
            const T* ks = okk ? kp + (int64_t)kg * a.k_ss + lc * V : zero;
            const T* vs = okv ? vp + (int64_t)kg * a.v_ss + lv * V : zero;
            const uint32_t roff = (uint32_t)((wid * IPW + i) * KPI * ROWB);
            attn_dma16(ks, __builtin_amdgcn_readfirstlane(kbase + roff));
            attn_dma16(vs, __builtin_amdgcn_readfirstlane(vbase + roff));
        }
    };

    // per-lane byte offsets of the transposed V reads: key row (grp*4 + l15/4) (+32*kp2, +16), column block db
    const int vrow = grp * 4 + (l15 >> 2);
    uint32_t voff[NDB];
#pragma unroll
    for (int db = 0; db < NDB; ++db) voff[db] = (uint32_t)(vrow * ROWB + ((db ^ vsw(vrow)) << 5) + (l15 & 3) * 8);

    constexpr int NSTAGE = HD <= 64 ? 3 : 2;
    constexpr int DPT = 2 * IPW;               // DMA instructions per tile per wave (K and V)
#pragma unroll
    for (int st = 0; st < NSTAGE - 1; ++st)
        if (st < ntiles) issue_tile(st, st);
    int stage = 0;
    for (int t = 0; t < ntiles; ++t) {
        if (NSTAGE == 3 && t + 1 < ntiles) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DPT) : "memory");   // tile t landed, t+1 may fly
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                     // ... for every wave; the stage of tile t-1 is drained
        if (t + NSTAGE - 1 < ntiles) {
            int ns = stage + NSTAGE - 1;
            if (ns >= NSTAGE) ns -= NSTAGE;
            issue_tile(t + NSTAGE - 1, ns);
        }
        const char* Kb = smem_raw + stage * 2 * TILE_B;
        const char* Vb = Kb + TILE_B;
        if (++stage == NSTAGE) stage = 0;
        const int t0 = t * kBKV;

        // ---- S^T = K Q^T for both 16-row query blocks ---------------------------------------------
        f32x4_t s[2][NKB];
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {
            s[0][kb] = f32x4_t{0.f, 0.f, 0.f, 0.f};
            s[1][kb] = f32x4_t{0.f, 0.f, 0.f, 0.f};
            const int r = kb * 16 + l15;
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {
                const uint4 kf = *reinterpret_cast<const uint4*>(Kb + r * ROWB + (((ks * 4 + grp) ^ (r & (CPR - 1))) << 4));
                s[0][kb] = AttnMma<T>::run(kf, qf[0][ks], s[0][kb]);
                s[1][kb] = AttnMma<T>::run(kf, qf[1][ks], s[1][kb]);
            }
        }
        // ---- softmax numerators (log2 domain, deferred rescale) ----------------------------------------
        uint4 pfrag[2][NKB / 2];
        const bool need_mask = a.causal_br || (t0 + kBKV > kvl);   // wave-uniform
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            if (need_mask) {
#pragma unroll
                for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int kg = t0 + kb * 16 + grp * 4 + r;
                        const bool ok = kg < kvl && (!a.causal_br || kg <= qi[qb] + shift);
                        s[qb][kb][r] = ok ? s[qb][kb][r] : -1e30f;
                    }
            }
            // lane-local maximum of the 16 scores: 8 x v_max3_f32.  Written in asm because fmaxf() on MFMA results makes
            // hipcc emit a canonicalising v_max per operand (12 extra VALU per query block); the first statement carries
            // the wait states an MFMA result needs before a VALU read (asm is opaque to the hazard recognizer)
            float tmax;
            asm volatile("s_nop 7\n\ts_nop 7\n\tv_max3_f32 %0, %1, %2, %3"
                         : "=v"(tmax) : "v"(s[qb][0][0]), "v"(s[qb][0][1]), "v"(s[qb][0][2]));
            tmax = max3_asm(tmax, s[qb][0][3], s[qb][1][0]);
#pragma unroll
            for (int kb = 1; kb < NKB; ++kb) {
                tmax = max3_asm(tmax, s[qb][kb][1], s[qb][kb][2]);
                if (kb + 1 < NKB) tmax = max3_asm(tmax, s[qb][kb][3], s[qb][kb + 1][0]);
                else tmax = max3_asm(tmax, s[qb][kb][3], s[qb][kb][3]);
            }
            if (__any(tmax > m_run[qb] + thr_raw)) {     // some row outgrew its reference maximum: raise it, rescale O and l
                tmax = fmaxf(tmax, __shfl_xor(tmax, 16, 64));
                tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
                const float m_new = fmaxf(m_run[qb], tmax);
                const float alpha = __builtin_amdgcn_exp2f((m_run[qb] - m_new) * sl2);
                m_run[qb] = m_new;
#pragma unroll
                for (int i = 0; i < NDB; ++i)
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[qb][i][r] *= alpha;
#pragma unroll
                for (int r = 0; r < 4; ++r) osum[qb][r] *= alpha;
            }
            // masked scores sit at -1e30, so exp2((s - m)*sl2) is exactly 0 for them — unless the whole row is
            // masked so far (m = -1e30 too); subtracting at least -1e29 keeps that case at exp2(-huge) = 0 as well
            const float nm = -fmaxf(m_run[qb], -1e29f) * sl2;
#pragma unroll
            for (int kp2 = 0; kp2 < NKB / 2; ++kp2) {
                float pf[8];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    pf[r] = __builtin_amdgcn_exp2f(fmaf(s[qb][2 * kp2][r], sl2, nm));
                    pf[4 + r] = __builtin_amdgcn_exp2f(fmaf(s[qb][2 * kp2 + 1][r], sl2, nm));
                }
                pfrag[qb][kp2] = pack<T>(pf);
            }
        }
        // ---- O^T += V^T P^T and l += 1^T P^T: V^T fragments by transposed LDS reads, shared by both query blocks --------
#pragma unroll
        for (int kp2 = 0; kp2 < NKB / 2; ++kp2) {
            osum[0] = AttnMma<T>::run(ones, pfrag[0][kp2], osum[0]);
            osum[1] = AttnMma<T>::run(ones, pfrag[1][kp2], osum[1]);
#pragma unroll
            for (int db = 0; db < NDB; ++db) {
                const char* vb = Vb + voff[db] + kp2 * 32 * ROWB;
                const v4s_t v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s_t*)(vb));
                const v4s_t v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s_t*)(vb + 16 * ROWB));
                const uint2 w0 = __builtin_bit_cast(uint2, v0), w1 = __builtin_bit_cast(uint2, v1);
                const uint4 vfrag = make_uint4(w0.x, w0.y, w1.x, w1.y);
                o[0][db] = AttnMma<T>::run(vfrag, pfrag[0][kp2], o[0][db]);
                o[1][db] = AttnMma<T>::run(vfrag, pfrag[1][kp2], o[1][db]);
            }
        }
    }

#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        if (qi[qb] >= a.q_len) continue;
        const float l = osum[qb][0];         // every row of the ones-product is the row sum
        const float inv = l > 0.f ? 1.0f / l : 0.f;
        T* op = (T*)a.out + (int64_t)b * a.o_sb + (int64_t)h * a.o_sh + (int64_t)qi[qb] * a.o_ss;
#pragma unroll
        for (int db = 0; db < NDB; ++db) {
            const int d = db * 16 + grp * 4;
            if (d + 3 < hd && ((a.o_ss | a.o_sh | a.o_sb) & 3) == 0) {
                float pk[8] = {o[qb][db][0] * inv, o[qb][db][1] * inv, o[qb][db][2] * inv, o[qb][db][3] * inv, 0, 0, 0, 0};
                const uint4 u = pack<T>(pk);
                *reinterpret_cast<uint2*>(op + d) = make_uint2(u.x, u.y);
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (d + r < hd) Tr<T>::st(op + d + r, o[qb][db][r] * inv);
            }
        }
    }
}

// --------------------------------------------------------------------------------------------
// (1d) flash attention v3p (head_dim <= 64).  attn_ver = 6 (addresses only, PIPE = false) is the SHIPPED DEFAULT since round 5
//   (validated on MI355X: bit-equal to v3 on every shape class of tests/test_kernels_gpu.py::test_flash_v3p_equals_v3 and against
//   the fp32 oracle in test_sdxl_attention_shapes_bf16; 863 -> 837 us at (16, 10 heads, 4096), 116.5 -> 112.1 us at (16, 20, 1024));
//   attn_ver = 5 (PIPE) stays an option: measured SLOWER than v3 (859 / 122 us) — the 204-VGPR form drops to 2 waves per SIMD.
//   Same data path, softmax and fragment layouts as v3 with the two changes the instruction budget of v3 asks for
//   (per 64-key tile and wave 1152 matrix-pipe cycles against ~1190 VALU cycles, MFMA busy 0.34-0.41):
//   * S(t+1) = K(t+1) Q^T is ISSUED before the softmax of tile t: the 16 S MFMAs of the next tile run on the matrix pipe
//     while this wave's VALU works through the 32 exp2 / max3 / packs of the current one (two named score sets, the loop
//     is unrolled by two so that neither set is indexed at run time).  K(t+1) therefore has to have landed one tile
//     earlier than in v3: the wave drains its DMA counter at the top of every iteration (the tile it waits for was
//     issued one full iteration before);
//   * the DMA source addresses are (per-lane base, loop-invariant) + (wave-uniform tile advance): v3 recomputes the 64-bit
//     byte offset of every key row per tile with quarter-rate integer multiplies (~270 VALU cycles per tile).
//   Bit-for-bit the same arithmetic per score as v3 (same MFMA order inside S, softmax and PV), so results are equal.
//   PIPE = false keeps v3's loop (S(t) inside iteration t, tile t+1 still in flight at the barrier, 3 waves per SIMD) and
//   takes only the address change — the two effects can be measured apart; PIPE = true needs 204 VGPRs (2 waves per SIMD).
// --------------------------------------------------------------------------------------------
// NW (round 6): waves per workgroup.  4 = 128 query rows per workgroup (3 workgroups per CU); 8 = 256 query rows per workgroup
// sharing every K / V tile (2 workgroups per CU = 4 waves per SIMD): per-wave arithmetic unchanged (bit-identical results), half
// the L2 -> LDS traffic per flop — at 4096 tokens x 10 heads x batch 16 the 4-wave form pulls 32 q-blocks x 64 tiles x 16 KB x
// 160 heads = 5.2 GB per launch through the L2s in 434 us = 12 TB/s, which IS the aggregate L2 rate of the part.
template <typename T, int HD, bool PIPE, int NW = 4>
__global__ __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(PIPE ? 2 : (NW >= 8 ? 4 : 3))))
void flash_attn3p_kernel(const AttnArgs a) {
    static_assert(HD == 64, "v3p: head_dim <= 64");
    static_assert(NW == 4 || NW == 8 || NW == 16, "waves per workgroup");
    constexpr int V = 8;
    constexpr int BQ2 = 32 * NW;         // query rows per block: NW waves x 32
    constexpr int CPR = HD / V;          // 16-byte chunks per key row
    constexpr int KPI = 64 / CPR;        // keys per DMA instruction
    constexpr int NI = kBKV / KPI;       // DMA instructions per tile per operand
    constexpr int IPW = NI / NW > 0 ? NI / NW : 1;   // ... per wave (NW = 16: only the first NI waves stage, one instruction each)
    constexpr int ROWB = HD * 2;         // bytes per key row
    constexpr int TILE_B = kBKV * ROWB;  // bytes of one K (or V) tile
    constexpr int NDB = HD / 16, NKB = kBKV / 16, NKS = HD / 32;
    constexpr int NSTAGE = 3;
    constexpr float THR_L2 = 8.0f;       // deferred-rescale threshold, log2 units
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, grp = lane >> 4;
    int bh, qblk;
    if (gridDim.y == 1 && a.xcd_heads > 0) {      // XCD-aware 1-D grid, as v3
        const int id = blockIdx.x, xcd = id & 7, j = id >> 3;
        const int nqb = a.xcd_heads;
        bh = (j / nqb) * 8 + xcd;
        qblk = j - (j / nqb) * nqb;
    } else {
        bh = blockIdx.y; qblk = blockIdx.x;
    }
    const int b = bh / a.n_heads, h = bh % a.n_heads;
    const int kvl = a.ragged ? a.kv_len_b[b] : a.kv_len;
    const int q0 = qblk * BQ2;
    const int hd = a.hd;
    const T* __restrict__ qp = (const T*)a.q + (int64_t)b * a.q_sb + (int64_t)h * a.q_sh;
    const T* __restrict__ kp = (const T*)a.k + (int64_t)b * a.k_sb + (int64_t)h * a.k_sh;
    const T* __restrict__ vp = (const T*)a.v + (int64_t)b * a.v_sb + (int64_t)h * a.v_sh;
    const int shift = kvl - a.q_len;
    const T* zero = reinterpret_cast<const T*>(g_attn_zero_page);
    const uint32_t lds0 = __builtin_amdgcn_readfirstlane((uint32_t)(size_t)(lds_void2_t*)smem_raw);
    const float sl2 = a.scale * 1.4426950408889634f;   // scores kept in the log2 domain
    const float thr_raw = THR_L2 / fmaxf(sl2, 1e-30f);

    int qi[2];
    uint4 qf[2][NKS];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        qi[qb] = q0 + wid * 32 + qb * 16 + l15;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            const int d = ks * 32 + grp * 8;
            qf[qb][ks] = (qi[qb] < a.q_len && d < hd) ? ld16(qp + (int64_t)qi[qb] * a.q_ss + d) : make_uint4(0, 0, 0, 0);
        }
    }
    f32x4_t o[2][NDB], osum[2];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        osum[qb] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < NDB; ++i) o[qb][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    }
    float m_run[2] = {-1e30f, -1e30f};
    const uint4 ones = make_uint4(OnesFrag<T>::kPair, OnesFrag<T>::kPair, OnesFrag<T>::kPair, OnesFrag<T>::kPair);

    int kv_end = kvl;
    if (a.causal_br) {
        const int lim = q0 + BQ2 - 1 + shift + 1;
        if (lim < kv_end) kv_end = lim;
        if (kv_end < 0) kv_end = 0;
    }
    const int ntiles = (kv_end + kBKV - 1) / kBKV;
    if (ntiles == 0) {                   // (causal blocks with no visible key: v3 stores zeros through its epilogue, so do we)
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            if (qi[qb] >= a.q_len) continue;
            T* op = (T*)a.out + (int64_t)b * a.o_sb + (int64_t)h * a.o_sh + (int64_t)qi[qb] * a.o_ss;
#pragma unroll
            for (int db = 0; db < NDB; ++db)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int d = db * 16 + grp * 4 + r;
                    if (d < hd) Tr<T>::st(op + d, 0.f);
                }
        }
        return;
    }

    auto vsw = [](int r) { return (r >> 1) & 3; };       // V image: 32-byte chunk c of key row r at chunk c ^ vsw(r)

    // ---- DMA coordinates: everything that depends on the lane is loop-invariant -----------------------------------
    const int skey = lane / CPR, spc = lane % CPR;
    int dkey[IPW];
    const T* kbase_l[IPW];
    const T* vbase_l[IPW];
    bool kin[IPW], vin[IPW];
#pragma unroll
    for (int i = 0; i < IPW; ++i) {
        const int key = (wid * IPW + i) * KPI + skey;           // key row inside a tile
        const int lc = spc ^ (key & (CPR - 1));                 // K: 16-byte chunks swizzled on the source side
        const int lv = ((((spc >> 1) ^ vsw(key)) << 1) | (spc & 1));   // V: 32-byte chunks
        dkey[i] = key;
        kin[i] = lc * V < hd;
        vin[i] = lv * V < hd;
        kbase_l[i] = kp + (int64_t)key * a.k_ss + lc * V;
        vbase_l[i] = vp + (int64_t)key * a.v_ss + lv * V;
    }
    auto issue_tile = [&](int t, int buf) {
        if constexpr (NW * IPW > NI) { if (wid * IPW >= NI) return; }   // wave-uniform: this wave stages nothing (its vmcnt stays 0)
        const uint32_t kbase = lds0 + (uint32_t)(buf * 2 * TILE_B);
        const uint32_t vbase = kbase + TILE_B;
        const int64_t kadv = (int64_t)t * kBKV * a.k_ss;        // wave-uniform
        const int64_t vadv = (int64_t)t * kBKV * a.v_ss;
        const int t0k = t * kBKV;
#pragma unroll
        for (int i = 0; i < IPW; ++i) {
            const bool inr = t0k + dkey[i] < kvl;
            const T* ks = (inr && kin[i]) ? kbase_l[i] + kadv : zero;
            const T* vs = (inr && vin[i]) ? vbase_l[i] + vadv : zero;
            const uint32_t roff = (uint32_t)((wid * IPW + i) * KPI * ROWB);
            attn_dma16(ks, __builtin_amdgcn_readfirstlane(kbase + roff));
            attn_dma16(vs, __builtin_amdgcn_readfirstlane(vbase + roff));
        }
    };

    const int vrow = grp * 4 + (l15 >> 2);
    uint32_t voff[NDB];
#pragma unroll
    for (int db = 0; db < NDB; ++db) voff[db] = (uint32_t)(vrow * ROWB + ((db ^ vsw(vrow)) << 5) + (l15 & 3) * 8);

    // S^T = K Q^T of the tile in ring slot `slot`, both 16-row query blocks
    auto compute_s = [&](int slot, f32x4_t (&s)[2][NKB]) {
        const char* Kb = smem_raw + slot * 2 * TILE_B;
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {
            s[0][kb] = f32x4_t{0.f, 0.f, 0.f, 0.f};
            s[1][kb] = f32x4_t{0.f, 0.f, 0.f, 0.f};
            const int r = kb * 16 + l15;
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {
                const uint4 kf = *reinterpret_cast<const uint4*>(Kb + r * ROWB + (((ks * 4 + grp) ^ (r & (CPR - 1))) << 4));
                s[0][kb] = AttnMma<T>::run(kf, qf[0][ks], s[0][kb]);
                s[1][kb] = AttnMma<T>::run(kf, qf[1][ks], s[1][kb]);
            }
        }
    };

    // softmax numerators of tile t from its scores + O^T += V^T P^T, l += 1^T P^T (v3's statements, unchanged)
    auto softmax_pv = [&](int t, int slot, f32x4_t (&s)[2][NKB]) {
        const char* Vb = smem_raw + slot * 2 * TILE_B + TILE_B;
        const int t0 = t * kBKV;
        uint4 pfrag[2][NKB / 2];
        const bool need_mask = a.causal_br || (t0 + kBKV > kvl);   // wave-uniform
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            if (need_mask) {
#pragma unroll
                for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int kg = t0 + kb * 16 + grp * 4 + r;
                        const bool ok = kg < kvl && (!a.causal_br || kg <= qi[qb] + shift);
                        s[qb][kb][r] = ok ? s[qb][kb][r] : -1e30f;
                    }
            }
            float tmax;
            asm volatile("s_nop 7\n\ts_nop 7\n\tv_max3_f32 %0, %1, %2, %3"
                         : "=v"(tmax) : "v"(s[qb][0][0]), "v"(s[qb][0][1]), "v"(s[qb][0][2]));
            tmax = max3_asm(tmax, s[qb][0][3], s[qb][1][0]);
#pragma unroll
            for (int kb = 1; kb < NKB; ++kb) {
                tmax = max3_asm(tmax, s[qb][kb][1], s[qb][kb][2]);
                if (kb + 1 < NKB) tmax = max3_asm(tmax, s[qb][kb][3], s[qb][kb + 1][0]);
                else tmax = max3_asm(tmax, s[qb][kb][3], s[qb][kb][3]);
            }
            if (__any(tmax > m_run[qb] + thr_raw)) {
                tmax = fmaxf(tmax, __shfl_xor(tmax, 16, 64));
                tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
                const float m_new = fmaxf(m_run[qb], tmax);
                const float alpha = __builtin_amdgcn_exp2f((m_run[qb] - m_new) * sl2);
                m_run[qb] = m_new;
#pragma unroll
                for (int i = 0; i < NDB; ++i)
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[qb][i][r] *= alpha;
#pragma unroll
                for (int r = 0; r < 4; ++r) osum[qb][r] *= alpha;
            }
            const float nm = -fmaxf(m_run[qb], -1e29f) * sl2;
#pragma unroll
            for (int kp2 = 0; kp2 < NKB / 2; ++kp2) {
                float pf[8];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    pf[r] = __builtin_amdgcn_exp2f(fmaf(s[qb][2 * kp2][r], sl2, nm));
                    pf[4 + r] = __builtin_amdgcn_exp2f(fmaf(s[qb][2 * kp2 + 1][r], sl2, nm));
                }
                pfrag[qb][kp2] = pack<T>(pf);
            }
        }
#pragma unroll
        for (int kp2 = 0; kp2 < NKB / 2; ++kp2) {
            osum[0] = AttnMma<T>::run(ones, pfrag[0][kp2], osum[0]);
            osum[1] = AttnMma<T>::run(ones, pfrag[1][kp2], osum[1]);
#pragma unroll
            for (int db = 0; db < NDB; ++db) {
                const char* vb = Vb + voff[db] + kp2 * 32 * ROWB;
                const v4s_t v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s_t*)(vb));
                const v4s_t v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s_t*)(vb + 16 * ROWB));
                const uint2 w0 = __builtin_bit_cast(uint2, v0), w1 = __builtin_bit_cast(uint2, v1);
                const uint4 vfrag = make_uint4(w0.x, w0.y, w1.x, w1.y);
                o[0][db] = AttnMma<T>::run(vfrag, pfrag[0][kp2], o[0][db]);
                o[1][db] = AttnMma<T>::run(vfrag, pfrag[1][kp2], o[1][db]);
            }
        }
    };

    if constexpr (!PIPE) {               // v3's loop order with the loop-invariant DMA addresses
        issue_tile(0, 0);
        if (1 < ntiles) issue_tile(1, 1);
        int stage = 0;
        f32x4_t sc[2][NKB];
        for (int t = 0; t < ntiles; ++t) {
            if (t + 1 < ntiles) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * IPW) : "memory");   // tile t landed, t+1 may fly
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (t + NSTAGE - 1 < ntiles) {
                int ns = stage + NSTAGE - 1;
                if (ns >= NSTAGE) ns -= NSTAGE;
                issue_tile(t + NSTAGE - 1, ns);
            }
            compute_s(stage, sc);
            softmax_pv(t, stage, sc);
            if (++stage == NSTAGE) stage = 0;
        }
    } else {
    // ---- prologue: tiles 0 and 1 requested, S(0) computed ----------------------------------------------------------
    issue_tile(0, 0);
    if (1 < ntiles) issue_tile(1, 1);
    if (1 < ntiles) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * IPW) : "memory");     // tile 0 landed, tile 1 may fly
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    f32x4_t sA[2][NKB], sB[2][NKB];
    compute_s(0, sA);

    // iteration t (scores of tile t in `cur`): K(t+1) landed for every wave -> tile t+2 requested into the slot tile t-1
    // left -> S(t+1) issued into `nxt` -> softmax(t), PV(t)
    auto iteration = [&](int t, int slot, f32x4_t (&cur)[2][NKB], f32x4_t (&nxt)[2][NKB]) {
        const bool more = t + 1 < ntiles;
        int s1 = slot + 1; if (s1 >= NSTAGE) s1 -= NSTAGE;
        int s2 = s1 + 1;   if (s2 >= NSTAGE) s2 -= NSTAGE;
        if (more) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces of tile t+1 (its only outstanding DMA) landed
            __syncthreads();                                   // ... and everybody's; iteration t-1 is finished everywhere
            if (t + 2 < ntiles) issue_tile(t + 2, s2);
            compute_s(s1, nxt);
        }
        softmax_pv(t, slot, cur);
    };
    int t = 0, slot = 0;
    while (true) {
        iteration(t, slot, sA, sB);
        if (++t >= ntiles) break;
        if (++slot == NSTAGE) slot = 0;
        iteration(t, slot, sB, sA);
        if (++t >= ntiles) break;
        if (++slot == NSTAGE) slot = 0;
    }
    }

#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        if (qi[qb] >= a.q_len) continue;
        const float l = osum[qb][0];         // every row of the ones-product is the row sum
        const float inv = l > 0.f ? 1.0f / l : 0.f;
        T* op = (T*)a.out + (int64_t)b * a.o_sb + (int64_t)h * a.o_sh + (int64_t)qi[qb] * a.o_ss;
#pragma unroll
        for (int db = 0; db < NDB; ++db) {
            const int d = db * 16 + grp * 4;
            if (d + 3 < hd && ((a.o_ss | a.o_sh | a.o_sb) & 3) == 0) {
                float pk[8] = {o[qb][db][0] * inv, o[qb][db][1] * inv, o[qb][db][2] * inv, o[qb][db][3] * inv, 0, 0, 0, 0};
                const uint4 u = pack<T>(pk);
                *reinterpret_cast<uint2*>(op + d) = make_uint2(u.x, u.y);
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (d + r < hd) Tr<T>::st(op + d + r, o[qb][db][r] * inv);
            }
        }
    }
}

#if SS_EXPERIMENTAL
// --------------------------------------------------------------------------------------------
// (1e) cross-attention against a SHORT context (round 5): head_dim 64, kv_len <= 64 — the UNet's attn2 (64 image-feature
//   tokens): 120 launches per forward.  Through the flash kernels such a launch is one K/V tile per workgroup: DMA the tile,
//   wait, S, softmax, PV, store — a chain of latencies with nothing to overlap, 70.9 us for 84 MB of Q-in / O-out = 1.2 TB/s
//   (profiles/round4_unet_batch16_kernel_trace.txt: 8.5 ms per batch-16 forward for 0.65 TFLOP).  The whole context of a
//   (batch, head) pair is 8 KB of K and 8 KB of V, so here it lives in REGISTERS as ready MFMA fragments (K: 8 x 16 B as the
//   first operand of S^T = K Q^T; V^T: 8 x 16 B in the key order the P fragments come out in) and a workgroup STREAMS query
//   rows through it: a persistent grid of 2 workgroups per CU, each walking a contiguous range of (pair, 128-row chunk) work
//   items — the fragments are reloaded only when the pair changes — with the next item's Q rows requested before the current
//   item is computed.  HBM-bound by design: Q in, O out, nothing else.
//   Arithmetic per score identical to flash v3 / v3p with a single tile (same MFMA order in S and PV, true row maximum,
//   exp2 in the log2 domain, P rounded to the tensor dtype, row sums of the ROUNDED P by the ones-MFMA), so results are
//   bit-equal to those kernels (tests/test_kernels_gpu.py::test_cross_attention_kv64_equals_flash) and the fp32-oracle
//   tolerances carry over.
//   MEASURED ON MI355X AND NOT ADOPTED (profiles/round5_cross_attn_b16.json): the premise was a misreading of the kernel trace —
//   the 70.9 us average there mixes 60 self-attention launches (116 us) with 60 cross-attention launches, which the flash path
//   already runs at 26.3 us = 3.2 TB/s (1024 tokens x 20 heads, batch 16) and 43.1 us = 3.9 TB/s (4096 x 10).  This kernel is
//   bit-equal to it (42 equality cases passed) but SLOWER: 33.7 / 46.8 us — 8 waves per CU with one 4 KB query tile in flight
//   each do not cover the HBM latency the way 12 resident flash workgroups do.  Kept only in the EXPERIMENTAL=1 build
//   (tuning knob attn_cross64 = 1 selects it there; tests/test_experimental_gpu.py).
// --------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2)))
void cross_attn64_kernel(const AttnArgs a, int n_items, int items_per_block, int chunks_per_pair) {
    constexpr int HD = 64, NDB = 4, NKB = 4, NKS = 2;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, grp = lane >> 4;
    const float sl2 = a.scale * 1.4426950408889634f;
    const uint4 ones = make_uint4(OnesFrag<T>::kPair, OnesFrag<T>::kPair, OnesFrag<T>::kPair, OnesFrag<T>::kPair);
    const int kvl = a.kv_len;
    const bool o_vec = ((a.o_ss | a.o_sh | a.o_sb) & 3) == 0;
    int it = blockIdx.x * items_per_block;
    int it_end = it + items_per_block;
    if (it_end > n_items) it_end = n_items;
    if (it >= it_end) return;

    uint4 kf[NKB][NKS];      // K[key = kb*16 + l15][d = ks*32 + grp*8 ..]
    uint4 vf[NKB / 2][NDB];  // V[key in the P-fragment order of (kp2, grp)][d = db*16 + l15], 8 keys per lane
    int cur_pair = -1;
    auto load_q = [&](int item, uint4 (&qf)[2][NKS]) {
        const int pair = item / chunks_per_pair, chunk = item - pair * chunks_per_pair;
        const int b = pair / a.n_heads, h = pair - b * a.n_heads;
        const T* __restrict__ qp = (const T*)a.q + (int64_t)b * a.q_sb + (int64_t)h * a.q_sh;
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            const int qi = chunk * 128 + wid * 32 + qb * 16 + l15;
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks)
                qf[qb][ks] = qi < a.q_len ? ld16(qp + (int64_t)qi * a.q_ss + ks * 32 + grp * 8) : make_uint4(0, 0, 0, 0);
        }
    };
    uint4 qf[2][NKS], qn[2][NKS];
    load_q(it, qf);
    for (; it < it_end; ++it) {
        const bool more = it + 1 < it_end;           // workgroup-uniform
        if (more) load_q(it + 1, qn);                // in flight under this item's work
        const int pair = it / chunks_per_pair, chunk = it - pair * chunks_per_pair;
        const int b = pair / a.n_heads, h = pair - b * a.n_heads;
        if (pair != cur_pair) {                      // (workgroup-uniform) the context of a new (batch, head) pair
            cur_pair = pair;
            const T* __restrict__ kp = (const T*)a.k + (int64_t)b * a.k_sb + (int64_t)h * a.k_sh;
            const T* __restrict__ vp = (const T*)a.v + (int64_t)b * a.v_sb + (int64_t)h * a.v_sh;
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb) {
                const int key = kb * 16 + l15;
#pragma unroll
                for (int ks = 0; ks < NKS; ++ks)
                    kf[kb][ks] = key < kvl ? ld16(kp + (int64_t)key * a.k_ss + ks * 32 + grp * 8) : make_uint4(0, 0, 0, 0);
            }
#pragma unroll
            for (int kp2 = 0; kp2 < NKB / 2; ++kp2)
#pragma unroll
                for (int db = 0; db < NDB; ++db) {
                    uint32_t w[4];
#pragma unroll
                    for (int j = 0; j < 8; j += 2) {
                        const int k0 = (2 * kp2 + (j >> 2)) * 16 + grp * 4 + (j & 3), k1 = k0 + 1;
                        const uint32_t lo = k0 < kvl ? (uint32_t)(vp + (int64_t)k0 * a.v_ss + db * 16 + l15)->v : 0u;
                        const uint32_t hi = k1 < kvl ? (uint32_t)(vp + (int64_t)k1 * a.v_ss + db * 16 + l15)->v : 0u;
                        w[j >> 1] = lo | (hi << 16);
                    }
                    vf[kp2][db] = make_uint4(w[0], w[1], w[2], w[3]);
                }
        }
        // ---- S^T = K Q^T ------------------------------------------------------------------------------
        f32x4_t s[2][NKB];
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {
            s[0][kb] = f32x4_t{0.f, 0.f, 0.f, 0.f};
            s[1][kb] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {
                s[0][kb] = AttnMma<T>::run(kf[kb][ks], qf[0][ks], s[0][kb]);
                s[1][kb] = AttnMma<T>::run(kf[kb][ks], qf[1][ks], s[1][kb]);
            }
        }
        // ---- softmax numerators: true row maximum (the four lane groups of a row), exp2 in the log2 domain ---------------
        uint4 pfrag[2][NKB / 2];
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            if (kvl < 64) {
#pragma unroll
                for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (kb * 16 + grp * 4 + r >= kvl) s[qb][kb][r] = -1e30f;
            }
            float tmax = -1e30f;
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
                for (int r = 0; r < 4; ++r) tmax = fmaxf(tmax, s[qb][kb][r]);
            tmax = fmaxf(tmax, __shfl_xor(tmax, 16, 64));
            tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
            const float nm = -fmaxf(tmax, -1e29f) * sl2;
#pragma unroll
            for (int kp2 = 0; kp2 < NKB / 2; ++kp2) {
                float pf[8];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    pf[r] = __builtin_amdgcn_exp2f(fmaf(s[qb][2 * kp2][r], sl2, nm));
                    pf[4 + r] = __builtin_amdgcn_exp2f(fmaf(s[qb][2 * kp2 + 1][r], sl2, nm));
                }
                pfrag[qb][kp2] = pack<T>(pf);
            }
        }
        // ---- O^T = V^T P^T, l = 1^T P^T ---------------------------------------------------------------------------------
        f32x4_t o[2][NDB], osum[2];
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            osum[qb] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < NDB; ++i) o[qb][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int kp2 = 0; kp2 < NKB / 2; ++kp2) {
            osum[0] = AttnMma<T>::run(ones, pfrag[0][kp2], osum[0]);
            osum[1] = AttnMma<T>::run(ones, pfrag[1][kp2], osum[1]);
#pragma unroll
            for (int db = 0; db < NDB; ++db) {
                o[0][db] = AttnMma<T>::run(vf[kp2][db], pfrag[0][kp2], o[0][db]);
                o[1][db] = AttnMma<T>::run(vf[kp2][db], pfrag[1][kp2], o[1][db]);
            }
        }
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            const int qi = chunk * 128 + wid * 32 + qb * 16 + l15;
            if (qi >= a.q_len) continue;
            const float l = osum[qb][0];
            const float inv = l > 0.f ? 1.0f / l : 0.f;
            T* op = (T*)a.out + (int64_t)b * a.o_sb + (int64_t)h * a.o_sh + (int64_t)qi * a.o_ss;
#pragma unroll
            for (int db = 0; db < NDB; ++db) {
                const int d = db * 16 + grp * 4;
                if (o_vec) {
                    float pk[8] = {o[qb][db][0] * inv, o[qb][db][1] * inv, o[qb][db][2] * inv, o[qb][db][3] * inv, 0, 0, 0, 0};
                    const uint4 u = pack<T>(pk);
                    *reinterpret_cast<uint2*>(op + d) = make_uint2(u.x, u.y);
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) Tr<T>::st(op + d + r, o[qb][db][r] * inv);
                }
            }
        }
        if (more) {
#pragma unroll
            for (int qb = 0; qb < 2; ++qb)
#pragma unroll
                for (int ks = 0; ks < NKS; ++ks) qf[qb][ks] = qn[qb][ks];
        }
    }
}

template <typename T>
static int cross_attn64_launch(const AttnArgs& a, int64_t batch, hipStream_t s) {
    const int cpp = cdiv(a.q_len, 128);
    const int64_t n_items = batch * a.n_heads * cpp;
    static int n_cu = 0;
    if (n_cu == 0) {
        int dev = 0;
        hipDeviceProp_t p;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess) n_cu = p.multiProcessorCount;
        if (n_cu <= 0) n_cu = 256;
    }
    const int64_t max_blocks = (int64_t)2 * n_cu;                 // 2 workgroups (8 waves) per CU: the kernel's occupancy
    const int ipb = (int)((n_items + max_blocks - 1) / max_blocks);
    const int grid = (int)((n_items + ipb - 1) / ipb);
    hipLaunchKernelGGL((cross_attn64_kernel<T>), dim3((unsigned)grid), dim3(256), 0, s, a, (int)n_items, ipb, cpp);
    SS_LAUNCH_CHECK("cross_attn64");
    return SS_OK;
}

#endif  // SS_EXPERIMENTAL (cross_attn64)

template <typename T, bool PIPE>
static int flash3p_launch(const AttnArgs& a, int64_t batch, hipStream_t s) {
    // ring of 3 (K, V) tiles — or as many as the context has: a one-tile context (the UNet's cross-attention, 64 image-feature tokens)
    // touches ring slot 0 only, and at 16 KB instead of 48 KB per workgroup the CU holds 4 workgroups (register-limited) instead of 3
    const int64_t kv_tiles = a.ragged ? 3 : cdiv(a.kv_len, kBKV);
    const size_t lds = (size_t)(kv_tiles < 3 && !PIPE ? (kv_tiles < 1 ? 1 : kv_tiles) : 3) * 2 * kBKV * 64 * 2;
    // 8 / 16 waves per workgroup (256 / 512 query rows share the K / V tiles) once there are enough such workgroups to fill the chip
    // four times; knob attn_waves: 0 = this rule, 4 / 8 / 16 = force.  Measured (tools/attn_waves_bench.py, profiles/round6_attn_waves.txt,
    // 4 / 8 / 16 waves): (16, 10 heads, 4096) 839 / 749 / 720 us; (16, 20, 1024) 120 / 112 / 124; (8, 10, 4096) 428 / 380 / 406;
    // (2, 10, 4096) 119 / 139 / 129 — small launches keep 4 waves.
    const int64_t bh = batch * a.n_heads;
    const int aw = tuning_get("attn_waves", 0);
    const bool many = a.ragged || a.kv_len >= 512;     // one- or two-tile contexts (the UNet's cross-attention) gain nothing from sharing tiles
    const bool w16 = !PIPE && (aw == 16 || (aw == 0 && many && a.q_len >= 2048 && (int64_t)cdiv(a.q_len, 512) * bh >= 1024));
    const bool w8 = !PIPE && !w16 && (aw == 8 || (aw == 0 && many && a.q_len >= 512 && (int64_t)cdiv(a.q_len, 256) * bh >= 1024));
    const int nqb = cdiv(a.q_len, w16 ? 512 : w8 ? 256 : 128);
    AttnArgs a2 = a;
    dim3 grid((unsigned)nqb, (unsigned)bh);
    a2.xcd_heads = 0;
    if (bh % 8 == 0 && nqb > 1 && tuning_get("attn_xcd", 1)) {
        a2.xcd_heads = nqb;
        grid = dim3((unsigned)(nqb * bh), 1);
    }
    if constexpr (!PIPE) {
        if (w16) {
            hipLaunchKernelGGL((flash_attn3p_kernel<T, 64, false, 16>), grid, dim3(1024), lds, s, a2);
            SS_LAUNCH_CHECK("flash_attn3p");
            return SS_OK;
        }
        if (w8) {
            hipLaunchKernelGGL((flash_attn3p_kernel<T, 64, false, 8>), grid, dim3(512), lds, s, a2);
            SS_LAUNCH_CHECK("flash_attn3p");
            return SS_OK;
        }
    }
    hipLaunchKernelGGL((flash_attn3p_kernel<T, 64, PIPE>), grid, dim3(256), lds, s, a2);
    SS_LAUNCH_CHECK("flash_attn3p");
    return SS_OK;
}

template <typename T, int HD, bool VSWZ>
static int flash3_launch_hd(const AttnArgs& a, int64_t batch, hipStream_t s) {
    const size_t lds = (size_t)(HD <= 64 ? 3 : 2) * 2 * kBKV * HD * 2;   // ring of (K, V) tiles
    const int nqb = cdiv(a.q_len, 128);
    const int64_t bh = batch * a.n_heads;
    AttnArgs a2 = a;
    dim3 grid((unsigned)nqb, (unsigned)bh);
    a2.xcd_heads = 0;
    if (bh % 8 == 0 && nqb > 1 && tuning_get("attn_xcd", 1)) {   // XCD k owns heads k, k+8, ...
        a2.xcd_heads = nqb;
        grid = dim3((unsigned)(nqb * bh), 1);
    }
    hipLaunchKernelGGL((flash_attn3_kernel<T, HD, VSWZ>), grid, dim3(256), lds, s, a2);
    SS_LAUNCH_CHECK("flash_attn3");
    return SS_OK;
}

#if SS_EXPERIMENTAL
template <typename T, int HD>
static int flash2_launch_hd(const AttnArgs& a, int64_t batch, hipStream_t s) {
    const size_t lds = (size_t)(HD <= 64 ? 3 : 2) * 2 * kBKV * HD * 2;   // ring of (K, V) tiles
    dim3 grid((unsigned)cdiv(a.q_len, 128), (unsigned)(batch * a.n_heads));
    hipLaunchKernelGGL((flash_attn2_kernel<T, HD>), grid, dim3(256), lds, s, a);
    SS_LAUNCH_CHECK("flash_attn2");
    return SS_OK;
}
#endif

template <typename T, int HD>
static int flash_launch_hd(const AttnArgs& a, int64_t batch, hipStream_t s) {
    constexpr int V = Tr<T>::kVec;
    const size_t lds = ((size_t)kBKV * (HD + V) + (size_t)HD * (kBKV + V)) * sizeof(T);
    dim3 grid((unsigned)cdiv(a.q_len, kBQ), (unsigned)(batch * a.n_heads));
    hipLaunchKernelGGL((flash_attn_kernel<T, HD>), grid, dim3(256), lds, s, a);
    SS_LAUNCH_CHECK("flash_attn");
    return SS_OK;
}

template <typename T>
int attention_launch(const AttnArgs& a, int64_t batch, hipStream_t s) {
    constexpr int V = Tr<T>::kVec;
    SS_REQUIRE(a.hd % V == 0 && a.hd > 0 && a.hd <= 128, "attention: head_dim %d unsupported", a.hd);
    SS_REQUIRE(a.q_ss % V == 0 && a.k_ss % V == 0 && a.v_ss % V == 0 && a.q_sh % V == 0 && a.k_sh % V == 0 &&
                   a.v_sh % V == 0 && a.q_sb % V == 0 && a.k_sb % V == 0 && a.v_sb % V == 0,
               "attention: strides must be multiples of %d elements", V);
    SS_REQUIRE(!a.causal_br || a.kv_len >= a.q_len, "attention: causal needs kv_len >= q_len");
    if (a.q_len == 0 || batch == 0) return SS_OK;
    SS_REQUIRE(a.kv_len > 0, "attention: kv_len == 0");
    if constexpr (V == 8) {
        // v2 (DMA-staged, 32 rows per wave) whenever there is enough query work to fill its 128-row blocks
        // v3 / v2 (DMA-staged, 32 rows per wave) whenever there is enough query work to fill their 128-row blocks.
        // attn_ver: 6 (default since round 5) = v3p without the S(t+1) prefetch for head_dim <= 64 (loop-invariant DMA addresses;
        // measured -3 ... -4 % per launch on the UNet's self-attention shapes and bit-equal to v3, profiles/round5_attn_ab.json),
        // v3 for head_dim 128; 3 = v3 with the swizzled V image everywhere, 4 = v3 with the linear V image,
        // 5 = v3p WITH the prefetch (measured slower than v3: 2 waves per SIMD)
        const int ver = tuning_get("attn_ver", 6);
#if SS_EXPERIMENTAL
        // option (1e): a short context (the UNet's cross-attention over 64 image-feature tokens) with K / V held in registers
        if (a.hd == 64 && a.kv_len <= 64 && !a.causal_br && !a.ragged && a.q_len >= 128 && ver >= 3 && tuning_get("attn_cross64", 0))
            return cross_attn64_launch<T>(a, batch, s);
#endif
        if (ver == 5 && a.q_len >= 32 && a.hd <= 64) return flash3p_launch<T, true>(a, batch, s);     // options: v3p (see (1d))
        if (ver == 6 && a.q_len >= 32 && a.hd <= 64) return flash3p_launch<T, false>(a, batch, s);
        if (ver >= 3 && a.q_len >= 32) {
            if (ver == 3 || ver >= 5) {
                if (a.hd <= 64) return flash3_launch_hd<T, 64, true>(a, batch, s);
                return flash3_launch_hd<T, 128, true>(a, batch, s);
            }
            if (a.hd <= 64) return flash3_launch_hd<T, 64, false>(a, batch, s);
            return flash3_launch_hd<T, 128, false>(a, batch, s);
        }
#if SS_EXPERIMENTAL
        if (ver == 2 && a.q_len >= 32) {
            if (a.hd <= 64) return flash2_launch_hd<T, 64>(a, batch, s);
            return flash2_launch_hd<T, 128>(a, batch, s);
        }
#else
        SS_REQUIRE(ver != 2, "attention: attn_ver 2 (flash v2) is only in the EXPERIMENTAL=1 build");
#endif
    }
    if (a.hd <= 64) return flash_launch_hd<T, 64>(a, batch, s);
    return flash_launch_hd<T, 128>(a, batch, s);
}

int attention_dev(const AttnArgs& a, int64_t batch, int dtype, hipStream_t s) {
    return SS_DISPATCH(dtype, attention_launch, a, batch, s);
}

// --------------------------------------------------------------------------------------------
// (2) decode attention (q_len = 1), split-KV, with RoPE + KV append fused in
// --------------------------------------------------------------------------------------------
// grid (n_heads, nsplit), 256 threads.  Thread t works for key group kg = t / LPK (LPK = hd/V lanes
// share one key, each holding V consecutive d) and streams keys kg, kg+NKG, ... of its block's
// chunk with a PRIVATE online softmax (m, l, acc[V]) — no block-level barrier in the key loop; the
// NKG groups are merged once through LDS.  All K/V packs of a 4-keys-per-thread sub-chunk are
// issued before the first use (one HBM round trip per 4*NKG keys).
//
// Fused (engine path, qkv_raw != NULL): q and the NEW token's k are rotated here in the model dtype
// exactly like rotate_half/apply_rotary_pos_emb (modeling_llama_xformer.py:158-173); the block that
// owns position kv_len stores the rotated k and v into the cache (the torch.cat of :239-242) and
// takes them from registers, so no other kernel (and no global read-after-write) is needed.
// partial record per (head, split): [m, l, o[hd]] fp32; attn_combine_kernel merges the splits.
template <typename T>
__device__ __forceinline__ uint4 rope_pack(const T* row_h, const T* cos_t, const T* sin_t, int pos, int hd, int d0) {
    // returns round(round(x*cos) + round(rot*sin)) for d in [d0, d0+V)
    constexpr int V = Tr<T>::kVec;
    const int half = hd >> 1;
    float x[V], xp[V], c[V], sn[V], o[V];
    unpack<T>(ld16(row_h + d0), x);
    unpack<T>(ld16(row_h + (d0 < half ? d0 + half : d0 - half)), xp);
    unpack<T>(ld16(cos_t + (int64_t)pos * hd + d0), c);
    unpack<T>(ld16(sin_t + (int64_t)pos * hd + d0), sn);
    const float sign = d0 < half ? -1.f : 1.f;
#pragma unroll
    for (int j = 0; j < V; ++j) o[j] = Tr<T>::rnd(Tr<T>::rnd(x[j] * c[j]) + Tr<T>::rnd(sign * xp[j] * sn[j]));
    return pack<T>(o);
}

struct DecodeArgs {
    const void* q;        // rotated q [n_heads*hd]            (plain mode)
    const void* qkv_raw;  // [3*n_heads*hd] pre-RoPE q|k|v      (fused mode) or NULL
    void *kc, *vc;        // cache planes [n_heads, cap, hd]
    const void *cos_t, *sin_t;
    float* part;
    const int32_t* kv_len_dev;  // entries already in the cache
    const int32_t* pos_dev;     // rope position of the new token (fused mode)
    const int32_t* done_flag;
    int hd, n_heads, cap, nsplit;
    float scale;
    // sequence batch (blockIdx.z): element / int strides between consecutive sequences
    int nb, state_stride;
    int64_t q_stride, cache_stride, part_stride, out_stride;
};

template <typename T>
__global__ __launch_bounds__(256) void attn_decode_kernel(const DecodeArgs a0) {
    DecodeArgs a = a0;
    {   // select this block's sequence
        const int b = blockIdx.z;
        const int64_t so = (int64_t)b * a.state_stride;
        if (a.done_flag) a.done_flag += so;
        a.kv_len_dev += so;
        if (a.pos_dev) a.pos_dev += so;
        if (a.q) a.q = (const T*)a.q + b * a.q_stride;
        if (a.qkv_raw) a.qkv_raw = (const T*)a.qkv_raw + b * a.q_stride;
        a.kc = (T*)a.kc + b * a.cache_stride;
        a.vc = (T*)a.vc + b * a.cache_stride;
        a.part += b * a.part_stride;
    }
    constexpr int V = Tr<T>::kVec;
    constexpr int KPT = 4;  // keys per thread per sub-chunk
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* sm = reinterpret_cast<float*>(smem_raw);  // [NKG][hd + 2]
    if (a.done_flag && *a.done_flag) return;
    const int h = blockIdx.x, sp = blockIdx.y, tid = threadIdx.x;
    const int hd = a.hd;
    const int LPK = hd / V, NKG = 256 / LPK;
    const int kg = tid / LPK, dl = tid % LPK, d0 = dl * V;
    const bool fused = a.qkv_raw != nullptr;
    const int n_old = *a.kv_len_dev;
    const int kv_len = n_old + (fused ? 1 : 0);
    const int chunk = (kv_len + a.nsplit - 1) / a.nsplit;
    const int k_lo = sp * chunk;
    const int k_hi = min(kv_len, k_lo + chunk);
    const T* kh = (const T*)a.kc + (int64_t)h * a.cap * hd;
    const T* vh = (const T*)a.vc + (int64_t)h * a.cap * hd;
    const int E = a.n_heads * hd;

    // The first sub-chunk's K/V packs are requested before anything else: their HBM/L2 round trip
    // overlaps the q/k RoPE arithmetic below (which has its own dependent loads: qkv row, cos/sin).
    auto load_kv = [&](int base, uint4 (&kk)[KPT], uint4 (&vv)[KPT]) {
#pragma unroll
        for (int i = 0; i < KPT; ++i) {
            const int key = base + i * NKG;
            if (key < k_hi && !(fused && key == n_old)) {
                kk[i] = ld16(kh + (int64_t)key * hd + d0);
                vv[i] = ld16(vh + (int64_t)key * hd + d0);
            } else {
                kk[i] = make_uint4(0, 0, 0, 0);
                vv[i] = make_uint4(0, 0, 0, 0);
            }
        }
    };
    uint4 kk[KPT], vv[KPT];
    int base = k_lo + kg;
    if (base < k_hi) load_kv(base, kk, vv);

    uint4 qv, knew = make_uint4(0, 0, 0, 0), vnew = make_uint4(0, 0, 0, 0);
    if (fused) {
        const T* raw = (const T*)a.qkv_raw;
        const int pos = *a.pos_dev;
        qv = rope_pack<T>(raw + (int64_t)h * hd, (const T*)a.cos_t, (const T*)a.sin_t, pos, hd, d0);
        if (n_old >= k_lo && n_old < k_hi) {  // this block owns the new position
            knew = rope_pack<T>(raw + E + (int64_t)h * hd, (const T*)a.cos_t, (const T*)a.sin_t, pos, hd, d0);
            vnew = ld16(raw + 2 * E + (int64_t)h * hd + d0);
            if (kg == (n_old - k_lo) % NKG && n_old < a.cap) {
                st16((T*)a.kc + ((int64_t)h * a.cap + n_old) * hd + d0, knew);
                st16((T*)a.vc + ((int64_t)h * a.cap + n_old) * hd + d0, vnew);
            }
        }
    } else {
        qv = ld16((const T*)a.q + (int64_t)h * hd + d0);
    }

    float m = -1e30f, l = 0.f, acc[V];
#pragma unroll
    for (int j = 0; j < V; ++j) acc[j] = 0.f;
    while (base < k_hi) {
#pragma unroll
        for (int i = 0; i < KPT; ++i) {
            const int key = base + i * NKG;
            const bool is_new = fused && key == n_old;
            const uint4 kx = is_new ? knew : kk[i];
            const uint4 vx = is_new ? vnew : vv[i];
            float s = dot_pack<T>(kx, qv, 0.f);
            for (int o = LPK >> 1; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
            if (key < k_hi) {  // uniform within the LPK-lane key group
                s *= a.scale;
                const float mn = fmaxf(m, s);
                const float al = expf(m - mn), p = expf(s - mn);
                float vf[V];
                unpack<T>(vx, vf);
                l = l * al + p;
#pragma unroll
                for (int j = 0; j < V; ++j) acc[j] = fmaf(p, vf[j], acc[j] * al);
                m = mn;
            }
        }
        base += NKG * KPT;
        if (base < k_hi) load_kv(base, kk, vv);
    }
    // ---- merge the NKG private softmaxes --------------------------------------------------------
    float* mine = sm + kg * (hd + 2);
    if (dl == 0) { mine[0] = m; mine[1] = l; }
#pragma unroll
    for (int j = 0; j < V; ++j) mine[2 + d0 + j] = acc[j];
    __syncthreads();
    float* rec = a.part + ((int64_t)h * a.nsplit + sp) * (hd + 2);
    if (tid < hd) {
        float mm = -1e30f;
        for (int g = 0; g < NKG; ++g) mm = fmaxf(mm, sm[g * (hd + 2)]);
        float lt = 0.f, ot = 0.f;
        for (int g = 0; g < NKG; ++g) {
            const float* r = sm + g * (hd + 2);
            const float w = r[1] > 0.f ? expf(r[0] - mm) : 0.f;
            lt = fmaf(w, r[1], lt);
            ot = fmaf(w, r[2 + tid], ot);
        }
        rec[2 + tid] = ot;
        if (tid == 0) { rec[0] = mm; rec[1] = lt; }
    }
}

template <typename T, int NS>
__global__ void attn_combine_kernel(const float* __restrict__ part, T* __restrict__ out,
                                    const int32_t* __restrict__ done_flag, int hd, int state_stride,
                                    int64_t part_stride, int64_t out_stride) {
    const int b = blockIdx.y;
    if (done_flag && done_flag[(int64_t)b * state_stride]) return;
    part += b * part_stride;
    out += b * out_stride;
    const int h = blockIdx.x, d = threadIdx.x;
    const float* rec = part + (int64_t)h * NS * (hd + 2);
    float ms[NS], ls[NS], os[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) {  // 3*NS independent loads in flight
        ms[s] = rec[s * (hd + 2)];
        ls[s] = rec[s * (hd + 2) + 1];
        os[s] = rec[s * (hd + 2) + 2 + d];
    }
    float m = -1e30f;
#pragma unroll
    for (int s = 0; s < NS; ++s) m = fmaxf(m, ms[s]);
    float l = 0.f, o = 0.f;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const float w = ls[s] > 0.f ? expf(ms[s] - m) : 0.f;
        l = fmaf(w, ls[s], l);
        o = fmaf(w, os[s], o);
    }
    Tr<T>::st(out + (int64_t)h * hd + d, o / l);
}

// KV splits per (head, slot): 16 at one slot (512 blocks of ~25 keys at S = 400 — enough to fill the chip, measured
// optimum), fewer when several story slots already multiply the block count (4 slots: 4 splits, -6 % token time)
static inline int decode_nsplit(int nb) {
    int n = tuning_get("attn_decode_nsplit", 0);
    if (n <= 0) n = 16 / (nb < 1 ? 1 : nb);
    return n <= 4 ? 4 : n <= 8 ? 8 : n <= 16 ? 16 : 32;
}

template <typename T>
int attn_decode_launch(const DecodeArgs& a0, void* out, int64_t n_heads, int64_t hd, hipStream_t s) {
    constexpr int V = Tr<T>::kVec;
    SS_REQUIRE(hd % V == 0 && 256 % (hd / V) == 0 && 64 % (hd / V) == 0 && hd <= 256,
               "attn_decode: head_dim %lld unsupported", (long long)hd);
    DecodeArgs a = a0;
    a.nsplit = decode_nsplit(a.nb);
    a.scale = 1.0f / sqrtf((float)hd);
    const int NKG = 256 / (int)(hd / V);
    const size_t lds = (size_t)NKG * (hd + 2) * sizeof(float);
    if (a.nb < 1) a.nb = 1;
    hipLaunchKernelGGL(attn_decode_kernel<T>, dim3((unsigned)n_heads, (unsigned)a.nsplit, (unsigned)a.nb), dim3(256),
                       lds, s, a);
    SS_LAUNCH_CHECK("attn_decode");
#define SS_COMBINE(NS)                                                                                          \
    hipLaunchKernelGGL((attn_combine_kernel<T, NS>), dim3((unsigned)n_heads, (unsigned)a.nb), dim3((unsigned)hd), 0, \
                       s, (const float*)a.part, (T*)out, a.done_flag, (int)hd, a.state_stride, a.part_stride,       \
                       a.out_stride)
    switch (a.nsplit) {
        case 4: SS_COMBINE(4); break;
        case 8: SS_COMBINE(8); break;
        case 16: SS_COMBINE(16); break;
        default: SS_COMBINE(32); break;
    }
#undef SS_COMBINE
    SS_LAUNCH_CHECK("attn_combine");
    return SS_OK;
}

// plain: rotated q given, cache already holds kv_len entries
int attn_decode_dev(const void* q, const void* kc, const void* vc, void* out, void* ws, const int32_t* kv_len_dev,
                    const int32_t* done_flag, int64_t n_heads, int64_t hd, int64_t cache_cap, int dtype,
                    hipStream_t s) {
    DecodeArgs a;
    a.q = q; a.qkv_raw = nullptr; a.kc = (void*)kc; a.vc = (void*)vc; a.cos_t = a.sin_t = nullptr;
    a.part = (float*)ws; a.kv_len_dev = kv_len_dev; a.pos_dev = nullptr; a.done_flag = done_flag;
    a.hd = (int)hd; a.n_heads = (int)n_heads; a.cap = (int)cache_cap; a.nsplit = 0; a.scale = 0.f;
    a.nb = 1; a.state_stride = 0; a.q_stride = a.cache_stride = a.part_stride = a.out_stride = 0;
    return SS_DISPATCH(dtype, attn_decode_launch, a, out, n_heads, hd, s);
}

// fused: pre-RoPE qkv row; rotates q/k, appends k/v at slot *kv_len_dev, attends over kv_len+1 keys
// nb sequences: qkv rows 3*E apart, caches cache_stride elements apart, state words state_stride ints apart,
// outputs E apart, partial records one workspace slab (ss_attn_decode_workspace_bytes) apart.
int attn_decode_fused_dev(const void* qkv_raw, void* kc, void* vc, const void* cos_t, const void* sin_t, void* out,
                          void* ws, const int32_t* kv_len_dev, const int32_t* pos_dev, const int32_t* done_flag,
                          int64_t n_heads, int64_t hd, int64_t cache_cap, int nb, int state_stride,
                          int64_t cache_stride, int dtype, hipStream_t s) {
    DecodeArgs a;
    a.nb = nb; a.state_stride = state_stride; a.q_stride = 3 * n_heads * hd; a.cache_stride = cache_stride;
    a.part_stride = (int64_t)(ss_attn_decode_workspace_bytes(n_heads, hd) / sizeof(float));
    a.out_stride = n_heads * hd;
    a.q = nullptr; a.qkv_raw = qkv_raw; a.kc = kc; a.vc = vc; a.cos_t = cos_t; a.sin_t = sin_t;
    a.part = (float*)ws; a.kv_len_dev = kv_len_dev; a.pos_dev = pos_dev; a.done_flag = done_flag;
    a.hd = (int)hd; a.n_heads = (int)n_heads; a.cap = (int)cache_cap; a.nsplit = 0; a.scale = 0.f;
    return SS_DISPATCH(dtype, attn_decode_launch, a, out, n_heads, hd, s);
}

}  // namespace ss

using namespace ss;

extern "C" {

int ss_attention(const void* q, const void* k, const void* v, void* out, int64_t batch, int64_t n_heads,
                 int64_t q_len, int64_t kv_len, int64_t hd, int64_t q_sb, int64_t q_sh, int64_t q_ss, int64_t k_sb,
                 int64_t k_sh, int64_t k_ss, int64_t v_sb, int64_t v_sh, int64_t v_ss, int64_t o_sb, int64_t o_sh,
                 int64_t o_ss, float scale, int causal_br, int dtype, void* stream) {
    AttnArgs a;
    a.q = q; a.k = k; a.v = v; a.out = out;
    a.q_len = (int)q_len; a.kv_len = (int)kv_len; a.hd = (int)hd; a.n_heads = (int)n_heads;
    a.q_sb = q_sb; a.q_sh = q_sh; a.q_ss = q_ss; a.k_sb = k_sb; a.k_sh = k_sh; a.k_ss = k_ss;
    a.v_sb = v_sb; a.v_sh = v_sh; a.v_ss = v_ss; a.o_sb = o_sb; a.o_sh = o_sh; a.o_ss = o_ss;
    a.scale = scale; a.causal_br = causal_br; a.xcd_heads = 0; a.ragged = 0;
    return attention_dev(a, batch, dtype, (hipStream_t)stream);
}

// The same with a key count PER batch element (host array, batch <= 8): the stacked forward of several story slots
// whose caches hold different lengths (LlamaEngine.prefill_batch), one launch instead of one per slot.
int ss_attention_ragged(const void* q, const void* k, const void* v, void* out, int64_t batch, int64_t n_heads, int64_t q_len,
                        const int32_t* host_kv_lens, int64_t hd, int64_t q_sb, int64_t q_sh, int64_t q_ss, int64_t k_sb,
                        int64_t k_sh, int64_t k_ss, int64_t v_sb, int64_t v_sh, int64_t v_ss, int64_t o_sb, int64_t o_sh,
                        int64_t o_ss, float scale, int causal_br, int dtype, void* stream) {
    SS_REQUIRE(host_kv_lens && batch >= 1 && batch <= 8, "attention_ragged: 1..8 batch elements");
    AttnArgs a;
    a.q = q; a.k = k; a.v = v; a.out = out;
    a.q_len = (int)q_len; a.hd = (int)hd; a.n_heads = (int)n_heads;
    a.q_sb = q_sb; a.q_sh = q_sh; a.q_ss = q_ss; a.k_sb = k_sb; a.k_sh = k_sh; a.k_ss = k_ss;
    a.v_sb = v_sb; a.v_sh = v_sh; a.v_ss = v_ss; a.o_sb = o_sb; a.o_sh = o_sh; a.o_ss = o_ss;
    a.scale = scale; a.causal_br = causal_br; a.xcd_heads = 0; a.ragged = 1;
    int mx = 0;
    for (int b = 0; b < 8; ++b) {
        a.kv_len_b[b] = b < batch ? host_kv_lens[b] : 0;
        if (b < batch) {
            SS_REQUIRE(host_kv_lens[b] > 0 && (!causal_br || host_kv_lens[b] >= q_len), "attention_ragged: kv_len[%d] = %d", b,
                       (int)host_kv_lens[b]);
            if (host_kv_lens[b] > mx) mx = host_kv_lens[b];
        }
    }
    a.kv_len = mx;
    return attention_dev(a, batch, dtype, (hipStream_t)stream);
}

size_t ss_attn_decode_workspace_bytes(int64_t n_heads, int64_t hd) {
    return (size_t)n_heads * 64 /* max nsplit */ * (size_t)(hd + 2) * sizeof(float);
}

int ss_attn_decode(const void* q, const void* kcache, const void* vcache, void* out, void* workspace,
                   const int32_t* kv_len_dev, int64_t n_heads, int64_t hd, int64_t cache_cap, int dtype,
                   void* stream) {
    return attn_decode_dev(q, kcache, vcache, out, workspace, kv_len_dev, nullptr, n_heads, hd, cache_cap, dtype,
                           (hipStream_t)stream);
}

}  // extern "C"
