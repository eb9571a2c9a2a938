// fp8 (OCP e4m3fn) GEMM path for the SDXL UNet's linear layers (SURVEY.md §8 ★ row; BASELINE configs[4] "fp8 MFMA
// SDXL UNet"): row-wise dynamic quantisation of the activations, per-output-channel quantisation of the weights,
// v_mfma_scale_f32_16x16x128_f8f6f4 with unit block scales (twice the bf16 MFMA rate), fp32 accumulation, fp32
// de-quantisation  acc * scale_a[m] * scale_w[n]  fused in front of the usual epilogue (bias / GELU / GEGLU / residual),
// bf16 out.  The reference has no fp8 path: parity is a BOUND against the bf16 path (tests/test_fp8_gpu.py), while the
// kernel itself is exact against its own quantised operands (products of e4m3 values are exact in fp32).
//
// Quantisation (both operand kinds): s = amax(row) / 448, q = RNE_e4m3(x * (448 / amax)); an all-zero row gets s = 0.
#include <stdio.h>

#include "ss_gemm_common.h"

namespace ss {

// one wave per row; 16-byte loads (8 elements of a 16-bit type per lane per pass), the second pass re-reads the row from
// L1/L2.  K % 8 == 0.  Optional fused LayerNorm (gamma/beta != null): quantises  LN(x)  and never writes the normalised
// bf16 row — the values are rounded to T first, exactly where the unfused pipeline rounds them.
template <typename T, bool LN>
__global__ __launch_bounds__(256) void quantize_rows_fp8_kernel(const T* __restrict__ x, int64_t ld, int M, int K,
                                                                uint8_t* __restrict__ q, float* __restrict__ scale,
                                                                const T* __restrict__ gamma, const T* __restrict__ beta,
                                                                float eps) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= M) return;
    const T* xr = x + (int64_t)row * ld;
    float mean = 0.f, rstd = 1.f;
    if constexpr (LN) {
        // two-pass statistics like layernorm_wave_kernel / ss_rowstats (mean, then sum of squared deviations): the
        // single-pass E[x^2] - mean^2 form cancels on rows with |mean| >> std (outlier channels of UNet hidden states)
        float s = 0.f;
        for (int k = lane * 8; k < K; k += 512) {
            float f[8];
            unpack<T>(*reinterpret_cast<const uint4*>(xr + k), f);
#pragma unroll
            for (int e = 0; e < 8; ++e) s += f[e];
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
        mean = s / (float)K;
        float ss2 = 0.f;
        for (int k = lane * 8; k < K; k += 512) {      // the row is re-read from L1/L2
            float f[8];
            unpack<T>(*reinterpret_cast<const uint4*>(xr + k), f);
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float d = f[e] - mean; ss2 += d * d; }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) ss2 += __shfl_xor(ss2, o, 64);
        const float var = ss2 / (float)K;
        rstd = rsqrtf(var + eps);
    }
    auto value = [&](const float (&f)[8], int k, float (&o)[8]) {
        if constexpr (LN) {
            float gm[8], bt[8];
            unpack<T>(*reinterpret_cast<const uint4*>(gamma + k), gm);
            unpack<T>(*reinterpret_cast<const uint4*>(beta + k), bt);
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = Tr<T>::rnd((f[e] - mean) * rstd * gm[e] + bt[e]);
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = f[e];
        }
    };
    float amax = 0.f;
    for (int k = lane * 8; k < K; k += 512) {
        float f[8], o[8];
        unpack<T>(*reinterpret_cast<const uint4*>(xr + k), f);
        value(f, k, o);
#pragma unroll
        for (int e = 0; e < 8; ++e) amax = fmaxf(amax, fabsf(o[e]));
    }
    amax = wave_max(amax);
    const float inv = amax > 0.f ? 448.0f / amax : 0.f;
    if (lane == 0) scale[row] = amax / 448.0f;
    uint8_t* qr = q + (int64_t)row * K;
    for (int k = lane * 8; k < K; k += 512) {
        float f[8], o[8];
        unpack<T>(*reinterpret_cast<const uint4*>(xr + k), f);
        value(f, k, o);
        int lo = 0, hi = 0;
        lo = __builtin_amdgcn_cvt_pk_fp8_f32(o[0] * inv, o[1] * inv, lo, false);
        lo = __builtin_amdgcn_cvt_pk_fp8_f32(o[2] * inv, o[3] * inv, lo, true);
        hi = __builtin_amdgcn_cvt_pk_fp8_f32(o[4] * inv, o[5] * inv, hi, false);
        hi = __builtin_amdgcn_cvt_pk_fp8_f32(o[6] * inv, o[7] * inv, hi, true);
        *reinterpret_cast<uint2*>(qr + k) = make_uint2((uint32_t)lo, (uint32_t)hi);
    }
}

template <typename T>
int quantize_launch(const void* x, int64_t ld, int64_t M, int64_t K, void* q, float* scale, const void* gamma, const void* beta,
                    float eps, hipStream_t s) {
    if constexpr (Tr<T>::kVec != 8) {
        set_error("ss_quantize_rows_fp8: 16-bit inputs only");
        return SS_EINVAL;
    } else {
        const dim3 grid((unsigned)((M + 3) / 4)), block(256);
        if (gamma)
            hipLaunchKernelGGL((quantize_rows_fp8_kernel<T, true>), grid, block, 0, s, (const T*)x, ld, (int)M, (int)K, (uint8_t*)q,
                               scale, (const T*)gamma, (const T*)beta, eps);
        else
            hipLaunchKernelGGL((quantize_rows_fp8_kernel<T, false>), grid, block, 0, s, (const T*)x, ld, (int)M, (int)K, (uint8_t*)q,
                               scale, (const T*)nullptr, (const T*)nullptr, 0.f);
        SS_LAUNCH_CHECK("quantize_rows_fp8");
        return SS_OK;
    }
}

// closed-form tile rule for the fp8 kernels (the shapes are the UNet's: M = batch x tokens, N, K multiples of 160 / 128)
static int pick_cfg_fp8(int64_t M, int64_t N, int64_t K) {
    const int forced = tuning_get("gemm_fp8_cfg", 0);
    if (forced) return forced;
    if (N % 160 != 0) {
        if (N % 16 != 0) return 88;
        return M * N >= 128 * 128 * 256 ? 80 : 86;
    }
    // measured on MI355X at UNet batch 8 (tools/kbench.py fp8): 128x160 (two workgroups per CU) wins while the K loop
    // is short (K <= 1280: 1.30-1.35 PFLOP/s at [8192, 10240 | 3840, 1280]); 256x160 wins once it dominates (K = 5120:
    // 2.0 PFLOP/s)
    if (M >= 2048 && K >= 2560) return 82;
    if (M >= 256) return 81;
    return 85;
}

}  // namespace ss

extern "C" {

int ss_quantize_rows_fp8(const void* x, int64_t ld, int64_t M, int64_t K, void* q_out, float* scale_out, const void* ln_gamma,
                         const void* ln_beta, float ln_eps, int dtype, void* stream) {
    SS_REQUIRE(x && q_out && scale_out && M > 0 && K > 0 && K % 8 == 0 && ld % 8 == 0 && ld >= K, "ss_quantize_rows_fp8: bad arguments");
    SS_REQUIRE((ln_gamma == nullptr) == (ln_beta == nullptr), "ss_quantize_rows_fp8: gamma and beta go together");
    return SS_DISPATCH(dtype, ss::quantize_launch, x, ld, M, K, q_out, scale_out, ln_gamma, ln_beta, ln_eps, (hipStream_t)stream);
}

int ss_gemm_fp8(const void* A8, const float* scale_a, const void* W8, const float* scale_w, void* C, int64_t M, int64_t N,
                int64_t K, int64_t ldc, const void* bias, const void* residual, int64_t ldr, int epilogue, void* stream) {
    SS_REQUIRE(A8 && scale_a && W8 && scale_w && C && M > 0 && N > 0, "ss_gemm_fp8: NULL / empty argument");
    SS_REQUIRE(K >= 128 && K % 128 == 0, "ss_gemm_fp8: K must be a multiple of 128 (K=%lld)", (long long)K);
    SS_REQUIRE((epilogue & ~(SS_EPI_BIAS | SS_EPI_GELU | SS_EPI_RESIDUAL | SS_EPI_GEGLU_PAIR)) == 0, "ss_gemm_fp8: unsupported epilogue %d", epilogue);
    SS_REQUIRE(!(epilogue & SS_EPI_BIAS) || bias, "ss_gemm_fp8: bias epilogue without bias");
    SS_REQUIRE(!(epilogue & SS_EPI_RESIDUAL) || residual, "ss_gemm_fp8: residual epilogue without residual");
    ss::GemmArgs g;
    g.A = A8; g.W = W8; g.C = C; g.bias = bias; g.residual = residual;
    g.M = (int)M; g.N = (int)N; g.K = (int)K; g.lda = K; g.ldw = K; g.ldc = ldc; g.ldr = ldr; g.epi = epilogue;
    g.rowvec = nullptr; g.rows_per_batch = 1; g.rowvec_ld = 0;
    g.conv_H = g.conv_W = g.conv_Cin = g.conv_stride = g.conv_up = g.conv_Ho = g.conv_Wo = 0;
    g.scale_a = scale_a; g.scale_w = scale_w;
    const int cfg = ss::pick_cfg_fp8(M, N, K);
    const int mt = (cfg == 85 || cfg == 86) ? 64 : (cfg == 82 || cfg >= 90) ? 256 : 128;
    g.swz = ss::tuning_get("gemm_fp8_swz", (int)((M + mt - 1) / mt) >= 16 ? 8 : 0);
    if (ss::tuning_get("gemm_fp8_debug", 0)) fprintf(stderr, "ss_gemm_fp8 [%lld,%lld,%lld] epi %d cfg %d swz %d\n", (long long)M, (long long)N, (long long)K, epilogue, cfg, g.swz);
    int rc = 1;
    if (cfg >= 90) {        // 4-wave / AGPR-accumulator 256x256 tile (ss_gemm_w4.inc); ineligible shapes take the 8-wave 256x160 tile
        rc = ss::gemm_w4_dispatch_fp8(cfg, g, (hipStream_t)stream);
        if (rc == 1) rc = ss::gemm_sp_dispatch_fp8(82, g, (hipStream_t)stream);
    } else {
        rc = ss::gemm_sp_dispatch_fp8(cfg, g, (hipStream_t)stream);
    }
    if (rc == 1) {
        ss::set_error("ss_gemm_fp8: no kernel for cfg %d / shape [%lld, %lld, %lld]", cfg, (long long)M, (long long)N, (long long)K);
        return SS_EINVAL;
    }
    return rc;
}

}  // extern "C"
