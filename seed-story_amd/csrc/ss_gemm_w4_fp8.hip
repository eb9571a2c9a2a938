// One-stream-per-SIMD MFMA GEMM kernels with fp8 (OCP e4m3) operands: 256x256 tile, 4 waves, AGPR accumulators,
// v_mfma_scale_f32_32x32x64_f8f6f4 (see ss_gemm_w4.inc).  bf16 results.
#include "ss_gemm_common.h"
#define SS_W4_T ::ss::fp8_t
#define SS_W4_CONV 0
#define SS_W4_FP8 1
#include "ss_gemm_w4.inc"
