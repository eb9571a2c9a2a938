// One-stream-per-SIMD MFMA GEMM kernels (4 waves, 32x32x16 fragments, AGPR accumulators), fp16 (see ss_gemm_w4.inc).
#include "ss_gemm_common.h"
#define SS_W4_T ::ss::f16_t
#define SS_W4_CONV 0
#include "ss_gemm_w4.inc"
