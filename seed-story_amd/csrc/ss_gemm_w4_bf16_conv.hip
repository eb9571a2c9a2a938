// One-stream-per-SIMD implicit-GEMM 3x3 conv kernels, bf16 (see ss_gemm_w4.inc).
#include "ss_gemm_common.h"
#define SS_W4_T ::ss::bf16_t
#define SS_W4_CONV 1
#include "ss_gemm_w4.inc"
