// Software-pipelined LDS-DMA GEMM kernels, fp8 (OCP e4m3) operands -> bf16 results (see ss_gemm_sp.inc).
#include "ss_gemm_common.h"
#define SS_SP_T ::ss::fp8_t
#define SS_SP_CONV 0
#define SS_SP_FP8 1
#include "ss_gemm_sp.inc"
