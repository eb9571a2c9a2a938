// Ping-pong 8-phase implicit-GEMM 3x3 convolution kernels, bf16 instantiations (see ss_gemm_pp.inc).
#include "ss_gemm_common.h"
#define SS_PP_T ::ss::bf16_t
#define SS_PP_CONV 1
#include "ss_gemm_pp.inc"
