// Ping-pong 8-phase GEMM kernels, f16 instantiations (see ss_gemm_pp.inc).
#include "ss_gemm_common.h"
#define SS_PP_T ::ss::f16_t
#define SS_PP_CONV 0
#include "ss_gemm_pp.inc"
