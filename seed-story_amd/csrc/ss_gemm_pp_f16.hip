// Ping-pong 8-phase 256x256 GEMM kernels, fp16 instantiations (see ss_gemm_pp.inc).
#include "ss_gemm_common.h"
#define SS_PP_T ::ss::f16_t
#include "ss_gemm_pp.inc"
