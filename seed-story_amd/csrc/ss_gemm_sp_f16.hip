// Software-pipelined LDS-DMA GEMM kernels, f16 instantiations (see ss_gemm_sp.inc).
#include "ss_gemm_common.h"
#define SS_SP_T ::ss::f16_t
#define SS_SP_CONV 0
#include "ss_gemm_sp.inc"
