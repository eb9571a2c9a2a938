// Software-pipelined LDS-DMA implicit-GEMM 3x3 conv kernels, f16 instantiations (see ss_gemm_sp.inc).
#include "ss_gemm_common.h"
#define SS_SP_T ::ss::f16_t
#define SS_SP_CONV 1
#include "ss_gemm_sp.inc"
