// Software-pipelined LDS-DMA implicit-GEMM 3x3 conv kernels, bf16 instantiations (see ss_gemm_sp.inc).
#include "ss_gemm_common.h"
#define SS_SP_T ::ss::bf16_t
#define SS_SP_CONV 1
#include "ss_gemm_sp.inc"
