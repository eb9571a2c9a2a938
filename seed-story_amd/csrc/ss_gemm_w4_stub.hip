// Default build: the one-stream-per-SIMD 4-wave GEMM family (ss_gemm_w4.inc: cfg 90-93, fp8 95 / 96) is NOT compiled in — no
// tile-table entry selects it (round 4: it reached MFMA busy 0.80 but the same wall time as the 8-wave tiles, and its fp8 tiles
// were slower than cfg 82).  `make EXPERIMENTAL=1` builds the five ss_gemm_w4_*.hip units instead of this file.  Here every
// dispatch answers 1 = "not eligible", which the callers already treat as "take the 8-wave 256x256 tile" (ss_gemm.hip
// gemm_dispatch_cfg, ss_fp8.hip ss_gemm_fp8).
#include "ss_gemm_common.h"

namespace ss {

template <typename T> int gemm_w4_dispatch(int cfg, const GemmArgs& g, hipStream_t s);
template <typename T> int gemm_w4_dispatch_conv(int cfg, const GemmArgs& g, hipStream_t s);
template <> int gemm_w4_dispatch<bf16_t>(int, const GemmArgs&, hipStream_t) { return 1; }
template <> int gemm_w4_dispatch<f16_t>(int, const GemmArgs&, hipStream_t) { return 1; }
template <> int gemm_w4_dispatch_conv<bf16_t>(int, const GemmArgs&, hipStream_t) { return 1; }
template <> int gemm_w4_dispatch_conv<f16_t>(int, const GemmArgs&, hipStream_t) { return 1; }
int gemm_w4_dispatch_fp8(int, const GemmArgs&, hipStream_t) { return 1; }

}  // namespace ss
