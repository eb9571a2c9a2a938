"""Gate-mode split-bf16 GEMM (fp32 tensors, gemm_f32_split = 1): a few launches of the big stacked-prefill products for a rocprofv3 --pmc pass
(the 7.4 x over-fetch of round 4 was 'fixed' by M-fastest bands and never re-counted).  SPLIT_ORDER = 0 | 16 selects the
plain N-fastest grid / the shipped bands of 16 N tiles."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "seed-story_amd"))
import torch  # noqa: E402
from seedstory import _lib, ops  # noqa: E402

DEV = "cuda:0"
_lib.set_tuning("gemm_f32_split", 1)
_lib.set_tuning("gemm_f32_split_order", int(os.environ.get("SPLIT_ORDER", "16")))
for (M, N, K) in [(7304, 12288, 4096), (7304, 22016, 4096), (7304, 4096, 11008)]:
    a = torch.randn(M, K, device=DEV)
    ws = [torch.randn(N, K, device=DEV) * 0.02 for _ in range(2)]
    for i in range(3):
        ops.gemm(a, ws[i % 2])
    torch.cuda.synchronize()
