import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "seed-story_amd"))
import torch
from seedstory import _lib, ops
_lib.set_tuning("gemm_autotune", 0)
for cfg, (M, N, K) in [(8, (4096, 4096, 4096)), (8, (2048, 10240, 1280)), (10, (2048, 1280, 1280)), (15, (8192, 5120, 640))]:
    _lib.set_tuning("gemm_cfg", cfg)
    a = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
    w = torch.randn(N, K, device="cuda", dtype=torch.bfloat16) * 0.02
    for _ in range(3):
        ops.gemm(a, w)
torch.cuda.synchronize()
