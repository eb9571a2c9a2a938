"""A handful of launches of the dominant MFMA kernels of the UNet forward at batch 8 — wrapped by the rocprofv3 --pmc
passes of tools/run_pmc.sh (separate counter passes, --kernel-trace only)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "seed-story_amd"))
import torch
from seedstory import _lib, ops
from seedstory.diffusion import _conv_w
dt = torch.bfloat16
DEV = "cuda:0"
_lib.set_tuning("gemm_autotune", 0)
CASES = [("ff1", 8192, 10240, 1280, (36, 0), True), ("ff1", 8192, 10240, 1280, (40, 0), True),
         ("n1280", 8192, 1280, 1280, (39, 4), False), ("n1280", 8192, 1280, 1280, (41, 4), False),
         ("ff2", 8192, 1280, 5120, (32, 4), False), ("ff2", 8192, 1280, 5120, (42, 4), False),
         ("qkv", 8192, 3840, 1280, (37, 4), False)]
for name, M, N, K, (cfg, swz), geglu in CASES:
    a = torch.randn(M, K, device=DEV, dtype=dt)
    ws = [torch.randn(N, K, device=DEV, dtype=dt) * 0.03 for _ in range(3)]
    b = torch.zeros(N, device=DEV, dtype=dt)
    _lib.set_tuning("gemm_cfg", cfg)
    _lib.set_tuning("gemm_xcd_swizzle", swz)
    for w in ws:
        if geglu:
            ops.gemm_geglu(a, w, b)
        else:
            ops.gemm(a, w, bias=b)
    torch.cuda.synchronize()
_lib.set_tuning("gemm_cfg", 30)
_lib.set_tuning("gemm_xcd_swizzle", 8)
x = torch.randn(8 * 32 * 32, 1280, device=DEV, dtype=dt)
for i in range(3):
    w = (torch.randn(1280, 1280, 3, 3, device=DEV, dtype=torch.float32) * 0.01).to(dt)
    ops.conv3x3(x, _conv_w(w), 8, 32, 32, bias=torch.zeros(1280, device=DEV, dtype=dt))
_lib.set_tuning("gemm_cfg", 0)
q = torch.randn(8, 4096, 640, device=DEV, dtype=dt)
k = torch.randn(8, 4096, 640, device=DEV, dtype=dt)
v = torch.randn(8, 4096, 640, device=DEV, dtype=dt)
for _ in range(3):
    ops.attention(q, k, v, 10)
torch.cuda.synchronize()
