"""A handful of launches of the dominant MFMA kernels of the UNet forward at batch 8 (the shipped tile table decides
the configuration) — wrapped by the rocprofv3 --pmc passes of tools/gpu_round2.sh (separate counter passes,
--kernel-trace only).  A marker kernel (softmax_rows on 8 elements) separates the cases in dispatch order:
ff1 bf16 | n1280+residual bf16 | ff2 bf16 | qkv bf16 | ff1 fp8 | ff2 fp8 | conv 1280 | attention 4096."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "seed-story_amd"))
import torch
from seedstory import ops
from seedstory.diffusion import _conv_w
dt = torch.bfloat16
DEV = "cuda:0"
NREP = 4


def marker():
    ops.softmax_rows_(torch.zeros(1, 8, device=DEV, dtype=dt), 1.0)


CASES = [("ff1", 8192, 10240, 1280, True, False), ("n1280", 8192, 1280, 1280, False, True), ("ff2", 8192, 1280, 5120, False, True),
         ("qkv", 8192, 3840, 1280, False, False)]
for name, M, N, K, geglu, res in CASES:
    a = torch.randn(M, K, device=DEV, dtype=dt)
    ws = [torch.randn(N, K, device=DEV, dtype=dt) * 0.03 for _ in range(NREP)]
    b = torch.zeros(N, device=DEV, dtype=dt)
    r = torch.randn(M, N, device=DEV, dtype=dt) if res else None
    marker()
    for w in ws:
        if geglu:
            ops.gemm_geglu(a, w, b)
        else:
            ops.gemm(a, w, bias=b, residual=r)
    torch.cuda.synchronize()
for name, M, N, K, geglu in [("ff1_fp8", 8192, 10240, 1280, True), ("ff2_fp8", 8192, 1280, 5120, False)]:
    a = torch.randn(M, K, device=DEV, dtype=dt)
    a8, sa = ops.quantize_rows_fp8(a)
    ws = [ops.quantize_rows_fp8(torch.randn(N, K, device=DEV, dtype=dt) * 0.03) for _ in range(NREP)]
    b = torch.zeros(N, device=DEV, dtype=dt)
    marker()
    for w8, sw in ws:
        ops.gemm_fp8(a8, sa, w8, sw, bias=b, geglu=geglu)
    torch.cuda.synchronize()
x = torch.randn(8 * 32 * 32, 1280, device=DEV, dtype=dt)
cw = [_conv_w((torch.randn(1280, 1280, 3, 3, device=DEV, dtype=torch.float32) * 0.01).to(dt)) for _ in range(NREP)]
marker()
for w in cw:
    ops.conv3x3(x, w, 8, 32, 32, bias=torch.zeros(1280, device=DEV, dtype=dt))
q = torch.randn(8, 4096, 640, device=DEV, dtype=dt)
k = torch.randn(8, 4096, 640, device=DEV, dtype=dt)
v = torch.randn(8, 4096, 640, device=DEV, dtype=dt)
marker()
for _ in range(NREP):
    ops.attention(q, k, v, 10)
torch.cuda.synchronize()
