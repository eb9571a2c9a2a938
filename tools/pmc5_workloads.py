"""Round-5 counter workloads (one per invocation, few launches each) for `rocprofv3 --pmc ... --kernel-trace`:
    python tools/pmc5_workloads.py attn    flash v3p (default) on the UNet's self-attention shapes at batch 16
    python tools/pmc5_workloads.py split   the split-bf16 gate-mode GEMM on the stacked-prefill shapes (fp32 tensors)
    python tools/pmc5_workloads.py gemv    = tools/gemv_pmc.py with GEMV_DTYPE=f32 (the gate-mode decode launch mix)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "seed-story_amd"))
import torch  # noqa: E402

from seedstory import _lib, ops  # noqa: E402

what = sys.argv[1]
if what == "attn":
    for (B, H, hd, L) in [(16, 10, 64, 4096), (16, 20, 64, 1024)]:
        E = H * hd
        q, k, v = (torch.randn(B, L, E, device="cuda", dtype=torch.bfloat16) for _ in range(3))
        for _ in range(3):
            ops.attention(q, k, v, H, None, False)
elif what == "split":
    _lib.set_tuning("gemm_f32_split", 1)
    for (M, N, K) in [(7304, 12288, 4096), (7304, 4096, 4096), (528, 12288, 4096)]:
        a = torch.randn(M, K, device="cuda")
        w = torch.randn(N, K, device="cuda") * 0.02
        for _ in range(3):
            ops.gemm(a, w)
    _lib.set_tuning("gemm_f32_split", 0)
elif what == "gemv":
    os.environ["GEMV_DTYPE"] = "f32"
    exec(open(os.path.join(ROOT, "tools", "gemv_pmc.py")).read())
torch.cuda.synchronize()
