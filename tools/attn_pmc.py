"""Three launches of the UNet self-attention shapes — wrapped by rocprofv3 --pmc passes (tools/run_attn_pmc.sh)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "seed-story_amd"))
import torch
from seedstory import ops
for (B, H, hd, L) in [(8, 10, 64, 4096), (8, 20, 64, 1024)]:
    E = H * hd
    q = torch.randn(B, L, E, device="cuda", dtype=torch.bfloat16)
    k = torch.randn(B, L, E, device="cuda", dtype=torch.bfloat16)
    v = torch.randn(B, L, E, device="cuda", dtype=torch.bfloat16)
    for _ in range(3):
        ops.attention(q, k, v, H, None, False)
torch.cuda.synchronize()
