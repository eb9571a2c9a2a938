"""Run 3 SDXL-base UNet forwards (batch 2, 128x128 latents) — to be wrapped by rocprofv3 --kernel-trace."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "seed-story_amd"))
import torch
from seedstory.diffusion import UNet2DConditionModel
DEV = "cuda:0"
dt = torch.bfloat16
unet = UNet2DConditionModel().to(DEV, dt).init_synthetic(1)
x = torch.randn(2, 4, 128, 128, device=DEV, dtype=dt)
ctx = torch.randn(2, 64, 2048, device=DEV, dtype=dt)
cond = {"text_embeds": torch.randn(2, 1280, device=DEV, dtype=dt), "time_ids": torch.tensor([[1024, 1024, 0, 0, 1024, 1024]] * 2, dtype=torch.float32)}
for _ in range(3):
    unet(x, 500.0, ctx, added_cond_kwargs=cond)
torch.cuda.synchronize()
