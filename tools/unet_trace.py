"""Run 3 SDXL-base UNet forwards (batch SS_UNET_BATCH, default 8 = 4 stories x CFG, 128x128 latents) — to be
wrapped by rocprofv3 --kernel-trace.  One untraced-equivalent warm-up forward runs the autotuner first."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "seed-story_amd"))
import torch
from seedstory.diffusion import UNet2DConditionModel
DEV = "cuda:0"
dt = torch.bfloat16
unet = UNet2DConditionModel().to(DEV, dt).init_synthetic(1)
B = int(os.environ.get("SS_UNET_BATCH", "8"))
if os.environ.get("SS_ATTN_WAVES"):
    from seedstory import _lib
    _lib.set_tuning("attn_waves", int(os.environ["SS_ATTN_WAVES"]))
if os.environ.get("KB_LNFOLD") is not None:
    unet.enable_lnfold(os.environ["KB_LNFOLD"] != "0")
x = torch.randn(B, 4, 128, 128, device=DEV, dtype=dt)
ctx = torch.randn(B, 64, 2048, device=DEV, dtype=dt)
cond = {"text_embeds": torch.randn(B, 1280, device=DEV, dtype=dt), "time_ids": torch.tensor([[1024, 1024, 0, 0, 1024, 1024]] * B, dtype=torch.float32)}
unet(x, 500.0, ctx, added_cond_kwargs=cond)   # autotune pass
torch.cuda.synchronize()
import time
from seedstory import ops
ops.softmax_rows_(torch.zeros(1, 8, device=DEV, dtype=dt), 1.0)   # marker kernel: trace_summary counts what follows it
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3):
    unet(x, 500.0, ctx, added_cond_kwargs=cond)
torch.cuda.synchronize()
print("wall ms per forward: %.2f" % ((time.perf_counter() - t0) / 3 * 1e3))
