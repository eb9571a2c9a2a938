"""Run 3 SDXL VAE decodes (one 128x128 latent -> 1024x1024 image, bf16) behind a marker kernel — to be wrapped by
rocprofv3 --kernel-trace; tools/trace_summary.py prints the per-kernel / per-grid breakdown of what follows the marker."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "seed-story_amd"))
import torch
from seedstory import ops
from seedstory.diffusion import AutoencoderKL
DEV, dt = "cuda:0", torch.bfloat16
vae = AutoencoderKL().to(DEV, dt).init_synthetic(2)
lat = torch.randn(1, 4, 128, 128, device=DEV, dtype=dt) * 0.5
vae.decode_nhwc(lat, prescale=1.0 / 0.13025)      # tile-table pass
torch.cuda.synchronize()
ops.transpose(torch.zeros(8, 8, device=DEV, dtype=dt))   # (the VAE's own mid-block attention launches softmax_rows: no marker; summary = all 4 decodes / 4)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3):
    vae.decode_nhwc(lat, prescale=1.0 / 0.13025)
torch.cuda.synchronize()
print("wall ms per decode: %.2f" % ((time.perf_counter() - t0) / 3 * 1e3))
