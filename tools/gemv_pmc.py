"""The decode token's GEMV launch mix for the counter passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, tools/gpu_round4.sh final):
per token 32 x (q|k|v + RMSNorm, o + residual, gate|up + RMSNorm + SiLU.mul, down + residual) + one lm_head = the 129 launches
`bench.py`'s roofline.mllm_decode_gemv averages over (13.215 GB per token, 102.4 MB per launch), at GEMV_NB sequence slots per
sweep (default 8: the MFMA form), over rotating weight copies (4 layer sets = 1.6 GB per cycle: nothing is served from the
256 MB Infinity Cache).  Starts in seconds once torch is paged in — `bench.py --mllm-only` under the counters costs a minute
per pass for the same kernels."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "seed-story_amd"))
from seedstory import _lib, ops  # noqa: E402

dev = "cuda:0"
dt = torch.float32 if os.environ.get("GEMV_DTYPE") == "f32" else torch.bfloat16
if dt == torch.float32:      # the gate-mode decode: fp32 weights through the split-bf16 MFMA form
    _lib.set_tuning("gemm_f32_split", int(os.environ.get("GEMV_SPLIT", "1")))
ES = 4 if dt == torch.float32 else 2
NB = int(os.environ.get("GEMV_NB", "8"))
TOKENS = int(os.environ.get("GEMV_TOKENS", "3"))
H, I, V, L, COPIES = 4096, 11008, 32066, 32, 4


def w(rows, cols):
    return torch.randn(rows, cols, device=dev, dtype=dt) * 0.02


sets = [(w(3 * H, H), w(H, H), w(2 * I, H), w(H, I)) for _ in range(COPIES)]
heads = [w(V, H) for _ in range(2)]
nw = torch.ones(H, device=dev, dtype=dt)
x = torch.randn(NB, H, device=dev, dtype=dt)
xi = torch.randn(NB, I, device=dev, dtype=dt)
res = torch.randn(NB, H, device=dev, dtype=dt)
torch.cuda.synchronize()
nbytes = 0
for t in range(TOKENS):
    for l in range(L):
        wqkv, wo, wgu, wd = sets[(t * L + l) % COPIES]
        ops.gemv_batched(wqkv, x, norm_w=nw, eps=1e-5)
        ops.gemv_batched(wo, x, residual=res)
        ops.gemv_batched(wgu, x, norm_w=nw, eps=1e-5, silu_mul=True)
        ops.gemv_batched(wd, xi, residual=res)
        nbytes += ES * (wqkv.numel() + wo.numel() + wgu.numel() + wd.numel())
    ops.gemv_batched(heads[t % 2], x)
    nbytes += ES * heads[0].numel()
torch.cuda.synchronize()
print(json.dumps({"gemv_pmc": True, "slots_per_sweep": NB, "tokens": TOKENS, "launches": TOKENS * (4 * L + 1),
                  "algorithmic_bytes_per_launch": round(nbytes / (TOKENS * (4 * L + 1))), "dtype": str(dt).split(".")[-1]}))
