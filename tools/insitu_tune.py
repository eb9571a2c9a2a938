"""In-situ refinement of the tile table: for the UNet's heaviest GEMM / conv shapes, try the other tile configurations INSIDE
the whole forward (batch 8) and keep what makes the forward faster.  The shipped table was measured per shape in
sustained isolation; neighbours (cache state, the kernel that follows) shift some choices.
    python tools/insitu_tune.py  -> gpurun_out/insitu_tune.json (+ the improved table at gpurun_out/tune_insitu.json)"""
import json, os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "seed-story_amd"))
import torch
from seedstory import _lib, tune
from seedstory.diffusion import UNet2DConditionModel
DEV, dt = "cuda:0", torch.bfloat16
UB = 8
unet = UNet2DConditionModel().to(DEV, dt).init_synthetic(1)
x = torch.randn(UB, 4, 128, 128, device=DEV, dtype=dt)
ctx = torch.randn(UB, 64, 2048, device=DEV, dtype=dt)
cond = {"text_embeds": torch.randn(UB, 1280, device=DEV, dtype=dt), "time_ids": torch.tensor([[1024, 1024, 0, 0, 1024, 1024]] * UB, dtype=torch.float32)}
_lib.set_tuning("gemm_autotune", 0)


def fwd_ms(n=3):
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        unet(x, 500.0, ctx, added_cond_kwargs=cond)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return statistics.median(ts)


unet(x, 500.0, ctx, added_cond_kwargs=cond)
torch.cuda.synchronize()
base = fwd_ms(5)
print("baseline forward %.3f ms" % base, flush=True)
table = {tuple(r[:8]): r for r in tune.export_table()}
# the forward's shapes: M = UB*1024 / UB*4096 / UB*16384 rows, 16-bit; heaviest first (flops = M*N*K)
keys = [k for k in table if k[0] == 1 and k[1] in (UB * 1024, UB * 4096, UB * 16384)]
if os.environ.get("KEYMODE") == "gemm":
    keys = [k for k in keys if k[4] == 0]
keys.sort(key=lambda k: -k[1] * k[2] * k[3])
keys = keys[:int(os.environ.get("NKEYS", "22"))]
GEMM_C = [60, 61, 62, 63, 64, 65, 69, 71, 72]
CONV_C = [61, 63, 65, 66, 69, 71]
log = []
for k in keys:
    cur = table[k]
    best = (base, cur[8], cur[9])
    tried = []
    for cfg in (CONV_C if k[4] else GEMM_C):
        for swz in (0, 4, 8):
            if (cfg, swz) == (cur[8], cur[9]):
                continue
            tune.import_table([list(k) + [cfg, swz]])
            try:
                t = fwd_ms(3)
            except Exception as ex:
                t = 1e9
            tried.append((cfg, swz, round(t, 3)))
            if t < best[0] - 0.12:
                t2 = fwd_ms(4)                      # confirm
                if t2 < best[0] - 0.10:
                    best = (t2, cfg, swz)
    tune.import_table([list(k) + [best[1], best[2]]])
    if (best[1], best[2]) != (cur[8], cur[9]):
        base = fwd_ms(5)
        print("shape %s: cfg %d/%d -> %d/%d, forward now %.3f ms" % (k, cur[8], cur[9], best[1], best[2], base), flush=True)
    log.append({"key": list(k), "was": [cur[8], cur[9]], "now": [best[1], best[2]], "forward_ms_after": round(base, 3),
                "fastest_alternatives": sorted(tried, key=lambda v: v[2])[:3]})
final = fwd_ms(7)
print("final forward %.3f ms" % final)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump({"final_forward_ms": final, "log": log}, open(os.path.join(ROOT, "gpurun_out", "insitu_tune.json"), "w"), indent=0)
tune.save_table(os.path.join(ROOT, "gpurun_out", "tune_insitu.json"), note="in-situ refined")
