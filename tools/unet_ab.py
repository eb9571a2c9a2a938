"""Interleaved A/B of whole SDXL-base UNet forwards (batch 8) under named variants: tile-table overrides and tuning knobs.
    python tools/unet_ab.py base w4=8192,10240,1280:91/8+8192,3840,1280:91/8 generic=knob:gemm_epi_generic:1 ...
A variant is  name=<item>+<item>...  with  <item> = M,N,K:cfg/swz  (GEMM table entry)  or  knob:<key>:<value>.
Prints the median / min forward time per variant over ROUNDS interleaved rounds -> gpurun_out/unet_ab.json"""
import json, os, statistics, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "seed-story_amd"))
import torch
from seedstory import _lib, tune
from seedstory.diffusion import UNet2DConditionModel
DEV, dt = "cuda:0", torch.bfloat16
UB = int(os.environ.get("SS_UNET_BATCH", "8"))
ROUNDS = int(os.environ.get("ROUNDS", "5"))
unet = UNet2DConditionModel().to(DEV, dt).init_synthetic(1)
x = torch.randn(UB, 4, 128, 128, device=DEV, dtype=dt)
ctx = torch.randn(UB, 64, 2048, device=DEV, dtype=dt)
cond = {"text_embeds": torch.randn(UB, 1280, device=DEV, dtype=dt), "time_ids": torch.tensor([[1024, 1024, 0, 0, 1024, 1024]] * UB, dtype=torch.float32)}
_lib.set_tuning("gemm_autotune", 0)
unet(x, 500.0, ctx, added_cond_kwargs=cond)
torch.cuda.synchronize()
table0 = {tuple(r[:8]): list(r) for r in tune.export_table()}


def parse(spec):
    name, _, body = spec.partition("=")
    entries, knobs = [], []
    for item in (body.split("+") if body else []):
        if item.startswith("knob:"):
            _, k, v = item.split(":")
            knobs.append((k, int(v)))
        elif item.startswith("lnfold:"):
            knobs.append(("__lnfold__", int(item.split(":")[1])))
        else:
            shape, cs = item.split(":")
            M, N, K = [int(v) for v in shape.split(",")]
            cfg, _, swz = cs.partition("/")
            entries.append([1, M, N, K, 0, 0, 0, 0, int(cfg), int(swz or 8)])
    return name, entries, knobs


variants = [parse(a) for a in sys.argv[1:]] or [("base", [], [])]
all_keys = {tuple(e[:8]) for _, es, _ in variants for e in es}
all_knobs = {k for _, _, ks in variants for k, _ in ks}


def apply(entries, knobs):
    tune.import_table([table0[k] for k in all_keys if k in table0])       # restore the shipped choices
    for k in all_knobs:
        if k != "__lnfold__":
            _lib.set_tuning(k, 0)
    if entries:
        tune.import_table(entries)
    lnf = 0
    for k, v in knobs:
        if k == "__lnfold__":
            lnf = v
        else:
            _lib.set_tuning(k, v)
    unet.enable_lnfold(bool(lnf))


def fwd_ms(n):
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        unet(x, 500.0, ctx, added_cond_kwargs=cond)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return ts


res = {name: [] for name, _, _ in variants}
for r in range(ROUNDS):
    for name, entries, knobs in variants:
        apply(entries, knobs)
        fwd_ms(1)
        res[name] += fwd_ms(3)
apply([], [])
out = {name: {"median_ms": round(statistics.median(v), 3), "min_ms": round(min(v), 3), "n": len(v)} for name, v in res.items()}
for name, v in out.items():
    print("%-14s median %.3f ms   min %.3f ms" % (name, v["median_ms"], v["min_ms"]), flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "unet_ab.json"), "w"), indent=0)
