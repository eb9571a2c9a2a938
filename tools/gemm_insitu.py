"""In-situ (sustained, back-to-back, cold weights) timing of the short-K GEMMs of the 1280-wide transformer blocks:
LN -> GEMM chains as in the UNet, per tile configuration and epilogue.  gpurun_out/gemm_insitu.json"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "seed-story_amd"))
import torch
from seedstory import _lib, ops
dt = torch.bfloat16
DEV = "cuda:0"
_lib.set_tuning("gemm_autotune", 0)


def run(M, N, K, cfg, swz, epi, cold, with_ln, iters=160):
    nW = max(2, int(320e6 // (N * K * 2))) if cold else 1
    nW = min(nW, 96)
    ws = [torch.randn(N, K, device=DEV, dtype=dt) * 0.03 for _ in range(nW)]
    h = torch.randn(M, K, device=DEV, dtype=dt)
    g, b = torch.ones(K, device=DEV, dtype=dt), torch.zeros(K, device=DEV, dtype=dt)
    bias = torch.randn(N, device=DEV, dtype=dt)
    res = torch.randn(M, N, device=DEV, dtype=dt)
    out = torch.empty(M, N, device=DEV, dtype=dt)
    _lib.set_tuning("gemm_cfg", cfg)
    _lib.set_tuning("gemm_xcd_swizzle", swz)

    def body(i):
        y = ops.layernorm(h, g, b, 1e-5) if with_ln else h
        if epi == "res":
            ops.gemm(y, ws[i % nW], bias=bias, residual=res, out=out)
        else:
            ops.gemm(y, ws[i % nW], out=out)
    for i in range(8):
        body(i)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for i in range(iters):
        body(i)
    e1.record()
    torch.cuda.synchronize()
    _lib.set_tuning("gemm_cfg", 0)
    return e0.elapsed_time(e1) / iters * 1e3


def ln_only(M, K, iters=160):
    h = torch.randn(M, K, device=DEV, dtype=dt)
    g, b = torch.ones(K, device=DEV, dtype=dt), torch.zeros(K, device=DEV, dtype=dt)
    for _ in range(8):
        ops.layernorm(h, g, b, 1e-5)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(iters):
        ops.layernorm(h, g, b, 1e-5)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


res = []
cfgs = [int(c) for c in os.environ.get("CFGS", "26,30,31,41,44,42,23,20").split(",")]
SHAPES = [tuple(int(v) for v in x.split("x")) for x in os.environ.get("SHAPES", "8192x1280x1280,8192x3840x1280,8192x1280x5120").split(",")]
for (M, N, K) in SHAPES:
    t_ln = ln_only(M, K)
    for cfg in cfgs:
        for epi in ("none", "res"):
            if N != 1280 and epi == "res":
                continue
            row = {"M": M, "N": N, "K": K, "cfg": cfg, "epi": epi, "ln_us": round(t_ln, 1)}
            row["warm_nolN"] = round(run(M, N, K, cfg, 4, epi, False, False), 1)
            row["cold_noLN"] = round(run(M, N, K, cfg, 4, epi, True, False), 1)
            row["cold_LN_minus_ln"] = round(run(M, N, K, cfg, 4, epi, True, True) - t_ln, 1)
            row["tflops_cold"] = round(2.0 * M * N * K / (row["cold_noLN"] * 1e-6) / 1e12)
            res.append(row)
            print(row, flush=True)
json.dump(res, open(os.path.join(ROOT, "gpurun_out", "gemm_insitu.json"), "w"), indent=0)
