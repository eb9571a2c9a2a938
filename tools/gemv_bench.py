"""Decode-projection micro-benchmark on MI355X: the LLaMA-7B GEMV shapes at 1..8 sequences per sweep, sustained over rotating
weight copies (the weights of 32 layers never sit in L2 / Infinity Cache), HIP events around batches of launches.
  python tools/gemv_bench.py [--knob gemv_mfma_min_nb=1 ...]  ->  one JSON line per (shape, nb) + gpurun_out/gemv_bench.json"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "seed-story_amd"))
from seedstory import _lib, ops  # noqa: E402

dev = "cuda:0"
dt = torch.float32 if os.environ.get("GEMV_DTYPE") == "f32" else torch.bfloat16      # f32 + gemm_f32_split=1: the gate-mode form
ES = 4 if dt == torch.float32 else 2
knobs = [a.split("=") for a in sys.argv[1:] if "=" in a and not a.startswith("--")]
NBS = [int(x) for x in os.environ.get("GEMV_NBS", "1,4,8").split(",")]
SHAPES = [("qkv+norm", 12288, 4096, dict(norm=True)), ("o+res", 4096, 4096, dict(res=True)), ("gate|up+norm silu", 11008, 4096, dict(norm=True, silu=True)),
          ("down+res", 4096, 11008, dict(res=True)), ("lm_head", 32066, 4096, {})]
out = []
for name, N, K, kw in SHAPES:
    rows = 2 * N if kw.get("silu") else N
    copies = max(2, int(1.2e9 // (rows * K * ES)))
    Ws = [torch.randn(rows, K, device=dev, dtype=dt) * 0.02 for _ in range(copies)]
    nw = torch.ones(K, device=dev, dtype=dt)
    for nb in NBS:
        x = torch.randn(nb, K, device=dev, dtype=dt)
        res = torch.randn(nb, N, device=dev, dtype=dt)
        for variant, kv in (("default", []), ("knobs", knobs)) if knobs else (("default", []),):
            for k, v in kv:
                _lib.set_tuning(k, int(v))
            call = lambda w: ops.gemv_batched(w, x, norm_w=nw if kw.get("norm") else None, eps=1e-5,  # noqa: E731
                                              residual=res if kw.get("res") else None, silu_mul=bool(kw.get("silu")))
            for w in Ws[:2]:
                call(w)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps = 3
            e0.record()
            for _ in range(reps):
                for w in Ws:
                    call(w)
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / (reps * copies)
            for k, v in kv:
                _lib.set_tuning(k, {"gemv_mfma_min_nb": 3, "gemv_nt": 1, "gemv_mfma_blocks": 256, "gemv_mfma_nt": 0}.get(k, 0))
            rec = {"shape": name, "N": N, "K": K, "nb": nb, "variant": variant if not kv else dict(kv), "us": round(us, 2),
                   "GBps": round(rows * K * ES / us / 1e3, 1), "MB": round(rows * K * ES / 1e6, 1), "dtype": str(dt).split(".")[-1]}
            out.append(rec)
            print(json.dumps(rec), flush=True)
    del Ws
    torch.cuda.empty_cache()
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/gemv_bench%s.json" % ("_f32" if ES == 4 else ""), "w"), indent=1)
