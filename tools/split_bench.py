"""fp32-tensor GEMMs of the MLLM half: exact fp32 MFMA chain vs the split-bf16 gate mode (`gemm_f32_split`), MI355X.
    python tools/split_bench.py        -> gpurun_out/split_bench.json"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "seed-story_amd"))
import torch  # noqa: E402

from seedstory import _lib, ops  # noqa: E402

DEV = "cuda:0"


def timed(fn, n=5):
    fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


res = []
for (M, N, K) in [(7304, 12288, 4096), (7304, 4096, 4096), (7304, 22016, 4096), (7304, 4096, 11008), (528, 12288, 4096), (528, 4096, 11008),
                  (1024, 4992, 1664), (2048, 4096, 4096)]:
    a = torch.randn(M, K, device=DEV)
    w = torch.randn(N, K, device=DEV) * 0.02
    row = {"M": M, "N": N, "K": K}
    ref = None
    for split in (0, 1):
        _lib.set_tuning("gemm_f32_split", split)
        us = timed(lambda: ops.gemm(a, w))
        y = ops.gemm(a, w)
        if ref is None:
            ref = y
        row["split_us" if split else "exact_us"] = round(us, 1)
        row["split_tflops" if split else "exact_tflops"] = round(2.0 * M * N * K / us / 1e6, 1)
    for tile in (1, 2):          # A/B: 256x128 / 128x256 with 8 waves instead of 128x128 with 4
        _lib.set_tuning("gemm_f32_split_tile", tile)
        us = timed(lambda: ops.gemm(a, w))
        yt = ops.gemm(a, w)
        row["split_tile%d_us" % tile] = round(us, 1)
        row["split_tile%d_rel" % tile] = float((yt - ref).norm() / ref.norm())
    _lib.set_tuning("gemm_f32_split_tile", 0)
    for order in (0, 1, 4, 8):       # A/B: the plain N-fastest grid (0) and M-fastest bands of `order` N tiles (default since round 5: 16)
        _lib.set_tuning("gemm_f32_split_order", order)
        us = timed(lambda: ops.gemm(a, w))
        yt = ops.gemm(a, w)
        row["order%d_us" % order] = round(us, 1)
        assert float((yt - ref).norm() / ref.norm()) < 1e-4
    _lib.set_tuning("gemm_f32_split_order", 16)
    _lib.set_tuning("gemm_f32_split", 0)
    row["speedup"] = round(row["exact_us"] / row["split_us"], 2)
    row["rel_split_vs_exact"] = float((y - ref).norm() / ref.norm())
    print(row)
    res.append(row)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(res, open(os.path.join(ROOT, "gpurun_out", "split_bench.json"), "w"), indent=0)
