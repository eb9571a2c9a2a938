"""Small-M LLaMA projections: regular tiles (ss_gemm) vs the split-K path, sustained over rotating weight copies."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "seed-story_amd"))
import torch
from seedstory import _lib, ops
DEV, dt = "cuda:0", torch.bfloat16
H, I = 4096, 11008


def timed(fn, n):
    for i in range(3):
        fn(i)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for i in range(n):
        fn(i)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


res = []
for M in (264, 460):
    for (N, K, r) in ((3 * H, H, False), (H, H, True), (2 * I, H, False), (H, I, True)):
        nW = max(2, min(24, int(400e6 // (N * K * 2))))
        ws = [torch.randn(N, K, device=DEV, dtype=dt) * 0.02 for _ in range(nW)]
        a = torch.randn(M, K, device=DEV, dtype=dt)
        rr = torch.randn(M, N, device=DEV, dtype=dt) if r else None
        row = {"M": M, "N": N, "K": K, "regular_us": round(timed(lambda i: ops.gemm(a, ws[i % nW], residual=rr), 40), 1)}
        for swz in (4, 0):
            _lib.set_tuning("gemm_splitk_swz", swz)
            for S in (0, 2, 3, 4, 6):
                _lib.set_tuning("gemm_splitk_s", S)
                row["splitk_swz%d_S%s_us" % (swz, S or "auto")] = round(timed(lambda i: ops.gemm_splitk(a, ws[i % nW], residual=rr), 40), 1)
        _lib.set_tuning("gemm_splitk_s", 0); _lib.set_tuning("gemm_splitk_swz", 4)
        row["weights_MB"] = round(N * K * 2 / 1e6, 1)
        print(row, flush=True)
        res.append(row)
        del ws
json.dump(res, open(os.path.join(ROOT, "gpurun_out", "splitk_bench.json"), "w"), indent=0)
