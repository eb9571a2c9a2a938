"""LayerNorm wave kernel: rows per wave (tuning knob layernorm_rows_per_wave = 1 | 2 | 4) — bit-equality and time per launch."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "seed-story_amd"))
import torch  # noqa: E402
from seedstory import _lib, ops  # noqa: E402

DEV = "cuda:0"
for rows, cols in ((16384, 1280), (65536, 640), (8192, 1280), (16385, 1280), (4096 * 16, 1664)):
    xs = [torch.randn(rows, cols, device=DEV).bfloat16() for _ in range(6)]
    w = torch.randn(cols, device=DEV).bfloat16()
    b = torch.randn(cols, device=DEV).bfloat16()
    ref = None
    line = "[%6d, %4d]" % (rows, cols)
    for R in (1, 2, 4):
        _lib.set_tuning("layernorm_rows_per_wave", R)
        y = ops.layernorm(xs[0], w, b, 1e-5)
        if ref is None:
            ref = y
            t = torch.nn.functional.layer_norm(xs[0].float(), (cols,), w.float(), b.float(), 1e-5)
            err = float((y.float() - t).norm() / t.norm())
        eq = bool(torch.equal(y, ref))
        for i in range(3):
            ops.layernorm(xs[i], w, b, 1e-5)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(60):
            ops.layernorm(xs[i % 6], w, b, 1e-5)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 60 * 1e3
        line += "  R=%d %6.1f us (%.2f TB/s)%s" % (R, us, 4.0 * rows * cols / us / 1e6, "" if eq else " NOT-EQUAL")
    print(line, " vs torch fp32 %.1e" % err)
_lib.set_tuning("layernorm_rows_per_wave", 1)
