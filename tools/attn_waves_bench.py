"""flash_attn3p (head dim 64): 4 vs 8 waves per workgroup (tuning knob attn_waves) on the UNet's self-attention shapes as the forward issues
them (fused q|k|v projection output, heads packed) — bit-equality and time per launch, interleaved."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "seed-story_amd"))
import torch  # noqa: E402
from seedstory import _lib, ops  # noqa: E402

DEV, dt = "cuda:0", torch.bfloat16
for (B, Hh, L, Lk) in [(16, 10, 4096, 4096), (16, 20, 1024, 1024), (8, 10, 4096, 4096), (2, 10, 4096, 4096), (2, 20, 1024, 1024), (16, 10, 4096, 64), (16, 20, 1024, 64),
                       (3, 10, 1000, 1000), (1, 16, 300, 700)]:
    E = Hh * 64
    qs = [torch.randn(B, L, E, device=DEV, dtype=dt) for _ in range(3)]
    ks = [torch.randn(B, Lk, E, device=DEV, dtype=dt) for _ in range(3)]
    vs = [torch.randn(B, Lk, E, device=DEV, dtype=dt) for _ in range(3)]
    ref, line = None, "B%2d h%2d L%4d Lk%4d " % (B, Hh, L, Lk)
    times = {4: [], 8: [], 16: []}
    for rnd in range(3):
        for w in (4, 8, 16):
            _lib.set_tuning("attn_waves", w)
            y = ops.attention(qs[0], ks[0], vs[0], Hh)
            if ref is None:
                ref = y
            eq = bool(torch.equal(y, ref))
            if not eq:
                line += " w%d NOT-EQUAL(%.2e)" % (w, float((y.float() - ref.float()).abs().max()))
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(12):
                ops.attention(qs[i % 3], ks[i % 3], vs[i % 3], Hh)
            e1.record()
            torch.cuda.synchronize()
            times[w].append(e0.elapsed_time(e1) / 12 * 1e3)
    fl = 4.0 * B * Hh * L * Lk * 64
    for w in (4, 8, 16):
        us = sorted(times[w])[1]
        line += "  w%d %7.1f us (%6.1f TF)" % (w, us, fl / us * 1e-6)
    print(line, flush=True)
_lib.set_tuning("attn_waves", 0)
