"""The stacked image-token block of the MLLM half (GRP slots x 66 rows through LlamaEngine.prefill_batch against a 400-token
cache), 3 passes behind a marker kernel — to be wrapped by rocprofv3 --kernel-trace and summarised by trace_summary.py.
  SS_BLOCK_SLOTS = slots per group (default 8)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "seed-story_amd"))
import torch  # noqa: E402

import bench  # noqa: E402
from seedstory import ops  # noqa: E402

dev, dt = "cuda:0", torch.bfloat16
GRP = int(os.environ.get("SS_BLOCK_SLOTS", "8"))
eng, _ = bench.build_engine(dev, dt, GRP)
z = [torch.zeros(66, bench.H, device=dev, dtype=dt)] * GRP


def one():
    for b in range(GRP):
        eng.select(b).set_lengths(400, 400)
    if GRP == 1:
        eng.select(0).prefill(z[0])
    else:
        eng.prefill_batch(z)


one()
torch.cuda.synchronize()
ops.softmax_rows_(torch.zeros(1, 8, device=dev, dtype=dt), 1.0)      # marker: trace_summary counts what follows it
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3):
    one()
torch.cuda.synchronize()
print("wall ms per block pass (%d slots x 66 rows): %.3f" % (GRP, (time.perf_counter() - t0) / 3 * 1e3))
