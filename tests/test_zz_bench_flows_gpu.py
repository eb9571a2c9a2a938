"""GPU flow tests of `bench.py` modes added in round 5, in a file that sorts LAST: the round-end run is `pytest -x`, and a whole-process
bench flow (40 GB of models, a subprocess) is the kind of test that should not stand in front of the kernel / parity tests."""
import pytest

pytestmark = pytest.mark.gpu


def test_bench_sink_flow_evicts_on_the_slab():
    """`bench.py --sink` end to end (BASELINE configs[4] as named, shortened): 2 lock-step stories of 10 steps with the 8-image window,
    2 Euler steps — steps 8 and 9 of every story evict on the KV slab (ss_llama_kv_gather) instead of re-prefilling, every other
    step is the 65-row continuation.  The JSON line must say so: workload names the attention sink, two evictions per story with
    the index arithmetic of seedstory/story.py (first eviction keeps 4 + 12 + 12 sink rows, the second extends them to 52)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--sink", "--story-len", "10", "--steps", "10", "--warmup", "0",
           "--stories-per-gpu", "2", "--diffusion-steps", "2", "--no-cpu-baseline", "--no-batch1", "--no-tolerance-modes", "--no-roofline"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    sink = line["config"]["attention_sink"]
    assert "ATTENTION SINK" in line["config"]["workload"] and line["value"] > 0
    assert sink["evictions_in_last_round_stories"] == 4                      # 2 stories x (step 8, step 9)
    # step 8: kv = 1 + 9 x 114 - 65 = 962 live rows, the evicted image's </img> at index 114 -> keep 28 + (962 - 115) = 875, drop 87;
    # step 9: kv = 28 + 8 x 114 + 114 - 65 = 989, </img> at 28 + 113 -> keep 52 + (989 - 142) = 899, drop 90
    assert sink["kv_rows_kept_per_eviction"] == (875 + 899) / 2 and sink["kv_rows_dropped_per_eviction"] == (87 + 90) / 2
    assert sink["cache_cap_rows"] >= 1 + 9 * 114 + 52 + 115

