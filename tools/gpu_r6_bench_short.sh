#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-tolerance-modes --no-batch1 > gpurun_out/r6_bench_short.json 2> gpurun_out/r6_bench_short.err; echo rc=$?
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r6_bench_short.json").read().strip().splitlines()[-1]); r=d["roofline"]
print(d["value"], d["ms_per_step"], "fwd", r["forward_ms"], "eager", r.get("forward_ms_eager"), "graph", r.get("forward_ms_graph_replay"), "frac", r["frac"], "fp8", r["unet_fp8"].get("speedup_vs_bf16_forward"))
PY
tail -3 gpurun_out/r6_bench_short.err
