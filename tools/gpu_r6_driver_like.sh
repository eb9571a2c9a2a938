#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
T0=$(date +%s)
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r6_bench_driver_like.json 2> gpurun_out/r6_bench_driver_like.err; echo "rc=$? wall=$(( $(date +%s) - T0 ))s"
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r6_bench_driver_like.json").read().strip().splitlines()[-1]); r=d["roofline"]
print(d["value"], d["ms_per_step"], d["steps"], "fwd", r["forward_ms"], "frac", r["frac"], "traffic", r["traffic"], "batch1", d["batch1"]["value"])
print(r["dominant_kernel"]); print(r["gemm_8192cubed_control"]); print(r.get("traffic_note"))
print(d["tolerance_modes"]["gate_mode"]["value_full_pipeline"], d["cpu_baseline"]["seconds_per_story_step"])
PY
