#!/bin/bash
# round 6: full GPU suite (complete output kept) + smoke + default bench
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r6_pytest_gpu_full.txt 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r6_pytest_gpu_full.txt
tail -5 gpurun_out/r6_pytest_gpu_full.txt
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -2
timeout 1200 python bench.py > gpurun_out/r6_bench.json 2> gpurun_out/r6_bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r6_bench.json').read().strip().splitlines()[-1])
r=d.get('roofline',{})
print(d['value'], d['ms_per_step'], 'forward_ms', r.get('forward_ms'), 'frac', r.get('frac'), r.get('dominant_kernel'))
print('batch1', d.get('batch1'))
print('control', r.get('gemm_8192cubed_control'))
print('gate', d.get('tolerance_modes',{}).get('gate_mode'))
PY
