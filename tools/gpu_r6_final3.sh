#!/bin/bash
# round 6, last GPU call: the pipelined-epilogue equality tests added after the closing suite, counters re-collected on the final build
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_sdxl_gpu.py -m gpu -x -q -k "pipelined or pingpong" > gpurun_out/r6e_pytest_pipelined.txt 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/r6e_pytest_pipelined.txt
bash tools/pmc_round6.sh > gpurun_out/r6e_pmc.log 2>&1; echo "pmc rc=$?"
python - <<'PY'
import json
d=json.load(open('gpurun_out/round6_pmc_summary.json'))['gemm_hbm_traffic']
for k,v in d.items(): print(k, v.get('cfg_swz'), v.get('profiled_us'), 'mfma', v.get('mfma_busy_frac'), 'clk', v.get('effective_clock_ghz'), 'overfetch', v.get('overfetch_ratio'))
PY
