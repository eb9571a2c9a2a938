#!/bin/bash
# round 6 closing sequence: PMC on the shipped table, full suite, smoke, default bench, records for BASELINE configs[1] / [2] / [4]
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
STAGE=${1:-all}
if [ $STAGE = all ] || [ $STAGE = pmc ]; then
  bash tools/pmc_round6.sh > gpurun_out/r6_pmc.log 2>&1; python - <<'PY'
import json
d=json.load(open('gpurun_out/round6_pmc_summary.json'))['gemm_hbm_traffic']
for k,v in d.items(): print(k, v.get('cfg_swz'), v.get('profiled_us'), 'mfma', v.get('mfma_busy_frac'), 'clk', v.get('effective_clock_ghz'), 'overfetch', v.get('overfetch_ratio'))
PY
fi
if [ $STAGE = all ] || [ $STAGE = suite ]; then
  timeout 1800 python -m pytest tests -m gpu -x -q > gpurun_out/r6_pytest_gpu_final.txt 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r6_pytest_gpu_final.txt; grep -E "passed|failed" gpurun_out/r6_pytest_gpu_final.txt | tail -1
  timeout 300 python __graft_entry__.py smoke 2>&1 | tail -1
fi
if [ $STAGE = all ] || [ $STAGE = bench ]; then
  timeout 1200 python bench.py > gpurun_out/r6_bench_final.json 2> gpurun_out/r6_bench_final.err; echo "bench rc=$?"
  timeout 600 python bench.py --mllm-only --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r6_cfg1_mllm_only.json 2>/dev/null; echo "cfg1 rc=$?"
  timeout 900 python bench.py --story-len 5 --steps 5 --warmup 2 --no-cpu-baseline --no-tolerance-modes > gpurun_out/r6_cfg2_story_len5.json 2>/dev/null; echo "cfg2 rc=$?"
  timeout 1200 python bench.py --sink --story-len 25 --no-cpu-baseline --no-tolerance-modes --no-batch1 > gpurun_out/r6_cfg4_sink_len25_bf16.json 2>/dev/null; echo "cfg4 rc=$?"
  timeout 1200 python bench.py --sink --story-len 25 --unet-fp8 --no-cpu-baseline --no-tolerance-modes --no-batch1 > gpurun_out/r6_cfg4_sink_len25_fp8.json 2>/dev/null; echo "cfg4 fp8 rc=$?"
  python - <<'PY'
import json
for f in ('r6_bench_final','r6_cfg1_mllm_only','r6_cfg2_story_len5','r6_cfg4_sink_len25_bf16','r6_cfg4_sink_len25_fp8'):
    try:
        d=json.loads(open('gpurun_out/%s.json'%f).read().strip().splitlines()[-1]); r=d.get('roofline') or {}
        print(f, d['value'], d['ms_per_step'], 'fwd', r.get('forward_ms'), 'frac', r.get('frac'), 'traffic', r.get('traffic'), 'batch1', (d.get('batch1') or {}).get('value'))
        if f=='r6_bench_final': print('  dom', r.get('dominant_kernel')); print('  ctl', r.get('gemm_8192cubed_control')); print('  gate', (d.get('tolerance_modes') or {}).get('gate_mode',{}).get('value_full_pipeline'))
    except Exception as e: print(f, 'parse failed', e)
PY
fi
