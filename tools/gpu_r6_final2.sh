#!/bin/bash
# round 6, closing sequence after the pipelined epilogue: forward trace base-lib vs this lib (same box, interleaved), full suite,
# smoke, the driver's bench command, rocprofv3 --kernel-trace --stats of the bench command.
cd "$GRAFT_REPO_ROOT" || exit 1
R=$PWD
mkdir -p gpurun_out
export TMPDIR=/tmp
STAGE=${1:-all}
LB=$R/seed-story_amd/lib/ab/libseedstory_hip_base.so
if [ $STAGE = all ] || [ $STAGE = trace ]; then
  for rep in 1 2; do for tab in base new; do
    if [ $tab = base ]; then export SEEDSTORY_HIP_LIB=$LB; else unset SEEDSTORY_HIP_LIB; fi
    rm -rf /tmp/tr_$tab
    (cd /tmp && export SS_UNET_BATCH=16 && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_$tab -o t -- python $R/tools/unet_trace.py > $R/gpurun_out/r6e_unet_trace_${tab}_$rep.log 2>&1)
    python tools/trace_summary.py $(find /tmp/tr_$tab -name "*kernel_trace.csv" | head -1) 3 > gpurun_out/r6e_unet_b16_trace_${tab}_$rep.txt 2>&1
    echo "$tab $rep: $(tail -1 gpurun_out/r6e_unet_trace_${tab}_$rep.log) | $(head -1 gpurun_out/r6e_unet_b16_trace_${tab}_$rep.txt)"
  done; done
  unset SEEDSTORY_HIP_LIB
  head -14 gpurun_out/r6e_unet_b16_trace_new_2.txt
fi
if [ $STAGE = all ] || [ $STAGE = suite ]; then
  timeout 1800 python -m pytest tests -m gpu -x -q > gpurun_out/r6e_pytest_gpu_final.txt 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r6e_pytest_gpu_final.txt; grep -E "passed|failed" gpurun_out/r6e_pytest_gpu_final.txt | tail -1
  timeout 300 python __graft_entry__.py smoke 2>&1 | tail -1
fi
if [ $STAGE = all ] || [ $STAGE = bench ]; then
  T0=$(date +%s)
  python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r6e_bench_driver_like.json 2> gpurun_out/r6e_bench_driver_like.err; echo "bench rc=$? wall=$(( $(date +%s) - T0 ))s"
  python - <<'PY'
import json
d=json.loads(open("gpurun_out/r6e_bench_driver_like.json").read().strip().splitlines()[-1]); r=d["roofline"]
print(d["value"], d["ms_per_step"], d["steps"], "fwd", r["forward_ms"], "frac", r["frac"], "traffic", r["traffic"], "batch1", d["batch1"]["value"])
print(r["dominant_kernel"]); print(str(r["gemm_8192cubed_control"])[:400])
print(d["tolerance_modes"]["gate_mode"]["value_full_pipeline"], d["cpu_baseline"]["seconds_per_story_step"])
PY
fi
if [ $STAGE = all ] || [ $STAGE = stats ]; then
  rm -rf gpurun_out/r6estats
  B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-tolerance-modes --no-batch1"
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r6estats -o b -- $B > gpurun_out/r6e_stats.log 2>&1; echo "prof rc=$?"
  f=$(find gpurun_out/r6estats -name "*kernel_stats.csv" | head -1); head -8 "$f" | cut -c1-200; cp "$f" gpurun_out/r6e_bench_kernel_stats.csv
  grep '^{"metric' gpurun_out/r6e_stats.log | tail -1 > gpurun_out/r6e_bench_under_tracer.json; cut -c1-160 gpurun_out/r6e_bench_under_tracer.json
  find gpurun_out/r6estats -name "*.csv" -size +1M -delete
fi
