#!/bin/bash
# round 6: the full GPU suite three times back to back with the complete output kept, then test_engine_gpu.py ten times
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
for i in 1 2 3; do
  timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r6_suite_run$i.txt 2>&1; echo "run $i rc=$?" | tee -a gpurun_out/r6_suite_run$i.txt
  tail -2 gpurun_out/r6_suite_run$i.txt
done
for i in $(seq 1 10); do
  timeout 600 python -m pytest tests/test_engine_gpu.py -m gpu -x -q > gpurun_out/r6_engine_loop_$i.txt 2>&1; echo "engine loop $i rc=$? $(tail -1 gpurun_out/r6_engine_loop_$i.txt)"
done | tee gpurun_out/r6_engine_loop_summary.txt
dmesg 2>/dev/null | tail -5
