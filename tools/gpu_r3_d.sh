#!/bin/bash
# round 3, call D: the whole GPU suite + smoke + the default bench (the judged sequence)
mkdir -p gpurun_out/summary
export TMPDIR=/tmp
(timeout 1700 python -m pytest tests -m gpu -q --tb=short > gpurun_out/r3d_pytest_all.log 2>&1; echo "rc=$?" >> gpurun_out/r3d_pytest_all.log)
tail -4 gpurun_out/r3d_pytest_all.log
timeout 300 python __graft_entry__.py smoke > gpurun_out/r3d_smoke.log 2>&1; tail -1 gpurun_out/r3d_smoke.log
timeout 1200 python bench.py > gpurun_out/r3d_bench_full.log 2>&1; tail -1 gpurun_out/r3d_bench_full.log > gpurun_out/summary/round3_bench.json; tail -c 600 gpurun_out/r3d_bench_full.log
