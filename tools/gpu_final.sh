#!/bin/bash
# End-of-round sequence: what the driver runs (pytest -m gpu, smoke, bench) + copies for profiles/.
mkdir -p gpurun_out/final
export TMPDIR=/tmp
(timeout 1500 python -m pytest tests -m gpu -q --tb=short > gpurun_out/final/pytest_all.log 2>&1; echo "rc=$?" >> gpurun_out/final/pytest_all.log)
tail -4 gpurun_out/final/pytest_all.log
timeout 300 python __graft_entry__.py smoke > gpurun_out/final/smoke.log 2>&1; tail -1 gpurun_out/final/smoke.log
timeout 1200 python bench.py > gpurun_out/final/bench.log 2>&1; tail -1 gpurun_out/final/bench.log > gpurun_out/final/round2_bench.json; cut -c1-700 gpurun_out/final/round2_bench.json
