#!/bin/bash
mkdir -p gpurun_out/final
export TMPDIR=/tmp
(timeout 1200 python -m pytest tests -m gpu -q --tb=short > gpurun_out/final/pytest_all.log 2>&1; echo "rc=$?" >> gpurun_out/final/pytest_all.log)
grep -E "passed|failed|rc=" gpurun_out/final/pytest_all.log | tail -3
timeout 200 python __graft_entry__.py smoke > gpurun_out/final/smoke.log 2>&1; tail -1 gpurun_out/final/smoke.log
