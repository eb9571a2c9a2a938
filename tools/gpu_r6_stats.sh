#!/bin/bash
# round 6: rocprofv3 --kernel-trace --stats of the bench command (shipped schedule), summary -> profiles/round6_bench_kernel_stats.csv
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out; rm -rf gpurun_out/r6stats
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-tolerance-modes --no-batch1"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r6stats -o b -- $B > gpurun_out/r6_stats.log 2>&1; echo "prof rc=$?"
f=$(find gpurun_out/r6stats -name "*kernel_stats.csv" | head -1); head -14 "$f" | cut -c1-220; cp "$f" gpurun_out/r6_bench_kernel_stats.csv
grep '^{"metric' gpurun_out/r6_stats.log | tail -1 > gpurun_out/r6_bench_under_tracer.json; cut -c1-160 gpurun_out/r6_bench_under_tracer.json
find gpurun_out/r6stats -name "*.csv" -size +1M -delete
