#!/bin/bash
# Standard GPU-box sequence: parity tests, smoke, bench, rocprofv3 kernel stats.  Outputs -> gpurun_out/
mkdir -p gpurun_out
if [ -z "$SKIP_TESTS" ]; then
(timeout 600 python -m pytest tests -m gpu -q --tb=short > gpurun_out/pytest_all.log 2>&1; echo "rc=$?" >> gpurun_out/pytest_all.log)
tail -4 gpurun_out/pytest_all.log
timeout 120 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; tail -1 gpurun_out/smoke.log
timeout 600 python bench.py "$@" > gpurun_out/bench.log 2>&1; tail -1 gpurun_out/bench.log
fi
export TMPDIR=/tmp
rm -rf gpurun_out/prof
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o bench -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > gpurun_out/prof.log 2>&1
tail -2 gpurun_out/prof.log
find gpurun_out/prof -name "*stats*" | head; 
f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -25 "$f"
# separate PMC passes (never combined with sys/hip traces): HBM read / write bytes per kernel
rm -rf gpurun_out/pmc
timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc/fetch -o pmc -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/pmc_fetch.log 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc/write -o pmc -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/pmc_write.log 2>&1
tail -2 gpurun_out/pmc_fetch.log
find gpurun_out/pmc -name "*.csv" | head
# keep only the small summaries (the raw traces / counter dumps are tens of MB and gpurun_out/ is capped at 64 MiB)
python tools/summarize_prof.py gpurun_out/summary round1_rocprof > /dev/null
rm -rf gpurun_out/pmc
find gpurun_out/prof -name "*kernel_trace.csv" -delete
