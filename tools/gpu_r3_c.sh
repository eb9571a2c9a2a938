#!/bin/bash
# round 3, call C: batched MLLM parts + RCCL world-1 tests, 4 vs 8 stories per GPU, kernel stats + GEMV PMC of the shipped schedule
mkdir -p gpurun_out/summary
export TMPDIR=/tmp
(timeout 1200 python -m pytest tests/test_engine_gpu.py tests/test_boundary_gpu.py -q -x --tb=short > gpurun_out/r3c_tests.log 2>&1; echo "rc=$?" >> gpurun_out/r3c_tests.log)
tail -3 gpurun_out/r3c_tests.log
B="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-batch1"
timeout 900 $B --stories-per-gpu 8 --save-tune-table gpurun_out/tune_b16.json > gpurun_out/r3c_bench_spg8.log 2>&1; tail -1 gpurun_out/r3c_bench_spg8.log | cut -c1-400
timeout 900 $B --stories-per-gpu 4 > gpurun_out/r3c_bench_spg4.log 2>&1; tail -1 gpurun_out/r3c_bench_spg4.log | cut -c1-400
rm -rf gpurun_out/r3; mkdir -p gpurun_out/r3
timeout 700 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3/stats_overlap -o b -- $B > gpurun_out/r3c_overlap.log 2>&1
timeout 700 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3/stats_serial -o b -- $B --no-overlap > gpurun_out/r3c_serial.log 2>&1
find gpurun_out/r3 -name "*kernel_trace.csv" -delete
B1="python bench.py --mllm-only --steps 1 --warmup 0 --no-cpu-baseline --no-batch1"
timeout 500 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/r3/fetch -o p -- $B1 > gpurun_out/r3c_fetch.log 2>&1
timeout 500 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/r3/write -o p -- $B1 > gpurun_out/r3c_write.log 2>&1
find gpurun_out/r3 -name "*kernel_trace.csv" -delete
python tools/summarize_round3.py gpurun_out/summary > gpurun_out/r3c_summary.log 2>&1; tail -2 gpurun_out/r3c_summary.log | cut -c1-600
rm -rf gpurun_out/r3
ls -la gpurun_out/summary
