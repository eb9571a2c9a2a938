#!/bin/bash
# Round-2 judged sequence on the GPU box: parity tests, smoke, bench, kernel stats (overlapped and serial schedule),
# counter passes.  Small summaries -> gpurun_out/summary (copied to profiles/ by hand).
mkdir -p gpurun_out/summary
export TMPDIR=/tmp
if [ -z "$SKIP_TESTS" ]; then
(timeout 1500 python -m pytest tests -m gpu -q --tb=short > gpurun_out/pytest_all.log 2>&1; echo "rc=$?" >> gpurun_out/pytest_all.log)
tail -5 gpurun_out/pytest_all.log
timeout 300 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; tail -1 gpurun_out/smoke.log
fi
if [ -z "$SKIP_BENCH" ]; then
timeout 1200 python bench.py > gpurun_out/bench_full.log 2>&1; tail -1 gpurun_out/bench_full.log > gpurun_out/summary/round2_bench.json; tail -c 1500 gpurun_out/bench_full.log
fi
rm -rf gpurun_out/r2; mkdir -p gpurun_out/r2
B="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-batch1"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2/stats_overlap -o b -- $B > gpurun_out/r2_overlap.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2/stats_serial -o b -- $B --no-overlap > gpurun_out/r2_serial.log 2>&1
find gpurun_out/r2 -name "*kernel_trace.csv" -delete
B1="python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-batch1"
timeout 500 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/r2/fetch -o p -- $B1 > gpurun_out/r2_fetch.log 2>&1
timeout 500 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/r2/write -o p -- $B1 > gpurun_out/r2_write.log 2>&1
find gpurun_out/r2 -name "*kernel_trace.csv" -delete
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" "SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d gpurun_out/r2/k_$i -o p -- python tools/pmc_kernels.py > gpurun_out/r2_k$i.log 2>&1
done
find gpurun_out/r2 -name "*kernel_trace.csv" -delete
python tools/summarize_round2.py gpurun_out/summary > gpurun_out/r2_summary.log 2>&1; tail -3 gpurun_out/r2_summary.log
rm -rf gpurun_out/r2
ls -la gpurun_out/summary
