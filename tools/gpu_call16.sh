#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/summary3
rm -rf gpurun_out/r2; mkdir -p gpurun_out/r2
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" "SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout 120 rocprofv3 --pmc $set --kernel-trace --output-format csv -d gpurun_out/r2/k_$i -o p -- python tools/pmc_kernels.py > gpurun_out/r2_k$i.log 2>&1
done
find gpurun_out/r2 -name "*kernel_trace.csv" -delete
python tools/summarize_round2.py gpurun_out/summary3 > gpurun_out/r2_summary.log 2>&1; tail -c 300 gpurun_out/r2_summary.log; echo
rm -rf gpurun_out/r2
(timeout 400 python bench.py --unet-fp8 --story-len 25 --kv-reuse --steps 4 --warmup 1 --no-cpu-baseline --no-batch1 > gpurun_out/bench_cfg4_fp8.log 2>&1); tail -1 gpurun_out/bench_cfg4_fp8.log | cut -c1-400
(timeout 400 python bench.py --kv-reuse --steps 5 --warmup 2 --no-cpu-baseline --no-batch1 > gpurun_out/bench_kvreuse.log 2>&1); tail -1 gpurun_out/bench_kvreuse.log | cut -c1-300
