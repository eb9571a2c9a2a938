#!/bin/bash
mkdir -p gpurun_out/summary
export TMPDIR=/tmp
(timeout 900 python -m pytest tests/test_fp8_gpu.py tests/test_sdxl_gpu.py tests/test_fulldim_gpu.py -m gpu -q --tb=short -s -k "unet_forward_fp8 or groupnorm or resnet or resblock or unet_forward_tiny or vae_decode_tiny or pipeline" 2>&1 | grep -E "UNet forward|passed|failed|FAILED|assert" | head -20)
rm -rf gpurun_out/r2; mkdir -p gpurun_out/r2
B1="python bench.py --mllm-only --steps 1 --warmup 0 --no-cpu-baseline --no-batch1"
timeout 500 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/r2/fetch -o p -- $B1 > gpurun_out/r2_fetch.log 2>&1; tail -2 gpurun_out/r2_fetch.log | cut -c1-300
timeout 500 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/r2/write -o p -- $B1 > gpurun_out/r2_write.log 2>&1
find gpurun_out/r2 -name "*kernel_trace.csv" -delete
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" "SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d gpurun_out/r2/k_$i -o p -- python tools/pmc_kernels.py > gpurun_out/r2_k$i.log 2>&1
done
find gpurun_out/r2 -name "*kernel_trace.csv" -delete
mkdir -p gpurun_out/r2k_raw; for d in gpurun_out/r2/k_*; do f=$(find $d -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp $f gpurun_out/r2k_raw/$(basename $d).csv; done
python tools/summarize_round2.py gpurun_out/summary2 > gpurun_out/r2_summary.log 2>&1; tail -c 600 gpurun_out/r2_summary.log
du -sh gpurun_out/r2k_raw; rm -rf gpurun_out/r2
