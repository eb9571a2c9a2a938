#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
R=$PWD
mkdir -p gpurun_out
export TMPDIR=/tmp
python tools/attn_waves_bench.py > gpurun_out/r6_attn_waves.txt 2>&1; tail -10 gpurun_out/r6_attn_waves.txt
timeout 1200 python -m pytest tests/test_kernels_gpu.py tests/test_sdxl_gpu.py tests/test_fp16_gpu.py tests/test_attn_processor.py tests/test_engine_gpu.py -m gpu -x -q -k "attention or attn or flash or pingpong or engine or lvlm or generate" 2>&1 | tail -3
for k in 4 0 4 0; do
  rm -rf /tmp/tr_a
  (cd /tmp && export SS_UNET_BATCH=16 SS_ATTN_WAVES=$k && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_a -o t -- python $R/tools/unet_trace.py > $R/gpurun_out/r6_unet_trace_aw.log 2>&1)
  python tools/trace_summary.py $(find /tmp/tr_a -name "*kernel_trace.csv" | head -1) 3 > gpurun_out/r6d_unet_b16_trace_aw$k.txt 2>&1
  echo "attn_waves=$k: $(grep 'wall ms' gpurun_out/r6_unet_trace_aw.log) $(head -1 gpurun_out/r6d_unet_b16_trace_aw$k.txt) flash ms/fwd: $(grep flash gpurun_out/r6d_unet_b16_trace_aw$k.txt | sed 's/.*ms\/fwd= *//' | paste -sd+ | bc)"
done
