#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
(timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_sdxl_gpu.py -m gpu -q -x --tb=short -k "(gemm or conv) and (70 or 71 or 72)" 2>&1 | tail -3)
rm -f seed-story_amd/seedstory/tune_gfx950.json
(timeout 900 python tools/kbench.py tune > gpurun_out/tune_stdout.txt 2> gpurun_out/tune_log.txt; echo "rc=$?" >> gpurun_out/tune_stdout.txt)
tail -2 gpurun_out/tune_stdout.txt
cp gpurun_out/tune_gfx950.json seed-story_amd/seedstory/tune_gfx950.json 2>/dev/null
(timeout 300 python tools/kbench.py unet --batch 8 > gpurun_out/unet_b8.log 2>&1); tail -1 gpurun_out/unet_b8.log
SEEDSTORY_TUNE_TABLE=/nonexistent python tools/kbench.py unet --batch 8 2>&1 | tail -1
rm -rf gpurun_out/utrace
(timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/utrace -o u -- python tools/kbench.py unet --batch 8 > gpurun_out/utrace.log 2>&1)
f=$(find gpurun_out/utrace -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && python tools/trace_summary.py $f 3 > gpurun_out/unet_b8_trace.txt; rm -rf gpurun_out/utrace
head -32 gpurun_out/unet_b8_trace.txt
(timeout 300 python tools/kbench.py vae 2>&1 | tail -1)
(timeout 300 python tools/kbench.py unet --batch 2 2>&1 | tail -1)
