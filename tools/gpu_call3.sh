#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
(timeout 300 python -m pytest tests/test_kernels_gpu.py tests/test_fulldim_gpu.py -m gpu -q -x --tb=short -k "attention" > gpurun_out/pytest_attn.log 2>&1; echo "rc=$?" >> gpurun_out/pytest_attn.log); tail -3 gpurun_out/pytest_attn.log
(timeout 300 python tools/kbench.py attn > gpurun_out/attn_ab.log 2>&1); tail -7 gpurun_out/attn_ab.log
(timeout 300 python tools/kbench.py unet --batch 8 > gpurun_out/unet_b8.log 2>&1); tail -1 gpurun_out/unet_b8.log
rm -rf gpurun_out/utrace
(timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/utrace -o u -- python tools/kbench.py unet --batch 8 > gpurun_out/utrace.log 2>&1)
f=$(find gpurun_out/utrace -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && python tools/trace_summary.py $f 3 > gpurun_out/unet_b8_trace.txt; rm -rf gpurun_out/utrace
head -45 gpurun_out/unet_b8_trace.txt
SEEDSTORY_TUNE_TABLE=/nonexistent python tools/kbench.py unet --batch 8 2>&1 | tail -1
