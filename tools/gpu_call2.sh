#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
(timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_sdxl_gpu.py -m gpu -q -x --tb=short -k "gemm or conv" > gpurun_out/pytest_kernels.log 2>&1; echo "rc=$?" >> gpurun_out/pytest_kernels.log)
tail -4 gpurun_out/pytest_kernels.log
rm -f seed-story_amd/seedstory/tune_gfx950.json
(timeout 600 python tools/kbench.py tune > gpurun_out/tune_stdout.txt 2> gpurun_out/tune_log.txt; echo "rc=$?" >> gpurun_out/tune_stdout.txt)
tail -2 gpurun_out/tune_stdout.txt
cp gpurun_out/tune_gfx950.json seed-story_amd/seedstory/tune_gfx950.json 2>/dev/null
(timeout 300 python tools/kbench.py unet --batch 8 > gpurun_out/unet_b8.log 2>&1); tail -1 gpurun_out/unet_b8.log
rm -rf gpurun_out/utrace
(timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/utrace -o u -- python tools/kbench.py unet --batch 8 > gpurun_out/utrace.log 2>&1)
f=$(find gpurun_out/utrace -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && python tools/trace_summary.py $f 6 > gpurun_out/unet_b8_trace.txt; rm -rf gpurun_out/utrace
head -30 gpurun_out/unet_b8_trace.txt
bash tools/run_pmc.sh > gpurun_out/run_pmc.log 2>&1; tail -5 gpurun_out/run_pmc.log
(timeout 1200 python -m pytest tests/test_fulldim_gpu.py tests/test_sdxl_gpu.py -m gpu -q --tb=short -s -k "full_size or lvlm or get_image_embeds or tiny" > gpurun_out/pytest_new.log 2>&1; echo "rc=$?" >> gpurun_out/pytest_new.log)
grep -E "HIP vs|img_gen_feat|passed|failed|Error" gpurun_out/pytest_new.log | tail -20
