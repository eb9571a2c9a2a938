#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
python tools/attn_waves_bench.py > gpurun_out/r6_attn_waves.txt 2>&1; cat gpurun_out/r6_attn_waves.txt | tail -12
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_sdxl_gpu.py tests/test_fp16_gpu.py -m gpu -x -q -k "attention or attn or flash or pingpong" 2>&1 | tail -3
