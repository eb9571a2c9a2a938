#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
python -m pytest tests/test_fp8_gpu.py -m gpu -q --tb=line -s 2>&1 | grep -E "transformer block|UNet forward|passed|failed|FAILED"
(timeout 600 python tools/kbench.py fp8 --batch 8 > gpurun_out/fp8_bench.log 2>&1); grep -E "unet_forward|'M'" gpurun_out/fp8_bench.log | cut -c1-400
