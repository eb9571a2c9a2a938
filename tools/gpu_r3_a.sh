#!/bin/bash
# round 3, call A: new full-dimension parity tests + UNet forward at batch 8 / 16
mkdir -p gpurun_out
export TMPDIR=/tmp
(timeout 1500 python -m pytest tests/test_frontend_full_gpu.py -q -s --tb=short > gpurun_out/r3a_frontend.log 2>&1; echo "rc=$?" >> gpurun_out/r3a_frontend.log)
tail -3 gpurun_out/r3a_frontend.log
(timeout 1500 python -m pytest tests/test_fulldim_gpu.py -q -s --tb=short -k "assembled" > gpurun_out/r3a_unet_full.log 2>&1; echo "rc=$?" >> gpurun_out/r3a_unet_full.log)
tail -3 gpurun_out/r3a_unet_full.log
timeout 600 python tools/kbench.py unet --batch 8 > gpurun_out/r3a_unet_b8.log 2>&1; tail -1 gpurun_out/r3a_unet_b8.log
timeout 900 python tools/kbench.py unet --batch 16 > gpurun_out/r3a_unet_b16.log 2>&1; tail -1 gpurun_out/r3a_unet_b16.log
lscpu | grep -i -E "model name|^CPU\(s\)" > gpurun_out/r3a_cpu.txt; grep -o -E "amx_bf16|avx512_bf16" /proc/cpuinfo | sort | uniq -c >> gpurun_out/r3a_cpu.txt; cat gpurun_out/r3a_cpu.txt
