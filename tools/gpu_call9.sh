#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
(timeout 900 python -m pytest tests/test_fp8_gpu.py -m gpu -q -x --tb=short -s 2>&1 | tail -40) > gpurun_out/pytest_fp8.log 2>&1
tail -30 gpurun_out/pytest_fp8.log
(timeout 600 python tools/kbench.py fp8 --batch 8 > gpurun_out/fp8_bench.log 2>&1); tail -12 gpurun_out/fp8_bench.log
(timeout 600 python -m pytest tests/test_fulldim_gpu.py tests/test_engine_gpu.py -m gpu -q --tb=short -k "vae_decode_full_size or gen_george_driver" 2>&1 | tail -5)
