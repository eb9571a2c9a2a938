#!/bin/bash
# round-2 GPU call 1: new kernels' parity first, then the tile table, A/B timings, full suite, a short bench
mkdir -p gpurun_out
export TMPDIR=/tmp
(timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_sdxl_gpu.py -m gpu -q -x --tb=short -k "gemm or conv or attention" > gpurun_out/pytest_kernels.log 2>&1; echo "rc=$?" >> gpurun_out/pytest_kernels.log)
tail -5 gpurun_out/pytest_kernels.log
(timeout 600 python tools/kbench.py tune > gpurun_out/tune_stdout.txt 2> gpurun_out/tune_log.txt; echo "rc=$?" >> gpurun_out/tune_stdout.txt)
tail -3 gpurun_out/tune_stdout.txt
cp gpurun_out/tune_gfx950.json seed-story_amd/seedstory/tune_gfx950.json 2>/dev/null
(timeout 300 python tools/kbench.py attn > gpurun_out/attn_ab.log 2>&1); tail -7 gpurun_out/attn_ab.log
(timeout 300 python tools/kbench.py unet --batch 8 > gpurun_out/unet_b8.log 2>&1); tail -2 gpurun_out/unet_b8.log
(timeout 1200 python -m pytest tests/test_fulldim_gpu.py -m gpu -q --tb=short -s > gpurun_out/pytest_fulldim.log 2>&1; echo "rc=$?" >> gpurun_out/pytest_fulldim.log)
tail -5 gpurun_out/pytest_fulldim.log
(timeout 900 python -m pytest tests -m gpu -q --tb=short --deselect tests/test_fulldim_gpu.py -k "not (gemm or conv or attention)" > gpurun_out/pytest_rest.log 2>&1; echo "rc=$?" >> gpurun_out/pytest_rest.log)
tail -5 gpurun_out/pytest_rest.log
(timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench_short.log 2>&1); tail -1 gpurun_out/bench_short.log | cut -c1-1500
