#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
(timeout 900 python -m pytest tests/test_boundary_gpu.py tests/test_checkpoints.py tests/test_preprocess.py tests/test_engine_gpu.py -m gpu -q --tb=short 2>&1 | tail -40) > gpurun_out/pytest_call8.log 2>&1
tail -25 gpurun_out/pytest_call8.log
(timeout 600 python -m pytest tests/test_fulldim_gpu.py -m gpu -q -x --tb=short -s -k "vae_decode_full_size" 2>&1 | tail -15) > gpurun_out/pytest_vae_full.log 2>&1
tail -8 gpurun_out/pytest_vae_full.log
cp seed-story_amd/seedstory/tune_gfx950.json gpurun_out/tune_prev.json
rm -f seed-story_amd/seedstory/tune_gfx950.json
(timeout 1200 python tools/kbench.py tune > gpurun_out/tune_stdout.txt 2> gpurun_out/tune_log.txt; echo "rc=$?" >> gpurun_out/tune_stdout.txt)
tail -2 gpurun_out/tune_stdout.txt
cp gpurun_out/tune_gfx950.json seed-story_amd/seedstory/tune_gfx950.json 2>/dev/null
(timeout 300 python tools/kbench.py unet --batch 8 2>&1 | tail -1)
SEEDSTORY_TUNE_TABLE=gpurun_out/tune_prev.json python tools/kbench.py unet --batch 8 2>&1 | tail -1
SEEDSTORY_TUNE_TABLE=/nonexistent python tools/kbench.py unet --batch 8 2>&1 | tail -1
