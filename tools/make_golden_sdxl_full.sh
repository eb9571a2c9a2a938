#!/bin/bash
# Regenerates tests/golden/sdxl_full_truth.safetensors: the fp32 CPU-oracle outputs of the whole de-tokenizer at real size
# (tests/test_fulldim_gpu.py::sdxl_full_truth, ~200 s of host time).  The 2.57 B UNet weights are drawn on the GPU, so this
# runs on the GPU box; the file is keyed on a checksum of every weight and input and is ignored (recomputed) when the key
# differs.  Usage, from the container:
#   /usr/local/graft/bin/gpurun --timeout 900 -- 'bash tools/make_golden_sdxl_full.sh'
#   cp gpurun_out/sdxl_full_truth.safetensors tests/golden/
set -e
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
SS_WRITE_GOLDEN=gpurun_out/sdxl_full_truth.safetensors python -m pytest tests/test_fulldim_gpu.py -q -s -m gpu \
    -k "sdxl_unet_assembled_full_size" 2>&1 | tail -40
