#!/bin/bash
# round 4: GEMM core micro-benchmarks (8-wave 16x16x32 tiles vs the 4-wave / AGPR-accumulator tiles), no torch
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
LIB=seed-story_amd/lib/libseedstory_hip.so
timeout 600 tools/bin/gemm_ubench $LIB "$@" > gpurun_out/ubench.log 2>&1
echo "rc=$?" >> gpurun_out/ubench.log
tail -120 gpurun_out/ubench.log
