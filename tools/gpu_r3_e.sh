#!/bin/bash
# round 3, call E: split-K small-M GEMM (tests + effect on the block continuation / MLLM half)
mkdir -p gpurun_out/summary
export TMPDIR=/tmp
(timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x --tb=short -k "splitk" > gpurun_out/r3e_splitk.log 2>&1; echo "rc=$?" >> gpurun_out/r3e_splitk.log); tail -3 gpurun_out/r3e_splitk.log
(timeout 900 python -m pytest tests/test_engine_gpu.py tests/test_fulldim_gpu.py -q -x --tb=short -k "llama or prefill or img_block or generate" > gpurun_out/r3e_engine.log 2>&1; echo "rc=$?" >> gpurun_out/r3e_engine.log); tail -3 gpurun_out/r3e_engine.log
B="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-batch1"
timeout 900 $B --save-tune-table gpurun_out/tune_spg4.json > gpurun_out/r3e_bench.log 2>&1; tail -1 gpurun_out/r3e_bench.log | cut -c1-300
timeout 600 $B --mllm-only > gpurun_out/r3e_mllm_only.log 2>&1; tail -1 gpurun_out/r3e_mllm_only.log | cut -c1-300
timeout 600 $B --mllm-only --no-splitk > gpurun_out/r3e_mllm_only_nosplit.log 2>&1; tail -1 gpurun_out/r3e_mllm_only_nosplit.log | cut -c1-300
