#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
export SS_BENCH_SINGLE_DEVICE=1 SS_BENCH_WATCHDOG_S=150
(timeout 240 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29521 bench.py --gpus 2 --steps 2 --warmup 1 --partition slots --stories-per-gpu 1 --mllm-only --no-cpu-baseline > gpurun_out/bench_2rank_slots_mllm.log 2>&1); grep -v "^\[W\|amdgpu.ids\|^\*\*\|OMP_NUM" gpurun_out/bench_2rank_slots_mllm.log | tail -c 1500
(timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29522 bench.py --gpus 2 --steps 2 --warmup 1 --partition slots --stories-per-gpu 1 --diffusion-steps 4 --no-cpu-baseline > gpurun_out/bench_2rank_slots.log 2>&1); grep -v "^\[W\|amdgpu.ids\|^\*\*\|OMP_NUM" gpurun_out/bench_2rank_slots.log | tail -c 2500
