#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
(timeout 900 python -m pytest tests/test_engine_gpu.py tests/test_fulldim_gpu.py -m gpu -q --tb=short -s -k "forward_call or kv_cache_head or lora" 2>&1 | grep -E "LoRA|passed|failed|FAILED|Error|assert" | head -20)
export SS_BENCH_SINGLE_DEVICE=1
(timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 2 --warmup 1 --partition slots --no-cpu-baseline > gpurun_out/bench_2rank_slots.log 2>&1); tail -c 1800 gpurun_out/bench_2rank_slots.log
(timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 2 --steps 2 --warmup 1 --stories-per-gpu 2 --no-cpu-baseline --no-batch1 > gpurun_out/bench_2rank_replicas.log 2>&1); tail -c 600 gpurun_out/bench_2rank_replicas.log
