cd $GRAFT_REPO_ROOT
export HSA_ENABLE_IPC_MODE_LEGACY=0
B="--gpus 2 --steps 2 --warmup 1 --stories-per-gpu 2 --diffusion-steps 2 --story-len 3 --no-cpu-baseline --no-batch1 --no-tolerance-modes --partition slots"
SS_BENCH_SINGLE_DEVICE=1 SS_BENCH_WATCHDOG_S=200 timeout 260 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29573 bench.py $B --no-roofline > gpurun_out/dbg_2rank_noroof.log 2>&1; echo "noroof rc=$?"; tail -3 gpurun_out/dbg_2rank_noroof.log | cut -c1-300
SS_BENCH_SINGLE_DEVICE=1 SS_BENCH_WATCHDOG_S=200 SS_ROOF_DEBUG=1 timeout 260 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29575 bench.py $B > gpurun_out/dbg_2rank_roof.log 2>&1; echo "roof rc=$?"; grep -n "ROOF\|Error\|error\|core dump" gpurun_out/dbg_2rank_roof.log | head -20; tail -3 gpurun_out/dbg_2rank_roof.log | cut -c1-300
