#!/bin/bash
# round 3, call B: producer-carried LayerNorm statistics (tests + A/B), cross-attention timing, per-kernel trace
mkdir -p gpurun_out
export TMPDIR=/tmp
(timeout 900 python -m pytest tests/test_lnfold_gpu.py tests/test_sdxl_gpu.py tests/test_engine_gpu.py -q -x --tb=short -s > gpurun_out/r3b_tests.log 2>&1; echo "rc=$?" >> gpurun_out/r3b_tests.log)
tail -3 gpurun_out/r3b_tests.log
(timeout 900 python -m pytest tests/test_fulldim_gpu.py -q -s --tb=short -k "transformer_block or assembled_full_size_bf16" > gpurun_out/r3b_full.log 2>&1; echo "rc=$?" >> gpurun_out/r3b_full.log)
tail -3 gpurun_out/r3b_full.log
for i in 1 2; do
KB_LNFOLD=1 timeout 600 python tools/kbench.py unet --batch 8 2>&1 | tail -1
KB_LNFOLD=0 timeout 600 python tools/kbench.py unet --batch 8 2>&1 | tail -1
done > gpurun_out/r3b_ab.log 2>&1; cat gpurun_out/r3b_ab.log
timeout 300 python tools/kbench.py cross > gpurun_out/r3b_cross.log 2>&1; tail -3 gpurun_out/r3b_cross.log
cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr -o t -- python $GRAFT_REPO_ROOT/tools/unet_trace.py > $GRAFT_REPO_ROOT/gpurun_out/r3b_trace.log 2>&1; cd $GRAFT_REPO_ROOT
python tools/trace_summary.py $(find /tmp/tr -name "*kernel_trace.csv" | head -1) 3 > gpurun_out/r3b_unet_trace.txt 2>&1; head -30 gpurun_out/r3b_unet_trace.txt
