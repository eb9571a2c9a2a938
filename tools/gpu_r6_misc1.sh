#!/bin/bash
# round 6: LayerNorm rows-per-wave A/B, batch-2 UNet trace (the reference's batch-1 loop), configs[0] on the GPU box's host
cd "$GRAFT_REPO_ROOT" || exit 1
R=$PWD
mkdir -p gpurun_out
export TMPDIR=/tmp
python tools/ln_bench.py > gpurun_out/r6_ln_bench.txt 2>&1; cat gpurun_out/r6_ln_bench.txt | tail -8
rm -rf /tmp/tr_b2
(cd /tmp && export SS_UNET_BATCH=2 && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_b2 -o t -- python $R/tools/unet_trace.py > $R/gpurun_out/r6_unet_trace_b2.log 2>&1)
python tools/trace_summary.py $(find /tmp/tr_b2 -name "*kernel_trace.csv" | head -1) 3 > gpurun_out/r6_unet_b2_trace.txt 2>&1
grep "wall ms" gpurun_out/r6_unet_trace_b2.log; head -45 gpurun_out/r6_unet_b2_trace.txt
python bench.py --config0 2>/dev/null | tail -1 > gpurun_out/r6_config0_gpu_host.json; cat gpurun_out/r6_config0_gpu_host.json | cut -c1-400
