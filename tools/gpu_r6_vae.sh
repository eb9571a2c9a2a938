#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
R=$PWD
export TMPDIR=/tmp
rm -rf /tmp/trv
(cd /tmp && timeout 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/trv -o t -- python $R/tools/vae_trace.py > $R/gpurun_out/r6_vae_trace.log 2>&1)
python tools/trace_summary.py $(find /tmp/trv -name "*kernel_trace.csv" | head -1) 3 > gpurun_out/r6_vae_decode_kernel_trace.txt 2>&1
tail -2 gpurun_out/r6_vae_trace.log; head -30 gpurun_out/r6_vae_decode_kernel_trace.txt
