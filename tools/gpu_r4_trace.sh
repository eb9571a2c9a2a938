#!/bin/bash
# round 4: per-kernel breakdown of the UNet forward (batch 8) -> gpurun_out/<name>.txt
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
export TMPDIR=/tmp
NAME=${1:-r4_unet_trace}
R=$PWD
rm -rf /tmp/tr; cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr -o t -- python $R/tools/unet_trace.py > $R/gpurun_out/${NAME}.log 2>&1; cd $R
python tools/trace_summary.py $(find /tmp/tr -name "*kernel_trace.csv" | head -1) 3 > gpurun_out/${NAME}.txt 2>&1
head -60 gpurun_out/${NAME}.txt
