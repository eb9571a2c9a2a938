#!/bin/bash
# round 6: per-kernel trace of the batch-16 UNet forward, shipped table vs the table in gpurun_out/ (or $1)
cd "$GRAFT_REPO_ROOT" || exit 1
R=$PWD
mkdir -p gpurun_out
NEW=${1:-$R/gpurun_out/tune_gfx950.json}
export TMPDIR=/tmp
for tab in old new; do
  if [ $tab = new ]; then export SEEDSTORY_TUNE_TABLE=$NEW; else unset SEEDSTORY_TUNE_TABLE; fi
  rm -rf /tmp/tr_$tab
  (cd /tmp && export SS_UNET_BATCH=16 && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_$tab -o t -- python $R/tools/unet_trace.py > $R/gpurun_out/r6_unet_trace_$tab.log 2>&1)
  python tools/trace_summary.py $(find /tmp/tr_$tab -name "*kernel_trace.csv" | head -1) 3 > gpurun_out/r6_unet_b16_trace_$tab.txt 2>&1
  tail -1 gpurun_out/r6_unet_trace_$tab.log; head -24 gpurun_out/r6_unet_b16_trace_$tab.txt
done
