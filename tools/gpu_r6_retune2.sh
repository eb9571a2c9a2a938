#!/bin/bash
# round 6: re-tune GEMM + conv entries with the ping-pong tiles (54/55/56), then trace the batch-16 forward: old, new, old, new
cd "$GRAFT_REPO_ROOT" || exit 1
R=$PWD
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python tools/retune.py --conv --out gpurun_out/tune_gfx950.json > gpurun_out/r6_retune2.txt 2>&1
tail -3 gpurun_out/r6_retune2.txt
for rep in 1 2; do
for tab in old new; do
  if [ $tab = new ]; then export SEEDSTORY_TUNE_TABLE=$R/gpurun_out/tune_gfx950.json; else unset SEEDSTORY_TUNE_TABLE; fi
  rm -rf /tmp/tr_$tab
  (cd /tmp && export SS_UNET_BATCH=16 && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_$tab -o t -- python $R/tools/unet_trace.py > $R/gpurun_out/r6_unet_trace_$tab.log 2>&1)
  python tools/trace_summary.py $(find /tmp/tr_$tab -name "*kernel_trace.csv" | head -1) 3 > gpurun_out/r6_unet_b16_trace_${tab}_$rep.txt 2>&1
  echo "$tab $rep: $(grep 'wall ms' gpurun_out/r6_unet_trace_$tab.log) $(head -1 gpurun_out/r6_unet_b16_trace_${tab}_$rep.txt)"
done
done
head -30 gpurun_out/r6_unet_b16_trace_new_2.txt
