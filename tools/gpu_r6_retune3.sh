#!/bin/bash
# round 6: new ping-pong tests, re-tune with cfg 57 / 58 among the candidates, forward trace shipped-table vs new-table (interleaved)
cd "$GRAFT_REPO_ROOT" || exit 1
R=$PWD
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_kernels_gpu.py tests/test_sdxl_gpu.py -m gpu -x -q -k "pingpong or persistent_multi_tile or dma_tile_configs" > gpurun_out/r6_pp_tests.txt 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r6_pp_tests.txt
timeout 1500 python tools/retune.py --conv --out gpurun_out/tune_gfx950.json > gpurun_out/r6_retune3.txt 2>&1
tail -2 gpurun_out/r6_retune3.txt
for rep in 1 2; do
for tab in old new; do
  if [ $tab = new ]; then export SEEDSTORY_TUNE_TABLE=$R/gpurun_out/tune_gfx950.json; else unset SEEDSTORY_TUNE_TABLE; fi
  rm -rf /tmp/tr_$tab
  (cd /tmp && export SS_UNET_BATCH=16 && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_$tab -o t -- python $R/tools/unet_trace.py > $R/gpurun_out/r6_unet_trace_$tab.log 2>&1)
  python tools/trace_summary.py $(find /tmp/tr_$tab -name "*kernel_trace.csv" | head -1) 3 > gpurun_out/r6c_unet_b16_trace_${tab}_$rep.txt 2>&1
  echo "$tab $rep: $(grep 'wall ms' gpurun_out/r6_unet_trace_$tab.log) $(head -1 gpurun_out/r6c_unet_b16_trace_${tab}_$rep.txt)"
done
done
head -16 gpurun_out/r6c_unet_b16_trace_new_2.txt
