#!/bin/bash
# round 6, step 3: re-tune the GEMM table with the ping-pong tiles, then bench old table vs new table on the same box
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 900 python tools/retune.py --out gpurun_out/tune_gfx950.json > gpurun_out/r6_retune.txt 2>&1
tail -5 gpurun_out/r6_retune.txt
for tab in old new; do
  if [ $tab = new ]; then export SEEDSTORY_TUNE_TABLE=$PWD/gpurun_out/tune_gfx950.json; else unset SEEDSTORY_TUNE_TABLE; fi
  timeout 900 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-tolerance-modes --no-batch1 > gpurun_out/r6_bench_$tab.json 2> gpurun_out/r6_bench_$tab.err
  python - <<PY
import json
try:
    d=json.loads(open('gpurun_out/r6_bench_$tab.json').read().strip().splitlines()[-1])
    r=d.get('roofline',{})
    print('$tab', d['value'], d['ms_per_step'], 'forward_ms', r.get('forward_ms'), 'frac', r.get('frac'), 'dom', r.get('dominant_kernel',{}))
except Exception as e:
    print('$tab parse failed', e)
PY
done
tail -3 gpurun_out/r6_bench_new.err
