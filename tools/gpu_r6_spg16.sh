#!/bin/bash
# round 6: 16 stories per GPU (UNet batch 32: two tiles per CU on the N = 1280 shapes) vs the shipped 8, same box
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 1200 python bench.py --stories-per-gpu 16 --steps 3 --warmup 2 --no-cpu-baseline --no-tolerance-modes --no-batch1 --save-tune-table gpurun_out/tune_spg16.json > gpurun_out/r6_spg16.json 2> gpurun_out/r6_spg16.err; echo "rc=$?"
timeout 1200 python bench.py --stories-per-gpu 8 --steps 4 --warmup 2 --no-cpu-baseline --no-tolerance-modes --no-batch1 > gpurun_out/r6_spg8.json 2>/dev/null; echo "rc=$?"
SEEDSTORY_TUNE_TABLE=$PWD/gpurun_out/tune_spg16.json timeout 1200 python bench.py --stories-per-gpu 16 --steps 3 --warmup 2 --no-cpu-baseline --no-tolerance-modes --no-batch1 > gpurun_out/r6_spg16b.json 2>/dev/null; echo "rc=$?"
python - <<'PY'
import json
for f in ('r6_spg16','r6_spg8','r6_spg16b'):
    try:
        d=json.loads(open('gpurun_out/%s.json'%f).read().strip().splitlines()[-1]); r=d.get('roofline') or {}
        print(f, d['value'], d['ms_per_step'], 'fwd', r.get('forward_ms'), 'batch', r.get('unet_batch'), 'frac', r.get('frac'), 'tuned', d.get('config',{}).get('tuned_in_this_process'))
    except Exception as e: print(f, 'parse failed', e)
PY
tail -3 gpurun_out/r6_spg16.err
