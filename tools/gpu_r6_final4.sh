#!/bin/bash
# round 6: BASELINE configs[1] / [2] / [4] lines re-taken on the final build
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 600 python bench.py --mllm-only --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r6f_cfg1_mllm_only.json 2>/dev/null; echo "cfg1 rc=$?"
timeout 900 python bench.py --story-len 5 --steps 5 --warmup 2 --no-cpu-baseline --no-tolerance-modes > gpurun_out/r6f_cfg2_story_len5.json 2>/dev/null; echo "cfg2 rc=$?"
timeout 1200 python bench.py --sink --story-len 25 --no-cpu-baseline --no-tolerance-modes --no-batch1 > gpurun_out/r6f_cfg4_sink_len25_bf16.json 2>/dev/null; echo "cfg4 rc=$?"
timeout 1200 python bench.py --sink --story-len 25 --unet-fp8 --no-cpu-baseline --no-tolerance-modes --no-batch1 > gpurun_out/r6f_cfg4_sink_len25_fp8.json 2>/dev/null; echo "cfg4 fp8 rc=$?"
python - <<'PY'
import json
for f in ('r6f_cfg1_mllm_only','r6f_cfg2_story_len5','r6f_cfg4_sink_len25_bf16','r6f_cfg4_sink_len25_fp8'):
    try:
        d=json.loads(open('gpurun_out/%s.json'%f).read().strip().splitlines()[-1]); r=d.get('roofline') or {}
        print(f, d['value'], d['ms_per_step'], 'fwd', r.get('forward_ms'), 'frac', r.get('frac'), 'batch1', (d.get('batch1') or {}).get('value'))
    except Exception as e: print(f, 'parse failed', e)
PY
