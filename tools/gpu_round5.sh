#!/bin/bash
# Round 5, first GPU calls (prepared at the end of round 4, when the GPU budget was spent):   tools/gpu_round5.sh <stage>
#   exp      the unadopted option kernels: equality-with-v3 tests of flash v3p (attn_ver 5 / 6) + the attention A/B timing
#            (tools/kbench.py attn: v2 / v3 / v3+XCD / v3p-6 / v3p-5 on the UNet's self-attention shapes at batch 8 and 16)
#   ab       bench A/B on one box, short runs: shipped default | attn_ver=6 | attn_ver=5 | 16 stories per GPU
#   tests    the whole `pytest -m gpu` suite with durations (the cached host-oracle truths should read "loaded")
#   bench    `python bench.py` (defaults) -> gpurun_out/r5_bench.json
# Everything else (rocprofv3 stats, counter passes, summaries): tools/gpu_round4.sh stages, unchanged.
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
export TMPDIR=/tmp
S="--steps 3 --warmup 1 --no-cpu-baseline --no-batch1 --no-tolerance-modes"
line() { grep '^{"metric' "$1" | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d.get('roofline') or {}; print(d['value'], d['ms_per_step'], r.get('forward_ms'), d['config'].get('knobs'), d['config'].get('stories_per_gpu'))"; }
case "$1" in
exp)
  SS_TEST_EXPERIMENTAL=1 timeout 600 python -m pytest tests/test_experimental_gpu.py -q -x > gpurun_out/r5_exp_tests.log 2>&1; echo "rc=$?" >> gpurun_out/r5_exp_tests.log; tail -5 gpurun_out/r5_exp_tests.log
  timeout 300 python tools/kbench.py attn > gpurun_out/r5_attn_ab.log 2>&1; tail -12 gpurun_out/r5_attn_ab.log;;
ab)
  for tag in "default" "attn6:--knob attn_ver=6" "attn5:--knob attn_ver=5" "spg16:--stories-per-gpu 16"; do
    name=${tag%%:*}; extra=""; [ "$tag" != "$name" ] && extra=${tag#*:}
    timeout 400 python bench.py $S $extra > gpurun_out/r5_ab_$name.log 2>&1; echo "$name: $(line gpurun_out/r5_ab_$name.log)"
  done;;
tests)
  timeout 1500 python -m pytest tests -q -m gpu --durations=25 > gpurun_out/r5_pytest.log 2>&1; echo "rc=$?" >> gpurun_out/r5_pytest.log; tail -45 gpurun_out/r5_pytest.log;;
bench)
  timeout 900 python bench.py > gpurun_out/r5_bench.log 2>&1; grep '^{"metric' gpurun_out/r5_bench.log | tail -1 > gpurun_out/r5_bench.json; cut -c1-600 gpurun_out/r5_bench.json;;
*) echo "unknown stage $1"; exit 2;;
esac
