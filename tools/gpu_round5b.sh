#!/bin/bash
# Round 5, the later GPU calls: the new kernels / modes of this round.   tools/gpu_round5b.sh <stage>
#   tests   new + touched tests (split-bf16 gate mode, v3p default, sink at full width, position-table bound, dtype policy)
#   lnfold  tools/dbg_lnfold_vec.py with the 4-wave folded tiles offered again + the rowpart producers
#   sink    bench --sink --story-len 25 (BASELINE configs[4] as named), bf16 and --unet-fp8, one whole story each, + the re-prefill comparator
#   gate    bench defaults without cpu baseline / batch1: tolerance_modes.gate_mode (full pipeline with the split-bf16 MLLM half)
#   cross   the (unadopted) short-context attention kernel: equality tests + kbench A/B + UNet forward A/B
#   split   split-bf16 GEMM tests + tools/split_bench.py (exact fp32 chain vs split, tile / order A/B)
#   traces  rocprofv3 --kernel-trace of the batch-16 UNet forward and of the VAE decode -> tools/trace_summary.py
#   final   smoke(), the default bench line (every leg), rocprofv3 --kernel-trace --stats of the bench command
# (counter passes: tools/pmc_round5.sh; the whole suite: tools/gpu_round5.sh tests)
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
export TMPDIR=/tmp
case "$1" in
tests)
  timeout 1200 python -m pytest -q -x tests/test_kernels_gpu.py tests/test_errors_gpu.py "tests/test_frontend_full_gpu.py" \
     "tests/test_fulldim_gpu.py::test_attention_sink_continuation_full_width" "tests/test_fulldim_gpu.py::test_sdxl_attention_shapes_bf16" \
     tests/test_engine_gpu.py -s --durations=10 > gpurun_out/r5b_tests.log 2>&1; echo "rc=$?" >> gpurun_out/r5b_tests.log
  grep -E "split-bf16|img_gen_feat|sink continuation|passed|failed|rc=|Error|error" gpurun_out/r5b_tests.log | tail -60;;
lnfold)
  timeout 300 python tools/dbg_lnfold_vec.py > gpurun_out/r5b_lnfold.log 2>&1; echo "rc=$?" >> gpurun_out/r5b_lnfold.log; cat gpurun_out/r5b_lnfold.log | tail -30;;
sink)
  for tag in "bf16" "fp8:--unet-fp8"; do
    name=${tag%%:*}; extra=""; [ "$tag" != "$name" ] && extra=${tag#*:}
    timeout 600 python bench.py --sink --story-len 25 --steps 25 --warmup 0 --no-cpu-baseline --no-batch1 --no-tolerance-modes $extra > gpurun_out/r5b_sink_$name.log 2>&1
    echo "rc=$?"; grep '^{"metric' gpurun_out/r5b_sink_$name.log | tail -1 > gpurun_out/r5b_sink_$name.json
    python -c "import json;d=json.load(open('gpurun_out/r5b_sink_$name.json'));print('$name', d['value'], d['ms_per_step'], d['config']['attention_sink'])" || tail -20 gpurun_out/r5b_sink_$name.log
  done
  timeout 600 python bench.py --story-len 25 --steps 25 --warmup 0 --no-cpu-baseline --no-batch1 --no-tolerance-modes --no-roofline > gpurun_out/r5b_len25_reprefill.log 2>&1
  grep '^{"metric' gpurun_out/r5b_len25_reprefill.log | tail -1 > gpurun_out/r5b_len25_reprefill.json
  python -c "import json;d=json.load(open('gpurun_out/r5b_len25_reprefill.json'));print('len25 re-prefill', d['value'], d['ms_per_step'])";;
gate)
  timeout 900 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-batch1 > gpurun_out/r5b_gate.log 2>&1; echo "rc=$?"
  grep '^{"metric' gpurun_out/r5b_gate.log | tail -1 > gpurun_out/r5b_gate.json
  python -c "import json;d=json.load(open('gpurun_out/r5b_gate.json'));print(d['value'], d['ms_per_step']);print(json.dumps(d['tolerance_modes'],indent=1)[:3000]);print(json.dumps(d['roofline'].get('gemm_8192cubed_control')))" || tail -30 gpurun_out/r5b_gate.log;;
cross)
  # the register-resident cross-attention kernel + the 4-deep GroupNorm loops: equality / oracle tests, then timings
  timeout 900 python -m pytest -q -x tests/test_kernels_gpu.py -k "cross_attention or v3p" "tests/test_fulldim_gpu.py::test_sdxl_attention_shapes_bf16" \
     tests/test_sdxl_gpu.py "tests/test_fulldim_gpu.py::test_sdxl_transformer_block_full_size" "tests/test_fulldim_gpu.py::test_sdxl_resblock_full_size" \
     "tests/test_fulldim_gpu.py::test_sdxl_unet_assembled_full_size_bf16" > gpurun_out/r5b_cross_tests.log 2>&1; echo "rc=$?" >> gpurun_out/r5b_cross_tests.log
  tail -6 gpurun_out/r5b_cross_tests.log
  timeout 200 python tools/kbench.py cross --batch 16 > gpurun_out/r5b_cross_kbench.log 2>&1; tail -3 gpurun_out/r5b_cross_kbench.log
  cp gpurun_out/cross_attn.json gpurun_out/r5b_cross_attn_b16.json 2>/dev/null
  for k in "attn_cross64=1" "attn_cross64=0"; do
    KB_KNOBS=$k timeout 300 python tools/kbench.py unet --batch 16 2>&1 | tail -1 | sed "s/^/$k /"
  done | tee gpurun_out/r5b_unet_ab.log;;
split)
  timeout 300 python -m pytest -q -x tests/test_kernels_gpu.py -k "f32_split" "tests/test_frontend_full_gpu.py::test_gate_mode_generate_hidden4096_img_gen_feat" -s 2>&1 | grep -E "split-bf16|img_gen_feat|passed|failed" | tail -16
  timeout 300 python tools/split_bench.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5b_split_bench.log;;
traces)
  # per-kernel traces of the batch-16 UNet forward and of one VAE decode (tools/trace_summary.py)
  R=$PWD; rm -rf /tmp/tr /tmp/trv
  (cd /tmp && export SS_UNET_BATCH=16 && timeout 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr -o t -- python $R/tools/unet_trace.py > $R/gpurun_out/r5b_unet_trace.log 2>&1)
  python tools/trace_summary.py $(find /tmp/tr -name "*kernel_trace.csv" | head -1) 3 > gpurun_out/r5b_unet_batch16_kernel_trace.txt 2>&1; head -14 gpurun_out/r5b_unet_batch16_kernel_trace.txt
  (cd /tmp && timeout 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/trv -o t -- python $R/tools/vae_trace.py > $R/gpurun_out/r5b_vae_trace.log 2>&1)
  python tools/trace_summary.py $(find /tmp/trv -name "*kernel_trace.csv" | head -1) 3 > gpurun_out/r5b_vae_decode_kernel_trace.txt 2>&1; head -40 gpurun_out/r5b_vae_decode_kernel_trace.txt;;
final)
  # the round's records: smoke(), the default bench line (every leg), rocprofv3 --kernel-trace --stats of the bench command
  timeout 300 python __graft_entry__.py smoke 2>&1 | tail -2
  timeout 900 python bench.py > gpurun_out/r5_bench.log 2>&1; echo "bench rc=$?"; grep '^{"metric' gpurun_out/r5_bench.log | tail -1 > gpurun_out/r5_bench.json
  python -c "import json;d=json.load(open('gpurun_out/r5_bench.json'));r=d['roofline'];print(d['value'],d['ms_per_step'],d['batch1'],r['forward_ms'],r['frac'],r['dominant_kernel']['avg_launch_us'],r['dominant_kernel']['frac'],r['mllm_decode_gemv']['avg_launch_us'],r['mllm_decode_gemv']['frac']);print(json.dumps(d['tolerance_modes'].get('gate_mode')));print(json.dumps(r.get('gemm_8192cubed_control')));print(d['cpu_baseline'])" || tail -30 gpurun_out/r5_bench.log
  rm -rf gpurun_out/r5; mkdir -p gpurun_out/r5
  B="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-batch1 --no-tolerance-modes --no-roofline"
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r5/stats_overlap -o b -- $B > gpurun_out/r5_overlap.log 2>&1; echo "prof rc=$?"
  find gpurun_out/r5 -name "*kernel_trace.csv" -delete
  f=$(find gpurun_out/r5 -name "*kernel_stats.csv" | head -1); head -12 "$f" | cut -c1-200; cp "$f" gpurun_out/r5_bench_kernel_stats.csv; grep '^{"metric' gpurun_out/r5_overlap.log | tail -1 | cut -c1-120;;
*) echo "unknown stage $1"; exit 2;;
esac
