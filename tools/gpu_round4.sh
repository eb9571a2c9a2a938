#!/bin/bash
# Round 4, everything that runs on the GPU box, by stage:   tools/gpu_round4.sh <stage> [args]
#   ubench <case>...   torch-free GEMM / conv micro-benchmark (tools/gemm_ubench.cpp) -> gpurun_out/ubench.log
#   tests              the whole `pytest -m gpu` suite with durations      -> gpurun_out/r4_pytest.log
#   bench              `python bench.py` (N = 1, defaults) + the driver-like command -> gpurun_out/r4_bench*.json
#   prof               rocprofv3 --kernel-trace --stats of the bench command (shipped schedule and --no-overlap) and the per-kernel
#                      trace of one UNet forward                           -> gpurun_out/r4/..., summarised by summarize_round4.py
#   pmc_gemv / prof_bench   the decode-GEMV traffic passes only / the two bench profiles only (no UNet trace)
#   final              see the stage
#   pmc                counter passes (each set in its own --pmc --kernel-trace run, never with sys/hip traces): the dominant MFMA
#                      kernels on the SHIPPED tile table through gemm_ubench, attention through tools/attn_pmc.py, decode GEMV
#                      FETCH_SIZE / WRITE_SIZE through bench.py --mllm-only -> gpurun_out/r4/..., summarised likewise
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
export TMPDIR=/tmp
LIB=seed-story_amd/lib/libseedstory_hip.so
stage=$1; shift
case "$stage" in
ubench)
  timeout 600 tools/bin/gemm_ubench $LIB "$@" > gpurun_out/ubench.log 2>&1; echo "rc=$?" >> gpurun_out/ubench.log; tail -120 gpurun_out/ubench.log;;
tests)
  timeout 1500 python -m pytest tests -q -m gpu --durations=25 > gpurun_out/r4_pytest.log 2>&1; echo "rc=$?" >> gpurun_out/r4_pytest.log; tail -45 gpurun_out/r4_pytest.log;;
bench)
  timeout 900 python bench.py > gpurun_out/r4_bench.log 2>&1; tail -1 gpurun_out/r4_bench.log > gpurun_out/r4_bench.json; cut -c1-600 gpurun_out/r4_bench.json
  timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-tolerance-modes > gpurun_out/r4_bench_driver_like.log 2>&1; tail -1 gpurun_out/r4_bench_driver_like.log > gpurun_out/r4_bench_driver_like.json; cut -c1-300 gpurun_out/r4_bench_driver_like.json;;
prof)
  rm -rf gpurun_out/r4/stats_overlap gpurun_out/r4/stats_serial; mkdir -p gpurun_out/r4
  B="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-batch1 --no-tolerance-modes"
  timeout 700 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r4/stats_overlap -o b -- $B > gpurun_out/r4_overlap.log 2>&1
  timeout 700 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r4/stats_serial -o b -- $B --no-overlap > gpurun_out/r4_serial.log 2>&1
  find gpurun_out/r4 -name "*kernel_trace.csv" -delete
  R=$PWD; rm -rf /tmp/tr; (cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr -o t -- python $R/tools/unet_trace.py > $R/gpurun_out/r4_unet_trace.log 2>&1)
  python tools/trace_summary.py $(find /tmp/tr -name "*kernel_trace.csv" | head -1) 3 > gpurun_out/r4_unet_batch8_kernel_trace.txt 2>&1; head -32 gpurun_out/r4_unet_batch8_kernel_trace.txt
  python tools/summarize_round4.py stats;;
pmc)
  rm -rf gpurun_out/r4/k_* gpurun_out/r4/a_* gpurun_out/r4/fetch gpurun_out/r4/write; mkdir -p gpurun_out/r4
  CASES="16384,10240,1280,16:69/0 8192,10240,1280,16:60/8 8192,3840,1280,0:60/8 8192,1280,1280,0,1:62/8 8192,1280,5120,0,1:62/4 c8,32,32,1280,1280,1,0,1:61/8 8192,8192,8192,0:60/8,91/8"
  i=0
  for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" "SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VMEM SQ_INSTS_MFMA" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
    i=$((i+1))
    UBENCH_PMC=1 timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d gpurun_out/r4/k_$i -o p -- tools/bin/gemm_ubench $LIB $CASES > gpurun_out/r4_kpmc_$i.log 2>&1
    timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d gpurun_out/r4/a_$i -o p -- python tools/attn_pmc.py > gpurun_out/r4_apmc_$i.log 2>&1
  done
  B1="python bench.py --mllm-only --steps 1 --warmup 0 --no-cpu-baseline --no-batch1 --no-tolerance-modes"
  timeout 500 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/r4/fetch -o p -- $B1 > gpurun_out/r4_fetch.log 2>&1
  timeout 500 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/r4/write -o p -- $B1 > gpurun_out/r4_write.log 2>&1
  echo "$CASES" > gpurun_out/r4/cases.txt
  python tools/summarize_round4.py pmc
  find gpurun_out/r4 -name "*.csv" -size +8M -delete;;
pmc_gemv)
  # only the decode-GEMV traffic passes (the GEMM / attention counter sets above are tied to the tile table, which did not change)
  rm -rf gpurun_out/r4/fetch gpurun_out/r4/write; mkdir -p gpurun_out/r4
  B1="python bench.py --mllm-only --steps 1 --warmup 0 --no-cpu-baseline --no-batch1 --no-tolerance-modes"
  timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/r4/fetch -o p -- $B1 > gpurun_out/r4_fetch.log 2>&1
  timeout 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/r4/write -o p -- $B1 > gpurun_out/r4_write.log 2>&1
  find gpurun_out/r4 -name "*.csv" -size +8M -delete; tail -1 gpurun_out/r4_fetch.log | cut -c1-200;;
prof_bench)
  rm -rf gpurun_out/r4/stats_overlap gpurun_out/r4/stats_serial; mkdir -p gpurun_out/r4
  B="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-batch1 --no-tolerance-modes --no-roofline"
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r4/stats_overlap -o b -- $B > gpurun_out/r4_overlap.log 2>&1
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r4/stats_serial -o b -- $B --no-overlap > gpurun_out/r4_serial.log 2>&1
  find gpurun_out/r4 -name "*kernel_trace.csv" -delete
  python tools/summarize_round4.py stats | cut -c1-600;;
final)
  # the last call of round 4 (4.7 GPU-minutes left): (A) kernel-trace --stats of the bench command at the SHIPPED default (8 stories per
  # GPU, MFMA-form decode GEMV) + the tile-table entries that run tuned in-process; (B) counters of the dominant kernel of that
  # default (ff1 GEGLU at UNet batch 16) through the torch-free micro-benchmark; (C) FETCH_SIZE / WRITE_SIZE of the decode token's
  # GEMV launch mix at 8 slots per sweep (tools/gemv_pmc.py).  Every step under a timeout cut to what is left of BUDGET seconds;
  # summarised by summarize_round4.py final (runs on the raw CSVs here as well, should the call end before it)
  BUDGET=${BUDGET:-245}
  step() {  # step <name> <max seconds> <command...>
    local name=$1 max=$2; shift 2
    local left=$((BUDGET - SECONDS)); [ $left -lt $max ] && max=$left
    if [ $max -lt 6 ]; then echo "$name skipped (budget)"; return; fi
    timeout $max "$@"; echo "$name rc=$? t=${SECONDS}s"
  }
  rm -rf gpurun_out/r4; mkdir -p gpurun_out/r4
  CASES="16384,10240,1280,16:69/0"; echo "$CASES" > gpurun_out/r4/cases.txt
  step A 165 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r4/stats_overlap -o b -- python bench.py --steps 1 --warmup 1 \
      --no-cpu-baseline --no-batch1 --no-tolerance-modes --no-roofline --save-tune-table gpurun_out/r4/tune_after_bench.json > gpurun_out/r4_overlap.log 2>&1
  find gpurun_out/r4 -name "*kernel_trace.csv" -delete
  export UBENCH_PMC=1
  i=0
  for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" "SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VMEM SQ_INSTS_MFMA" "TCC_HIT_sum TCC_MISS_sum"; do
    i=$((i+1))
    step B$i 20 rocprofv3 --pmc $set --kernel-trace --output-format csv -d gpurun_out/r4/k_$i -o p -- tools/bin/gemm_ubench $LIB $CASES > gpurun_out/r4_kpmc_$i.log 2>&1
    if [ $i = 2 ]; then   # the GEMV traffic passes go before the optional GEMM counter sets
      step C1 40 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/r4/fetch -o p -- python tools/gemv_pmc.py > gpurun_out/r4_fetch.log 2>&1
      step C2 40 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/r4/write -o p -- python tools/gemv_pmc.py > gpurun_out/r4_write.log 2>&1
    fi
  done
  python tools/summarize_round4.py final 2>&1 | cut -c1-1500
  find gpurun_out/r4 -name "*.csv" -size +8M -delete;;
final2)
  # after `final`: the 17 shapes that call tuned in-process are in the shipped table now.  (A) kernel-trace --stats of the bench command,
  # (T) per-kernel trace of the batch-16 UNet forward, (J) a bench line of the final code (traffic fields from the committed PMC
  # record), (S) the --no-overlap stats — in that order, under the same budget guard
  BUDGET=${BUDGET:-195}
  step() { local name=$1 max=$2; shift 2; local left=$((BUDGET - SECONDS)); [ $left -lt $max ] && max=$left
           if [ $max -lt 6 ]; then echo "$name skipped (budget)"; return; fi; timeout $max "$@"; echo "$name rc=$? t=${SECONDS}s"; }
  rm -rf gpurun_out/r4/stats_overlap gpurun_out/r4/stats_serial /tmp/tr; mkdir -p gpurun_out/r4
  B="python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-batch1 --no-tolerance-modes --no-roofline"
  step A 70 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r4/stats_overlap -o b -- $B > gpurun_out/r4_overlap.log 2>&1
  find gpurun_out/r4 -name "*kernel_trace.csv" -delete
  R=$PWD
  (cd /tmp && export SS_UNET_BATCH=16 && step T 60 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr -o t -- python $R/tools/unet_trace.py > $R/gpurun_out/r4_unet_trace.log 2>&1)
  python tools/trace_summary.py $(find /tmp/tr -name "*kernel_trace.csv" | head -1) 3 > gpurun_out/r4_unet_batch16_kernel_trace.txt 2>&1; head -12 gpurun_out/r4_unet_batch16_kernel_trace.txt
  step J 100 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-batch1 --no-tolerance-modes > gpurun_out/r4_bench_final_code.log 2>&1
  tail -1 gpurun_out/r4_bench_final_code.log > gpurun_out/r4_bench_final_code.json; cut -c1-200 gpurun_out/r4_bench_final_code.json
  step S 70 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r4/stats_serial -o b -- $B --no-overlap > gpurun_out/r4_serial.log 2>&1
  find gpurun_out/r4 -name "*kernel_trace.csv" -delete
  python tools/summarize_round4.py final 2>&1 | cut -c1-600;;
*) echo "unknown stage $stage"; exit 2;;
esac
