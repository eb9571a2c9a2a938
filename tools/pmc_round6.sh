#!/bin/bash
# Round 6 counter passes over the torch-free GEMM / conv micro-benchmark ON THE SHIPPED TILES (tools/gemm_ubench in UBENCH_PMC mode:
# 3 launches per case over rotating weights).  Each counter set in its own `rocprofv3 --pmc ... --kernel-trace` pass (never with
# sys / hip traces).  Cases = name:ubench-argument; the cfg/swz of each case is read from the shipped table by tools/summarize_pmc6.py --cases.
cd "$(dirname "$0")/.." && mkdir -p gpurun_out && export TMPDIR=/tmp
R=$PWD
rm -rf gpurun_out/r6pmc; mkdir -p gpurun_out/r6pmc
L=$R/seed-story_amd/lib/libseedstory_hip.so
U=$R/tools/bin/gemm_ubench
python tools/summarize_pmc6.py --cases > gpurun_out/r6pmc/cases.txt
cat gpurun_out/r6pmc/cases.txt
i=0
for set in "SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" "FETCH_SIZE" "WRITE_SIZE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  while read -r name arg; do
    (cd /tmp && UBENCH_PMC=1 timeout 120 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/r6pmc/${name}_$i -o p -- $U $L $arg > $R/gpurun_out/r6pmc/${name}_$i.log 2>&1)
  done < gpurun_out/r6pmc/cases.txt
done
python tools/summarize_pmc6.py gpurun_out/r6pmc
find gpurun_out/r6pmc -name "*.csv" -size +2M -delete
