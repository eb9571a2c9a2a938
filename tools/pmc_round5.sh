#!/bin/bash
# Round 5 counter passes (each set alone with --kernel-trace; never with sys / hip traces): flash v3p, the split-bf16 gate-mode GEMM and GEMV
cd "$(dirname "$0")/.." && mkdir -p gpurun_out && export TMPDIR=/tmp
rm -rf gpurun_out/r5pmc; mkdir -p gpurun_out/r5pmc
i=0
for set in "SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  for wl in attn split gemv; do
    timeout 150 rocprofv3 --pmc $set --kernel-trace --output-format csv -d gpurun_out/r5pmc/${wl}_$i -o p -- python tools/pmc5_workloads.py $wl > gpurun_out/r5pmc_${wl}_$i.log 2>&1
  done
done
python tools/summarize_pmc5.py gpurun_out/r5pmc | head -120
find gpurun_out/r5pmc -name "*.csv" -size +4M -delete
