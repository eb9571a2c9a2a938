#!/bin/bash
# separate counter passes (never combined with sys/hip traces) over tools/attn_pmc.py; summaries -> gpurun_out/attn_pmc.txt
export TMPDIR=/tmp
rm -rf gpurun_out/apmc; mkdir -p gpurun_out
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_VMEM" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d gpurun_out/apmc/$tag -o p -- python tools/attn_pmc.py > /dev/null 2>&1
done
python tools/pmc_summary.py $(find gpurun_out/apmc -name "*counter_collection.csv") > gpurun_out/attn_pmc.txt 2>&1
rm -rf gpurun_out/apmc
cat gpurun_out/attn_pmc.txt
