#!/bin/bash
# separate counter passes (never combined with sys/hip traces) over tools/pmc_kernels.py; summary -> gpurun_out/pmc_kernels.txt
export TMPDIR=/tmp
rm -rf gpurun_out/kpmc; mkdir -p gpurun_out
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d gpurun_out/kpmc/p$i -o p -- python tools/pmc_kernels.py > gpurun_out/kpmc_p$i.log 2>&1
done
python tools/pmc_summary2.py $(find gpurun_out/kpmc -name "*counter_collection.csv") > gpurun_out/pmc_kernels.txt 2>&1
rm -rf gpurun_out/kpmc
tail -80 gpurun_out/pmc_kernels.txt
