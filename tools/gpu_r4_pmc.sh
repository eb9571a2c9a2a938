#!/bin/bash
# round 4: counter passes (each set in its own rocprofv3 --pmc --kernel-trace run; never combined with sys/hip traces) over the
# torch-free GEMM micro-benchmark.   tools/gpu_r4_pmc.sh <out name> <case> ...   -> gpurun_out/<out name>.txt
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
export TMPDIR=/tmp UBENCH_PMC=1
OUT=$1; shift
rm -rf gpurun_out/kpmc
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout 120 rocprofv3 --pmc $set --kernel-trace --output-format csv -d gpurun_out/kpmc/p$i -o p -- tools/bin/gemm_ubench seed-story_amd/lib/libseedstory_hip.so "$@" > gpurun_out/kpmc_p$i.log 2>&1
done
python3 tools/pmc_summary.py gemm_sp_kernel,gemm_w4_kernel $(find gpurun_out/kpmc -name "*counter_collection.csv") > gpurun_out/$OUT.txt 2>&1
rm -rf gpurun_out/kpmc
tail -150 gpurun_out/$OUT.txt
