#!/bin/bash
# Developer tool (GPU box): PMC counters of the half-precision network kernels over a short guided solve.
# Usage: bash tools/gpu_net_pmc.sh [tag]   -> gpurun_out/<tag>_net_pmc.txt
TAG=${1:-dev}
export TMPDIR=/tmp
out=gpurun_out/${TAG}_net_pmc.txt
rm -f $out
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rm -rf gpurun_out/np_$TAG
  timeout 300 rocprofv3 --pmc $grp --output-format csv -d gpurun_out/np_$TAG -- python3 tools/gpu_guided_bench.py --spp 8 --train-spp 8 --net-precision 16 > gpurun_out/${TAG}_net_pmc$i.log 2>&1
  f=$(find gpurun_out/np_$TAG -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && python3 tools/pmc_summary.py "$f" net_forward_h net_train_h grid_grad | tee -a $out
done
rm -rf gpurun_out/np_$TAG
