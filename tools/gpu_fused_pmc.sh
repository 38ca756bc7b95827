#!/bin/bash
# Developer tool (GPU box): PMC counters of the fused guided sample kernel (guiding phase, 16 samples in one launch).
# Usage: bash tools/gpu_fused_pmc.sh [tag] [extra gpu_guided_bench.py args]  -> gpurun_out/<tag>_fused_pmc.txt
TAG=${1:-dev}; shift
export TMPDIR=/tmp
out=gpurun_out/${TAG}_fused_pmc.txt
rm -f $out
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_MISC SQ_INSTS_FLAT" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rm -rf gpurun_out/fp_$TAG
  timeout 300 rocprofv3 --pmc $grp --output-format csv -d gpurun_out/fp_$TAG -- python3 tools/gpu_guided_bench.py --spp 16 --train-spp 0 --net-precision 16 "$@" > gpurun_out/${TAG}_fused_pmc$i.log 2>&1
  f=$(find gpurun_out/fp_$TAG -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && python3 tools/pmc_summary.py "$f" guided_sample | sed 's/.*FusedNet)[ ]*//; s/^void wost::guided_sample_kernel[^ ]* *//' | tee -a $out
done
grep -o '"walk_steps":[^,]*' gpurun_out/${TAG}_fused_pmc1.log | tee -a $out
rm -rf gpurun_out/fp_$TAG
