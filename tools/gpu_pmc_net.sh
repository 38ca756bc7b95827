#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/pmcn
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_F32 SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp --output-format csv -d gpurun_out/pmcn/p$i -- python3 tools/gpu_net_probe.py pixel > gpurun_out/pmcn/p$i.log 2>&1
  f=$(find gpurun_out/pmcn/p$i -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && python3 tools/pmc_summary.py "$f" backward_wgrad
  rm -rf gpurun_out/pmcn/p$i
done
