#!/bin/bash
# PMC passes for the guided kernels (each counter group in its own run; no trace domains mixed in)
export TMPDIR=/tmp
mkdir -p gpurun_out/pmcg
ARGS="tools/gpu_guided_bench.py --spp ${SPP:-4} --train-spp ${TSPP:-2}"
i=0
for grp in \
 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
 "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS" \
 "GRBM_GUI_ACTIVE" ; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp --output-format csv -d gpurun_out/pmcg/p$i -- python3 $ARGS > gpurun_out/pmcg/p$i.log 2>&1
  f=$(find gpurun_out/pmcg/p$i -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && python3 tools/pmc_summary.py "$f" "$@" | tee gpurun_out/pmcg/p$i.summary.txt
  rm -rf gpurun_out/pmcg/p$i
done
