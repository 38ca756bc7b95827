#!/bin/bash
# PMC passes for the walk kernel (each counter group in its own run; no trace domains mixed in)
export TMPDIR=/tmp
mkdir -p gpurun_out/pmc
ARGS="bench.py --steps 1 --warmup 0 --no-cpu-baseline --spp ${SPP:-32} --steps-per-round ${SPR:-256}"
timeout 60 rocprofv3 -L > gpurun_out/pmc/counters.txt 2>&1
i=0
for grp in \
 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
 "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM" \
 "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum" \
 "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE" ; do
  i=$((i+1))
  timeout 180 rocprofv3 --pmc $grp --output-format csv -d gpurun_out/pmc/p$i -- python3 $ARGS > gpurun_out/pmc/p$i.log 2>&1
  f=$(find gpurun_out/pmc/p$i -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && python3 tools/pmc_summary.py "$f" | tee gpurun_out/pmc/p$i.summary.txt
done
