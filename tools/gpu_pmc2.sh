#!/bin/bash
# memory-pipeline counters for the walk kernel (TA/TCP/TD), one group per run
export TMPDIR=/tmp
mkdir -p gpurun_out/pmc2
ARGS="bench.py --steps 1 --warmup 0 --no-cpu-baseline --spp ${SPP:-64} ${EXTRA}"
i=0
for grp in \
 "TA_TA_BUSY_sum TA_TOTAL_WAVEFRONTS_sum" \
 "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
 "TCP_GATE_EN1_sum TCP_GATE_EN2_sum" \
 "TCP_PENDING_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum" \
 "TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum" \
 "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum" \
 "TCP_TOTAL_ACCESSES_sum TCP_TA_TCP_STATE_READ_sum" \
 "TD_TD_BUSY_sum TD_TC_STALL_sum" \
 "GRBM_GUI_ACTIVE" \
 "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU" ; do
  i=$((i+1))
  timeout 120 rocprofv3 --pmc $grp --output-format csv -d gpurun_out/pmc2/p$i -- python3 $ARGS > gpurun_out/pmc2/p$i.log 2>&1
  f=$(find gpurun_out/pmc2/p$i -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && python3 tools/pmc_summary.py "$f" | grep -v init_kernel
done
