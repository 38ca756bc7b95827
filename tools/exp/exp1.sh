#!/bin/bash
# round-3 experiment 1: what binds walk_round_kernel?  baseline, sensitivity variants, occupancy, TA/TCP counters
export TMPDIR=/tmp
O=gpurun_out/r03_exp1
mkdir -p $O
B="bench.py --steps 2 --warmup 1 --no-extras --no-cpu-baseline --no-1spp"
val() { python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        r = json.loads(l); print('$1', round(r['value']/1e9,3), 'e9 steps/s', r['ms_per_step'], 'ms', r.get('time_to_1spp_ms'), json.dumps(r.get('scheduler')))"; }
echo "== baseline" | tee $O/log.txt
python bench.py --no-extras --no-cpu-baseline 2>$O/base.err | val base | tee -a $O/log.txt
python $B 2>>$O/base.err | val base2 | tee -a $O/log.txt
echo "== variants" | tee -a $O/log.txt
for v in valu30 valu60 loads track; do
  WOST_LIB=$PWD/elaina_amd/lib/variants/$v.so python $B 2>$O/$v.err | val $v | tee -a $O/log.txt
  grep WOST_TRACK $O/$v.err | tail -2 | tee -a $O/log.txt
done
echo "== occupancy (LDS pad)" | tee -a $O/log.txt
for pad in 2000 8000 16000; do
  WOST_EXP_LDS_PAD=$pad python $B 2>/dev/null | val pad$pad | tee -a $O/log.txt
done
echo "== shard probe" | tee -a $O/log.txt
python tools/gpu_shard_probe.py 2>/dev/null | tee -a $O/log.txt
echo "== PMC TA/TCP" | tee -a $O/log.txt
BA="bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-1spp --no-extras"
i=0
for grp in "TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum" "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
  "TCP_GATE_EN1_sum TCP_GATE_EN2_sum" "TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum" \
  "TCP_TA_TCP_STATE_READ_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum" "TD_TD_BUSY_sum TD_TC_STALL_sum" \
  "TCP_TOTAL_ACCESSES_sum TCP_TOTAL_READ_sum TCP_TOTAL_CACHE_ACCESSES_sum" \
  "SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
  "SQ_INST_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC" \
  "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 240 rocprofv3 --pmc $grp --output-format csv -d $O/pmc$i -- python3 $BA > $O/pmc$i.log 2>&1
  f=$(find $O/pmc$i -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && python3 tools/pmc_summary.py "$f" walk_round | grep -v 'calls=7 ' | tee -a $O/log.txt || { echo "pass $i ($grp) failed"; tail -3 $O/pmc$i.log; } | tee -a $O/log.txt
  rm -rf $O/pmc$i
done
echo "== gather microbench with counters" | tee -a $O/log.txt
./tools/micro/gather_rate 2>&1 | head -8 | tee -a $O/log.txt
for grp in "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TOTAL_ACCESSES_sum" "TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum" "GRBM_GUI_ACTIVE"; do
  timeout 120 rocprofv3 --pmc $grp --output-format csv -d $O/mg -- ./tools/micro/gather_rate > $O/mg.log 2>&1
  f=$(find $O/mg -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && python3 - "$f" <<'PY' | tee -a $O/log.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:60]:
    print(r["Kernel_Name"][:40], r["Counter_Name"], r["Counter_Value"], r.get("Grid_Size"), r.get("Dispatch_Id"))
PY
  rm -rf $O/mg
done
