#!/bin/bash
# round 5, trip aa: config 2 as ONE persistent launch (refill = 1: a lane takes the next pixel when its pixel's 256 samples are done)
# at 6 / 5 / 4 / 3 / 2 resident blocks per CU (WOST_EXP_LDS_PAD), against the rounds
export TMPDIR=/tmp
O=gpurun_out/r05_aa; mkdir -p $O
B="python bench.py --no-extras --no-cpu-baseline --no-1spp --steps 3 --warmup 1"
P='import sys,json
for l in sys.stdin:
    if l.startswith("{"):
        d=json.loads(l); print(round(d["value"]/1e9,3),"e9", round(d["ms_per_step"],2),"ms", d["roofline"].get("launches"))'
echo "== rounds" | tee -a $O/persistent.txt; $B 2>/dev/null | python -c "$P" | tee -a $O/persistent.txt
for pad in 0 6000 14000 27000 54000; do
  echo "== rounds, LDS pad $pad" | tee -a $O/persistent.txt; WOST_EXP_LDS_PAD=$pad $B 2>/dev/null | python -c "$P" | tee -a $O/persistent.txt
  echo "== refill=1, LDS pad $pad" | tee -a $O/persistent.txt; WOST_EXP_LDS_PAD=$pad $B --opt refill=1 2>/dev/null | python -c "$P" | tee -a $O/persistent.txt
done
WOST_TRACE_LAUNCHES=1 WOST_EXP_LDS_PAD=14000 $B --opt refill=1 --steps 1 --warmup 0 2>&1 | grep "^launch" | head -8 | tee -a $O/persistent.txt
