#!/bin/bash
# (a record: ran at commit 45cb138, whose wost_net.hip still had the fused-loss kernel, its three-launch check and the -D variants used here)
# round 5, trip c: the half-precision run-to-run difference bisected in place -- the fused-loss forward launched three times on the
# same inputs (WOST_NET_FUSED_LOSS=3, tools/probes/repro_probe3.py), clean / bad triples per build variant
export TMPDIR=/tmp
O=gpurun_out/r05_c; mkdir -p $O
for v in h_base h_gridglobal h_wglobal h_bothglobal h_fence h_768; do rm -f $O/summary.txt.tmp;
  WOST_LIB=elaina_amd/lib/variants/$v/libwost_hip.so REPS=${REPS:-10} timeout 600 python tools/probes/repro_probe3.py > $O/$v.log 2>&1
  echo "$v: clean $(grep -c 'TRIPLE CLEAN' $O/$v.log) bad $(grep -c 'TRIPLE BAD' $O/$v.log)" | tee -a $O/summary.txt
  grep 'TRIPLE BAD' $O/$v.log | head -5 | tee -a $O/summary.txt
  grep 'raw outputs differ' $O/$v.log | head -6 | tee -a $O/summary.txt
done
# what the refined (SAH) leaf assignment of the 2-D builder is worth on configs 2 and 3: WOST_TREE_REFINE=0 = plain Morton groups
B="python bench.py --no-extras --no-cpu-baseline --steps 5 --warmup 2"
P='import sys,json
for l in sys.stdin:
    if l.startswith("{"):
        d=json.loads(l); print(round(d["value"]/1e9,3),"e9", round(d["ms_per_step"],2),"ms create", round(d.get("create_ms",0),1), d.get("scheduler",{}).get("visits_per_step"))'
for cfg in 2 3; do for r in 1 0; do echo "== config $cfg refine $r" | tee -a $O/refine.txt; WOST_TREE_REFINE=$r $B --config $cfg 2>/dev/null | python -c "$P" | tee -a $O/refine.txt; done; done
