#!/bin/bash
# round 5, trip x: the fused guided kernel at 14 / 16 waves per CU in half precision (128 VGPRs and scratch) against the 12 it has
export TMPDIR=/tmp
O=gpurun_out/r05_x; mkdir -p $O
P='import sys,json
for l in sys.stdin:
    if l.startswith("{"):
        d=json.loads(l); print(round(d["value"]/1e9,4),"e9", round(d["ms_per_step"],1),"ms")'
for v in default fused896 fused1024; do
  echo "== $v" | tee -a $O/fused_threads.txt
  if [ $v = default ]; then python bench.py --config 4 --net-precision 16 --steps 2 --warmup 1 2>/dev/null | python -c "$P" | tee -a $O/fused_threads.txt
  else WOST_LIB=elaina_amd/lib/variants/$v/libwost_hip.so python bench.py --config 4 --net-precision 16 --steps 2 --warmup 1 2>/dev/null | python -c "$P" | tee -a $O/fused_threads.txt; fi
done
