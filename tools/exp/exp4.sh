#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r03_exp4
mkdir -p $O
./tools/micro/half_wave | tee $O/half_wave.txt
val() { python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        r = json.loads(l); print('$1', round(r['value']/1e9,3), 'e9 steps/s', round(r['ms_per_step'],2), 'ms')"; }
B="bench.py --steps 3 --warmup 1 --no-extras --no-cpu-baseline --no-1spp"
for v in base lb5 lb4 base lb5; do
  if [ $v = base ]; then python $B 2>/dev/null | val base | tee -a $O/log.txt; else WOST_LIB=$PWD/elaina_amd/lib/variants/$v.so python $B 2>/dev/null | val $v | tee -a $O/log.txt; fi
done
for bw in "2 8" "4 8" "3 6" "3 12"; do set -- $bw
  WOST_LIB=$PWD/elaina_amd/lib/variants/lb5.so python $B --opt trav_burst=$1 --opt wait_weight=$2 2>/dev/null | val "lb5 burst $1 weight $2" | tee -a $O/log.txt
done
