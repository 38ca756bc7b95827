#!/bin/bash
mkdir -p gpurun_out/r06_i
for n in 4 5 6; do
  for scene in ladybug fille; do
    WOST_LIB=elaina_amd/lib/variants/burst$n/libwost_hip.so python tools/exp/r06_sweep.py $scene "trav_burst=3" "trav_burst=$n" "trav_burst=$n,wait_weight=4" "trav_burst=$n,wait_weight=6" 2>&1 | grep -v amdgpu.ids | sed "s/^/unrolled $n: /" | tee -a gpurun_out/r06_i/burst.txt
  done
done
