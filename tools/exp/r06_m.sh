#!/bin/bash
# round 6, trip m: the first-tile deviation of net_forward_h_kernel (kernel WITHOUT the recomputed first tile), twelve full-size
# config-4 solves per setting (1280 Adam steps = 3840 forward launches each), the self check's counts summed per setting
mkdir -p gpurun_out/r06_m
export WOST_LIB=elaina_amd/lib/variants/noredo/libwost_hip.so
for pre in "" "burn1" "lds" "icache23" "lds23"; do
  echo "== WOST_NET_CHECK3_PRE='$pre'"
  for k in 1 2 3 4 5 6 7 8 9 10 11 12; do
    WOST_NET_CHECK3=1 WOST_NET_CHECK3_PRE="$pre" python tools/gpu_guided_bench.py --net-precision 16 --spp 256 --train-spp 256 2>&1 | grep "CHECK3 after\|distinct units" | sed "s/^/   run $k: /"
  done
done 2>&1 | tee gpurun_out/r06_m/check3_settings.txt
