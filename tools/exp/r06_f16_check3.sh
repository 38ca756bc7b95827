#!/bin/bash
# EXPERIMENTS 26: the first-tile deviation of net_forward_h_kernel with kernels in front of chosen launches of the three-launch self check.
# Needs the variant library without the recomputed first tile:  python tools/build_variant.py noredo "-DWOST_H_NO_REDO" wost_net.hip
# Usage (GPU box, repo root):  bash tools/exp/r06_f16_check3.sh [solves per setting, default 12]
N=${1:-12}
export WOST_LIB=elaina_amd/lib/variants/noredo/libwost_hip.so
for pre in "" "burn1" "lds" "icache23" "lds23"; do
  echo "== WOST_NET_CHECK3_PRE='$pre'"
  for k in $(seq 1 $N); do
    WOST_NET_CHECK3=1 WOST_NET_CHECK3_PRE="$pre" python tools/gpu_guided_bench.py --net-precision 16 --spp 256 --train-spp 256 2>&1 | grep "CHECK3 after\|distinct units" | sed "s/^/   run $k: /"
  done
done
