#!/bin/bash
# round 6, trip k: the half-precision first-tile deviation -- what runs in front of the three launches of the self check
# (the kernel WITHOUT the first-tile redo), config 4 at full size, 256 samples (1280 Adam steps = 3840 forward launches per run)
mkdir -p gpurun_out/r06_k
export WOST_LIB=elaina_amd/lib/variants/noredo/libwost_hip.so
run() { echo "== $1: WOST_NET_CHECK3_PRE='$2'"; WOST_NET_CHECK3=1 WOST_NET_CHECK3_PRE="$2" python tools/gpu_guided_bench.py --net-precision 16 --spp 256 --train-spp 256 2>&1 | grep "CHECK3"; }
{
run "baseline (no redo)" ""
run "baseline again" ""
run "NaNs in all LDS in front of every launch" "lds"
run "NaNs in all LDS in front of launches 2 and 3" "lds23"
run "a heavy matrix kernel in front of launch 1" "burn1"
run "instruction caches invalidated in front of launches 2 and 3" "icache23"
run "matrix kernel in front of 1 + icache in front of 2, 3" "burn1,icache23"
run "lds + icache in front of 2 and 3" "lds23,icache23"
echo "== variant 2 of the self check (a discarded launch of the same kernel in front)"; WOST_NET_CHECK3=2 python tools/gpu_guided_bench.py --net-precision 16 --spp 256 --train-spp 256 2>&1 | grep CHECK3
unset WOST_LIB
echo "== the shipped kernel (first tile recomputed)"; WOST_NET_CHECK3=1 python tools/gpu_guided_bench.py --net-precision 16 --spp 256 --train-spp 256 2>&1 | grep CHECK3
} 2>&1 | tee gpurun_out/r06_k/check3.txt
