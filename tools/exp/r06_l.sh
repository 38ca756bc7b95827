#!/bin/bash
mkdir -p gpurun_out/r06_l
run() { echo "== $1"; WOST_NET_CHECK3=1 python tools/gpu_guided_bench.py --net-precision 16 --spp 256 --train-spp 256 2>&1 | grep "CHECK3"; }
{
export WOST_LIB=elaina_amd/lib/variants/noredo/libwost_hip.so
for k in 1 2 3 4 5 6; do run "no redo, run $k"; done
export WOST_LIB=elaina_amd/lib/variants/noredo_nosettle/libwost_hip.so
for k in 1 2 3; do run "no redo, no settle padding (the kernels of round 3), run $k"; done
} 2>&1 | tee gpurun_out/r06_l/check3_more.txt
