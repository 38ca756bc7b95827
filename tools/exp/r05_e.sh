#!/bin/bash
# (a record: ran at commit 45cb138, whose wost_net.hip still had the fused-loss kernel, its three-launch check and the -D variants used here)
# round 5, trip e: the half-precision difference at three waves per SIMD (blocks of 768: every triple differs) bisected in place,
# and the standalone chain at 768 threads / with LDS gathers beside it
export TMPDIR=/tmp
O=gpurun_out/r05_e; mkdir -p $O
for v in h768_base h768_settle h768_wglobal h768_gridglobal h768_nogather h768_nogather_wglobal h768_fence h512_base; do
  WOST_LIB=elaina_amd/lib/variants/$v/libwost_hip.so REPS=${REPS:-6} timeout 600 python tools/probes/repro_probe3.py > $O/$v.log 2>&1
  echo "$v: clean $(grep -c 'TRIPLE CLEAN' $O/$v.log) bad $(grep -c 'TRIPLE BAD' $O/$v.log)" | tee -a $O/summary.txt
  grep 'TRIPLE BAD' $O/$v.log | head -3 | tee -a $O/summary.txt
done
timeout 900 ./tools/micro/mfma_chain 10000 524288 0 2>&1 | tee $O/mfma_chain.txt
timeout 900 ./tools/micro/mfma_chain 10000 524288 7 2>&1 | tee -a $O/mfma_chain.txt
timeout 900 ./tools/micro/mfma_chain 10000 524288 6 2>&1 | tee -a $O/mfma_chain.txt
