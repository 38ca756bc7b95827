#!/bin/bash
# round 5, trip c: the half-precision run-to-run difference bisected in place -- the fused-loss forward launched three times on the
# same inputs (WOST_NET_FUSED_LOSS=3, tools/probes/repro_probe3.py), clean / bad triples per build variant
export TMPDIR=/tmp
O=gpurun_out/r05_c; mkdir -p $O
for v in h_base h_gridglobal h_wglobal h_bothglobal h_fence h_768; do
  WOST_LIB=elaina_amd/lib/variants/$v/libwost_hip.so REPS=${REPS:-10} timeout 600 python tools/probes/repro_probe3.py > $O/$v.log 2>&1
  echo "$v: clean $(grep -c 'TRIPLE CLEAN' $O/$v.log) bad $(grep -c 'TRIPLE BAD' $O/$v.log)" | tee -a $O/summary.txt
  grep 'TRIPLE BAD' $O/$v.log | head -5 | tee -a $O/summary.txt
  grep 'raw outputs differ' $O/$v.log | head -6 | tee -a $O/summary.txt
done
