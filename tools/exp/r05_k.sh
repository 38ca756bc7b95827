#!/bin/bash
# round 5, trip k: where in the launch the forward kernel's first launch differs, and whether a discarded launch in front moves it
export TMPDIR=/tmp
O=gpurun_out/r05_k; mkdir -p $O
REPS=3 timeout 1500 python tools/probes/check3_cfg4.py > $O/check3_1.log 2>&1; grep -E "CHECK3" $O/check3_1.log | tail -4
WOST_NET_CHECK3=2 REPS=3 timeout 1500 python tools/probes/check3_cfg4.py > $O/check3_2.log 2>&1; grep -E "CHECK3" $O/check3_2.log | tail -4
