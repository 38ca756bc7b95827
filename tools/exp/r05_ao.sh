#!/bin/bash
# round 5, trip ao: the half-precision mode's reproducibility at the final binary (wost_net.hip changed after trip w) -- 30 full-size config-4 pairs, the three-launch self
# check over 8 solves
export TMPDIR=/tmp
O=gpurun_out/r05_ao; mkdir -p $O
ONLY_F16=0 BOTH_F16=${PAIRS:-30} timeout 3000 python tools/probes/repro_cfg4.py > $O/pairs.log 2>&1
echo "both f16: identical pairs $(grep -c 'field equal True.*weights equal True' $O/pairs.log) of $(grep -c 'field equal' $O/pairs.log)" | tee -a $O/summary.txt
grep -o "steps [0-9]* / [0-9]*" $O/pairs.log | sort | uniq -c | tee -a $O/summary.txt
REPS=8 timeout 1500 python tools/probes/check3_cfg4.py > $O/check3.log 2>&1; grep -E "CHECK3" $O/check3.log | tail -2 | tee -a $O/summary.txt
INF=16 REPS=4 timeout 1500 python tools/probes/check3_cfg4.py > $O/check3_both.log 2>&1; grep -E "CHECK3" $O/check3_both.log | tail -2 | tee -a $O/summary.txt
