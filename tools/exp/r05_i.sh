#!/bin/bash
# round 5, trip i: which side of the half-precision mode deviates -- full-size config 4 solve pairs with half-precision inference only,
# with half-precision training only
export TMPDIR=/tmp
O=gpurun_out/r05_i; mkdir -p $O
ONLY_F16=${PAIRS:-20} BOTH_F16=0 INF_F16=${PAIRS:-20} timeout 3000 python tools/probes/repro_cfg4.py > $O/pairs_sides.log 2>&1
echo "fp32 inference + f16 training: identical $(grep 'inference f32 training f16' $O/pairs_sides.log | grep -c 'field equal True.*weights equal True') of $(grep -c 'inference f32 training f16' $O/pairs_sides.log)" | tee -a $O/summary.txt
echo "f16 inference + fp32 training: identical $(grep 'inference f16 training f32' $O/pairs_sides.log | grep -c 'field equal True.*weights equal True') of $(grep -c 'inference f16 training f32' $O/pairs_sides.log)" | tee -a $O/summary.txt
grep "equal False" $O/pairs_sides.log | cut -c1-200 | tee -a $O/summary.txt
