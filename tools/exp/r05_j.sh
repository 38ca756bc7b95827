#!/bin/bash
# round 5, trip j: which training kernel deviates at full size (three launches per Adam step, compared on the device)
export TMPDIR=/tmp
O=gpurun_out/r05_j; mkdir -p $O
REPS=${REPS:-16} timeout 3000 python tools/probes/check3_cfg4.py > $O/check3.log 2>&1
grep -E "CHECK3|solve" $O/check3.log | tail -40
