#!/bin/bash
# round 5, trip n: the standalone chain on full-mantissa operands, back to back and behind a one-block kernel / an idle gap
export TMPDIR=/tmp
O=gpurun_out/r05_n; mkdir -p $O
for before in 0 4 6; do for mode in 0 7 1; do
  timeout 600 ./tools/micro/mfma_chain ${LAUNCHES:-3000} 524288 $mode $before 1 2>&1 | tee -a $O/mfma_chain_random.txt
done; done
