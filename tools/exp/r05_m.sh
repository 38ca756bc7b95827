#!/bin/bash
# round 5, trip m: the standalone chain with ANOTHER kernel in front of every launch (LDS garbage / register garbage / other code / a
# one-block kernel): does the first tile of a wave go wrong as in the production kernel?
export TMPDIR=/tmp
O=gpurun_out/r05_m; mkdir -p $O
for before in 4 5 6 1 2 3; do for mode in 0 7; do
  timeout 600 ./tools/micro/mfma_chain ${LAUNCHES:-3000} 524288 $mode $before 2>&1 | grep -v calibration | tee -a $O/mfma_chain_before.txt
done; done
