#!/bin/bash
# scheduler constants of the 3-D walk kernel (developer scratch)
for cfg in "16 2 4" "24 3 4" "32 3 4" "64 3 4" "32 2 4" "32 1 4"; do set -- $cfg
  echo "weight $1 burst $2 blocks/CU $3: $(WOST3_WAIT_WEIGHT=$1 WOST3_TRAV_BURST=$2 WOST3_BLOCKS_PER_CU=$3 python tools/scratch/bench3d.py 2>&1 | grep -o 'kernel [0-9.]* ms' | tr '\n' ' ')"
done
