#!/bin/bash
mkdir -p gpurun_out/r06_e
python tools/exp/r06_shards.py "under_frac=0" "" "under_frac=0.0625" "under_frac=0.25" "under_frac=0.25,long_cap=65536" "under_frac=0.5,long_cap=131072" \
  "under_frac=0.125,long_thin=0" "under_frac=0.125,long_thin=8192" "under_frac=0.125,long_thin=16384" "under_frac=0.25,long_cap=65536,long_thin=8192" \
  "under_frac=0.125,long_priority=1" "under_frac=0.125,quad_fill=2" "under_frac=0.125,quad_fill=4" "under_frac=0,quad_fill=4" "under_frac=0.125,tail_sort=0" 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06_e/shards.txt
