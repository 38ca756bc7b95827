#!/bin/bash
mkdir -p gpurun_out/r06_v
python tools/exp/r06_shards.py "" "wait_weight=16" "wait_weight=64" "wait_weight=512" "trav_burst=1" "trav_burst=2" "trav_burst=5" "wait_weight=64,trav_burst=1" "wait_weight=64,trav_burst=2" "wait_weight=16,trav_burst=2" "wait_weight=4,trav_burst=5" "block_size=64" "block_size=128" 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06_v/shard_scheduler.txt
