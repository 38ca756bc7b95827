#!/bin/bash
mkdir -p gpurun_out/r06_d
for scene in ladybug fille; do
python tools/exp/r06_sweep.py $scene "" "resident_blocks=1280" "resident_blocks=1152" "resident_blocks=1024" "resident_blocks=1408" \
  "wait_weight=4" "wait_weight=6" "wait_weight=12" "wait_weight=16" "trav_burst=2" "trav_burst=4" "trav_burst=5" \
  "long_steps=1152" "long_thin=1024" "long_thin=4096" "long_steps=1152,long_thin=4096" "quad_fill=0.5" "quad_fill=2" \
  "resident_blocks=1280,long_steps=1152" "steps_per_round=192" "steps_per_round=384" 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06_d/sweep_$scene.txt
done
