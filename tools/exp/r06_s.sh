#!/bin/bash
mkdir -p gpurun_out/r06_s
python tools/exp/r06_sweep.py fille "" "trace" "long_steps=768" "long_steps=1280" "long_steps=1536" "long_steps=2048" "long_thin=512" "long_thin=4096" "long_thin=8192" "long_steps=1536,long_thin=4096" "long_steps=2048,long_thin=8192" "long_steps=0" "tail_sort=0" "steps_per_round=512" 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06_s/fille_tail.txt
