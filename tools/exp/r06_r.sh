#!/bin/bash
mkdir -p gpurun_out/r06_r
python tools/exp/r06_1spp.py "" "trace" "few_leave=1" "few_leave=1,few_round=8" "few_leave=1,few_round=12" "few_leave=1,few_round=24" "few_leave=1,few_round=32" "few_leave=1,few_round=64" "few_leave=1,few_round=16,trace" "few_leave=1,few_round=16,quad_fill=2" "few_leave=1,few_round=16,quad_fill=4" 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06_r/one_spp.txt
