#!/bin/bash
mkdir -p gpurun_out/r06_h
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "persistent or quad" > gpurun_out/r06_h/pytest.log 2>&1; tail -3 gpurun_out/r06_h/pytest.log
python bench.py --no-extras --no-cpu-baseline 2>gpurun_out/r06_h/bench.err | tee gpurun_out/r06_h/bench.json | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(j['value'], j['ms_per_step'], j['time_to_1spp_ms']); print(j['roofline']['pass_launches'])"
python tools/exp/r06_sweep.py ladybug "" "" "long_steps=1280" "long_thin=1024" "long_thin=4096" "long_steps=896" "trav_burst=5" "wait_weight=4,trav_burst=5" "wait_weight=6,trav_burst=4" 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06_h/sweep.txt
python tools/exp/r06_sweep.py fille "" "" "long_steps=1280" "trav_burst=5" "wait_weight=4,trav_burst=5" "wait_weight=4,trav_burst=4" "wait_weight=4" 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06_h/sweep_fille.txt
