#!/bin/bash
mkdir -p gpurun_out/r06_p
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "persistent or refill or config1 or round_length or quad or strays or emissive or full_size_properties_config2" 2>&1 | tail -3
{
python tools/exp/r06_sweep.py ladybug "" "" "trav_burst=4" "trav_burst=6" "trav_burst=5,wait_weight=8" "trav_burst=5,wait_weight=4" "resident_blocks=1280" "trace"
python tools/exp/r06_sweep.py fille "" "" "trav_burst=4" "trav_burst=6" "trav_burst=5,wait_weight=4"
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06_p/persist_variant.txt
