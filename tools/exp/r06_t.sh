#!/bin/bash
mkdir -p gpurun_out/r06_t
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "persistent or config1 or round_length or block_size or thin or sharded" 2>&1 | tail -2
{
python tools/exp/r06_sweep.py ladybug "roomy=0" "roomy=-1" "roomy=0" "roomy=-1" "roomy=1,persist=0" "roomy=0,persist=0"
python tools/exp/r06_shards.py "roomy=0" "roomy=-1" "roomy=1"
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06_t/roomy.txt
