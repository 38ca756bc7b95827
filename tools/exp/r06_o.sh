#!/bin/bash
mkdir -p gpurun_out/r06_o
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "persistent or refill or config1 or round_length or quad or random_scenes or strays or source or emissive or ragged" 2>&1 | tail -3
{
python tools/exp/r06_sweep.py ladybug "" "" "resident_blocks=1280" "persist=0"
WOST_LIB=elaina_amd/lib/variants/waves5/libwost_hip.so python tools/exp/r06_sweep.py ladybug "resident_blocks=1280" "resident_blocks=1280" "persist=0" | sed 's/^/5 waves, 96 VGPRs: /'
python tools/exp/r06_sweep.py fille "" "resident_blocks=1280"
WOST_LIB=elaina_amd/lib/variants/waves5/libwost_hip.so python tools/exp/r06_sweep.py fille "resident_blocks=1280" | sed 's/^/5 waves, 96 VGPRs: /'
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06_o/spills.txt
