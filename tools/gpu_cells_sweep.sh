#!/bin/bash
# cell-list kernel: parity tests, then bench of a few compile-time variants (rebuilt on the box)
export TMPDIR=/tmp
TAG=${TAG:-r02b}
mkdir -p gpurun_out/$TAG
python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -8 | tee gpurun_out/$TAG/pytest_parity.log
BARGS="bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras"
for cfg in "5 1" "6 1" "4 1" "4 2" "4 4" "8 1"; do
  set -- $cfg
  WOST_HIPCC_DEFS="-DWOST_CELLS_WAVES=$1 -DWOST_CELLS_SCAN_UNROLL=$2" python -c "
from elaina_amd import build as b
import os
os.utime(os.path.join(b.CSRC, 'wost_cells.h'))
b.build_library()" > gpurun_out/$TAG/build_$1_$2.log 2>&1
  echo "== waves $1 unroll $2" | tee -a gpurun_out/$TAG/sweep.txt
  python $BARGS 2>/dev/null | grep -v amdgpu.ids | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        r = json.loads(l); print(r['value'], r['ms_per_step'], r['time_to_1spp_ms'], r['roofline']['launches'], r['rel_l2_vs_oracle'] if 'rel_l2_vs_oracle' in r else '')" | tee -a gpurun_out/$TAG/sweep.txt
  python bench.py --config 3 --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-1spp 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        r = json.loads(l); print('fille', r['value'], r['ms_per_step'])" | tee -a gpurun_out/$TAG/sweep.txt
done
echo "== tree kernel" | tee -a gpurun_out/$TAG/sweep.txt
python $BARGS --opt accel=0 2>/dev/null | grep -v amdgpu.ids | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        r = json.loads(l); print(r['value'], r['ms_per_step'])" | tee -a gpurun_out/$TAG/sweep.txt
