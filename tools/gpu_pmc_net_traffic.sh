#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/pmct
for grp in "FETCH_SIZE" "WRITE_SIZE"; do
  timeout 300 rocprofv3 --pmc $grp --output-format csv -d gpurun_out/pmct/p -- python3 tools/gpu_net_probe.py pixel > gpurun_out/pmct/p.log 2>&1
  f=$(find gpurun_out/pmct/p -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && python3 tools/pmc_summary.py "$f" net_ grid_grad weight_grad
  rm -rf gpurun_out/pmct/p
done
