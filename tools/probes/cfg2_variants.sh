#!/bin/bash
# config 2 with the default library and developer variants (LIBS = variant names under elaina_amd/lib/variants)
export TMPDIR=/tmp
mkdir -p gpurun_out/cfg2
for v in ${LIBS:-default}; do
  lib=elaina_amd/lib/variants/$v/libwost_hip.so
  [ $v = default ] && lib=elaina_amd/lib/libwost_hip.so
  echo "== $v" | tee -a gpurun_out/cfg2/variants.txt
  WOST_LIB=$lib python bench.py --steps ${STEPS:-10} --warmup 2 --no-cpu-baseline --no-1spp --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps({k:d[k] for k in ('value','ms_per_step')}), d['config']['walk_steps_per_pass'], json.dumps(d.get('scheduler')))" | tee -a gpurun_out/cfg2/variants.txt
done
