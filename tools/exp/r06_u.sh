#!/bin/bash
mkdir -p gpurun_out/r06_u
for v in default fused512 fused640 fused1024; do
  if [ $v = default ]; then unset WOST_LIB; else export WOST_LIB=elaina_amd/lib/variants/$v/libwost_hip.so; fi
  python tools/gpu_guided_bench.py --net-precision 16 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$v f16: solve %.4f s train %.4f s field_crc %d' % (j['solve_s'], j['train_s'], j['field_crc']))"
done 2>&1 | tee gpurun_out/r06_u/fused_threads.txt
