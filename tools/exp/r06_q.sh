#!/bin/bash
mkdir -p gpurun_out/r06_q
for prec in 32 16; do for o in 0 1 1; do
python tools/gpu_guided_bench.py --net-precision $prec --net-opt grid_overlap=$o 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('precision $prec grid_overlap $o: solve %.4f s train %.4f s launches %d field_crc %d params_crc %d' % (j['solve_s'], j['train_s'], j['kernel_launches'], j['field_crc'], j['params_crc']))"
done; done 2>&1 | tee gpurun_out/r06_q/overlap.txt
