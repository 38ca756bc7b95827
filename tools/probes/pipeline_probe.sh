#!/bin/bash
# config 4 (ladybug guided, 1024^2, 256 spp all trained) in the exact and the reordered training orders
# CFGS = "pipeline:train_group ..." PRECS = "16 32"
export TMPDIR=/tmp
mkdir -p gpurun_out/pipe
for prec in ${PRECS:-16}; do
  for cfg in ${CFGS:-0:1 0:4 0:8 0:16 1:1 1:4 1:8 1:16}; do
    python tools/gpu_guided_bench.py --net-precision $prec --pipeline ${cfg%%:*} --train-group ${cfg##*:} --repeat 2 ${EXTRA} 2>&1 | grep -v amdgpu.ids | tail -1 | tee -a gpurun_out/pipe/cfg4.jsonl
  done
done
