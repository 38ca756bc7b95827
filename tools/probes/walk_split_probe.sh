#!/bin/bash
# where the guiding-phase walk of config 4 spends its time: 64 guiding samples with the untrained network (train-spp 0),
# default build against developer variants without the network / without the mixture / without both
export TMPDIR=/tmp
mkdir -p gpurun_out/split
for v in default nonet novmm neither; do
  lib=elaina_amd/lib/variants/$v/libwost_hip.so
  [ $v = default ] && lib=elaina_amd/lib/libwost_hip.so
  for prec in 16 32; do
    echo "== $v f$prec" | tee -a gpurun_out/split/split.txt
    WOST_LIB=$lib python tools/gpu_guided_bench.py --net-precision $prec --spp 64 --train-spp 0 --repeat 2 2>&1 | grep -v amdgpu.ids | tail -1 | tee -a gpurun_out/split/split.txt
  done
done
