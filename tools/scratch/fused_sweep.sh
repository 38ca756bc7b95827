#!/bin/bash
# timing of the fused guided sample kernel under variations (developer scratch)
export TMPDIR=/tmp
run() {
  tag=$1; shift
  rm -rf gpurun_out/fs_$tag
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/fs_$tag -- python3 tools/gpu_guided_bench.py --spp 16 --train-spp 16 --net-precision 16 "$@" > gpurun_out/fs_$tag.log 2>&1
  f=$(find gpurun_out/fs_$tag -name "*kernel_stats.csv" | head -1)
  echo "$tag: $(python3 tools/print_kernel_stats.py $f | grep guided_sample | cut -c60-110)"
  rm -rf gpurun_out/fs_$tag
}
WOST_GUIDED_DEEP=1000 run nodeep
WOST_GUIDED_DEEP=32 run deep32
WOST_GUIDED_DEEP=24 run deep24
WOST_GUIDED_DEEP=16 run deep16
WOST_GUIDED_DEEP=12 run deep12
WOST_GUIDED_DEEP=8 run deep8
WOST_GUIDED_DEEP=16 WOST_GUIDED_ORDER=4 run deep16order4
