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
for i in 1 2; do WOST_GUIDED_TAIL_CHUNK=0 run chunk0_$i; WOST_GUIDED_TAIL_CHUNK=4 run chunk4_$i; done
