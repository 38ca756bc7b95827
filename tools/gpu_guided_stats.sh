#!/bin/bash
# Developer tool (GPU box): per-kernel time of a short guided solve (32 trained samples of config 4).
# Usage: bash tools/gpu_guided_stats.sh [tag]   -> gpurun_out/<tag>_guided_kernel_stats.csv
TAG=${1:-dev}
export TMPDIR=/tmp
rm -rf gpurun_out/gs_$TAG
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/gs_$TAG -- python3 tools/gpu_guided_bench.py --spp 32 --train-spp 32 $GARGS > gpurun_out/${TAG}_guided_stats.log 2>&1
f=$(find gpurun_out/gs_$TAG -name "*kernel_stats.csv" | head -1)
cp "$f" gpurun_out/${TAG}_guided_kernel_stats.csv
rm -rf gpurun_out/gs_$TAG
python3 tools/print_kernel_stats.py gpurun_out/${TAG}_guided_kernel_stats.csv | head -16
grep walk_steps gpurun_out/${TAG}_guided_stats.log | cut -c1-300
