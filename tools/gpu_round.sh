#!/bin/bash
# one GPU-box trip: gpu tests, smoke, bench, rocprof kernel trace (summary -> gpurun_out/)
set -x
export TMPDIR=/tmp
mkdir -p gpurun_out
python -m pytest tests -x -q -m gpu 2>&1 | tail -15 | tee gpurun_out/pytest_gpu.log
python __graft_entry__.py smoke 2>&1 | tail -3 | tee gpurun_out/smoke.log
python bench.py --steps 2 --warmup 1 2>&1 | tee gpurun_out/bench.log
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/bench_prof.log 2>&1
find gpurun_out/prof -name '*stats*' | head; 
f=$(find gpurun_out/prof -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" gpurun_out/kernel_stats.csv && head -12 "$f"
