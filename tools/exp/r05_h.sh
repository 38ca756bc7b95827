#!/bin/bash
# round 5, trip h: the device build of the 2-D trees against the host builder, then config 1-3 parity on device-built trees
export TMPDIR=/tmp
O=gpurun_out/r05_h; mkdir -p $O
python -m pytest tests/test_gpu_build2.py -x -q -m gpu -s > $O/pytest_build2.log 2>&1; tail -40 $O/pytest_build2.log | cut -c1-250
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "config1 or full_size or random_scenes or closest_point or large_neumann or silhouette" > $O/pytest_parity.log 2>&1; tail -4 $O/pytest_parity.log
