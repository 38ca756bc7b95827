#!/bin/bash
# round 5, trip y: the random-scene fuzzers with EVERY 2-D tree built on the device (WOST_DEVICE_BUILD_MIN=1: meshes of 4 .. 400 segments,
# closed and open, emissive, with sources and masks -- tree shapes around the leaf-size and arity boundaries) against the oracle
export TMPDIR=/tmp
export WOST_DEVICE_BUILD_MIN=1
O=gpurun_out/r05_y; mkdir -p $O
timeout 900 python tools/fuzz/fuzz_parity.py 0 240 2>&1 | tail -4 | tee $O/fuzz_parity_device_trees.txt
timeout 900 python tools/fuzz/fuzz_guided.py 0 24 2>&1 | tail -3 | tee -a $O/fuzz_parity_device_trees.txt
timeout 600 python tools/fuzz/fuzz_queries.py 1 2 3 4 5 6 7 8 2>&1 | awk '{print}' | grep -v "HIP!=brute 0 (dist 0)" | tail -12 | tee -a $O/fuzz_parity_device_trees.txt
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "tiny_meshes or ties or dirichlet_only or emissive or edge_settings" 2>&1 | tail -2 | tee -a $O/fuzz_parity_device_trees.txt
