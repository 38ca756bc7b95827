#!/bin/bash
mkdir -p gpurun_out/r06_j
python -m pytest tests/test_guided_integrator.py -x -q -m gpu -k "fused or training_end_to_end or frozen or config4 or shard" > gpurun_out/r06_j/pytest.log 2>&1; tail -4 gpurun_out/r06_j/pytest.log
for prec in 16 32; do for o in 0 1; do
python tools/gpu_guided_bench.py --net-precision $prec --opt walk_order=$o --repeat 2 2>/dev/null | tee -a gpurun_out/r06_j/cfg4.txt
done; done
python tools/gpu_guided_bench.py --net-precision 16 --opt walk_order=0 --train-group 16 --pipeline 1 --repeat 2 2>/dev/null | tee -a gpurun_out/r06_j/cfg4.txt
python tools/gpu_guided_bench.py --net-precision 16 --opt walk_order=1 --train-group 16 --pipeline 1 --repeat 2 2>/dev/null | tee -a gpurun_out/r06_j/cfg4.txt
for o in 0 1; do
python tools/gpu_guided_bench.py --net-precision 16 --frame 2048 --spp 128 --train-spp 16 --one-shard-of 8 --opt walk_order=$o 2>/dev/null | tee -a gpurun_out/r06_j/cfg5.txt
done
