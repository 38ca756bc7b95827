#!/bin/bash
# round 6, trip a: the persistent first launch -- parity tests, then timing against the rounds
mkdir -p gpurun_out/r06_a
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "persistent or refill or config1 or round_length" > gpurun_out/r06_a/pytest.log 2>&1; tail -5 gpurun_out/r06_a/pytest.log
python tools/exp/r06_persist.py ladybug 256 > gpurun_out/r06_a/persist_ladybug.txt 2>&1; cat gpurun_out/r06_a/persist_ladybug.txt
python tools/exp/r06_persist.py fille 256 > gpurun_out/r06_a/persist_fille.txt 2>&1; grep -v "^launch" gpurun_out/r06_a/persist_fille.txt
python tools/exp/r06_1spp.py > gpurun_out/r06_a/one_spp.txt 2>&1; cat gpurun_out/r06_a/one_spp.txt
