#!/bin/bash
# round 6, trip c: what the persistent launch hands over -- long remainders beside the rounds, sorted first round
mkdir -p gpurun_out/r06_c
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "persistent or refill or config1 or round_length or quad" > gpurun_out/r06_c/pytest.log 2>&1; tail -5 gpurun_out/r06_c/pytest.log
python tools/exp/r06_persist.py ladybug 256 > gpurun_out/r06_c/persist_ladybug.txt 2>&1; cat gpurun_out/r06_c/persist_ladybug.txt
python tools/exp/r06_persist.py fille 256 > gpurun_out/r06_c/persist_fille.txt 2>&1; grep -v "^launch" gpurun_out/r06_c/persist_fille.txt
