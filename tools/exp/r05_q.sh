#!/bin/bash
# round 5, trip q: guided 3-D with the global-atomic grid-gradient levels in one launch -- parity, then the bench scenes
export TMPDIR=/tmp
O=gpurun_out/r05_q; mkdir -p $O
python -m pytest tests/test_guided_3d.py tests/test_gpu_far_trees.py -x -q -m gpu -k "guided3 or net3 or 3d or trained or frozen" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
python tools/probes/bench3d_guided_only.py 2>&1 | tail -8 | tee $O/guided3d.txt
