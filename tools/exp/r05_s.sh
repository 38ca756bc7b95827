#!/bin/bash
# round 5, trip s: time-to-1spp with a step limit in the refill launch (long walks finished by the rounds that follow: thin waves, quads)
export TMPDIR=/tmp
O=gpurun_out/r05_s; mkdir -p $O
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "refill" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
P='import sys,json
for l in sys.stdin:
    if l.startswith("{"):
        d=json.loads(l); print(d.get("time_to_1spp_ms"))'
for k in 0 6 8 12 16 24 32 48; do
  echo "== refill_steps $k" | tee -a $O/onespp.txt
  for cfg in 2 3; do python bench.py --config $cfg --steps 1 --warmup 0 --no-extras --no-cpu-baseline --opt refill_steps=$k 2>/dev/null | python -c "$P" | tee -a $O/onespp.txt; done
done
