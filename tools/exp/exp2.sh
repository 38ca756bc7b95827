#!/bin/bash
# round-3 experiment 2: the quad kernel -- parity, then what it buys in under-filled launches
export TMPDIR=/tmp
O=gpurun_out/r03_exp2
mkdir -p $O
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "quad or refill or random_scenes" 2>&1 | tail -8 | tee $O/pytest.log
val() { python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        r = json.loads(l); print('$1', round(r['value']/1e9,3), 'e9 steps/s', round(r['ms_per_step'],2), 'ms', r.get('time_to_1spp_ms'))"; }
echo "== cfg2 full frame" | tee -a $O/log.txt
for q in "quad=0" "quad=-1" "quad=-1 --opt quad_fill=2" "quad=-1 --opt quad_fill=4"; do
  python bench.py --steps 2 --warmup 1 --no-extras --no-cpu-baseline --opt $q 2>/dev/null | val "$q" | tee -a $O/log.txt
done
echo "== shard probe" | tee -a $O/log.txt
for q in "quad=0" "quad=-1 quad_fill=1" "quad=-1 quad_fill=1.5" "quad=-1 quad_fill=2" "quad=-1 quad_fill=3" "quad=-1 quad_fill=6" "quad=1"; do
  echo "-- $q" | tee -a $O/log.txt
  python tools/gpu_shard_probe.py ladybug 256 $q 2>/dev/null | tee -a $O/log.txt
done
echo "== launches trace, shard of 8, quad auto fill 2" | tee -a $O/log.txt
WOST_TRACE_LAUNCHES=1 python tools/gpu_shard_probe.py ladybug 256 quad=-1 quad_fill=2 2>&1 | grep -E "^launch|shard 0 of 8" | tail -40 | tee -a $O/log.txt
