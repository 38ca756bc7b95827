#!/bin/bash
# round 5, trip d: the packed tree image in the round kernel -- parity, then config 2 / 3 against the fp32 records
export TMPDIR=/tmp
O=gpurun_out/r05_d; mkdir -p $O
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "config1 or full_size or variants or round_length or random_scenes or closest_point or refill or large_neumann or emissive" > $O/pytest.log 2>&1; tail -5 $O/pytest.log
B="python bench.py --no-extras --no-cpu-baseline --steps 5 --warmup 2 --opt pair=0"
P='import sys,json
for l in sys.stdin:
    if l.startswith("{"):
        d=json.loads(l); print(round(d["value"]/1e9,3),"e9", round(d["ms_per_step"],2),"ms", d.get("time_to_1spp_ms"), d.get("scheduler"))'
for cfg in 2 3; do
  echo "== config $cfg packed" | tee -a $O/packed.txt;   $B --config $cfg 2>/dev/null | python -c "$P" | tee -a $O/packed.txt
  echo "== config $cfg fp32 records" | tee -a $O/packed.txt; WOST_LIB=elaina_amd/lib/variants/unpacked/libwost_hip.so $B --config $cfg 2>/dev/null | python -c "$P" | tee -a $O/packed.txt
done
