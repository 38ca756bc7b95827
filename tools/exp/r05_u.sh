#!/bin/bash
# round 5, trip u: node records at a 128-byte stride (one line per fetch) against the 96-byte records; the tree build after the
# wave-level reduction of the split costs
export TMPDIR=/tmp
O=gpurun_out/r05_u; mkdir -p $O
python -m pytest tests/test_gpu_build2.py -x -q -m gpu -s 2>&1 | grep -E "passed|failed|^.ladybug|^.fille|soup_70000" | tail -6 | tee $O/build2.txt
B="python bench.py --no-extras --no-cpu-baseline --steps 5 --warmup 2"
P='import sys,json
for l in sys.stdin:
    if l.startswith("{"):
        d=json.loads(l); print(round(d["value"]/1e9,3),"e9", round(d["ms_per_step"],2),"ms", d.get("time_to_1spp_ms"), d.get("scheduler",{}).get("visits_per_step"))'
for cfg in 2 3; do
  echo "== config $cfg, 96-byte records" | tee -a $O/node_stride.txt; $B --config $cfg 2>/dev/null | python -c "$P" | tee -a $O/node_stride.txt
  echo "== config $cfg, 128-byte stride" | tee -a $O/node_stride.txt; WOST_LIB=elaina_amd/lib/variants/node32/libwost_hip.so $B --config $cfg 2>/dev/null | python -c "$P" | tee -a $O/node_stride.txt
done
WOST_LIB=elaina_amd/lib/variants/node32/libwost_hip.so python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "config1 or closest_point or random_scenes" 2>&1 | tail -2 | tee -a $O/node_stride.txt
