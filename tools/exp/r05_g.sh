#!/bin/bash
# round 5, trip g: config 4 in half precision, full-size solve PAIRS -- fused guided kernel at two waves per SIMD (default from here
# on) against three (the round-4 build)
export TMPDIR=/tmp
O=gpurun_out/r05_g; mkdir -p $O
ONLY_F16=0 BOTH_F16=${PAIRS:-20} timeout 1500 python tools/probes/repro_cfg4.py > $O/pairs_512.log 2>&1
echo "512 threads: identical pairs $(grep -c 'field equal True.*weights equal True' $O/pairs_512.log) of $(grep -c 'field equal' $O/pairs_512.log)" | tee -a $O/summary.txt
WOST_LIB=elaina_amd/lib/variants/fused768/libwost_hip.so ONLY_F16=0 BOTH_F16=${PAIRS:-20} timeout 1500 python tools/probes/repro_cfg4.py > $O/pairs_768.log 2>&1
echo "768 threads: identical pairs $(grep -c 'field equal True.*weights equal True' $O/pairs_768.log) of $(grep -c 'field equal' $O/pairs_768.log)" | tee -a $O/summary.txt
P='import sys,json
for l in sys.stdin:
    if l.startswith("{"):
        d=json.loads(l); print(round(d["value"]/1e9,4),"e9", round(d["ms_per_step"],1),"ms")'
echo "cfg4 f16 at 512:" | tee -a $O/summary.txt; python bench.py --config 4 --net-precision 16 --steps 2 --warmup 1 2>/dev/null | python -c "$P" | tee -a $O/summary.txt
echo "cfg4 f16 at 768:" | tee -a $O/summary.txt; WOST_LIB=elaina_amd/lib/variants/fused768/libwost_hip.so python bench.py --config 4 --net-precision 16 --steps 2 --warmup 1 2>/dev/null | python -c "$P" | tee -a $O/summary.txt
