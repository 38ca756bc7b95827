#!/bin/bash
# round 5, trip l: the first-tile redo of net_forward_h_kernel -- three-launch self check at full size (no launch in front), with and
# without mfma_settle, the kernel without the redo as control; then full-size config 4 pairs in half precision
export TMPDIR=/tmp
O=gpurun_out/r05_l; mkdir -p $O
REPS=${REPS:-6} timeout 1500 python tools/probes/check3_cfg4.py > $O/check3_redo.log 2>&1; echo "redo + settle:" | tee -a $O/summary.txt; grep -E "CHECK3" $O/check3_redo.log | tail -2 | tee -a $O/summary.txt
WOST_LIB=elaina_amd/lib/variants/redo_nosettle/libwost_hip.so REPS=${REPS:-6} timeout 1500 python tools/probes/check3_cfg4.py > $O/check3_redo_nosettle.log 2>&1; echo "redo, no settle:" | tee -a $O/summary.txt; grep -E "CHECK3" $O/check3_redo_nosettle.log | tail -2 | tee -a $O/summary.txt
WOST_LIB=elaina_amd/lib/variants/noredo/libwost_hip.so REPS=2 timeout 1500 python tools/probes/check3_cfg4.py > $O/check3_noredo.log 2>&1; echo "no redo (control):" | tee -a $O/summary.txt; grep -E "CHECK3" $O/check3_noredo.log | tail -2 | tee -a $O/summary.txt
ONLY_F16=0 BOTH_F16=${PAIRS:-25} timeout 2500 python tools/probes/repro_cfg4.py > $O/pairs.log 2>&1
echo "both f16, redo + settle: identical pairs $(grep -c 'field equal True.*weights equal True' $O/pairs.log) of $(grep -c 'field equal' $O/pairs.log)" | tee -a $O/summary.txt
grep -o "steps [0-9]* / [0-9]*" $O/pairs.log | sort | uniq -c | tee -a $O/summary.txt
P='import sys,json
for l in sys.stdin:
    if l.startswith("{"):
        d=json.loads(l); print(round(d["value"]/1e9,4),"e9", round(d["ms_per_step"],1),"ms")'
echo "cfg4 f16:" | tee -a $O/summary.txt; python bench.py --config 4 --net-precision 16 --steps 2 --warmup 1 2>/dev/null | python -c "$P" | tee -a $O/summary.txt
echo "cfg4 f16 no settle:" | tee -a $O/summary.txt; WOST_LIB=elaina_amd/lib/variants/redo_nosettle/libwost_hip.so python bench.py --config 4 --net-precision 16 --steps 2 --warmup 1 2>/dev/null | python -c "$P" | tee -a $O/summary.txt
