#!/bin/bash
# round 5, trip b: the pair kernel (parity, then config 2 against the round kernel, scheduler weights, 5 waves per SIMD) and the
# matrix-instruction chain reproducer
export TMPDIR=/tmp
O=gpurun_out/r05_b; mkdir -p $O
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "pair or config1 or full_size or variants or round_length" > $O/pytest_pair.log 2>&1; tail -5 $O/pytest_pair.log
B="python bench.py --no-extras --no-cpu-baseline --no-1spp --steps 5 --warmup 2"
for o in "pair=0" "pair=-1" "pair=-1 wait_weight=4" "pair=-1 wait_weight=6" "pair=-1 wait_weight=12" "pair=-1 wait_weight=16" "pair=-1 pair_fill=0.5" "pair=-1 pair_fill=0.25"; do
  a=""; for kv in $o; do a="$a --opt $kv"; done
  echo "== $o" | tee -a $O/pair_sweep.txt
  $B $a 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(round(d['value']/1e9,3),'e9', round(d['ms_per_step'],2),'ms', d.get('scheduler'), d['roofline'].get('launches'), d['roofline'].get('kernel_ms_per_launch'))
" | tee -a $O/pair_sweep.txt
done
echo "== pair5 (5 waves per SIMD)" | tee -a $O/pair_sweep.txt
WOST_LIB=elaina_amd/lib/variants/pair5/libwost_hip.so $B 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(round(d['value']/1e9,3),'e9', round(d['ms_per_step'],2),'ms', d.get('scheduler'))
" | tee -a $O/pair_sweep.txt
echo "== config 3 pair=0 / auto" | tee -a $O/pair_sweep.txt
for o in "pair=0" "pair=-1"; do $B --config 3 --opt $o 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(round(d['value']/1e9,3),'e9', round(d['ms_per_step'],2),'ms', d.get('scheduler'))
" | tee -a $O/pair_sweep.txt; done
timeout 900 ./tools/micro/mfma_chain 20000 2>&1 | tee $O/mfma_chain.txt
