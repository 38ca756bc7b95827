#!/bin/bash
# (a record: ran at commit 45cb138, whose wost_net.hip still had the fused-loss kernel, its three-launch check and the -D variants used here)
# round 5, trip f: where the three launches part (blocks of 768): the hidden activations of every layer dumped and compared
export TMPDIR=/tmp
O=gpurun_out/r05_f; mkdir -p $O
WOST_LIB=elaina_amd/lib/variants/h768_dump/libwost_hip.so REPS=3 timeout 900 python tools/probes/repro_probe3.py > $O/h768_dump.log 2>&1
grep -E "^DUMP|^   tile" $O/h768_dump.log | head -80
