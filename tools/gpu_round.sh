#!/bin/bash
# one GPU-box trip: gpu tests, smoke, bench, rocprof kernel trace + PMC traffic of the bench command
export TMPDIR=/tmp
TAG=${TAG:-r01}
mkdir -p gpurun_out/$TAG
python -m pytest tests -x -q -m gpu 2>&1 | tail -5 | tee gpurun_out/$TAG/pytest_gpu.log
python __graft_entry__.py smoke 2>&1 | tail -2 | tee gpurun_out/$TAG/smoke.log
python bench.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/$TAG/bench.json
BARGS="bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-1spp"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$TAG/trace -- python3 $BARGS > gpurun_out/$TAG/bench_trace.log 2>&1
f=$(find gpurun_out/$TAG/trace -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" gpurun_out/$TAG/kernel_stats.csv && head -5 "$f"
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" \
 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
 "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM" \
 "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 180 rocprofv3 --pmc $grp --output-format csv -d gpurun_out/$TAG/pmc$i -- python3 $BARGS > gpurun_out/$TAG/pmc$i.log 2>&1
  f=$(find gpurun_out/$TAG/pmc$i -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && python3 tools/pmc_summary.py "$f" | tee -a gpurun_out/$TAG/pmc_summary.txt
done
python3 - <<PY
import json, re
tot = {}
for line in open("gpurun_out/$TAG/pmc_summary.txt"):
    m = re.search(r"walk_round_kernel.*?(FETCH_SIZE|WRITE_SIZE)\s+calls=(\d+)\s+sum=([0-9.e+]+)", line)
    if m:
        tot[m.group(1)] = (int(m.group(2)), float(m.group(3)))
if "FETCH_SIZE" in tot and "WRITE_SIZE" in tot:
    calls = tot["FETCH_SIZE"][0]
    # MI355X_MICROARCH.md "HBM": FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports
    # half of the bytes of 16-B-per-lane reads -> doubled; WRITE_SIZE is exact
    per_launch = (2.0 * tot["FETCH_SIZE"][1] + tot["WRITE_SIZE"][1]) * 1024.0 / calls
    json.dump({"kernel": "walk_round_kernel", "launches": calls, "fetch_kib_sum": tot["FETCH_SIZE"][1],
               "write_kib_sum": tot["WRITE_SIZE"][1], "hbm_bytes_per_launch": per_launch,
               "command": "rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE -- python3 $BARGS",
               "correction": "2*FETCH_SIZE + WRITE_SIZE, KiB -> bytes"},
              open("gpurun_out/$TAG/walk_round_traffic.json", "w"), indent=1)
PY
rm -rf gpurun_out/$TAG/pmc[0-9] gpurun_out/$TAG/trace
