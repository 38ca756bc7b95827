#!/bin/bash
# one GPU-box trip: gpu tests, smoke, bench, rocprof kernel trace + PMC traffic of the bench command
export TMPDIR=/tmp
TAG=${TAG:-r01}
mkdir -p gpurun_out/$TAG
python -m pytest tests -x -q -m gpu 2>&1 | tail -5 | tee gpurun_out/$TAG/pytest_gpu.log
python __graft_entry__.py smoke 2>&1 | tail -2 | tee gpurun_out/$TAG/smoke.log
python bench.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/$TAG/bench.json
BARGS="bench.py --steps 1 --warmup 0 --no-cpu-baseline"
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
rm -rf gpurun_out/$TAG/pmc[0-9] gpurun_out/$TAG/trace
