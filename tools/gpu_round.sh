#!/bin/bash
# one GPU-box trip: gpu tests, smoke, bench, rocprof kernel trace + PMC passes of the bench command.
# TAG names the output directory under gpurun_out/; STAGES selects what runs (default: all).
export TMPDIR=/tmp
TAG=${TAG:-r06}
STAGES=${STAGES:-"tests smoke micro bench trace pmc pmc_guided pmc3d pmc3d_1024 pmc_guided3d build3 build2"}
mkdir -p gpurun_out/$TAG
has() { [[ " $STAGES " == *" $1 "* ]]; }
if has tests; then
  # the whole log with the slowest tests named: the driver's limit for this command is 1 200 s, ours 600 s
  python -m pytest tests -x -q -m gpu -rs --durations=40 > gpurun_out/$TAG/pytest_gpu.log 2>&1; tail -60 gpurun_out/$TAG/pytest_gpu.log
  # a green run with skipped GPU tests is a broken environment, not a pass
  if grep -qE '[0-9]+ skipped' gpurun_out/$TAG/pytest_gpu.log; then echo "GPU TESTS WERE SKIPPED" | tee -a gpurun_out/$TAG/pytest_gpu.log; fi
fi
if has smoke; then python __graft_entry__.py smoke 2>&1 | tail -3 | tee gpurun_out/$TAG/smoke.log; fi
if has micro; then ./tools/micro/valu_rate 2>&1 | tee gpurun_out/$TAG/valu_rate.txt; fi
if has bench; then python bench.py 2>gpurun_out/$TAG/bench.err | grep -v amdgpu.ids | tee gpurun_out/$TAG/bench.json; tail -5 gpurun_out/$TAG/bench.err; fi
BARGS="bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-1spp --no-extras"
if has trace; then
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$TAG/trace -- python3 $BARGS > gpurun_out/$TAG/bench_trace.log 2>&1
  f=$(find gpurun_out/$TAG/trace -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" gpurun_out/$TAG/kernel_stats.csv && head -5 "$f"
fi
if has pmc; then
  i=0
  rm -f gpurun_out/$TAG/pmc_summary.txt
  for grp in "FETCH_SIZE" "WRITE_SIZE" \
   "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
   "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM" \
   "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    timeout 180 rocprofv3 --pmc $grp --output-format csv -d gpurun_out/$TAG/pmc$i -- python3 $BARGS > gpurun_out/$TAG/pmc$i.log 2>&1
    f=$(find gpurun_out/$TAG/pmc$i -name '*counter_collection.csv' | head -1)
    [ -n "$f" ] && python3 tools/pmc_summary.py "$f" | tee -a gpurun_out/$TAG/pmc_summary.txt
  done
  python3 tools/pmc_derive.py gpurun_out/$TAG/pmc_summary.txt gpurun_out/$TAG/bench_trace.log gpurun_out/$TAG "$BARGS"
fi
# the guided configuration's dominant kernel (config 4 at 32 spp, 16 of them trained): kernel trace + PMC passes per precision
if has pmc_guided; then
  for prec in 16 32; do
    GARGS="bench.py --config 4 --spp 32 --train-spp 16 --steps 1 --warmup 0 --net-precision $prec"
    timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$TAG/gtrace$prec -- python3 $GARGS > gpurun_out/$TAG/guided_trace_f$prec.log 2>&1
    f=$(find gpurun_out/$TAG/gtrace$prec -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" gpurun_out/$TAG/guided_kernel_stats_f$prec.csv
    rm -f gpurun_out/$TAG/guided_pmc_f$prec.txt
    i=0
    for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
               "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU SQ_VALU_MFMA_BUSY_CYCLES" \
               "GRBM_GUI_ACTIVE"; do
      i=$((i+1))
      timeout 300 rocprofv3 --pmc $grp --output-format csv -d gpurun_out/$TAG/gpmc$i -- python3 $GARGS > gpurun_out/$TAG/gpmc$i.log 2>&1
      f=$(find gpurun_out/$TAG/gpmc$i -name '*counter_collection.csv' | head -1)
      [ -n "$f" ] && python3 tools/pmc_summary.py "$f" guided_sample grid_grad net_train net_forward optimizer | tee -a gpurun_out/$TAG/guided_pmc_f$prec.txt
      rm -rf gpurun_out/$TAG/gpmc$i
    done
    python3 tools/pmc_derive_guided.py gpurun_out/$TAG/guided_pmc_f$prec.txt gpurun_out/$TAG/guided_kernel_stats_f$prec.csv gpurun_out/$TAG/guided_trace_f$prec.log gpurun_out/$TAG/guided_sample_f$prec.json "$GARGS"
    rm -rf gpurun_out/$TAG/gtrace$prec
  done
fi
# the 3-D uniform kernel on the two bench scenes (Dirichlet icosphere, Neumann shell): kernel trace + PMC passes
if has pmc3d; then
  A3="tools/probes/bench3d_only.py"
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$TAG/trace3d -- python3 $A3 > gpurun_out/$TAG/bench3d_trace.log 2>&1
  f=$(find gpurun_out/$TAG/trace3d -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" gpurun_out/$TAG/kernel_stats_3d.csv
  rm -f gpurun_out/$TAG/pmc_summary_3d.txt
  i=0
  for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
             "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_ANY" "GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    timeout 300 rocprofv3 --pmc $grp --output-format csv -d gpurun_out/$TAG/p3d$i -- python3 $A3 > gpurun_out/$TAG/p3d$i.log 2>&1
    f=$(find gpurun_out/$TAG/p3d$i -name '*counter_collection.csv' | head -1)
    [ -n "$f" ] && PMC_NAME_WIDTH=90 python3 tools/pmc_summary.py "$f" walk3_kernel | tee -a gpurun_out/$TAG/pmc_summary_3d.txt
    rm -rf gpurun_out/$TAG/p3d$i
  done
  python3 tools/pmc_derive_3d.py gpurun_out/$TAG/pmc_summary_3d.txt gpurun_out/$TAG/kernel_stats_3d.csv gpurun_out/$TAG/walk3_valu.json "$A3"
  rm -rf gpurun_out/$TAG/trace3d
fi
# the same two scenes at 1024^2 (a million walkers: the full-chip frames)
if has pmc3d_1024; then
  A3="tools/probes/bench3d_only.py"
  export BENCH3D_FRAME=1024
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$TAG/trace3d -- python3 $A3 > gpurun_out/$TAG/bench3d_1024_trace.log 2>&1
  f=$(find gpurun_out/$TAG/trace3d -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" gpurun_out/$TAG/kernel_stats_3d_1024.csv
  rm -f gpurun_out/$TAG/pmc_summary_3d_1024.txt
  i=0
  for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
             "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_ANY" "GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    timeout 300 rocprofv3 --pmc $grp --output-format csv -d gpurun_out/$TAG/p3d$i -- python3 $A3 > gpurun_out/$TAG/p3d$i.log 2>&1
    f=$(find gpurun_out/$TAG/p3d$i -name '*counter_collection.csv' | head -1)
    [ -n "$f" ] && PMC_NAME_WIDTH=90 python3 tools/pmc_summary.py "$f" walk3_kernel | tee -a gpurun_out/$TAG/pmc_summary_3d_1024.txt
    rm -rf gpurun_out/$TAG/p3d$i
  done
  python3 tools/pmc_derive_3d.py gpurun_out/$TAG/pmc_summary_3d_1024.txt gpurun_out/$TAG/kernel_stats_3d_1024.csv gpurun_out/$TAG/walk3_valu_1024.json "BENCH3D_FRAME=1024 $A3"
  rm -rf gpurun_out/$TAG/trace3d
  unset BENCH3D_FRAME
fi
# GuidedIntegrator<3> on the two bench scenes (256^2, 16 spp, 8 trained): kernel trace + PMC passes of its walk kernels
if has pmc_guided3d; then
  AG="tools/probes/bench3d_guided_only.py"
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$TAG/traceg3 -- python3 $AG > gpurun_out/$TAG/guided3d_trace.log 2>&1
  f=$(find gpurun_out/$TAG/traceg3 -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" gpurun_out/$TAG/kernel_stats_guided3d.csv
  rm -f gpurun_out/$TAG/pmc_summary_guided3d.txt
  i=0
  for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY" "GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    timeout 300 rocprofv3 --pmc $grp --output-format csv -d gpurun_out/$TAG/pg3$i -- python3 $AG > gpurun_out/$TAG/pg3$i.log 2>&1
    f=$(find gpurun_out/$TAG/pg3$i -name '*counter_collection.csv' | head -1)
    [ -n "$f" ] && PMC_NAME_WIDTH=90 python3 tools/pmc_summary.py "$f" g3_ | tee -a gpurun_out/$TAG/pmc_summary_guided3d.txt
    rm -rf gpurun_out/$TAG/pg3$i
  done
  python3 tools/pmc_derive_guided3d.py gpurun_out/$TAG/pmc_summary_guided3d.txt gpurun_out/$TAG/kernel_stats_guided3d.csv gpurun_out/$TAG/guided3d_valu.json "$AG"
  rm -rf gpurun_out/$TAG/traceg3
fi
# the triangle LBVH built on the device (81 920 triangles, five builds each way): kernel trace of the build kernels
if has build3; then
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$TAG/traceb3 -- python3 tools/probes/build3_only.py > gpurun_out/$TAG/build3_trace.log 2>&1
  f=$(find gpurun_out/$TAG/traceb3 -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" gpurun_out/$TAG/kernel_stats_build3.csv
  grep mesh gpurun_out/$TAG/build3_trace.log | tail -1
  rm -rf gpurun_out/$TAG/traceb3
fi
# the segment LBVH built on the device (ladybug, fille; five builds each way): kernel trace of the build kernels
if has build2; then
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$TAG/traceb2 -- python3 tools/probes/build2_only.py > gpurun_out/$TAG/build2_trace.log 2>&1
  f=$(find gpurun_out/$TAG/traceb2 -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" gpurun_out/$TAG/kernel_stats_build2.csv
  grep mesh_build2 gpurun_out/$TAG/build2_trace.log | tail -1
  rm -rf gpurun_out/$TAG/traceb2
fi
rm -rf gpurun_out/$TAG/pmc[0-9] gpurun_out/$TAG/trace
