#!/bin/bash
# round 5, trip ac: GuidedIntegrator<3> on the bench scenes after the fused sample kernel and the boxed grid gradient: every kernel by
# (name, blocks), launches and total time (tools/probes/bench3d_guided_only.py under rocprofv3 --kernel-trace)
export TMPDIR=/tmp
O=gpurun_out/r05_ac; mkdir -p $O
cd /tmp
timeout 400 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/trace -- python3 $GRAFT_REPO_ROOT/tools/probes/bench3d_guided_only.py > $GRAFT_REPO_ROOT/$O/run.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find $O/trace -name '*kernel_trace.csv' | head -1)
python3 - "$f" <<'PY' | tee $O/guided3d_kernels.txt
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
g = collections.OrderedDict()
tot = 0
for r in rows:
    n = r["Kernel_Name"]
    key = (n.split("(")[0][-48:], int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])))
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    g.setdefault(key, []).append(d)
    tot += d
print("kernel, blocks: launches, total ms, mean us, max us   (all kernels: %.1f ms)" % (tot / 1e6))
for k, v in sorted(g.items(), key=lambda kv: -sum(kv[1])):
    if sum(v) > 0.004 * tot: print(k, len(v), round(sum(v) / 1e6, 2), round(sum(v) / len(v) / 1e3, 1), round(max(v) / 1e3, 1))
PY
grep "^{" $O/run.log | tee -a $O/guided3d_kernels.txt
rm -rf $O/trace
