#!/bin/bash
# round 5, trip ab: where the 3-D grid gradient's time goes (504 launches = 28 % of GuidedIntegrator<3>'s GPU time): the launches of
# tools/probes/bench3d_guided_only.py by LDS size and grid size
export TMPDIR=/tmp
O=gpurun_out/r05_ab; mkdir -p $O
cd /tmp
timeout 400 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/trace -- python3 $GRAFT_REPO_ROOT/tools/probes/bench3d_guided_only.py > $GRAFT_REPO_ROOT/$O/run.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find $O/trace -name '*kernel_trace.csv' | head -1)
python3 - "$f" <<'PY' | tee $O/grid_grad_launches.txt
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
g = collections.OrderedDict()
for r in rows:
    n = r["Kernel_Name"]
    if "grid_grad" not in n and "grid_bin3" not in n and "net_backward_wgrad" not in n and "vmm3_loss" not in n and "optimizer_kernel" not in n: continue
    key = (n.split("(")[0][-40:], int(r["LDS_Block_Size"]), int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])))
    g.setdefault(key, []).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
print("kernel, LDS bytes, blocks: launches, mean us, min us, max us")
for k, v in g.items():
    print(k, len(v), round(sum(v) / len(v) / 1e3, 1), round(min(v) / 1e3, 1), round(max(v) / 1e3, 1))
PY
grep "^{" $O/run.log | tee -a $O/grid_grad_launches.txt
rm -rf $O/trace
