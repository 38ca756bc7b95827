#!/usr/bin/env python3
"""VALU figures of the guided 3-D walk kernels (g3_fused_kernel; g3_separate_kernel, g3_sample_kernel, g3_tail_kernel) per bench scene from PMC
summaries of tools/probes/bench3d_guided_only.py.  Usage: pmc_derive_guided3d.py pmc_summary.txt kernel_stats.csv out.json "<command>"
The Dirichlet icosphere runs the instantiations with NTREE = false, the Neumann shell those with NTREE = true (template
arguments EMISSIVE, NTREE, SOURCE of g3_fused / g3_separate / g3_tail; NTREE of g3_sample).  The 256^2 scenes run g3_fused_kernel (one launch
per sample), the 1024^2 scenes of the same command the launches per depth."""
import csv
import json
import os
import re
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from elaina_amd.build import source_id  # noqa: E402

summary, stats_csv, out_path, cmd = sys.argv[1:5]


def scene_of(name):
    m = re.search(r"g3_(separate|tail|fused)_kernel<(\w+), (\w+), (\w+)>", name) or re.search(r"g3_(sample)_kernel<()(\w+)>", name)
    if not m:
        return None, None
    return ("neumann_shell_1280" if m.group(3) == "true" else "dirichlet_icosphere_1280"), "g3_%s_kernel" % m.group(1)


per = {}
for line in open(summary):
    m = re.search(r"(g3_\w+_kernel<[^>]*>).*?(\w+)\s+calls=(\d+)\s+sum=([0-9.e+]+)", line)
    if not m:
        continue
    scene, kern = scene_of(m.group(1))
    if scene:
        c = per.setdefault(scene, {}).setdefault(kern, {})
        c[m.group(2)] = (int(m.group(3)), float(m.group(4)))
try:
    rows = list(csv.DictReader(open(stats_csv)))
except Exception:
    rows = []
total_ns = sum(float(r["TotalDurationNs"]) for r in rows) or 1.0
out = {"kernels": "g3_fused_kernel (256^2) / g3_separate_kernel / g3_sample_kernel / g3_tail_kernel (1024^2)", "source_id": source_id(), "scenes": {},
       "source": "rocprofv3 --pmc passes of `%s` (tools/gpu_round.sh, stage pmc_guided3d): pipe_busy = 4 SQ_ACTIVE_INST_VALU / (1024 SIMDs x "
                 "GRBM_GUI_ACTIVE / 8), lane_efficiency = SQ_THREAD_CYCLES_VALU / (64 SQ_INSTS_VALU), wait_share = SQ_WAIT_ANY / SQ_WAVE_CYCLES; "
                 "share_of_gpu_time from the kernel trace of the same command (both scenes, network and training kernels included)" % cmd}
for scene, kerns in per.items():
    e = {}
    for kern, tot in kerns.items():
        cyc = tot.get("GRBM_GUI_ACTIVE", (0, 0.0))[1] / 8.0
        k = {"launches": tot.get("SQ_INSTS_VALU", (0, 0))[0],
             "pipe_busy": 4.0 * tot["SQ_ACTIVE_INST_VALU"][1] / (1024.0 * cyc) if cyc and "SQ_ACTIVE_INST_VALU" in tot else None,
             "lane_efficiency": tot["SQ_THREAD_CYCLES_VALU"][1] / (64.0 * tot["SQ_INSTS_VALU"][1]) if "SQ_THREAD_CYCLES_VALU" in tot and tot.get("SQ_INSTS_VALU", (0, 0))[1] else None,
             "wait_share": tot["SQ_WAIT_ANY"][1] / tot["SQ_WAVE_CYCLES"][1] if "SQ_WAIT_ANY" in tot and tot.get("SQ_WAVE_CYCLES", (0, 0))[1] else None,
             "valu_wave_instructions": tot.get("SQ_INSTS_VALU", (0, None))[1]}
        ns = 0.0
        for r in rows:
            s2, k2 = scene_of(r["Name"])
            if s2 == scene and k2 == kern:
                ns += float(r["TotalDurationNs"])
        k["share_of_gpu_time"] = ns / total_ns
        k["total_ms"] = ns / 1e6
        e[kern] = k
    out["scenes"][scene] = e
json.dump(out, open(out_path, "w"), indent=1)
print(json.dumps(out))
