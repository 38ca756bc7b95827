#!/usr/bin/env python3
"""VALU figures of walk3_kernel per instantiation from PMC summaries of tools/probes/bench3d_only.py (one solve of each of the
two 3-D bench scenes).  Usage: pmc_derive_3d.py pmc_summary.txt kernel_stats.csv out.json "<command>"
The Dirichlet-only scene runs walk3_kernel<false, false, false, WAVE>, the Neumann shell walk3_kernel<false, false, true, WAVE>
(template arguments EMISSIVE, SOURCE, NTREE, WAVE)."""
import csv
import json
import os
import re
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from elaina_amd.build import source_id  # noqa: E402

summary, stats_csv, out_path, cmd = sys.argv[1:5]
per = {}
for line in open(summary):
    m = re.search(r"walk3_kernel<(\w+), (\w+), (\w+)(?:, (\w+))?>.*?(\w+)\s+calls=(\d+)\s+sum=([0-9.e+]+)", line)
    if m:
        key = "neumann_shell_1280" if m.group(3) == "true" else "dirichlet_icosphere_1280"
        per.setdefault(key, {})[m.group(5)] = (int(m.group(6)), float(m.group(7)))
out = {"kernel": "walk3_kernel", "source_id": source_id(), "scenes": {},
       "source": "rocprofv3 --pmc passes of `%s` (tools/gpu_round.sh, stage pmc3d): pipe_busy = 4 SQ_ACTIVE_INST_VALU / (1024 SIMDs x GRBM_GUI_ACTIVE / 8), "
                 "lane_efficiency = SQ_THREAD_CYCLES_VALU / (64 SQ_INSTS_VALU), wait_share = SQ_WAIT_ANY / SQ_WAVE_CYCLES" % cmd}
try:
    rows = list(csv.DictReader(open(stats_csv)))
except Exception:
    rows = []
for key, tot in per.items():
    cyc = tot.get("GRBM_GUI_ACTIVE", (0, 0.0))[1] / 8.0
    e = {"launches": tot.get("SQ_INSTS_VALU", (0, 0))[0],
         "pipe_busy": 4.0 * tot["SQ_ACTIVE_INST_VALU"][1] / (1024.0 * cyc) if cyc and "SQ_ACTIVE_INST_VALU" in tot else None,
         "lane_efficiency": tot["SQ_THREAD_CYCLES_VALU"][1] / (64.0 * tot["SQ_INSTS_VALU"][1]) if "SQ_THREAD_CYCLES_VALU" in tot else None,
         "wait_share": tot["SQ_WAIT_ANY"][1] / tot["SQ_WAVE_CYCLES"][1] if "SQ_WAIT_ANY" in tot and tot.get("SQ_WAVE_CYCLES", (0, 0))[1] else None,
         "valu_wave_instructions": tot.get("SQ_INSTS_VALU", (0, None))[1], "vmem_read_instructions": tot.get("SQ_INSTS_VMEM_RD", (0, None))[1],
         "lds_conflict_ratio": tot["SQ_LDS_BANK_CONFLICT"][1] / tot["SQ_ACTIVE_INST_LDS"][1] if tot.get("SQ_ACTIVE_INST_LDS", (0, 0))[1] else None}
    for r in rows:
        m = re.search(r"walk3_kernel<(\w+), (\w+), (\w+)(?:, (\w+))?>", r["Name"])
        if m and (m.group(3) == "true") == key.startswith("neumann"):
            e["avg_launch_ms"] = float(r["AverageNs"]) / 1e6
            e["wave_pools"] = m.group(4) == "true"
    out["scenes"][key] = e
json.dump(out, open(out_path, "w"), indent=1)
print(json.dumps(out))
