#!/usr/bin/env python3
"""Derive the VALU / matrix-pipe / LDS figures of guided_sample_kernel (and the training kernels' shares) from PMC
summaries of a guided bench run.
Usage: pmc_derive_guided.py pmc_summary.txt kernel_stats.csv bench_log out.json "<command>"
pmc_summary.txt: tools/pmc_summary.py output for "guided_sample" (all counter groups concatenated)."""
import csv
import json
import os
import re
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from elaina_amd.build import source_id  # noqa: E402

summary, stats_csv, bench_log, out_path, cmd = sys.argv[1:6]
tot = {}
for line in open(summary):
    m = re.search(r"(\w+)\s+calls=(\d+)\s+sum=([0-9.e+]+)", line)
    if m and "guided_sample" in line:
        tot[m.group(1)] = tot.get(m.group(1), 0.0) + float(m.group(3))
steps = None
try:
    for line in open(bench_log):
        if line.startswith("{"):
            steps = json.loads(line)["config"]["walk_steps_per_pass"]
except Exception:
    pass
share = None
try:
    rows = list(csv.DictReader(open(stats_csv)))
    total = sum(float(r["TotalDurationNs"]) for r in rows)
    mine = sum(float(r["TotalDurationNs"]) for r in rows if "guided_sample" in r["Name"])
    share = mine / total if total else None
    top = sorted(((float(r["TotalDurationNs"]) / total, r["Name"][:70]) for r in rows), reverse=True)[:8]
except Exception:
    top = []
cyc = tot.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
out = {"kernel": "guided_sample_kernel", "source_id": source_id(), "share_of_gpu_time": share,
       "pipe_busy": 4.0 * tot["SQ_ACTIVE_INST_VALU"] / (1024.0 * cyc) if cyc and "SQ_ACTIVE_INST_VALU" in tot else None,
       "lane_efficiency": tot["SQ_THREAD_CYCLES_VALU"] / (64.0 * tot["SQ_INSTS_VALU"]) if "SQ_THREAD_CYCLES_VALU" in tot else None,
       "mfma_busy": tot["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * cyc) if cyc and "SQ_VALU_MFMA_BUSY_CYCLES" in tot else None,
       "lds_conflict_ratio": tot["SQ_LDS_BANK_CONFLICT"] / tot["SQ_ACTIVE_INST_LDS"] if tot.get("SQ_ACTIVE_INST_LDS") else None,
       "walk_steps": steps,
       "lane_instr_per_step": (tot["SQ_THREAD_CYCLES_VALU"] / steps) if steps and "SQ_THREAD_CYCLES_VALU" in tot else None,
       "top_kernels_by_gpu_time": top,
       "source": "rocprofv3 --pmc passes of `%s` (tools/gpu_round.sh, stage pmc_guided): pipe_busy = 4 SQ_ACTIVE_INST_VALU / (1024 SIMDs x "
                 "GRBM_GUI_ACTIVE / 8), lane_efficiency = SQ_THREAD_CYCLES_VALU / (64 SQ_INSTS_VALU), mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / "
                 "(1024 x kernel cycles), lds_conflict_ratio = SQ_LDS_BANK_CONFLICT / SQ_ACTIVE_INST_LDS, lane_instr_per_step = active "
                 "lane-instructions / walk steps" % cmd}
json.dump(out, open(out_path, "w"), indent=1)
print(json.dumps(out))
