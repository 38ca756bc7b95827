#!/usr/bin/env python3
"""Derive the per-launch HBM traffic and the VALU figures of walk_round_kernel from a PMC summary
(tools/pmc_summary.py output) and the JSON line of the profiled bench run.
Usage: pmc_derive.py pmc_summary.txt bench_trace.log out_dir "<bench args>"
Writes out_dir/walk_round_traffic.json and out_dir/walk_round_valu.json (copied to profiles/ by hand)."""
import json
import os
import re
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from elaina_amd.build import source_id  # noqa: E402

summary, bench_log, out_dir, bargs = sys.argv[1:5]
tot = {}
for line in open(summary):
    m = re.search(r"walk_(?:round|cells)_kernel.*?(\w+)\s+calls=(\d+)\s+sum=([0-9.e+]+)", line)
    # the ordinary instantiation: the SLACK one (a handful of strayed walkers per solve) has the same name up to its last
    # template argument and a millionth of the counts
    if m and (m.group(1) not in tot or float(m.group(3)) > tot[m.group(1)][1]):
        tot[m.group(1)] = (int(m.group(2)), float(m.group(3)))
# the counters above are those of the instantiation with the largest sums: since round 6 the persistent first launch of the solve
# (one call per pass); the walk steps THAT launch took come from the bench line of the traced run (roofline.launch.walk_steps;
# the launch ends when the pixel queue runs dry, so its share of the pass varies by a fraction of a percent between runs)
steps = None
try:
    for line in open(bench_log):
        if line.startswith("{"):
            j = json.loads(line)
            steps = j["config"]["walk_steps_per_pass"]
            if j.get("roofline", {}).get("launch"):
                steps = j["roofline"]["launch"]["walk_steps"]
except Exception:
    pass
if "FETCH_SIZE" in tot and "WRITE_SIZE" in tot:
    calls = tot["FETCH_SIZE"][0]
    # MI355X_MICROARCH.md "HBM": FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports
    # half of the bytes of 16-B-per-lane reads -> doubled; WRITE_SIZE is exact
    per_launch = (2.0 * tot["FETCH_SIZE"][1] + tot["WRITE_SIZE"][1]) * 1024.0 / calls
    json.dump({"kernel": "walk_round_kernel", "source_id": source_id(), "launches": calls, "fetch_kib_sum": tot["FETCH_SIZE"][1],
               "write_kib_sum": tot["WRITE_SIZE"][1], "hbm_bytes_per_launch": per_launch,
               "command": "rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE -- python3 " + bargs,
               "correction": "2*FETCH_SIZE + WRITE_SIZE, KiB -> bytes"},
              open(out_dir + "/walk_round_traffic.json", "w"), indent=1)
need = ("SQ_ACTIVE_INST_VALU", "SQ_INSTS_VALU", "SQ_THREAD_CYCLES_VALU", "GRBM_GUI_ACTIVE")
if all(k in tot for k in need):
    # SQ_ACTIVE_INST_VALU counts quad-cycles summed over the SIMDs (MI355X_MICROARCH.md, cycle
    # constants); GRBM_GUI_ACTIVE is the sum of the kernel's cycles over the 8 XCDs; 256 CUs x 4 SIMDs
    kernel_cycles = tot["GRBM_GUI_ACTIVE"][1] / 8.0
    pipe_busy = 4.0 * tot["SQ_ACTIVE_INST_VALU"][1] / (1024.0 * kernel_cycles)
    lane_eff = tot["SQ_THREAD_CYCLES_VALU"][1] / (64.0 * tot["SQ_INSTS_VALU"][1])
    out = {"kernel": "walk_round_kernel", "source_id": source_id(), "pipe_busy": pipe_busy, "lane_efficiency": lane_eff,
           "valu_wave_instructions": tot["SQ_INSTS_VALU"][1], "salu_wave_instructions": tot.get("SQ_INSTS_SALU", (0, None))[1],
           "walk_steps": steps,
           "lane_instr_per_step": (tot["SQ_INSTS_VALU"][1] * 64.0 * lane_eff / steps) if steps else None,
           "source": "rocprofv3 --pmc passes of `python3 %s` (tools/gpu_round.sh): pipe_busy = 4 SQ_ACTIVE_INST_VALU / "
                     "(1024 SIMDs x GRBM_GUI_ACTIVE / 8), lane_efficiency = SQ_THREAD_CYCLES_VALU / (64 SQ_INSTS_VALU), "
                     "lane_instr_per_step = active lane-instructions / walk steps" % bargs}
    json.dump(out, open(out_dir + "/walk_round_valu.json", "w"), indent=1)
    print(json.dumps(out))
