"""Developer tool: per-depth GPU time of the guided walk phase from a rocprofv3 kernel trace
(kernel_trace.csv of `rocprofv3 --kernel-trace -- python3 tools/gpu_guided_bench.py --spp 4 --train-spp 0`)."""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
depth = -1
acc = defaultdict(lambda: defaultdict(float))
gaps = defaultdict(float)
prev_end = None
for r in rows:
    name = r["Kernel_Name"]
    dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    if "begin_sample_kernel" in name:
        depth = -1
    if "separate_kernel" in name:
        depth += 1
    key = "separate" if "separate_kernel" in name else "sample" if "sample_kernel" in name else "net" if "net_forward" in name else None
    if key and depth >= 0:
        acc[depth][key] += dur
        if prev_end is not None:
            gaps[depth] += (int(r["Start_Timestamp"]) - prev_end) / 1e3
    prev_end = int(r["End_Timestamp"])
tot = 0.0
print("depth  separate   net    sample   idle-gap   (us, summed over samples)")
for d in sorted(acc):
    a = acc[d]
    s = a["separate"] + a["net"] + a["sample"]
    tot += s + gaps[d]
    if d < 14 or d % 8 == 0:
        print("%4d  %8.0f %7.0f %8.0f %9.0f" % (d, a["separate"], a["net"], a["sample"], gaps[d]))
print("kernels+gaps %.1f ms; depth>=10: kernels %.1f ms gaps %.1f ms" % (
    tot / 1e3, sum(sum(acc[d].values()) for d in acc if d >= 10) / 1e3, sum(gaps[d] for d in gaps if d >= 10) / 1e3))
