#!/usr/bin/env python3
"""Sum rocprofv3 --pmc counter_collection.csv per kernel and counter.
Usage: pmc_summary.py counter_collection.csv [kernel-name substrings; default: the walk kernels]"""
import csv
import sys
from collections import defaultdict

wanted = sys.argv[2:] or ["walk_round", "walk_cells", "init_kernel"]
tot = defaultdict(float)
calls = defaultdict(int)
with open(sys.argv[1]) as f:
    for row in csv.DictReader(f):
        k = (row["Kernel_Name"][:int(__import__("os").environ.get("PMC_NAME_WIDTH", "60"))], row["Counter_Name"])
        tot[k] += float(row["Counter_Value"])
        calls[k] += 1
for (k, c), v in sorted(tot.items()):
    if any(w in k for w in wanted):
        print("%-62s %-32s calls=%-5d sum=%.6g avg=%.6g" % (k, c, calls[(k, c)], v, v / calls[(k, c)]))
