#!/usr/bin/env python3
"""Sum rocprofv3 --pmc counter_collection.csv per kernel and counter."""
import csv
import sys
from collections import defaultdict

tot = defaultdict(float)
calls = defaultdict(int)
with open(sys.argv[1]) as f:
    for row in csv.DictReader(f):
        k = (row["Kernel_Name"][:60], row["Counter_Name"])
        tot[k] += float(row["Counter_Value"])
        calls[k] += 1
for (k, c), v in sorted(tot.items()):
    if "walk_round" in k or "init_kernel" in k:
        print("%-62s %-32s calls=%-5d sum=%.6g avg=%.6g" % (k, c, calls[(k, c)], v, v / calls[(k, c)]))
