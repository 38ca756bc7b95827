"""print selected rows of a rocprofv3 kernel_stats.csv: name-substring filters as arguments"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    if len(sys.argv) <= 2 or any(k in r["Name"] for k in sys.argv[2:]):
        print("%-64s calls %5s avg %10.1f us  %5.1f %%" % (r["Name"][:64], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
