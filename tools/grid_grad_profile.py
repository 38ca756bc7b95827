import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
d=[(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3 for r in rows if "grid_grad" in r["Kernel_Name"]]
import collections
acc=collections.defaultdict(list)
for i,x in enumerate(d): acc[i%5].append(x)
for k in sorted(acc): print("launch %d of a batch: avg %.1f us (n=%d)"%(k,sum(acc[k])/len(acc[k]),len(acc[k])))
