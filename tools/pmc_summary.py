#!/usr/bin/env python
"""Mean per-launch value of every counter in rocprofv3 --pmc output (counter_collection.csv files under a dir),
for kernels whose name contains a pattern."""
import csv
import glob
import sys
root, pat = sys.argv[1], sys.argv[2]
agg = {}
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if pat not in r["Kernel_Name"]:
            continue
        k = (r["Counter_Name"], r["Kernel_Name"][:96])
        a = agg.setdefault(k, [0, 0.0])
        a[0] += 1
        a[1] += float(r["Counter_Value"])
for (c, k), (n, s) in sorted(agg.items()):
    print(f"{c:28s} launches={n:4d} mean={s / n:.6g}   {k}")
