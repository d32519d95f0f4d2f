#!/usr/bin/env python3
import csv, glob, sys, collections
path = sys.argv[1]; pat = sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for f in glob.glob(path + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if pat not in r["Kernel_Name"]: continue
        key = (r["Kernel_Name"][:60], r.get("Grid_Size", ""))
        a = acc[key][r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
for key, cs in acc.items():
    print(key)
    for c, (n, v) in sorted(cs.items()): print(f"   {c:32s} n={n:4d} mean={v/n:16.1f}")
