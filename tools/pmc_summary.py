#!/usr/bin/env python3
"""Sum a rocprofv3 --pmc counter per kernel from the counter_collection CSV: prints per-kernel dispatches, total and mean."""
import csv, glob, sys, collections
path, counter = sys.argv[1], sys.argv[2]
files = glob.glob(path + "/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(lambda: [0, 0.0])
for f in files:
    for r in csv.DictReader(open(f)):
        if r.get("Counter_Name") != counter: continue
        k = r["Kernel_Name"]; acc[k][0] += 1; acc[k][1] += float(r["Counter_Value"])
for k, (n, v) in sorted(acc.items(), key=lambda kv: -kv[1][1])[:int(sys.argv[3]) if len(sys.argv) > 3 else 12]:
    print(f"{n:7d} {v:16.1f} {v/n:14.2f}  {k[:120]}")
