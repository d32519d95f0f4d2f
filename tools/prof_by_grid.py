#!/usr/bin/env python3
"""Per (kernel, grid) durations from a rocprofv3 --kernel-trace CSV: separates the GEMM shapes that share one kernel symbol."""
import csv, glob, sys, collections
path, pat = sys.argv[1], sys.argv[2]
rows = collections.defaultdict(list)
for f in glob.glob(path + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if pat not in r["Kernel_Name"]: continue
        g = r.get("Grid_Size") or r.get("Grid_Size_X") or "?"
        rows[(r["Kernel_Name"][:70], g)].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for (k, g), v in sorted(rows.items()):
    v.sort()
    print(f"{k:70s} grid={g:>8s} n={len(v):5d} mean={sum(v)/len(v)/1e3:8.2f}us p10={v[len(v)//10]/1e3:8.2f} p50={v[len(v)//2]/1e3:8.2f} p90={v[len(v)*9//10]/1e3:8.2f}")
