#!/usr/bin/env python3
"""Per-queue busy spans of one mid-run step from a rocprofv3 kernel-trace CSV (steps are delimited by adamw_multi_kernel)."""
import csv, glob, sys, collections
path = sys.argv[1]
rows = []
for f in glob.glob(path + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")))
rows.sort()
ends = [r[1] for r in rows if "adamw_multi_kernel" in r[2]]
a, b = ends[-5], ends[-4]
step = [r for r in rows if a < r[0] <= b or (r[0] <= a < r[1])]
print(f"step window {(b-a)/1e3:.1f} us, {len(step)} kernels overlap it")
byq = collections.defaultdict(list)
for r in step: byq[r[3]].append(r)
for q, rs in sorted(byq.items()):
    busy = sum(min(e, b) - max(s, a) for s, e, n, _ in rs)
    first, last = min(max(s, a) for s, e, n, _ in rs), max(min(e, b) for s, e, n, _ in rs)
    top = collections.Counter()
    for s, e, n, _ in rs: top[n[:48]] += e - s
    print(f"queue {q}: {len(rs):4d} kernels, busy {busy/1e3:8.1f} us, active from +{(first-a)/1e3:7.1f} to +{(last-a)/1e3:7.1f}; top: " +
          ", ".join(f"{k} {v/1e3:.0f}us" for k, v in top.most_common(3)))
