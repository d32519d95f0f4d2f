#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace CSV: per-kernel count / total / average duration, sorted by total time."""
import csv, glob, sys, collections
path = sys.argv[1]
files = glob.glob(path + "/**/*kernel_trace.csv", recursive=True)
rows = collections.defaultdict(lambda: [0, 0.0])
tmin, tmax = None, None
for f in files:
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]; s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        rows[name][0] += 1; rows[name][1] += (e - s)
        tmin = s if tmin is None else min(tmin, s); tmax = e if tmax is None else max(tmax, e)
tot = sum(v[1] for v in rows.values())
print(f"# files={len(files)} kernels={sum(v[0] for v in rows.values())} total_kernel_ms={tot/1e6:.3f} span_ms={(tmax-tmin)/1e6:.3f}")
print(f"{'count':>8} {'total_ms':>10} {'avg_us':>10} {'pct':>6}  name")
for name, (c, t) in sorted(rows.items(), key=lambda kv: -kv[1][1])[:int(sys.argv[2]) if len(sys.argv) > 2 else 60]:
    print(f"{c:8d} {t/1e6:10.3f} {t/c/1e3:10.2f} {100*t/tot:6.2f}  {name[:150]}")
