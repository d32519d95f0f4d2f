#!/usr/bin/env python3
"""From a rocprofv3 kernel-trace CSV of bench.py: per kernel (name, grid) the launches per step, mean duration and a crude
CU-time estimate = duration x min(1, workgroups / 256) — which kernels of the step's side branches cost the most GPU time."""
import csv, glob, sys, collections
path = sys.argv[1]; steps = float(sys.argv[2]) if len(sys.argv) > 2 else 28.0
acc = collections.defaultdict(lambda: [0, 0.0, 0.0])
for f in glob.glob(path + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        g = int(r.get("Grid_Size", r.get("Grid_Size_X", "0")) or 0); wgs_threads = int(r.get("Workgroup_Size", r.get("Workgroup_Size_X", "256")) or 256)
        wgs = max(1, g // max(1, wgs_threads))
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        key = (name[:70], wgs, wgs_threads)
        a = acc[key]; a[0] += 1; a[1] += d; a[2] += d * min(1.0, wgs / 256.0)
tot = sum(a[2] for a in acc.values())
print(f"{'n/step':>7} {'avg_us':>8} {'wgs':>6} {'thr':>4} {'cu_us/step':>11} {'pct':>5}  name")
for key, a in sorted(acc.items(), key=lambda kv: -kv[1][2])[:45]:
    print(f"{a[0]/steps:7.1f} {a[1]/a[0]:8.1f} {key[1]:6d} {key[2]:4d} {a[2]/steps:11.1f} {100*a[2]/tot:5.1f}  {key[0]}")
