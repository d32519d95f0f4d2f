#!/usr/bin/env python3
"""From a rocprofv3 kernel-trace CSV: for the LAST complete step, list the kernels that start after the final CXR-encoder GEMM
(the serial tail of the two-stream step) with their start offset, duration and queue."""
import csv, glob, sys
path = sys.argv[1]
rows = []
for f in glob.glob(path + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")))
rows.sort()
# steps end with adamw_multi_kernel
ends = [i for i, r in enumerate(rows) if "adamw_multi_kernel" in r[2]]
i1 = ends[-3]; i0 = ends[-4] + 1          # a replay in the middle of the timed region
step = rows[i0:i1 + 1]
t0 = step[0][0]
last_vit = max(i for i, r in enumerate(step) if "gemm_bf16_nt_v6_kernel<1>" in r[2] or "layernorm_fwd_reg_kernel<true, 3>" in r[2])
print(f"step span {(step[-1][1]-t0)/1e3:.1f} us, {len(step)} kernels; CXR encoder ends at {(step[last_vit][1]-t0)/1e3:.1f} us")
busy = 0
for s, e, n, q in step[last_vit + 1:]:
    busy += e - s
tail = step[-1][1] - step[last_vit][1]
print(f"tail {tail/1e3:.1f} us, sum of kernel time in tail {busy/1e3:.1f} us, {len(step)-last_vit-1} kernels")
import collections
agg = collections.defaultdict(lambda: [0, 0])
for s, e, n, q in step[last_vit + 1:]:
    k = n[:70]; agg[k][0] += 1; agg[k][1] += e - s
for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:25]:
    print(f"{c:4d} {t/1e3:8.1f} us  {k}")
if "--list" in sys.argv:
    for s, e, n, q in step[last_vit + 1:]:
        print(f"{(s-t0)/1e3:9.1f} +{(e-s)/1e3:7.1f} q{q} {n[:90]}")
