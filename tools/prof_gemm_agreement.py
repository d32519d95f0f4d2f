#!/usr/bin/env python3
"""From a rocprofv3 kernel-trace CSV of `bench.py`: mean duration of the CXR-encoder block GEMMs (v6<1> + v7<1>) over (a) all
launches, (b) the LAST n launches — bench.py measures its roofline figure with HIP events around exactly those (the eager
encoder passes it issues right after the timed region), so (b) is the number that must agree with `roofline.avg_launch_us`."""
import csv, glob, sys
path = sys.argv[1]; n = int(sys.argv[2]) if len(sys.argv) > 2 else 240
rows = []
for f in glob.glob(path + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "gemm_bf16_nt_v6_kernel<1>" in k or "gemm_bf16_nt_v7_kernel<1" in k:
            rows.append((int(r["Start_Timestamp"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
rows.sort()
d = [x[1] for x in rows]
print(f"# CXR-encoder block GEMMs (v6<1> + v7<1>): {len(d)} launches, mean {sum(d)/len(d):.2f} us; last {n} launches (the eager passes bench.py "
      f"times with HIP events): mean {sum(d[-n:])/n:.2f} us")
