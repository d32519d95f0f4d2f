#!/usr/bin/env python3
"""The kernels of ONE step in launch order, from a `rocprofv3 --kernel-trace` CSV:  prof_sequence.py DIR [DELIM] [SKIP...]
The trace is cut at every dispatch whose name contains DELIM (default: the optimiser's `adamw_multi_kernel`, the last launch of a
training step) and the last complete segment is printed: index, queue, start offset (us), duration (us), grid x workgroup, short name.
Names containing one of SKIP are counted but not listed (e.g. the frozen encoder's kernels, to read the training branch alone)."""
import csv, glob, re, sys, collections
path = sys.argv[1]
delim = sys.argv[2] if len(sys.argv) > 2 else "adamw_multi_kernel"
skip = sys.argv[3:]
rows = []
for f in glob.glob(path + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?"),
                     r.get("Grid_Size", r.get("Grid_Size_X", "?")), r.get("Workgroup_Size", r.get("Workgroup_Size_X", "?"))))
rows.sort()
cuts = [i for i, r in enumerate(rows) if delim in r[2]]
if len(cuts) < 2:
    sys.exit(f"fewer than two dispatches of {delim!r} in the trace")
seg = rows[cuts[-2] + 1: cuts[-1] + 1]


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    n = re.sub(r"at::native::", "", n)
    return n[:110]


t0 = seg[0][0]
hidden = collections.Counter()
print(f"# {len(seg)} dispatches between the last two {delim}; span {(seg[-1][1] - t0) / 1e3:.0f} us, kernel time {sum(e - s for s, e, *_ in seg) / 1e3:.0f} us")
for i, (s, e, n, q, g, w) in enumerate(seg):
    hit = next((k for k in skip if k in n), None)
    if hit:
        hidden[hit] += 1
        continue
    print(f"{i:4d} q{q:>2} {(s - t0) / 1e3:9.1f} {(e - s) / 1e3:7.1f}  {g:>9}x{w:<5} {short(n)}")
for k, c in hidden.items():
    print(f"# not listed: {c} x {k}")
