#!/usr/bin/env python3
"""Reductions bench.py reads back (profiles/rocprof_gemm.json, profiles/traffic.json), written from ONE rocprofv3 run each:

  python3 tools/make_profile_json.py gemm <kernel-trace dir> <bench line file> <out json> <tag>
      mean duration of the CXR-encoder block GEMMs (gemm_bf16_nt_v6_kernel<1> + gemm_bf16_nt_v7_kernel<1>) over every launch of
      the trace, next to the in-kernel launch clocks the SAME run's bench line carries (roofline.avg_launch_us): the two numbers
      of `roofline.frac` / `roofline.frac_rocprof` side by side, with their gap.
  python3 tools/make_profile_json.py traffic <FETCH_SIZE pmc dir> <WRITE_SIZE pmc dir> <out json> <tag>
      HBM bytes per launch of the same kernels: FETCH_SIZE doubled (gfx950 reports half the bytes of a 16-B-per-lane stream,
      /opt/skills/guides/MI355X_MICROARCH.md), WRITE_SIZE as is, KiB -> bytes; per kernel and launch-weighted mean."""
import collections, csv, datetime, glob, json, os, subprocess, sys

GEMMS = ("gemm_bf16_nt_v6_kernel<1>", "gemm_bf16_nt_v7_kernel<1")      # (v7 carries a second template argument: <1, false> / <1, true>)


def commit():
    try:
        return subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True, cwd=os.path.dirname(os.path.abspath(__file__))).stdout.strip() or None
    except Exception:
        return None


def gemm(trace_dir, bench_file, out, tag):
    per = collections.defaultdict(list)
    for f in glob.glob(trace_dir + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            for g in GEMMS:
                if g in r["Kernel_Name"]:
                    per[g].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    allv = [x for v in per.values() for x in v]
    line = None
    for ln in open(bench_file):
        if ln.startswith("{") and '"roofline"' in ln:
            line = json.loads(ln)
    doc = {"tag": tag, "kernels": list(GEMMS), "launches": len(allv), "avg_launch_us": round(sum(allv) / max(len(allv), 1), 3),
           "per_kernel_avg_us": {g: round(sum(v) / len(v), 3) for g, v in per.items() if v},
           "source": f"rocprofv3 --kernel-trace --stats of `python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-hbm-table` (profiles/{tag}_kerneltrace_bench_teacher.txt)",
           "commit": commit(), "date": datetime.date.today().isoformat()}
    if line is not None:
        rf = line["roofline"]
        doc["same_run_in_kernel_avg_launch_us"] = rf["avg_launch_us"]
        doc["same_run_flops_per_launch"] = rf["algorithmic_flops_per_launch"]
        doc["gap_us_per_launch"] = round(doc["avg_launch_us"] - rf["avg_launch_us"], 3)
        doc["reading"] = ("rocprofv3 times a dispatch from its queue packet to its completion signal; the in-kernel clocks run from the first "
                          "workgroup's first instruction to the last workgroup's last store: the gap is per-dispatch launch / completion latency "
                          "(under the profiler every dispatch is also serialised: the captured graph's branches do not overlap)")
    json.dump(doc, open(out, "w"), indent=1)
    print(json.dumps(doc))


def traffic(fetch_dir, write_dir, out, tag):
    def per_kernel(path, counter):
        acc = collections.defaultdict(lambda: [0, 0.0])
        for f in glob.glob(path + "/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                if r.get("Counter_Name") != counter:
                    continue
                for g in GEMMS:
                    if g in r["Kernel_Name"]:
                        acc[g][0] += 1
                        acc[g][1] += float(r["Counter_Value"])
        return acc
    fe, wr = per_kernel(fetch_dir, "FETCH_SIZE"), per_kernel(write_dir, "WRITE_SIZE")
    nf, nw = sum(v[0] for v in fe.values()), sum(v[0] for v in wr.values())
    fetch_kib = sum(v[1] for v in fe.values()) / max(nf, 1)
    write_kib = sum(v[1] for v in wr.values()) / max(nw, 1)
    old = {}
    if os.path.exists(out):
        try:
            old = json.load(open(out))
        except Exception:
            old = {}
    hist = old.get("history", [])
    if "vit_gemm_hbm_bytes_per_launch" in old:
        hist.append({"round": old.get("round"), "fetch_size_kib_raw": old.get("fetch_size_kib_per_launch_raw"),
                     "write_size_kib": old.get("write_size_kib_per_launch"), "bytes_per_launch": old.get("vit_gemm_hbm_bytes_per_launch")})
    doc = {"tag": tag, "round": int(tag[1:3]) if tag[1:3].isdigit() else None, "kernel": " + ".join(GEMMS) + " (CXR-encoder block GEMMs, B=64, 224x224: 4 shapes x 12 layers per step)",
           "method": "rocprofv3 --pmc FETCH_SIZE and, in a separate pass, --pmc WRITE_SIZE (no trace domains), `bench.py --steps 3 --warmup 1 --eager "
                     "--no-cpu-baseline --no-hbm-table` (tools/collect_profiles.sh); mean per dispatch; FETCH_SIZE doubled per "
                     "/opt/skills/guides/MI355X_MICROARCH.md (gfx950 reports half the bytes of a 16-B-per-lane stream; LDS-DMA loads alike); counters in KiB",
           "source": f"profiles/{tag}_pmc_fetch_write_bench_teacher.txt", "commit": commit(), "date": datetime.date.today().isoformat(),
           "dispatches": {"FETCH_SIZE": nf, "WRITE_SIZE": nw},
           "fetch_size_kib_per_launch_raw": round(fetch_kib, 2), "write_size_kib_per_launch": round(write_kib, 2),
           "vit_gemm_hbm_bytes_per_launch": int((2 * fetch_kib + write_kib) * 1024),
           "algorithmic_bytes_per_launch": old.get("algorithmic_bytes_per_launch", 142500000), "algorithmic_split": old.get("algorithmic_split"),
           "per_kernel": {g: {"fetch_kib_raw": round(fe[g][1] / max(fe[g][0], 1), 2), "write_kib": round(wr[g][1] / max(wr[g][0], 1), 2),
                              "hbm_bytes_per_launch": int((2 * fe[g][1] / max(fe[g][0], 1) + wr[g][1] / max(wr[g][0], 1)) * 1024)} for g in GEMMS},
           "history": hist}
    json.dump(doc, open(out, "w"), indent=1)
    print(json.dumps({k: v for k, v in doc.items() if k != "history"}))


if __name__ == "__main__":
    {"gemm": gemm, "traffic": traffic}[sys.argv[1]](*sys.argv[2:6])
