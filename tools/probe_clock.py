#!/usr/bin/env python3
"""Is the GEMM loop power/clock limited?  Runs a GEMM back to back for a few seconds and samples rocm-smi (sclk, power);
then times the same launch with idle gaps in between (cool chip)."""
import sys, os, subprocess, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimodal_edema_prediction_amd import functional as Fn
dev = "cuda"; M, D, F = 64 * 257, 768, 3072
a = torch.randn(M, D, device=dev).bfloat16(); w = torch.randn(F, D, device=dev).bfloat16(); bias = torch.randn(F, device=dev)
out = torch.empty(M, F, device=dev, dtype=torch.bfloat16)
def smi(tag):
    for cmd in (["rocm-smi", "--showclocks", "--showpower"], ["amd-smi", "metric", "-c", "-p"]):
        try:
            r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=20)
            lines = [l for l in r.stdout.splitlines() if any(k in l.lower() for k in ("sclk", "power", "gfx", "clk", "socket"))]
            print(f"--- {tag}: {' '.join(cmd)} rc={r.returncode}"); print("\n".join(lines[:24]), flush=True)
            if r.returncode == 0: return
        except Exception as e:
            print(tag, cmd, "failed:", e, flush=True)
smi("idle")
stop = False
def loop():
    while not stop:
        for _ in range(200): Fn.gemm(a, w, bias=bias, act=0, out=out)
        torch.cuda.synchronize()
t = threading.Thread(target=loop); t.start()
time.sleep(1.5); smi("under load (1.5 s)"); time.sleep(1.0); smi("under load (2.5+ s)")
stop = True; t.join()
# timing: back to back vs spaced
def timed(n, gap):
    ts = []
    for i in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); Fn.gemm(a, w, bias=bias, act=0, out=out); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
        if gap: time.sleep(gap)
    ts.sort(); return ts[len(ts) // 2], ts[0]
for _ in range(300): Fn.gemm(a, w, bias=bias, act=0, out=out)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(2000): Fn.gemm(a, w, bias=bias, act=0, out=out)
e1.record(); torch.cuda.synchronize()
print(f"back to back, 2000 launches: {e0.elapsed_time(e1) / 2000 * 1e3:.1f} us each")
time.sleep(2.0)
print("single launches after a 20 ms gap: median %.1f us, min %.1f us" % timed(40, 0.02))
print("single launches, no gap (sync each): median %.1f us, min %.1f us" % timed(40, 0))
