#!/usr/bin/env python3
"""Cost of the fix-up a stream-K / split-K balancing of the 256 x 256 block GEMMs would add (csrc/dbg_fixup_bench.hip), at the
partial counts the step's shapes would produce, next to the measured upper bound of what balancing can save
(tools/bench_gemm_wgs.py: the chip's tile throughput is flat from ~192 resident workgroups on)."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimodal_edema_prediction_amd.abi import lib, ptr, stream
L = lib()
L.medp_dbg_splitk_fixup.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
dev = "cuda"
def timeit(fn, n=40):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
# (shape, partial slabs written, owner tiles, slabs an owner reads) under an even split of the remainder tiles' K-tiles over 256 workgroups
cases = [("qkv  585 tiles: 73 remainder tiles x 12 K-tiles over 256 workgroups (3.4 each)", 256 - 73, 73, 3),
         ("fc1  780 tiles: 12 ragged remainder tiles x 12 K-tiles (1 each over 144 workgroups)", 144 - 12, 12, 11),
         ("proj 195 tiles x 12 K-tiles over 256 workgroups (9.1 each)", 255, 195, 1),
         ("fc2  195 tiles x 48 K-tiles over 256 workgroups (36.6 each)", 255, 195, 1),
         ("2-way split of 73 tiles (146 workgroups, pairwise)", 73, 73, 1)]
ws = [torch.empty(256 * 65536, device=dev) for _ in range(3)]           # 3 x 64 MiB: rotate so that the slabs are not all cache-resident
out = torch.empty(256 * 65536, device=dev, dtype=torch.bfloat16)
i = [0]
for name, writers, owners, per in cases:
    def run():
        i[0] += 1
        L.medp_dbg_splitk_fixup(ptr(ws[i[0] % 3]), ptr(out), writers, owners, per, stream())
    t = timeit(run)
    mb = (writers * 0.262144 + owners * per * 0.262144 + owners * 0.131072)
    print(f"{name}: {t:6.1f} us for {writers} slabs written + {owners} x {per} read + {owners} bf16 tiles ({mb:.0f} MB moved)", flush=True)
