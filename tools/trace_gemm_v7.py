#!/usr/bin/env python3
"""Per-workgroup timeline of the persistent GEMM (debug hook medp_dbg_gemm_v7_trace): for every tile of every workgroup the
wall-clock stamps {tile top, K-loop done, epilogue issued, next tile landed}.  Prints per-phase statistics per tile ordinal."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from multimodal_edema_prediction_amd import functional as Fn
from multimodal_edema_prediction_amd.abi import lib
dev = "cuda"; M, D = 64 * 257, 768
N = int(sys.argv[1]) if len(sys.argv) > 1 else 3072
act = int(sys.argv[2]) if len(sys.argv) > 2 else 0
Mx = int(sys.argv[3]) if len(sys.argv) > 3 else M
a = [torch.randn(Mx, D, device=dev).bfloat16() for _ in range(4)]; w = torch.randn(N, D, device=dev).bfloat16(); bias = torch.randn(N, device=dev)
out = [torch.empty(Mx, N, device=dev, dtype=torch.bfloat16) for _ in range(4)]
L = lib()
L.medp_dbg_gemm_v7_trace.argtypes = [ctypes.c_void_p]
buf = torch.zeros(256 * 40 + 256 * 8 * 16, dtype=torch.int64, device=dev)
for i in range(20): Fn.gemm(a[i % 4], w, bias=bias, act=act, out=out[i % 4])
torch.cuda.synchronize()
L.medp_dbg_gemm_v7_trace(buf.data_ptr())
for i in range(4): Fn.gemm(a[i % 4], w, bias=bias, act=act, out=out[i % 4])
torch.cuda.synchronize()
L.medp_dbg_gemm_v7_trace(None)
raw = buf.cpu().numpy().astype(np.int64)
t = raw[:256 * 40].reshape(256, 8, 5)
ph = raw[256 * 40:].reshape(256, 8, 16)
t0 = t[:, 0, 1][t[:, 0, 1] > 0].min()
print(f"N={N} act={act} M={Mx}; times in us relative to the first tile top; 10-ns ticks")
for k in range(8):
    live = t[:, k, 1] > 0
    if not live.any(): break
    top, kdone, epi, landed = [(t[live, k, c] - t0) / 100.0 for c in (1, 2, 3, 4)]
    m0 = t[live, k, 0] >> 32
    rag = (m0 + 256 > Mx)
    print(f"tile #{k}: {live.sum():3d} WGs ({rag.sum()} ragged)  top {top.mean():6.1f} (min {top.min():6.1f} max {top.max():6.1f})  "
          f"K-loop {np.mean(kdone - top):5.2f} (max {np.max(kdone - top):5.2f})  epilogue {np.mean(epi - kdone):5.2f}  "
          f"end wait {np.mean(landed - epi):5.2f}  end {landed.mean():6.1f} (max {landed.max():6.1f})")
    if rag.any():
        print(f"          ragged only: K-loop {np.mean((kdone - top)[rag]):5.2f}  epilogue {np.mean((epi - kdone)[rag]):5.2f}  end wait {np.mean((landed - epi)[rag]):5.2f}")
# per XCD end time
last = np.array([t[b][t[b, :, 4] > 0][-1, 4] if (t[b, :, 4] > 0).any() else t0 for b in range(256)])     # (fewer than 256 workgroups may run)
print("per-XCD last end:", " ".join(f"{(last[x::8].max() - t0) / 100.0:6.1f}" for x in range(8)))
print("per-XCD tiles   :", " ".join(f"{(t[x::8, :, 1] > 0).sum():6d}" for x in range(8)))

if ph.any():
    names = ["load", "bar1", "mfma", "bar2"]
    nk = D // 64
    ntile = (t[:, :, 1] > 0).sum(axis=1)                     # tiles per workgroup
    ran = ntile > 0
    per = ph[ran] / (ntile[ran, None, None] * nk)            # ticks per K-tile
    for grp, waves in (("group 0 (waves 0-3)", slice(0, 4)), ("group 1 (waves 4-7)", slice(4, 8))):
        m = per[:, waves, :].mean(axis=(0, 1))
        print(grp + ": ticks per K-tile  " + "  ".join(f"P{k // 4 + 1}.{names[k % 4]} {m[k]:6.1f}" for k in range(16)) + f"   sum {m.sum():7.1f}")
