#!/usr/bin/env python3
"""Split-K sweep of DuETT's skinny GEMMs (forward shapes of both axes at cfg3, B 64): MEDP_GEMM_SPLITK is read once per process, so
this script is run once per value:   for s in 0 1 2 3 4 6 8; do MEDP_GEMM_SPLITK=$s python tools/bench_gemm_splitk.py; done
(0 = the heuristic of splitk_slices).  torch.matmul (hipBLASLt) beside each."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimodal_edema_prediction_amd import functional as Fn
from multimodal_edema_prediction_amd.abi import lib
dev = "cuda"; R = 3
def timeit(fn, n=50):
    for i in range(6): fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n): fn(i)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
S = os.environ.get("MEDP_GEMM_SPLITK", "heuristic")
shapes = [("event qkv", 3136, 72, 2328), ("event out", 3136, 2328, 24), ("event ff1", 3136, 512, 2328), ("event ff2", 3136, 2328, 512),
          ("time  qkv", 6208, 72, 1176), ("time  out", 6208, 1176, 24), ("time  ff1", 6208, 512, 1176), ("time  ff2", 6208, 1176, 512),
          ("img_proj", 16384, 256, 768), ("ts_proj", 6144, 256, 1176)]
row = []
for name, M, N, K in shapes:
    a = [torch.randn(M, K, device=dev).bfloat16() for _ in range(R)]; w = torch.randn(N, K, device=dev).bfloat16()
    out = [torch.empty(M, N, device=dev) for _ in range(R)]
    wsb = lib().medp_gemm_nt_workspace_bytes(M, N, K)
    t = timeit(lambda i: Fn.gemm(a[i % R], w, out=out[i % R]))
    tb = timeit(lambda i: torch.matmul(a[i % R], w.t()))
    row.append(f"{name} {t:5.1f} (lib {tb:5.1f}, ws {wsb >> 20} MB)")
print(f"SPLITK={S}: " + " | ".join(row), flush=True)
