#!/usr/bin/env python3
"""Timing-only ablation of the persistent block GEMM (python -m multimodal_edema_prediction_amd.build --ablate builds the libraries):
   MEDP_HIP_LIB=.../libmedp_hip_nomfma.so   K-loop with its staging stream, LDS fragment reads, barriers, epilogue — no MFMA
   MEDP_HIP_LIB=.../libmedp_hip_noloads.so  K-loop with MFMA, LDS reads, barriers, epilogue — no global -> LDS staging after the prologue
against the product library.  RESULTS OF THE ABLATED BUILDS ARE WRONG; only the times mean something: which side binds the K-loop."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimodal_edema_prediction_amd import functional as Fn
dev = "cuda"; R = 3
def timeit(fn, n=40):
    for i in range(6): fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n): fn(i)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
print("library:", os.environ.get("MEDP_HIP_LIB", "product"))
for M, N, K in ((16448, 2304, 768), (16448, 3072, 768), (16384, 3072, 768), (16384, 3072, 3072)):
    a = [torch.randn(M, K, device=dev).bfloat16() for _ in range(R)]; w = torch.randn(N, K, device=dev).bfloat16()
    bias = torch.randn(N, device=dev); out = [torch.empty(M, N, device=dev, dtype=torch.bfloat16) for _ in range(R)]
    t = timeit(lambda i: Fn.gemm(a[i % R], w, bias=bias, out=out[i % R]))
    print(f"M={M} N={N} K={K}: {t:7.1f} us", flush=True)
