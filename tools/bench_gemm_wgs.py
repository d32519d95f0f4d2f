#!/usr/bin/env python3
"""Chip-level tile throughput of the persistent block GEMM against the number of resident workgroups (medp_gemm_persistent_cap):
is an idle CU in the last round lost throughput, or is the GEMM bound by what the chip delivers to ALL CUs together (L2 / Infinity
Cache -> LDS), so that fewer workgroups each run faster?  Shapes: many full tiles (M 16384 x N 3072 = 768, N 6144 = 1536) so that
rounds quantise little; K = 768 and 3072."""
import sys, os
os.environ["MEDP_V7_FEWEST_WGS"] = "0"          # always `cap` workgroups (read once, at the first launch)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimodal_edema_prediction_amd import functional as Fn
from multimodal_edema_prediction_amd.abi import lib
dev = "cuda"; R = 3
def timeit(fn, n=40):
    for i in range(6): fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n): fn(i)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for M, N, K in ((16384, 6144, 768), (16384, 3072, 768), (16448, 2304, 768), (16448, 3072, 768), (16384, 3072, 3072)):
    a = [torch.randn(M, K, device=dev).bfloat16() for _ in range(R)]; w = torch.randn(N, K, device=dev).bfloat16()
    bias = torch.randn(N, device=dev); out = [torch.empty(M, N, device=dev, dtype=torch.bfloat16) for _ in range(R)]
    tiles = ((M + 255) // 256) * ((N + 255) // 256)
    row = []
    for cap in (256, 240, 224, 208, 200, 192, 176, 160, 128):
        lib().medp_gemm_persistent_cap(cap)
        t = timeit(lambda i: Fn.gemm(a[i % R], w, bias=bias, out=out[i % R]))
        row.append(f"{cap}: {t:6.1f} us {tiles / t:5.2f} t/us")
    lib().medp_gemm_persistent_cap(0)
    print(f"M={M} N={N} K={K} tiles {tiles}: " + " | ".join(row), flush=True)
