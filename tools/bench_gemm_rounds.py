#!/usr/bin/env python3
"""Tile-quantisation screen of the block GEMMs (K = 768, bf16 out + bias): the same kernels at shapes whose 256 x 256 tile counts
are exact multiples of 256 (full rounds) against the step's shapes (585 / 780 / 195 tiles), with torch.matmul (hipBLASLt, stream-K)
beside them.  Upper bound of what balancing the last round can buy, before any fix-up cost."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimodal_edema_prediction_amd import functional as Fn
dev = "cuda"
R = 4
def timeit(fn, n=60):
    for i in range(8): fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n): fn(i)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
shapes = [(16384, 1024, 768), (16384, 2048, 768), (16384, 3072, 768), (16384, 2304, 768), (16448, 2304, 768), (16448, 3072, 768),
          (16448, 768, 768), (16384, 768, 768), (16448, 768, 3072), (16384, 1024, 3072), (16384, 768, 3072)]
for M, N, K in shapes:
    a = [torch.randn(M, K, device=dev).bfloat16() for _ in range(R)]; w = torch.randn(N, K, device=dev).bfloat16()
    bias = torch.randn(N, device=dev); out = [torch.empty(M, N, device=dev, dtype=torch.bfloat16) for _ in range(R)]
    t = timeit(lambda i: Fn.gemm(a[i % R], w, bias=bias, out=out[i % R]))
    wt = w.t()
    tb = timeit(lambda i: torch.matmul(a[i % R], wt, out=out[i % R]))
    tiles = ((M + 255) // 256) * ((N + 255) // 256)
    print(f"M={M} N={N} K={K}: tiles {tiles:4d} ({tiles / 256:.2f} rounds)  ours {t:6.1f} us ({2*M*N*K/t/1e6:6.0f} TF)  per round-tile {t / -(-tiles // 256):5.1f} us   hipBLASLt {tb:6.1f} us ({2*M*N*K/tb/1e6:6.0f} TF)", flush=True)
