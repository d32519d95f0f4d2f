#!/usr/bin/env python3
"""Weight-gradient GEMMs dW[N,K] = dY[M,N]^T X[M,K] of the training-form DuETT at the student step's shapes: medp_gemm_bf16_tn (+ its slab sum)
against torch.matmul (hipBLASLt) on the same bf16 operands, rotating operand sets."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimodal_edema_prediction_amd import functional as Fn

SHAPES = [("event qkv", 3136, 72, 2328), ("event out", 3136, 2328, 24), ("event ff1", 3136, 512, 2328), ("event ff2", 3136, 2328, 512),
          ("time qkv", 6208, 72, 1176), ("time out", 6208, 1176, 24), ("time ff1", 6208, 512, 1176), ("time ff2", 6208, 1176, 512),
          ("img_proj", 16448, 256, 768), ("perceiver", 448, 256, 256)]


def t_us(fn, sets, n=60):
    for i in range(6):
        fn(*sets[i % len(sets)])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        fn(*sets[i % len(sets)])
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


for name, M, N, K in SHAPES:
    sets = [(torch.randn(M, N, device="cuda").bfloat16(), torch.randn(M, K, device="cuda").bfloat16()) for _ in range(6)]
    ours = t_us(lambda dy, x: Fn.gemm_tn(dy, x), sets)
    lib = t_us(lambda dy, x: torch.matmul(dy.t(), x), sets)
    gf = 2.0 * M * N * K / 1e9
    print(f"{name:10s} M {M:5d} N {N:5d} K {K:5d}  {gf:6.2f} GFLOP   ours {ours:6.1f} us ({gf / ours * 1e3:6.1f} TF)   torch.matmul {lib:6.1f} us ({gf / lib * 1e3:6.1f} TF)", flush=True)
