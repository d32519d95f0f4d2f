#!/usr/bin/env python3
"""Correctness screen for the GEMM variant selected by MEDP_GEMM_VARIANT (big-M path): ragged M/N/K, all epilogues, repeated
runs (a staging race shows up as rare wrong tiles), against an fp32 torch product of the same bf16 operands."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimodal_edema_prediction_amd import functional as Fn

torch.manual_seed(0)
dev = "cuda"
bad = 0
shapes = [(16448, 2304, 768), (16448, 768, 3072), (2050, 260, 776), (4096, 512, 64), (2048, 256, 72), (3000, 1000, 200),
          (16448, 768, 768), (8192, 4096, 1024), (2304, 3072, 136)]
for (m, n, k) in shapes:
    a = torch.randn(m, k, device=dev).bfloat16(); w = torch.randn(n, k, device=dev).bfloat16()
    bias = torch.randn(n, device=dev); scale = torch.rand(n, device=dev) + 0.5; res = torch.randn(m, n, device=dev)
    ref0 = a.float() @ w.float().T
    for rep in range(6):
        mode = rep % 3
        if mode == 0:
            out = Fn.gemm(a, w, out_dtype=torch.float32); ref = ref0
        elif mode == 1:
            out = Fn.gemm(a, w, bias=bias, act=1, out_dtype=torch.bfloat16).float()
            ref = torch.nn.functional.gelu(ref0 + bias).bfloat16().float()
        else:
            out = Fn.gemm(a, w, bias=bias, scale=scale, residual=res, out_dtype=torch.float32); ref = (ref0 + bias) * scale + res
        err = (out - ref).abs().max().item(); tol = 2e-2 * max(1.0, ref.abs().max().item()) if mode == 1 else 1e-3 * k ** 0.5 + 1e-3
        ok = err <= tol
        bad += (not ok)
        if not ok or rep == 0:
            print(f"M={m} N={n} K={k} mode={mode} rep={rep}: max err {err:.3e} tol {tol:.2e} {'ok' if ok else 'FAIL'}", flush=True)
print("FAILED" if bad else "ALL OK")
sys.exit(1 if bad else 0)
