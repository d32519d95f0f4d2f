#!/usr/bin/env python3
"""Yardstick: torch.matmul (hipBLASLt) on the four CXR-encoder block-GEMM shapes — run under `rocprofv3 --kernel-trace` to read the
library's kernel names (macro-tile, wave tiling, prefetch depth are spelled out in them) and durations beside our v6 / v7."""
import torch
M = 64 * 257
for n, k in ((2304, 768), (768, 768), (3072, 768), (768, 3072)):
    a = torch.randn(M, k, device="cuda").bfloat16()
    w = torch.randn(n, k, device="cuda").bfloat16()
    for _ in range(10):
        torch.matmul(a, w.T)
torch.cuda.synchronize()
