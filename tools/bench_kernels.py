#!/usr/bin/env python3
"""Micro-benchmarks of the hot kernels at the cfg3 (B=64, 224²) shapes; prints TFLOP/s per GEMM shape and attention."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimodal_edema_prediction_amd import functional as Fn

def timeit(fn, iters=200, warm=20):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3

def main():
    dev = "cuda"
    M = 64 * 257
    for (m, n, k, tag) in [(M, 2304, 768, "qkv"), (M, 768, 768, "proj"), (M, 3072, 768, "fc1"), (M, 768, 3072, "fc2"),
                           (4096, 4096, 4096, "4k^3"), (8192, 8192, 8192, "8k^3"), (64 * 49, 512, 2328, "duett ff1"),
                           (64 * 49, 2328, 512, "duett ff2"), (64 * 256, 256, 768, "img_proj")]:
        a = torch.randn(m, k, device=dev).bfloat16(); w = torch.randn(n, k, device=dev).bfloat16()
        out = torch.empty(m, n, device=dev, dtype=torch.bfloat16)
        t = timeit(lambda: Fn.gemm(a, w, out=out))
        print(f"gemm {tag:10s} M={m} N={n} K={k}: {t*1e6:8.1f} us  {2*m*n*k/t/1e12:7.1f} TFLOP/s", flush=True)
        if "--gemm-only" not in sys.argv:
            t2 = timeit(lambda: torch.matmul(a, w.T))
            print(f"   (torch/hipBLASLt same shape: {t2*1e6:8.1f} us  {2*m*n*k/t2/1e12:7.1f} TFLOP/s)", flush=True)
    if "--gemm-only" in sys.argv:
        return
    B, S, H = 64, 257, 12
    qkv = torch.randn(B * S, 3 * H * 64, device=dev).bfloat16()
    t = timeit(lambda: Fn.attn_dh64(qkv, B, S, H, 0.125))
    print(f"attn_dh64 B={B} S={S} H={H}: {t*1e6:8.1f} us  {4*B*H*S*S*64/t/1e12:7.1f} TFLOP/s", flush=True)
    x = torch.randn(M, 768, device=dev); w = torch.ones(768, device=dev); b = torch.zeros(768, device=dev)
    t = timeit(lambda: Fn.layernorm(x, w, b, 1e-6))
    print(f"layernorm {M}x768 f32->bf16: {t*1e6:8.1f} us  {(M*768*6)/t/1e9:7.1f} GB/s", flush=True)

if __name__ == "__main__":
    main()
