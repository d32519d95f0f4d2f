#!/usr/bin/env python3
"""qkv / fc1 / fc1 without GELU with rotating operands (as tools/bench_gemm_instep.py) — for A/B runs under MEDP_GEMM_V7 /
MEDP_V7_DEBUG."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimodal_edema_prediction_amd import functional as Fn
dev = "cuda"; M, D, F = 64 * 257, 768, 3072; R = 4
def mk(*shape, dtype=torch.bfloat16): return [torch.randn(*shape, device=dev).to(dtype) for _ in range(R)]
cases = {
    "qkv": dict(a=mk(M, D), w=torch.randn(3 * D, D, device=dev).bfloat16(), bias=torch.randn(3 * D, device=dev), out=mk(M, 3 * D), act=0),
    "fc1": dict(a=mk(M, D), w=torch.randn(F, D, device=dev).bfloat16(), bias=torch.randn(F, device=dev), out=mk(M, F), act=1),
    "fc1_nogelu": dict(a=mk(M, D), w=torch.randn(F, D, device=dev).bfloat16(), bias=torch.randn(F, device=dev), out=mk(M, F), act=0),
    "fc1_noragged": dict(a=mk(M - 64, D), w=torch.randn(F, D, device=dev).bfloat16(), bias=torch.randn(F, device=dev), out=mk(M - 64, F), act=0),
}
for name, c in cases.items():
    run = lambda i: Fn.gemm(c["a"][i % R], c["w"], bias=c["bias"], act=c["act"], out=c["out"][i % R])
    for i in range(8): run(i)
    torch.cuda.synchronize()
    n = 80
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n): run(i)
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / n * 1e3
    print(f"{name:13s} {t:7.1f} us", flush=True)
