#!/usr/bin/env python3
"""The four CXR-encoder block GEMMs with their REAL epilogues (bias / GELU / LayerScale + fp32 residual, bf16 or fp32 out) and
rotating operand sets (> 256 MB in total), so the Infinity Cache cannot hold the outputs as it does when one launch is
repeated on the same buffers.  Closer to what the step sees than tools/bench_kernels.py."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimodal_edema_prediction_amd import functional as Fn

dev = "cuda"
M, D, F = int(os.environ.get("M", 64 * 257)), 768, 3072
R = int(os.environ.get("ROT", "4"))
def mk(*shape, dtype=torch.bfloat16): return [torch.randn(*shape, device=dev).to(dtype) for _ in range(R)]
cases = {
    "qkv":  dict(a=mk(M, D), w=torch.randn(3 * D, D, device=dev).bfloat16(), bias=torch.randn(3 * D, device=dev), out=mk(M, 3 * D), act=0),
    "proj": dict(a=mk(M, D), w=torch.randn(D, D, device=dev).bfloat16(), bias=torch.randn(D, device=dev), scale=torch.rand(D, device=dev),
                 res=mk(M, D, dtype=torch.float32), act=0),
    "fc1":  dict(a=mk(M, D), w=torch.randn(F, D, device=dev).bfloat16(), bias=torch.randn(F, device=dev), out=mk(M, F), act=1),
    "fc1_nogelu": dict(a=mk(M, D), w=torch.randn(F, D, device=dev).bfloat16(), bias=torch.randn(F, device=dev), out=mk(M, F), act=0),
    "fc2":  dict(a=mk(M, F), w=torch.randn(D, F, device=dev).bfloat16(), bias=torch.randn(D, device=dev), scale=torch.rand(D, device=dev),
                 res=mk(M, D, dtype=torch.float32), act=0),
}
def run(c, i):
    j = i % R
    if "res" in c:   # in-place residual stream, as vit.hip does
        Fn.gemm(c["a"][j], c["w"], bias=c["bias"], scale=c["scale"], residual=c["res"][j], out=c["res"][j])
    else:
        Fn.gemm(c["a"][j], c["w"], bias=c["bias"], act=c["act"], out=c["out"][j])
for name, c in cases.items():
    for i in range(8): run(c, i)
    torch.cuda.synchronize()
    n = 80
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n): run(c, i)
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / n * 1e-3
    mm, nn, kk = c["a"][0].shape[0], c["w"].shape[0], c["w"].shape[1]
    print(f"{name:11s} M={mm} N={nn} K={kk}: {t*1e6:8.1f} us  {2*mm*nn*kk/t/1e12:7.1f} TFLOP/s", flush=True)
