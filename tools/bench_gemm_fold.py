#!/usr/bin/env python3
"""The four CXR-encoder block GEMMs with and without the LayerNorm-fold epilogues (medp_dbg_gemm_fold), rotating operands, next to the
LayerNorm launch the fold removes: does producer + consumer cost less than plain + plain + LayerNorm?"""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimodal_edema_prediction_amd import functional as Fn
from multimodal_edema_prediction_amd.abi import lib, ptr, stream, check
dev = "cuda"; M, D, F = 64 * 257, 768, 3072; R = 4
L = lib()
L.medp_dbg_gemm_fold.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int] * 3 + [ctypes.c_void_p] * 3 + [ctypes.c_int] * 2 + \
    [ctypes.c_void_p] * 3 + [ctypes.c_int, ctypes.c_void_p, ctypes.c_float, ctypes.c_int, ctypes.c_void_p]
def fold(A, W, C, M_, N, K, bias=None, scale=None, residual=None, act=0, out_bf16=0, c2=None, so=None, si=None, cs=None):
    check(L.medp_dbg_gemm_fold(ptr(A), ptr(W), ptr(C), M_, N, K, ptr(bias), ptr(scale), ptr(residual), act, out_bf16, ptr(c2), ptr(so), ptr(si),
                               3 if si is not None else 0, ptr(cs), 1e-6, D if si is not None else 0, stream()), "fold")
def timeit(fn, n=60):
    for i in range(8): fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n): fn(i)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
mk = lambda *sh, dt=torch.bfloat16: [torch.randn(*sh, device=dev).to(dt) for _ in range(R)]
Mpad = (M + 255) // 256 * 256
stats = [torch.rand(Mpad, 3, 2, device=dev) * 100 + 300 for _ in range(R)]
for s_ in stats: s_[:, :, 0] = torch.randn(Mpad, 3, device=dev)
xb, x32, att, f = mk(M, D), mk(M, D, dt=torch.float32), mk(M, D), mk(M, F)
qkv, fo = mk(M, 3 * D), mk(M, F)
c2 = mk(M, D)
w_qkv, w_proj, w_fc1, w_fc2 = [torch.randn(n, k, device=dev).bfloat16() * 0.05 for n, k in ((3 * D, D), (D, D), (F, D), (D, F))]
b3, bD, bF, ls = torch.randn(3 * D, device=dev), torch.randn(D, device=dev), torch.randn(F, device=dev), torch.rand(D, device=dev)
cs3, csF = torch.randn(3 * D, device=dev), torch.randn(F, device=dev)
lw, lb = torch.ones(D, device=dev), torch.zeros(D, device=dev)
rows = [
    ("qkv  plain", lambda i: Fn.gemm(xb[i % R], w_qkv, bias=b3, out=qkv[i % R])),
    ("qkv  consumer", lambda i: fold(xb[i % R], w_qkv, qkv[i % R], M, 3 * D, D, bias=b3, out_bf16=1, si=stats[i % R], cs=cs3)),
    ("fc1  plain (GELU)", lambda i: Fn.gemm(xb[i % R], w_fc1, bias=bF, act=1, out=fo[i % R])),
    ("fc1  consumer (GELU)", lambda i: fold(xb[i % R], w_fc1, fo[i % R], M, F, D, bias=bF, act=1, out_bf16=1, si=stats[i % R], cs=csF)),
    ("proj plain", lambda i: Fn.gemm(att[i % R], w_proj, bias=bD, scale=ls, residual=x32[i % R], out=x32[i % R])),
    ("proj producer", lambda i: fold(att[i % R], w_proj, x32[i % R], M, D, D, bias=bD, scale=ls, residual=x32[i % R], c2=c2[i % R], so=stats[i % R])),
    ("fc2  plain", lambda i: Fn.gemm(f[i % R], w_fc2, bias=bD, scale=ls, residual=x32[i % R], out=x32[i % R])),
    ("fc2  producer", lambda i: fold(f[i % R], w_fc2, x32[i % R], M, D, F, bias=bD, scale=ls, residual=x32[i % R], c2=c2[i % R], so=stats[i % R])),
    ("layernorm (fp32 -> bf16)", lambda i: Fn.layernorm(x32[i % R], lw, lb, 1e-6)),
]
for name, fn in rows:
    print(f"{name:28s} {timeit(fn):7.1f} us", flush=True)
