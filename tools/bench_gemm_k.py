import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimodal_edema_prediction_amd import functional as Fn
dev = "cuda"; M = 64 * 257
for N, outdt in ((2304, torch.bfloat16), (768, torch.float32)):
    for K in (64, 128, 256, 768, 1536, 3072):
        R = 4
        a = [torch.randn(M, K, device=dev).bfloat16() for _ in range(R)]; w = torch.randn(N, K, device=dev).bfloat16()
        bias = torch.randn(N, device=dev); out = [torch.empty(M, N, device=dev, dtype=outdt) for _ in range(R)]
        for i in range(8): Fn.gemm(a[i % R], w, bias=bias, out=out[i % R])
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(60): Fn.gemm(a[i % R], w, bias=bias, out=out[i % R])
        e1.record(); torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / 60 * 1e3
        print(f"N={N} out={str(outdt)[6:]} K={K:5d}: {t:7.1f} us", flush=True)
