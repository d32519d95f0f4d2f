#!/usr/bin/env python3
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimodal_edema_prediction_amd import functional as Fn
from tools.bench_kernels import timeit
B, S, H = (int(x) for x in (sys.argv[1:4] if len(sys.argv) > 3 else (64, 257, 12)))
qkv = (torch.randn(B * S, 3 * H * 64, device="cuda") * 0.5).bfloat16()
t = timeit(lambda: Fn.attn_dh64(qkv, B, S, H, 0.125))
print(f"attn_dh64 diag={os.environ.get('MEDP_ATTN_DIAG','0')} B={B} S={S} H={H}: {t*1e6:8.1f} us  {4*B*H*S*S*64/t/1e12:7.1f} TFLOP/s", flush=True)
