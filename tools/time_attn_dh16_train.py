#!/usr/bin/env python3
"""DuETT's training-form attention at the student step's shapes (B 64: the event-axis encoder attends over 49 tokens, the time-axis encoder over 97;
2 heads of 12): MFMA kernels (attention_dh16_train.hip) against the fp32 VALU kernels (attention_small.hip), isolated launches."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimodal_edema_prediction_amd import functional as Fn
from multimodal_edema_prediction_amd.abi import check, lib, ptr, stream
from tools.bench_kernels import timeit
H, dh = 2, 12
D = H * dh
for P in (0.0, 0.1):
  for B, N in ((64, 97), (64, 49)):
      qkv = torch.randn(B, N, 3 * D, device="cuda")
      do = torch.randn(B, N, D, device="cuda")
      o = torch.empty(B, N, D, device="cuda"); lse = torch.empty(B * H * N, device="cuda"); delta = torch.empty_like(lse); dqkv = torch.empty_like(qkv)
      f = lambda: check(lib().medp_attn_dh16_train_fwd(ptr(qkv), 3 * D, ptr(o), D, ptr(lse), 0, B, N, H, dh, dh ** -0.5, P, 1, 2, stream()), "f")
      b = lambda: check(lib().medp_attn_dh16_train_bwd(ptr(do), D, ptr(qkv), 3 * D, ptr(lse), ptr(delta), ptr(dqkv), 3 * D, 0, B, N, H, dh, dh ** -0.5, P, 1, 2, stream()), "b")
      fv = lambda: Fn.attn_small_fwd(qkv[..., :D], qkv[..., D:2 * D], qkv[..., 2 * D:], B, N, N, H, dh, dh ** -0.5, q_batch_stride=N * 3 * D, kv_batch_stride=N * 3 * D, dropout_p=P, seed=1, stream_id=2)
      base = dqkv.data_ptr()
      bv = lambda: check(lib().medp_attn_small_bwd(ptr(do), D, ptr(qkv), 3 * D, N * 3 * D, qkv.data_ptr() + 4 * D, qkv.data_ptr() + 8 * D, 3 * D, N * 3 * D, base, 3 * D,
                                                   base + 4 * D, 3 * D, base + 8 * D, 0, N * 3 * D, B, N, N, H, dh, dh ** -0.5, P, 1, 2, stream()), "bv")
      print(f"p={P} B={B} N={N}: MFMA fwd {timeit(f)*1e6:6.1f} us, bwd (2 launches) {timeit(b)*1e6:6.1f} us | VALU fwd {timeit(fv)*1e6:6.1f} us, bwd {timeit(bv)*1e6:6.1f} us", flush=True)
