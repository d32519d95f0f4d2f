#!/usr/bin/env python3
"""Perceiver attention cores at the step's shapes (B 64, 4 heads of 64, 7 queries over 256 / 96 / 7 keys): forward and backward, isolated
launches.  MEDP_ATTN_FEWQ=0 selects the wave-per-query kernels, the default the thread-per-key ones (csrc/attention_small.hip)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimodal_edema_prediction_amd import functional as Fn
from tools.bench_kernels import timeit
B, H, dh, Lq = 64, 4, 64, 7
D = H * dh
for Lk in (256, 96, 7):
    q = torch.randn(Lq, D, device="cuda")
    kv = torch.randn(B, Lk + 1, 2 * D, device="cuda")
    do = torch.randn(B, Lq, D, device="cuda")
    k, v = kv[:, 1:, :D], kv[:, 1:, D:]
    kw = dict(q_batch_stride=0, kv_batch_stride=kv.stride(0), dropout_p=0.1, seed=1, stream_id=2)
    dkv = torch.empty_like(kv)
    tf = timeit(lambda: Fn.attn_small_fwd(q, k, v, B, Lq, Lk, H, dh, 0.125, **kw))
    tb = timeit(lambda: Fn.attn_small_bwd(do, q, k, v, B, Lq, Lk, H, dh, 0.125, dkv_out=dkv[:, 1:, :], **kw))
    mb = B * Lk * 2 * D * 4 / 1e6
    print(f"FEWQ={os.environ.get('MEDP_ATTN_FEWQ', '1')} Lk={Lk:4d}: fwd {tf*1e6:6.1f} us ({mb / tf / 1e6:5.2f} TB/s of K+V)  bwd {tb*1e6:6.1f} us ({2 * mb / tb / 1e6:5.2f} TB/s of K+V+dK+dV)", flush=True)
