#!/usr/bin/env python3
"""Time medp_duett_embed_fwd stage 1 (fused psi build) at cfg3 shapes; MEDP_PSI_NB selects rows per workgroup."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimodal_edema_prediction_amd.abi import check, lib, ptr, stream
from multimodal_edema_prediction_amd.main_architecture_duett import load_duett_backbone
B, T, V, E = 64, 96, 48, 24
m = load_duett_backbone("synthetic", d_static_num=8, d_time_series_num=V, n_timesteps=T, freeze=True).cuda()
w = m._prepare()[0]
xs_static, xs_ts, xs_times = torch.randn(B, 8, device="cuda"), torch.zeros(B, T, 2 * V + 1, device="cuda"), torch.rand(B, T, device="cuda")
xs_ts[:, :, :V] = torch.randn(B, T, V, device="cuda")
xs_ts[:, :, V:2 * V] = torch.randint(0, 4, (B, T, V), device="cuda").float()
n = B * (T + 1) * (V + 1) * E
xe, h, temb = torch.empty(n, device="cuda"), torch.empty(n, device="cuda", dtype=torch.bfloat16), torch.empty(n, device="cuda")
f = lambda: check(lib().medp_duett_embed_fwd(ctypes.byref(w), ptr(xs_static), ptr(xs_ts), ptr(xs_times), B, T, ptr(xe), ptr(h), ptr(temb), None, None, 1, stream()), "embed")
for _ in range(10): f()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(100): f()
e1.record(); torch.cuda.synchronize()
print("MEDP_PSI_NB", os.environ.get("MEDP_PSI_NB", "default"), f"{e0.elapsed_time(e1) * 10:.2f} us")
