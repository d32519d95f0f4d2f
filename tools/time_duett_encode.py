#!/usr/bin/env python3
"""The frozen DuETT encoder (one C call, medp_duett_encode) at the step's shapes (B 64, T 96, V 48): time per call, alone on the GPU."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimodal_edema_prediction_amd.cohort import CohortCfg, make_batch
from multimodal_edema_prediction_amd.main_architecture_duett import load_duett_backbone
from tools.bench_kernels import timeit
B, T, V = 64, 96, 48
m = load_duett_backbone("synthetic", d_static_num=8, d_time_series_num=V, n_timesteps=T, freeze=True).cuda().eval()
bt = make_batch(CohortCfg(n_timesteps=T, n_vars=V, d_static=8, image_size=28, n_labels=7), 0, B, mode="teacher")
x = tuple(tuple(t.cuda() for t in bt[k]) for k in ("x_ts", "x_static", "bin_ends"))
xin = m.feats_to_input(x, B)
with torch.no_grad():
    t = timeit(lambda: m.encode(xin))
print(f"duett.encode (frozen, B={B}, T={T}, V={V}): {t*1e6:.1f} us per call", flush=True)
