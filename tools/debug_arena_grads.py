#!/usr/bin/env python3
"""Debug aid: one forward/backward of the teacher with a trainable DuETT, gradients left to autograd (None before backward)
against gradients accumulated into a dp.FlatGradArena — prints every parameter whose gradient differs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimodal_edema_prediction_amd import dp, engine
from multimodal_edema_prediction_amd.cohort import CohortCfg, make_batch
from multimodal_edema_prediction_amd.losses_duett import DualPathologyLoss
from multimodal_edema_prediction_amd.main_architecture_duett import CXREncoder, PatchDualPathologyPerceiver, TeacherModel, load_duett_backbone

dev = torch.device("cuda")
Tn, V, DS, K, B = 32, 16, 8, 7, 4
batch = make_batch(CohortCfg(n_timesteps=Tn, n_vars=V, d_static=DS, image_size=224, n_labels=K), 0, B, mode="teacher")
loss_fn = DualPathologyLoss(torch.ones(K)).to(dev)

def build():
    torch.manual_seed(0)
    bb = load_duett_backbone("synthetic", d_static_num=DS, d_time_series_num=V, n_timesteps=Tn, freeze=False)
    te = TeacherModel(bb, CXREncoder("synthetic", freeze=True), PatchDualPathologyPerceiver(K, bb.d_representation, dropout=0.0, head_dropout=0.0),
                      cxr_return_patches=True, d_img=768, use_aux_cxr=False, patch_dual_pathology_mode=True).to(dev)
    engine._set_train_with_frozen_eval(te)
    return te

def fb(te):
    b = engine._move_lists(batch, dev)
    out = te(b["x_ts"], b["x_static"], b["bin_ends"], b["pixel_values"])
    L = loss_fn(out["img_logits"], out["ts_logits"], out["fusion_logits"], b["y_multi"], b["y_multi_mask"])
    L["total"].backward()
    torch.cuda.synchronize()
    return float(L["total"])

a = build(); la = fb(a)
b_ = build()
params = [p for p in b_.parameters() if p.requires_grad]
used = dp.find_used_parameters(params, lambda: fb(b_))
arena = dp.FlatGradArena(params, used=used)
arena.bind(zero=True)
lb = fb(b_)
print("loss", la, lb)
na, nb = dict(a.named_parameters()), dict(b_.named_parameters())
bad = 0
for k in na:
    ga, gb = na[k].grad, nb[k].grad
    if (ga is None) != (gb is None):
        print("presence differs", k); bad += 1; continue
    if ga is None: continue
    if not torch.equal(ga, gb):
        d = (ga - gb).abs().max().item(); bad += 1
        print(f"{k:60s} max|diff| {d:.3e}  max|g| {ga.abs().max().item():.3e}")
print("differing tensors:", bad)
