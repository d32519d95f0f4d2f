"""The headline sizes against the ORACLE itself ("none of the configs is compared at its headline size"):
BASELINE.json configs[2] — the full teacher at B = 64, 224x224, T = 96, V = 48 — and configs[3] — the student at the same
T / V — one forward + loss + backward each, HIP path vs the fp32 CPU oracle with autograd on the same seeded cohort and
weights.  ≈ 3 TFLOP of fp32 on the host cores (about half a minute on the GPU box's 16 cores)."""
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

T, V, DS, K, B = 96, 48, 8, 7, 64
DEV = "cuda"


def _cos(a, b):
    a, b = a.detach().float().cpu().flatten(), b.detach().float().flatten()
    return float(torch.dot(a, b) / (a.norm() * b.norm() + 1e-30)), float(a.norm() / (b.norm() + 1e-30))


def test_teacher_cfg3_full_size_against_oracle():
    from multimodal_edema_prediction_amd import engine
    from multimodal_edema_prediction_amd.cohort import CohortCfg, make_batch
    from multimodal_edema_prediction_amd.losses_duett import DualPathologyLoss
    from multimodal_edema_prediction_amd.main_architecture_duett import (CXREncoder, PatchDualPathologyPerceiver, TeacherModel,
                                                                           load_duett_backbone)
    from oracle import duett_ref, fusion_ref, losses_ref, vit_ref
    from oracle.step_ref import split_teacher_sd
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    torch.manual_seed(0)
    backbone = load_duett_backbone("synthetic", d_static_num=DS, d_time_series_num=V, n_timesteps=T, freeze=True)
    cxr = CXREncoder("synthetic", freeze=True)
    per = PatchDualPathologyPerceiver(K, backbone.d_representation, dropout=0.0, head_dropout=0.0)
    torch.nn.init.normal_(per.correction_head[-1].weight, std=0.05)
    teacher = TeacherModel(backbone, cxr, per, cxr_return_patches=True, d_img=768, use_aux_cxr=False, patch_dual_pathology_mode=True).to(DEV)
    sd = {k: v.detach().float().cpu().clone() for k, v in teacher.state_dict().items()}
    batch = make_batch(CohortCfg(n_timesteps=T, n_vars=V, d_static=DS, image_size=224, n_labels=K, seed=1234), 0, B, mode="teacher")

    engine._set_train_with_frozen_eval(teacher)
    b = engine._move_lists(batch, DEV)
    out = teacher(b["x_ts"], b["x_static"], b["bin_ends"], b["pixel_values"])
    L = DualPathologyLoss(torch.ones(K)).to(DEV)(out["img_logits"], out["ts_logits"], out["fusion_logits"], b["y_multi"], b["y_multi_mask"])
    teacher.zero_grad()
    L["total"].backward()

    train = {k: v for k, v in sd.items() if v.is_floating_point() and not k.startswith(("duett.", "cxr."))}
    for v in train.values():
        v.requires_grad_(True)
    dsd, vsd = split_teacher_sd(sd)
    xin = duett_ref.feats_to_input((batch["x_ts"], batch["x_static"], list(batch["bin_ends"])), max_len=T)
    with torch.no_grad():
        ts_tokens = duett_ref.encode(dsd, duett_ref.DuettCfg(d_static_num=DS, d_time_series_num=V, n_timesteps=T), xin)
        _, patches = vit_ref.vit_forward(vsd, vit_ref.VitCfg(), batch["pixel_values"])
    ref = fusion_ref.teacher_fusion_forward(sd, ts_tokens, patches, 4)
    Lr = losses_ref.dual_pathology_loss(ref["img_logits"], ref["ts_logits"], ref["fusion_logits"], batch["y_multi"], batch["y_multi_mask"],
                                        torch.ones(K), None, 0.5, 0.5, 1.0)
    Lr["total"].backward()
    for k in ("img_logits", "ts_logits", "fusion_logits", "scaled_correction"):
        err = float((out[k].detach().float().cpu() - ref[k].detach()).abs().max())
        assert err <= 3e-2, (k, err)                                             # bf16 mode: logits <= 3e-2 abs
    assert abs(float(L["total"]) - float(Lr["total"])) <= 1e-2 * abs(float(Lr["total"]))
    named = dict(teacher.named_parameters())
    n = 0
    for k, r in train.items():
        if r.grad is None:
            assert named[k].grad is None, k
            continue
        c, ratio = _cos(named[k].grad, r.grad)
        assert c > 0.99 and abs(ratio - 1) < 0.1, (k, c, ratio)
        n += 1
    assert n >= 60
    # the same step under the fp32 kernel mode, the CXR encoder included (cxr_train.forward_fp32): the headline size at fp32 tolerances
    from multimodal_edema_prediction_amd import functional as Fn
    with Fn.precision_mode("fp32"), torch.no_grad():
        out32 = teacher(b["x_ts"], b["x_static"], b["bin_ends"], b["pixel_values"])
        L32 = DualPathologyLoss(torch.ones(K)).to(DEV)(out32["img_logits"], out32["ts_logits"], out32["fusion_logits"], b["y_multi"], b["y_multi_mask"])
    for k in ("img_logits", "ts_logits", "fusion_logits", "scaled_correction"):
        err = float((out32[k].float().cpu() - ref[k].detach()).abs().max())
        assert err <= 2e-4, ("fp32", k, err)
    assert abs(float(L32["total"]) - float(Lr["total"])) <= 2e-5 * abs(float(Lr["total"]))


def test_student_cfg4_shapes_full_size_against_oracle():
    from multimodal_edema_prediction_amd import functional as Fn
    from multimodal_edema_prediction_amd.cohort import CohortCfg, make_batch
    from multimodal_edema_prediction_amd.losses_duett import StudentKDLoss
    from multimodal_edema_prediction_amd.main_architecture_duett import StudentModel, load_duett_backbone
    from oracle import duett_ref, losses_ref
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    torch.manual_seed(1)
    student = StudentModel(load_duett_backbone("synthetic", d_static_num=DS, d_time_series_num=V, n_timesteps=T, freeze=False),
                           pool="mean", head_hidden=128, head_dropout=0.0).to(DEV).train()
    sd = {k: v.detach().float().cpu().clone() for k, v in student.state_dict().items()}
    batch = make_batch(CohortCfg(n_timesteps=T, n_vars=V, d_static=DS, image_size=32, n_labels=K, seed=1234), 0, B, mode="student")
    z_t = torch.linspace(-2.0, 2.0, B)
    xd = (tuple(x.to(DEV) for x in batch["x_ts"]), tuple(x.to(DEV) for x in batch["x_static"]), tuple(x.to(DEV) for x in batch["bin_ends"]))

    def hip(mode):
        student.load_state_dict(sd, strict=True)                       # (train-mode forward moves the BatchNorm running statistics)
        student.zero_grad()
        with Fn.precision_mode(mode):
            z = student(*xd)
            L = StudentKDLoss("vanilla_kl", 4.0, 0.5)(z, z_t.to(DEV), batch["y"].to(DEV))
            L["total"].backward()
        return z.detach().float().cpu(), float(L["total"]), {k: p.grad.detach().float().cpu().clone() for k, p in student.named_parameters() if p.grad is not None}

    ref_sd = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and "running_" not in k else v.clone()) for k, v in sd.items()}
    xin = duett_ref.feats_to_input((batch["x_ts"], batch["x_static"], list(batch["bin_ends"])), max_len=T)
    zr = duett_ref.student_forward(ref_sd, duett_ref.DuettCfg(d_static_num=DS, d_time_series_num=V, n_timesteps=T), xin, "mean", training=True)
    Lr = losses_ref.student_kd_loss(zr, z_t, batch["y"], 4.0, 0.5)
    Lr["total"].backward()

    z16, l16, g16 = hip("bf16")
    assert float((z16 - zr.detach()).abs().max()) <= 3e-2
    assert abs(l16 - float(Lr["total"])) <= 1e-2 * abs(float(Lr["total"]))
    z32, l32, g32 = hip("fp32")
    assert float((z32 - zr.detach()).abs().max()) <= 1e-4                        # fp32 kernel mode at the headline size
    assert abs(l32 - float(Lr["total"])) <= 1e-5 * abs(float(Lr["total"]))
    n = 0
    gmax = max(float(r.grad.abs().max()) for r in ref_sd.values() if torch.is_tensor(r) and r.requires_grad and r.grad is not None)
    for k, r in ref_sd.items():
        if not (torch.is_tensor(r) and r.requires_grad) or r.grad is None:
            continue
        c, ratio = _cos(g16[k], r.grad)
        assert c > 0.99 and abs(ratio - 1) < 0.1, ("bf16", k, c, ratio)
        # element-wise: 5e-4 of the tensor's largest element, with the fp32 summation floor of a 6144-row reduction (relative to the
        # model's largest gradient element) under it
        err, scale = float((g32[k].double() - r.grad.double()).abs().max()), float(r.grad.abs().max())
        assert err <= 5e-4 * scale + 2e-6 * gmax, ("fp32", k, err, scale, gmax)
        n += 1
    assert n > 100
