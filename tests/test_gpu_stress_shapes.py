"""BASELINE.json configs[4] (stress) shapes — CXR 512x512 (1297 ViT tokens, bicubic position grid, 5 LDS key chunks) and
DuETT T=256 / V=96 (event tokens 6168 wide, time tokens 2328 wide, 257-key attention) — at a small batch, HIP path vs the
CPU oracle on the same seeded weights and inputs.  Same tolerances as the cfg1/cfg3 parity tests."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from multimodal_edema_prediction_amd import engine  # noqa: E402
from multimodal_edema_prediction_amd.cohort import CohortCfg, make_batch  # noqa: E402
from multimodal_edema_prediction_amd.losses_duett import DualPathologyLoss, StudentKDLoss  # noqa: E402
from multimodal_edema_prediction_amd.main_architecture_duett import (CXREncoder, PatchDualPathologyPerceiver, StudentModel,  # noqa: E402
                                                                       TeacherModel, load_duett_backbone)
from oracle import duett_ref, losses_ref, step_ref, vit_ref  # noqa: E402

T, V, DS, K, B = 256, 96, 8, 7, 2
DEV = "cuda"


def test_stress_teacher_forward_and_loss():
    torch.manual_seed(0)
    backbone = load_duett_backbone("synthetic", d_static_num=DS, d_time_series_num=V, n_timesteps=T, freeze=True)
    cxr = CXREncoder("synthetic", freeze=True)
    per = PatchDualPathologyPerceiver(K, backbone.d_representation, dropout=0.0, head_dropout=0.0)
    torch.nn.init.normal_(per.correction_head[-1].weight, std=0.05)
    teacher = TeacherModel(backbone, cxr, per, cxr_return_patches=True, d_img=768, use_aux_cxr=False, patch_dual_pathology_mode=True).to(DEV)
    sd = {k: v.detach().float().cpu().clone() for k, v in teacher.state_dict().items()}
    ccfg = CohortCfg(n_timesteps=T, n_vars=V, d_static=DS, image_size=512, n_labels=K)
    batch = make_batch(ccfg, 0, B, mode="teacher")
    loss_fn = DualPathologyLoss(torch.ones(K)).to(DEV)
    engine._set_train_with_frozen_eval(teacher)
    b = engine._move_lists(batch, DEV)
    out = teacher(b["x_ts"], b["x_static"], b["bin_ends"], b["pixel_values"], return_attn=True)
    assert out["img_attn"].shape == (B, K, 1296) and out["ts_attn"].shape == (B, K, T)
    L = loss_fn(out["img_logits"], out["ts_logits"], out["fusion_logits"], b["y_multi"], b["y_multi_mask"])
    L["total"].backward()
    dcfg = duett_ref.DuettCfg(d_static_num=DS, d_time_series_num=V, n_timesteps=T)
    ref = step_ref.teacher_forward(sd, dcfg, vit_ref.VitCfg(), batch, return_attn=True)
    for k in ("img_logits", "ts_logits", "fusion_logits", "scaled_correction"):
        err = float((out[k].detach().cpu() - ref[k]).abs().max())
        assert err < 3e-2, (k, err)
    assert float((out["img_attn"].cpu() - ref["img_attn"]).abs().max()) < 5e-3
    Lr = losses_ref.dual_pathology_loss(ref["img_logits"], ref["ts_logits"], ref["fusion_logits"], batch["y_multi"], batch["y_multi_mask"],
                                        torch.ones(K))
    assert abs(float(L["total"]) - float(Lr["total"])) <= 1e-2 * abs(float(Lr["total"]))
    assert all(torch.isfinite(p.grad).all() for p in teacher.parameters() if p.grad is not None)


def test_stress_student_forward_backward():
    torch.manual_seed(1)
    backbone = load_duett_backbone("synthetic", d_static_num=DS, d_time_series_num=V, n_timesteps=T, freeze=False)
    student = StudentModel(backbone, pool="mean", head_hidden=128, head_dropout=0.0).to(DEV).train()
    ssd = {k: v.detach().cpu().clone() for k, v in student.state_dict().items()}
    ccfg = CohortCfg(n_timesteps=T, n_vars=V, d_static=DS, image_size=32, n_labels=K)
    batch = make_batch(ccfg, 7, 4, mode="student")
    b = engine._move_lists(batch, DEV)
    z_t = torch.tensor([0.3, -0.2, 0.1, 0.7])
    z_s = student(b["x_ts"], b["x_static"], b["bin_ends"])
    L = StudentKDLoss()(z_s, z_t.to(DEV), b["y"])
    L["total"].backward()
    train = {k: v.clone().requires_grad_(True) for k, v in ssd.items() if v.is_floating_point() and "running_" not in k}
    sdr = {**ssd, **train}
    dcfg = duett_ref.DuettCfg(d_static_num=DS, d_time_series_num=V, n_timesteps=T)
    xin = duett_ref.feats_to_input((batch["x_ts"], batch["x_static"], list(batch["bin_ends"])), max_len=T)
    z_ref = duett_ref.student_forward(sdr, dcfg, xin, "mean", training=True)
    Lr = losses_ref.student_kd_loss(z_ref, z_t, batch["y"])
    Lr["total"].backward()
    assert float((z_s.detach().cpu() - z_ref.detach()).abs().max()) < 3e-2
    assert abs(float(L["total"]) - float(Lr["total"])) <= 1e-2 * abs(float(Lr["total"]))
    named = dict(student.named_parameters())
    for k in ("head.0.weight", "duett.embedding_layers.95.4.weight", "duett.full_event_embedding.weight",
              "duett.time_transformers.1.layers.1.1.ff.2.weight", "duett.event_transformers.0.layers.0.1.to_k.weight",
              "duett.tab_encoder.0.weight", "duett.full_time_embedding.3.weight"):
        g, w = named[k].grad.float().cpu(), train[k].grad
        cos = float((g * w).sum() / (g.norm() * w.norm() + 1e-30))
        assert cos > 0.99, (k, cos)
