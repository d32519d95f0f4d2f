"""SURVEY §8(f1), second half: `--unfreeze_cxr` — the CXR encoder trains inside the teacher step.  The product path composes
the ViT forward from autograd nodes over the HIP kernels (cxr_train.py); the check is the CPU oracle with autograd through
its ViT restatement on the same seeded inputs: tokens, loss and the gradient of every ViT tensor."""
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pytestmark = pytest.mark.gpu


def test_trainable_cxr_encoder_matches_oracle_autograd():
    from multimodal_edema_prediction_amd.main_architecture_duett import CXREncoder
    from oracle import vit_ref
    dev = torch.device("cuda")
    torch.manual_seed(0)
    enc = CXREncoder("synthetic", freeze=False).to(dev).train()
    # non-trivial LayerScale / biases so every gradient path carries signal
    g = torch.Generator().manual_seed(5)
    with torch.no_grad():
        for k, p in enc.backbone.named_parameters():
            if k.endswith("lambda1"):
                p.copy_(0.5 + torch.rand(p.shape, generator=g))
            elif k.endswith(".bias"):
                p.copy_(0.05 * torch.randn(p.shape, generator=g))
    sd = {k: v.detach().float().cpu().clone() for k, v in enc.backbone.state_dict().items()}
    B = 2
    px = torch.randn(B, 3, 224, 224, generator=g)
    wsel = torch.randn(257, 768, generator=g) / 50.0              # a fixed linear read-out as the loss

    tok = enc.forward_bf16(px.to(dev))
    assert tok.dtype == torch.float32 and tok.requires_grad
    loss = (tok * wsel.to(dev)).sum()
    enc.zero_grad()
    loss.backward()

    for v in sd.values():
        if v.is_floating_point():
            v.requires_grad_(True)
    cls_ref, patches_ref = vit_ref.vit_forward(sd, vit_ref.VitCfg(), px)
    tok_ref = torch.cat([cls_ref.unsqueeze(1), patches_ref], 1)
    loss_ref = (tok_ref * wsel).sum()
    loss_ref.backward()

    err = (tok.detach().float().cpu() - tok_ref.detach()).abs()
    assert float(err.max()) < 6e-2 and float(err.mean()) < 6e-3, (float(err.max()), float(err.mean()))
    assert abs(float(loss.detach()) - float(loss_ref.detach())) <= 2e-2 * abs(float(loss_ref.detach())) + 1e-2
    named = dict(enc.backbone.named_parameters())
    n = 0
    for k, v in sd.items():
        if k not in named or v.grad is None or float(v.grad.norm()) == 0.0:
            continue
        if k.endswith("attention.key.bias"):
            # softmax is invariant to a constant added to every key (q.(k+b) shifts all scores of a query alike): the exact
            # gradient is zero and both sides hold rounding noise only
            assert float(named[k].grad.norm()) < 1e-3 * float(named[k.replace("key.bias", "value.bias")].grad.norm()) + 1e-4
            continue
        gk = named[k].grad
        assert gk is not None, k
        gk, want = gk.float().cpu(), v.grad
        cos = float((gk * want).sum() / (gk.norm() * want.norm() + 1e-30))
        rel = float((gk - want).norm() / (want.norm() + 1e-30))
        assert cos > 0.98 and rel < 0.25, (k, cos, rel)
        n += 1
    assert n >= 12 * 13 + 4, n


def test_teacher_step_with_trainable_cxr_matches_oracle_autograd():
    """The whole teacher (patch-dual fusion head on top) with `--unfreeze_cxr`: loss, logits, and gradients of the fusion head
    AND of the CXR encoder against the oracle's autograd; DuETT stays frozen."""
    from multimodal_edema_prediction_amd import engine
    from multimodal_edema_prediction_amd.cohort import CohortCfg, make_batch
    from multimodal_edema_prediction_amd.losses_duett import DualPathologyLoss
    from multimodal_edema_prediction_amd.main_architecture_duett import (CXREncoder, PatchDualPathologyPerceiver, TeacherModel,
                                                                           load_duett_backbone)
    from oracle import duett_ref, fusion_ref, losses_ref, vit_ref
    from oracle.step_ref import split_teacher_sd
    dev = torch.device("cuda")
    T, V, DS, K, B = 32, 16, 8, 7, 2
    torch.manual_seed(0)
    backbone = load_duett_backbone("synthetic", d_static_num=DS, d_time_series_num=V, n_timesteps=T, freeze=True)
    cxr = CXREncoder("synthetic", freeze=False)
    per = PatchDualPathologyPerceiver(K, backbone.d_representation, dropout=0.0, head_dropout=0.0)
    torch.nn.init.normal_(per.correction_head[-1].weight, std=0.05)
    teacher = TeacherModel(backbone, cxr, per, cxr_return_patches=True, d_img=768, use_aux_cxr=False,
                           patch_dual_pathology_mode=True).to(dev)
    sd = {k: v.detach().float().cpu().clone() for k, v in teacher.state_dict().items()}
    batch = make_batch(CohortCfg(n_timesteps=T, n_vars=V, d_static=DS, image_size=224, n_labels=K), 0, B, mode="teacher")

    engine._set_train_with_frozen_eval(teacher)
    b = engine._move_lists(batch, dev)
    out = teacher(b["x_ts"], b["x_static"], b["bin_ends"], b["pixel_values"])
    L = DualPathologyLoss(torch.ones(K)).to(dev)(out["img_logits"], out["ts_logits"], out["fusion_logits"], b["y_multi"], b["y_multi_mask"])
    teacher.zero_grad()
    L["total"].backward()

    dcfg = duett_ref.DuettCfg(d_static_num=DS, d_time_series_num=V, n_timesteps=T)
    train = {k: v for k, v in sd.items() if v.is_floating_point() and not k.startswith("duett.")}
    for v in train.values():
        v.requires_grad_(True)
    dsd, vsd = split_teacher_sd(sd)
    with torch.no_grad():
        xin = duett_ref.feats_to_input((batch["x_ts"], batch["x_static"], list(batch["bin_ends"])), max_len=T)
        ts_tokens = duett_ref.encode(dsd, dcfg, xin)
    _, patches = vit_ref.vit_forward(vsd, vit_ref.VitCfg(), batch["pixel_values"])
    ref = fusion_ref.teacher_fusion_forward(sd, ts_tokens, patches, 4)
    Lr = losses_ref.dual_pathology_loss(ref["img_logits"], ref["ts_logits"], ref["fusion_logits"], batch["y_multi"],
                                        batch["y_multi_mask"], torch.ones(K), None, 0.5, 0.5, 1.0)
    Lr["total"].backward()

    assert abs(float(L["total"].detach()) - float(Lr["total"].detach())) <= 1e-2 * abs(float(Lr["total"].detach()))
    assert float((out["fusion_logits"].detach().float().cpu() - ref["fusion_logits"].detach()).abs().max()) < 3e-2
    named = dict(teacher.named_parameters())
    n_vit = n_head = 0
    for k, v in train.items():
        if k not in named or v.grad is None or float(v.grad.norm()) == 0.0 or k.endswith("attention.key.bias"):
            continue
        g = named[k].grad
        assert g is not None, k
        g, want = g.float().cpu(), v.grad
        cos = float((g * want).sum() / (g.norm() * want.norm() + 1e-30))
        rel = float((g - want).norm() / (want.norm() + 1e-30))
        assert cos > 0.98 and rel < (0.35 if want.numel() == 1 else 0.25), (k, cos, rel)
        if k.startswith("cxr."):
            n_vit += 1
        else:
            n_head += 1
    assert n_vit >= 150 and n_head >= 40, (n_vit, n_head)
