"""SURVEY.md §8(f2) on the HIP path: the `DualPathologyPerceiver` teacher (`TeacherModel(dual_pathology_mode=True)`: CLS -> frozen
pretrained linear CXR head -> kept columns -> per-pathology residual fusion) against the fixture the reference's own code produced
(tests/golden/make_golden_dual.py runs the reference's commented-out class text inside its imported module), and the
student-from-checkpoint path (`best.pt` -> frozen dual teacher, reference training_duett/trainer.py:770-822)."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

from helpers import load_npz, load_shapes, synth_state_dict, t  # noqa: E402
from tests_dual_common import cxr_head_state  # noqa: E402

B, T, V, DS, K = 8, 32, 16, 8, 7
DEV = "cuda"


def _build(tmp_path, dropout=0.0):
    from multimodal_edema_prediction_amd.cohort import PATHOLOGY_LABELS
    from multimodal_edema_prediction_amd.main_architecture_duett import (CXREncoder, DualPathologyPerceiver, DuettFeatureExtractor,
                                                                           TeacherModel)
    gold = load_npz("teacher_dual_cfg1.npz")
    shapes = load_shapes("shapes.json")
    head_ckpt = os.path.join(str(tmp_path), "cxr_head.pt")
    torch.save(cxr_head_state(), head_ckpt)
    backbone = DuettFeatureExtractor(d_static_num=DS, d_time_series_num=V, d_target=1, pretrain=False, masked_transform_timesteps=T,
                                     max_len=T, aug_noise=0.0, aug_mask=0.0, transformer_dropout=0.0)
    for p in backbone.parameters():
        p.requires_grad = False
    backbone.eval()
    cxr = CXREncoder("synthetic", freeze=True, return_patches=False)
    per = DualPathologyPerceiver(n_pathologies=K, d_ts=backbone.d_representation, d_latent=256, n_heads=4, dropout=dropout,
                                 head_dropout=dropout)
    teacher = TeacherModel(backbone, cxr, per, head_hidden=128, head_dropout=0.0, cxr_return_patches=False, d_img=768, use_aux_cxr=False,
                           dual_pathology_mode=True, pretrained_cxr_head_ckpt=head_ckpt, pathology_labels=tuple(PATHOLOGY_LABELS))
    assert sorted(teacher.state_dict()) == sorted(shapes["teacher_dual"])              # the reference's key set
    assert teacher.cxr_head_keep_idx.tolist() == list(gold["keep_idx"])                 # label lookup, not the synthetic fill
    sd = synth_state_dict(shapes["teacher_dual"], seed=5)
    for k, v in synth_state_dict(shapes["vit"], seed=3).items():
        sd["cxr.backbone." + k] = v
    hs = cxr_head_state()["classifier_state_dict"]
    sd["pretrained_cxr_head.weight"], sd["pretrained_cxr_head.bias"] = hs["1.weight"], hs["1.bias"]
    sd["cxr_head_keep_idx"] = t(gold["keep_idx"]).long()
    teacher.load_state_dict(sd, strict=True)
    return teacher.to(DEV), gold, sd


def _batch():
    from multimodal_edema_prediction_amd.cohort import CohortCfg, make_batch
    return make_batch(CohortCfg(n_timesteps=T, n_vars=V, d_static=DS, image_size=224, seed=1234), 100, B, mode="teacher")


def maxerr(a, b):
    return float((a.detach().float().cpu() - torch.as_tensor(b)).abs().max())


def test_dual_teacher_forward_against_reference_fixture(tmp_path):
    from multimodal_edema_prediction_amd import engine
    teacher, gold, _ = _build(tmp_path)
    teacher.eval()
    b = engine._move_lists(_batch(), DEV)
    with torch.no_grad():
        out = teacher(b["x_ts"], b["x_static"], b["bin_ends"], b["pixel_values"], return_attn=True)
    assert {"fwd:" + k for k in out} == {k for k in gold if k.startswith("fwd:")}
    for k in ("main_logit", "img_logits", "ts_logits", "fusion_logits", "residuals"):
        assert maxerr(out[k], gold["fwd:" + k]) < 3e-2, (k, maxerr(out[k], gold["fwd:" + k]))       # bf16 MFMA operands
    assert maxerr(out["ts_attn"], gold["fwd:ts_attn"]) < 5e-3
    assert maxerr(out["ts_tokens"], gold["fwd:ts_tokens"]) < 6e-2


def test_dual_teacher_loss_gradients_and_engine_step(tmp_path):
    from multimodal_edema_prediction_amd import engine
    from multimodal_edema_prediction_amd.losses_duett import DualPathologyLoss
    from multimodal_edema_prediction_amd.optim import FusedAdamW
    teacher, gold, sd = _build(tmp_path)
    tb = _batch()
    loss_fn = DualPathologyLoss(torch.ones(K), None, 0.5, 0.5, 1.0).to(DEV)
    engine._set_train_with_frozen_eval(teacher)
    assert not teacher.pretrained_cxr_head.training and not teacher.cxr.training           # frozen sub-modules in eval()
    b = engine._move_lists(tb, DEV)
    out = teacher(b["x_ts"], b["x_static"], b["bin_ends"], b["pixel_values"])
    L = loss_fn(out["img_logits"], out["ts_logits"], out["fusion_logits"], b["y_multi"], b["y_multi_mask"])
    for k in ("total", "img_total", "ts_total", "fus_total"):
        assert abs(float(L[k]) - float(gold["loss:" + k])) <= 1e-2 * abs(float(gold["loss:" + k])) + 1e-4, k
    teacher.zero_grad()
    L["total"].backward()
    named = dict(teacher.named_parameters())
    unused = sorted(k for k, p in named.items() if p.requires_grad and p.grad is None)
    assert unused == sorted(str(s) for s in gold["unused_parameters"])
    for key in gold:
        if key.startswith("grad:"):
            g, r = named[key[5:]].grad.float().cpu().flatten(), t(gold[key]).flatten()
            cos = float(torch.dot(g, r) / (g.norm() * r.norm() + 1e-30))
            assert cos > 0.995 and abs(float(g.norm() / r.norm()) - 1) < 0.1, (key, cos)
        elif key.startswith("gsum:"):
            g = named[key[5:]].grad.double()
            assert abs(float(g.abs().sum()) - gold[key][1]) <= 0.1 * gold[key][1] + 1e-6, key
    # one engine step with the fused AdamW against the reference engine + torch.optim.AdamW
    teacher.load_state_dict(sd, strict=True)
    opt = FusedAdamW([p for p in teacher.parameters() if p.requires_grad], lr=8e-5, weight_decay=5e-2)
    step = engine.train_teacher_dual_pathology_batch(tb, teacher, loss_fn, opt, torch.device(DEV))
    assert abs(step["loss"] - float(gold["step:loss"])) <= 1e-2 * abs(float(gold["step:loss"]))
    for key in gold:
        if key.startswith("post:"):
            p = named[key[5:]].detach().double()
            assert abs(float(p.abs().sum()) - gold[key][1]) <= 2e-4 * gold[key][1] + 1e-6, key
            assert abs(float(p.sum()) - gold[key][0]) <= 2e-4 * gold[key][1] + 1e-5, key


def test_student_kd_from_a_dual_teacher_checkpoint(tmp_path):
    """trainer.py:770-822, 856-865: `best.pt` of a dual teacher -> frozen teacher rebuilt from the checkpoint's own `args` ->
    one `train_student_batch`; the rebuilt teacher's logits equal the saved model's, nothing in it trains."""
    from multimodal_edema_prediction_amd import checkpoint, engine, train_synthetic
    from multimodal_edema_prediction_amd.cohort import PATHOLOGY_LABELS
    from multimodal_edema_prediction_amd.losses_duett import StudentKDLoss
    from multimodal_edema_prediction_amd.main_architecture_duett import StudentModel, load_duett_backbone
    from multimodal_edema_prediction_amd.optim import FusedAdamW
    teacher, gold, _ = _build(tmp_path)
    head_ckpt = os.path.join(str(tmp_path), "cxr_head.pt")
    args = {"perceiver_type": "dual", "pathology_labels": ",".join(PATHOLOGY_LABELS), "cxr_model_name": "synthetic", "d_latent": 256,
            "n_perceiver_heads": 4, "perceiver_dropout": 0.1, "head_hidden": 128, "head_dropout": 0.1,
            "pretrained_cxr_head_ckpt": head_ckpt}
    opt = torch.optim.AdamW([p for p in teacher.parameters() if p.requires_grad], lr=1e-4)
    path = os.path.join(str(tmp_path), "run", "best.pt")
    checkpoint.save_ckpt(path, teacher, opt, epoch=3, metric=0.71, args=args)
    rebuilt = train_synthetic.build_teacher_from_ckpt(checkpoint.load_ckpt(path), d_static=DS, n_vars=V, duett_ckpt="synthetic",
                                                      n_timesteps=T, cxr_model_name_fallback="synthetic").to(DEV)
    assert not any(p.requires_grad for p in rebuilt.parameters()) and not rebuilt.training
    tb = _batch()
    b = engine._move_lists(tb, DEV)
    teacher.eval()
    with torch.no_grad():
        z0 = teacher(b["x_ts"], b["x_static"], b["bin_ends"], b["pixel_values"])["main_logit"]
        z1 = rebuilt(b["x_ts"], b["x_static"], b["bin_ends"], b["pixel_values"])["main_logit"]
    assert torch.equal(z0, z1)
    assert maxerr(z1, gold["fwd:main_logit"]) < 3e-2
    with pytest.raises(NotImplementedError):
        train_synthetic.build_teacher_from_ckpt({"args": dict(args, perceiver_type="dual_patch"), "model": {}}, d_static=DS, n_vars=V,
                                                duett_ckpt="synthetic", n_timesteps=T, cxr_model_name_fallback="synthetic")
    torch.manual_seed(1)
    student = StudentModel(load_duett_backbone("synthetic", d_static_num=DS, d_time_series_num=V, n_timesteps=T, freeze=False),
                           pool="mean", head_hidden=128, head_dropout=0.0).to(DEV)
    sopt = FusedAdamW([p for p in student.parameters() if p.requires_grad], lr=1e-3, weight_decay=5e-2)
    before = {k: v.detach().clone() for k, v in rebuilt.state_dict().items()}
    outs = [engine.train_student_batch(tb, tb, student, rebuilt, StudentKDLoss("vanilla_kl", 4.0, 0.5), sopt, torch.device(DEV)) for _ in range(3)]
    assert outs[-1]["loss"] < outs[0]["loss"] and np.isfinite(outs[-1]["kd"])
    for k, v in rebuilt.state_dict().items():
        assert torch.equal(v, before[k]), k
