"""FP32 KERNEL MODE (functional.set_precision("fp32"); SURVEY.md §7 "always keep an fp32 kernel mode for tight checks"): the
op-level (trainable) paths with fp32 GEMM operands, checked at §8(d)'s fp32 tolerances — logits <= 1e-4 abs, loss <= 1e-5
rel, gradients ELEMENT-WISE — against the CPU oracle and the reference's fixtures:
  * the student (DuETT trained end to end, BatchNorm batch statistics): the whole model runs op-level, so the comparison is
    end to end, fixtures of the reference's own `train_student_batch` included;
  * the WHOLE teacher from the pixels on: the frozen CXR encoder runs its fp32 form too (cxr_train.forward_fp32: fp32 GEMMs, fp32
    small-attention kernel), DuETT its op-level form in fp32."""
import math
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

from multimodal_edema_prediction_amd import functional as Fn  # noqa: E402

DEV = "cuda"


def _close(got, want, rtol, atol, what):
    got, want = got.detach().double().cpu(), torch.as_tensor(want).double()
    err = (got - want).abs()
    tol = atol + rtol * want.abs()
    assert bool((err <= tol).all()), f"{what}: max err {float(err.max()):.3e} at |ref| {float(want.abs().max()):.3e}"


def _grad_close(g, r, what):
    """element-wise: |g - r| <= 2e-4 * max|r| + 1e-7 for every element (fp32 sums of ~1e3 terms in two different orders)"""
    g, r = g.detach().double().cpu(), r.detach().double()
    scale = float(r.abs().max())
    err = float((g - r).abs().max())
    assert err <= 2e-4 * scale + 1e-7, f"{what}: max |dg| {err:.3e}, max |g| {scale:.3e}"


@pytest.mark.parametrize("M,N,K", [(257, 768, 768), (300, 72, 2328), (130, 7, 96), (64, 24, 408)])
def test_fp32_gemms(M, N, K):
    g = torch.Generator().manual_seed(M)
    a, w = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) / math.sqrt(K)
    bias, scale, res = torch.randn(N, generator=g), 1 + 0.1 * torch.randn(N, generator=g), torch.randn(M, N, generator=g)
    ad, wd = a.to(DEV), w.to(DEV)
    with Fn.precision_mode("fp32"):
        y = Fn.gemm(ad, wd, bias=bias.to(DEV), scale=scale.to(DEV), residual=res.to(DEV), act=1)
        dy = torch.randn(M, N, generator=g)
        dw = Fn.gemm_tn(dy.to(DEV), ad)
    want = torch.nn.functional.gelu(a.double() @ w.double().T + bias.double()) * scale.double() + res.double()
    _close(y, want, 2e-6, 2e-5, "gemm_f32_nt")              # K fp32 FMAs per output: ~sqrt(K) * 2^-24 * |a||w|
    _close(dw, dy.double().T @ a.double(), 5e-6, 5e-6 * math.sqrt(M), "gemm_f32_tn")
    assert Fn.precision() == "bf16"                     # the context manager restores the mode


def test_student_fp32_end_to_end():
    from helpers import load_npz, load_shapes, synth_state_dict, t
    from multimodal_edema_prediction_amd.cohort import CohortCfg, make_batch
    from multimodal_edema_prediction_amd.losses_duett import StudentKDLoss
    from multimodal_edema_prediction_amd.main_architecture_duett import DuettFeatureExtractor, StudentModel
    from oracle import duett_ref, losses_ref
    import json
    from helpers import GOLDEN_DIR
    META = json.load(open(os.path.join(GOLDEN_DIR, "meta.json")))
    B, T, V, DS = META["B"], META["T"], META["V"], META["DS"]
    gold = load_npz("student_step_cfg1.npz")
    shapes = load_shapes("shapes.json")
    bb = DuettFeatureExtractor(d_static_num=DS, d_time_series_num=V, d_target=1, pretrain=False, masked_transform_timesteps=T, max_len=T,
                               aug_noise=0.0, aug_mask=0.0, transformer_dropout=0.0)
    student = StudentModel(bb, pool="mean", head_hidden=128, head_dropout=0.0)
    sd = synth_state_dict(shapes["student"], seed=2)
    student.load_state_dict(sd, strict=True)
    student = student.to(DEV).train()
    tb = make_batch(CohortCfg(n_timesteps=T, n_vars=V, d_static=DS, image_size=224, seed=META["cohort_seed"]), META["teacher_batch_start"], B,
                    mode="student")
    z_t, y = t(gold["z_t"]), tb["y"]
    assert torch.equal(y, t(gold["y"]))
    with Fn.precision_mode("fp32"):
        z_s = student(tuple(x.to(DEV) for x in tb["x_ts"]), tuple(x.to(DEV) for x in tb["x_static"]), tuple(x.to(DEV) for x in tb["bin_ends"]))
        L = StudentKDLoss("vanilla_kl", 4.0, 0.5)(z_s, z_t.to(DEV), y.to(DEV))
        student.zero_grad()
        L["total"].backward()
    # oracle, same weights, autograd through the DuETT restatement
    ref_sd = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and "running_" not in k else v.clone()) for k, v in sd.items()}
    dcfg = duett_ref.DuettCfg(d_static_num=DS, d_time_series_num=V, n_timesteps=T)
    xin = duett_ref.feats_to_input((tb["x_ts"], tb["x_static"], list(tb["bin_ends"])), max_len=T)
    zr = duett_ref.student_forward(ref_sd, dcfg, xin, "mean", training=True)
    Lr = losses_ref.student_kd_loss(zr, z_t, y, 4.0, 0.5)
    Lr["total"].backward()
    _close(z_s, zr.detach(), 0.0, 1e-4, "student logits (fp32 mode)")
    assert abs(float(L["total"]) - float(Lr["total"])) <= 1e-5 * abs(float(Lr["total"]))
    # ... and the numbers the REFERENCE's own train_student_batch produced (tests/golden/make_golden.py)
    _close(z_s, gold["z_s_train"], 0.0, 1e-4, "student logits vs the reference's")
    assert abs(float(L["total"]) - float(gold["total"])) <= 1e-5 * abs(float(gold["total"]))
    named = dict(student.named_parameters())
    for key in gold:
        if key.startswith("grad:"):
            _grad_close(named[key[5:]].grad, t(gold[key]), "reference " + key)
    n_checked = 0
    for k, p in student.named_parameters():
        r = ref_sd[k]
        if p.grad is None:
            assert r.grad is None, k
            continue
        _grad_close(p.grad, r.grad, k)
        n_checked += 1
    assert n_checked > 100


def test_teacher_fp32_from_the_pixels():
    import test_gpu_model as Tm
    from multimodal_edema_prediction_amd import engine
    from multimodal_edema_prediction_amd.losses_duett import DualPathologyLoss
    from oracle import duett_ref, fusion_ref, losses_ref
    from oracle.step_ref import split_teacher_sd
    teacher = Tm.build_teacher()
    sd = {k: v.detach().float().cpu().clone() for k, v in teacher.state_dict().items()}
    tb = Tm.make_batch(Tm.CCFG, Tm.META["teacher_batch_start"], Tm.B, mode="teacher")
    loss_fn = DualPathologyLoss(torch.ones(Tm.K), None, 0.5, 0.5, 1.0).to(DEV)
    engine._set_train_with_frozen_eval(teacher)
    b = engine._move_lists(tb, DEV)
    with Fn.precision_mode("fp32"):
        tok = teacher.cxr.forward_bf16(b["pixel_values"])                 # fp32 tokens in this mode
        assert tok.dtype == torch.float32
        out = teacher(b["x_ts"], b["x_static"], b["bin_ends"], b["pixel_values"])
        L = loss_fn(out["img_logits"], out["ts_logits"], out["fusion_logits"], b["y_multi"], b["y_multi_mask"])
        teacher.zero_grad()
        L["total"].backward()
    train = {k: v for k, v in sd.items() if v.is_floating_point() and not k.startswith(("duett.", "cxr."))}
    for v in train.values():
        v.requires_grad_(True)
    dsd, vsd = split_teacher_sd(sd)
    xin = duett_ref.feats_to_input((tb["x_ts"], tb["x_static"], list(tb["bin_ends"])), max_len=Tm.T)
    with torch.no_grad():
        ts_tokens = duett_ref.encode(dsd, duett_ref.DuettCfg(d_static_num=Tm.DS, d_time_series_num=Tm.V, n_timesteps=Tm.T), xin)
    from oracle import vit_ref
    with torch.no_grad():
        _, patches = vit_ref.vit_forward(vsd, vit_ref.VitCfg(), tb["pixel_values"])          # the oracle's own encoder, from the pixels
    tok_err = float((tok.cpu()[:, 1:] - patches).abs().max())
    assert tok_err <= 2e-4 * float(patches.abs().max()), tok_err                             # 12 fp32 layers, two summation orders
    ref = fusion_ref.teacher_fusion_forward(sd, ts_tokens, patches, 4)
    Lr = losses_ref.dual_pathology_loss(ref["img_logits"], ref["ts_logits"], ref["fusion_logits"], tb["y_multi"], tb["y_multi_mask"],
                                        torch.ones(Tm.K), None, 0.5, 0.5, 1.0)
    Lr["total"].backward()
    for k in ("img_logits", "ts_logits", "fusion_logits", "scaled_correction"):
        _close(out[k], ref[k].detach(), 0.0, 1e-4, k)
    assert abs(float(L["total"]) - float(Lr["total"])) <= 1e-5 * abs(float(Lr["total"]))
    named = dict(teacher.named_parameters())
    n = 0
    for k, r in train.items():
        if r.grad is None:
            continue
        _grad_close(named[k].grad, r.grad, k)
        n += 1
    assert n >= 60
