"""BASELINE metric, second half: "AUROC parity vs CPU ref" — the experiment SURVEY.md §8(d) / BASELINE.md §3 define: the same
teacher (same weights) is TRAINED for 200 identical steps (dropout / augmentation off) on a learnable synthetic cohort twice —
by the HIP engine step (bf16 MFMA operands, fused AdamW) and by the fp32 CPU oracle step — and both models then score the same
1536 held-out items.  Contract tolerances, asserted as written there: per-label AUROC within 0.005, macro AUROC within 0.003.
The frozen encoders' tokens of the oracle side are computed ONCE per batch on the CPU (by the oracle's own encoders) and reused
over the 200 steps — they do not change — and the images are 112x112 (65 ViT tokens) so that the CPU side stays in budget."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pytestmark = pytest.mark.gpu


def run_parity(lr=5e-4, n_steps=200, n_train_b=16, n_eval_b=32):
    """Train twice (HIP engine step / fp32 CPU oracle step), score the held-out items twice; returns the diagnostics."""
    from multimodal_edema_prediction_amd import engine, evaluator
    from multimodal_edema_prediction_amd.cohort import CohortCfg, make_batch
    from multimodal_edema_prediction_amd.losses_duett import DualPathologyLoss
    from multimodal_edema_prediction_amd.main_architecture_duett import (CXREncoder, PatchDualPathologyPerceiver, TeacherModel,
                                                                           load_duett_backbone)
    from multimodal_edema_prediction_amd.optim import FusedAdamW
    from oracle import duett_ref, fusion_ref, losses_ref, optim_ref, vit_ref
    from oracle.step_ref import split_teacher_sd

    dev = torch.device("cuda")
    T, V, DS, K, B, IMG = 32, 16, 8, 7, 16, 112
    torch.manual_seed(0)
    backbone = load_duett_backbone("synthetic", d_static_num=DS, d_time_series_num=V, n_timesteps=T, freeze=True)
    cxr = CXREncoder("synthetic", freeze=True)
    per = PatchDualPathologyPerceiver(K, backbone.d_representation, dropout=0.0, head_dropout=0.0)
    torch.nn.init.normal_(per.correction_head[-1].weight, std=0.05)
    teacher = TeacherModel(backbone, cxr, per, cxr_return_patches=True, d_img=768, use_aux_cxr=False,
                           patch_dual_pathology_mode=True).to(dev)
    sd = {k: v.detach().float().cpu().clone() for k, v in teacher.state_dict().items()}
    ccfg = CohortCfg(n_timesteps=T, n_vars=V, d_static=DS, image_size=IMG, n_labels=K, learnable=True)
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    train_b = [make_batch(ccfg, i * B, B, mode="teacher") for i in range(n_train_b)]
    eval_b = [make_batch(ccfg, 10_000 + i * B, B, mode="teacher") for i in range(n_eval_b)]

    # ---- HIP: the engine step ------------------------------------------------------------------------------------------
    loss_fn = DualPathologyLoss(torch.ones(K)).to(dev)
    opt = FusedAdamW([p for p in teacher.parameters() if p.requires_grad], lr=lr, weight_decay=5e-2)
    hip_losses = [engine.train_teacher_dual_pathology_batch(train_b[s % n_train_b], teacher, loss_fn, opt, dev)["loss"]
                  for s in range(n_steps)]
    teacher.eval()
    hip_logits = []
    with torch.no_grad():
        for b in eval_b:
            bb = engine._move_lists(b, dev)
            hip_logits.append(teacher(bb["x_ts"], bb["x_static"], bb["bin_ends"], bb["pixel_values"])["fusion_logits"].float().cpu())
    hip_logits = torch.cat(hip_logits).numpy()

    # ---- CPU oracle: frozen encoders evaluated once per batch, then the same optimiser steps on the fusion head ---------
    dcfg, vcfg = duett_ref.DuettCfg(d_static_num=DS, d_time_series_num=V, n_timesteps=T), vit_ref.VitCfg()
    dsd, vsd = split_teacher_sd(sd)

    def encoders(batch):
        with torch.no_grad():
            xin = duett_ref.feats_to_input((batch["x_ts"], batch["x_static"], list(batch["bin_ends"])), max_len=T)
            return duett_ref.encode(dsd, dcfg, xin), vit_ref.vit_forward(vsd, vcfg, batch["pixel_values"])[1]

    enc_train = [encoders(b) for b in train_b]
    train = {k: v for k, v in sd.items() if v.is_floating_point() and not k.startswith(("duett.", "cxr."))}
    state = {"step": 0, "m": {}, "v": {}}
    ref_losses = []
    for s in range(n_steps):
        b, (ts_tok, patches) = train_b[s % n_train_b], enc_train[s % n_train_b]
        for v in train.values():
            v.requires_grad_(True)
            v.grad = None
        out = fusion_ref.teacher_fusion_forward(sd, ts_tok, patches, 4)
        L = losses_ref.dual_pathology_loss(out["img_logits"], out["ts_logits"], out["fusion_logits"], b["y_multi"],
                                           b["y_multi_mask"], torch.ones(K), None, 0.5, 0.5, 1.0)
        L["total"].backward()
        state["step"] += 1
        with torch.no_grad():
            for k, p in train.items():
                if p.grad is not None:
                    m = state["m"].setdefault(k, torch.zeros_like(p))
                    v2 = state["v"].setdefault(k, torch.zeros_like(p))
                    optim_ref.adamw_step(p, p.grad, m, v2, state["step"], lr, weight_decay=5e-2)
        for v in train.values():
            v.requires_grad_(False)
        ref_losses.append(float(L["total"].detach()))
    ref_logits = []
    with torch.no_grad():
        for b in eval_b:
            ts_tok, patches = encoders(b)
            ref_logits.append(fusion_ref.teacher_fusion_forward(sd, ts_tok, patches, 4)["fusion_logits"])
    ref_logits = torch.cat(ref_logits).numpy()

    y = torch.cat([b["y_multi"] for b in eval_b]).numpy()
    mk = torch.cat([b["y_multi_mask"] for b in eval_b]).numpy() > 0
    a_hip, a_ref = [], []
    for k in range(K):
        yy = y[mk[:, k], k]
        if yy.min() == yy.max():
            continue
        a_hip.append(evaluator.auroc(yy, hip_logits[mk[:, k], k]))
        a_ref.append(evaluator.auroc(yy, ref_logits[mk[:, k], k]))
    return {"hip_losses": np.array(hip_losses), "ref_losses": np.array(ref_losses), "hip_logits": hip_logits, "ref_logits": ref_logits,
            "a_hip": np.array(a_hip), "a_ref": np.array(a_ref), "n_eval": len(y), "K": K}


def test_trained_teacher_auroc_matches_cpu_oracle():
    # The contract point is the REFERENCE'S OWN learning rate, 8e-5 (run.py; ADVICE r2: round 2 asserted at 5e-5).  The sweep
    # (tools/auroc_sweep.py, profiles/r02_auroc_sweep.txt) brackets it: over 512 training items the two runs stay one trajectory at
    # 5e-5 (max per-label difference 0.0009) and at 1e-4 (0.0042); at 2e-4 / 5e-4 over 256 items the run memorises the pool, AdamW
    # amplifies bf16 rounding into different minima and the held-out AUROCs of BOTH runs are noise around 0.5 (differences 0.15 / 0.47)
    # — the bound holds on the low-LR side up to ~1e-4, BASELINE.md states that range.
    # 1536 held-out items (the contract asks for >= 512): at the cohort's 2 % prevalences 512 items hold one or two positives of
    # the rare labels, an AUROC that moves by 0.002 per rank of a single item — noise, not parity
    r = run_parity(lr=8e-5, n_steps=200, n_train_b=32, n_eval_b=96)          # 512 training items, 1536 held-out items
    hl, rl, a_hip, a_ref = r["hip_losses"], r["ref_losses"], r["a_hip"], r["a_ref"]
    print("per-label AUROC hip", np.round(a_hip, 4), "oracle", np.round(a_ref, 4), "max |logit diff|",
          float(np.abs(r["hip_logits"] - r["ref_logits"]).max()), "loss first/last", rl[:3], rl[-3:])
    assert len(a_hip) == r["K"] and r["n_eval"] == 1536
    assert np.mean(rl[-16:]) < 0.9 * np.mean(rl[:16])                     # a real training run, not a flat line
    np.testing.assert_allclose(hl, rl, rtol=3e-2, atol=2e-2)                 # one trajectory, all 200 steps
    corr = float(np.corrcoef(r["hip_logits"].ravel(), r["ref_logits"].ravel())[0, 1])
    assert corr > 0.99, corr
    assert float(np.mean(a_ref)) > 0.58                                      # the cohort is learnable and was (partly) learnt
    assert float(np.max(np.abs(a_hip - a_ref))) <= 0.005, (a_hip, a_ref)     # BASELINE.md §3 / SURVEY §8(d): per label
    assert abs(float(np.mean(a_hip)) - float(np.mean(a_ref))) <= 0.003, (a_hip, a_ref)      # macro
