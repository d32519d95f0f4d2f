"""fp32 CPU restatement of one full training step of the hot path (the caller side):
`train_teacher_dual_pathology_batch` (training_duett/engine.py:135-190) and `train_student_batch` (:270-301), from a
state_dict keyed like the reference's TeacherModel / StudentModel.  Dropout/augmentation off (parity form).
Used by tests, by `__graft_entry__.smoke()` and by bench.py's `cpu_baseline` leg only."""
from __future__ import annotations

import torch

from . import duett_ref, fusion_ref, losses_ref, optim_ref, vit_ref


def split_teacher_sd(sd):
    dsd = {k[len("duett."):]: v for k, v in sd.items() if k.startswith("duett.")}
    vsd = {k[len("cxr.backbone."):]: v for k, v in sd.items() if k.startswith("cxr.backbone.")}
    return dsd, vsd


def teacher_forward(sd, dcfg, vcfg, batch, n_heads=4, **kw):
    """TeacherModel.forward, patch-dual branch (model file :1075-1129); DuETT and CXR encoders frozen (no grad)."""
    dsd, vsd = split_teacher_sd(sd)
    xin = duett_ref.feats_to_input((batch["x_ts"], batch["x_static"], list(batch["bin_ends"])), max_len=dcfg.n_timesteps)
    with torch.no_grad():
        ts_tokens = duett_ref.encode(dsd, dcfg, xin)
        _, patches = vit_ref.vit_forward(vsd, vcfg, batch["pixel_values"])
    return fusion_ref.teacher_fusion_forward(sd, ts_tokens, patches, n_heads, **kw)


def teacher_step(sd, dcfg, vcfg, batch, opt_state, lr_of, weight_decay=5e-2, alphas=(0.5, 0.5, 1.0), n_heads=4):
    """One optimiser step in place on `sd`.  `opt_state`: {"step": int, "m": {}, "v": {}}; `lr_of(name) -> lr`."""
    train = {k: v for k, v in sd.items() if v.is_floating_point() and not k.startswith(("duett.", "cxr."))}
    for v in train.values():
        v.requires_grad_(True)
        v.grad = None
    out = teacher_forward(sd, dcfg, vcfg, batch, n_heads)
    K = batch["y_multi"].shape[1]
    L = losses_ref.dual_pathology_loss(out["img_logits"], out["ts_logits"], out["fusion_logits"], batch["y_multi"],
                                       batch["y_multi_mask"], torch.ones(K), None, *alphas)
    L["total"].backward()
    opt_state["step"] += 1
    with torch.no_grad():
        for k, p in train.items():
            if p.grad is None:
                continue
            m = opt_state["m"].setdefault(k, torch.zeros_like(p))
            v = opt_state["v"].setdefault(k, torch.zeros_like(p))
            optim_ref.adamw_step(p, p.grad, m, v, opt_state["step"], lr_of(k), weight_decay=weight_decay)
    for v in train.values():
        v.requires_grad_(False)
    return {"loss": float(L["total"]), "out": {k: v.detach() for k, v in out.items()}, "losses": L}


def student_step(ssd, dcfg, batch, z_t, opt_state, lr_of, weight_decay=5e-2, T=4.0, alpha=0.5, pool="mean"):
    """train_student_batch: student in train mode (BatchNorm batch statistics), KD loss, AdamW on every parameter that got a grad."""
    train = {k: v for k, v in ssd.items() if v.is_floating_point() and "running_" not in k}
    for v in train.values():
        v.requires_grad_(True)
        v.grad = None
    xin = duett_ref.feats_to_input((batch["x_ts"], batch["x_static"], list(batch["bin_ends"])), max_len=dcfg.n_timesteps)
    z_s = duett_ref.student_forward(ssd, dcfg, xin, pool, training=True)
    L = losses_ref.student_kd_loss(z_s, z_t, batch["y"], T, alpha)
    L["total"].backward()
    opt_state["step"] += 1
    with torch.no_grad():
        for k, p in train.items():
            if p.grad is None:
                continue
            m = opt_state["m"].setdefault(k, torch.zeros_like(p))
            v = opt_state["v"].setdefault(k, torch.zeros_like(p))
            optim_ref.adamw_step(p, p.grad, m, v, opt_state["step"], lr_of(k), weight_decay=weight_decay)
    for v in train.values():
        v.requires_grad_(False)
    return {"loss": float(L["total"]), "z_s": z_s.detach(), "losses": L}
