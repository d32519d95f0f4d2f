"""Per-batch step functions: host-side mirror of the reference's `training_duett/engine.py` (same names, argument order,
return dicts).  These are the CALLERS of the hot path; the reference's own file runs unmodified against this package's
modules as well — this copy exists so bench.py / tests have the step arithmetic on the GPU box, where
/root/reference is absent."""
from __future__ import annotations

import torch

from . import autograd_ops as A  # noqa: F401


def _set_train_with_frozen_eval(teacher, accelerator=None):
    """engine.py:7-20: train() everywhere except sub-modules whose parameters are all frozen."""
    teacher.train()
    unwrapped = accelerator.unwrap_model(teacher) if accelerator is not None else getattr(teacher, "module", teacher)
    for attr in ("duett", "cxr", "pretrained_cxr_head"):
        mod = getattr(unwrapped, attr, None)
        if mod is None:
            continue
        params = list(mod.parameters(recurse=True))
        if params and not any(p.requires_grad for p in params):
            mod.eval()


def _move_lists(batch: dict, device) -> dict:
    """engine.py:23-36."""
    out = {"x_ts": tuple(t.to(device) for t in batch["x_ts"]), "x_static": tuple(t.to(device) for t in batch["x_static"]),
           "bin_ends": tuple(t.to(device) for t in batch["bin_ends"]), "y": batch["y"].to(device)}
    if "pixel_values" in batch:
        out["pixel_values"] = batch["pixel_values"].to(device)
    if "y_multi" in batch:
        out["y_multi"] = batch["y_multi"].to(device)
        out["y_multi_mask"] = batch["y_multi_mask"].to(device)
    return out


def _backward(loss, accelerator):
    if accelerator is not None:
        accelerator.backward(loss)
    else:
        loss.backward()


def train_teacher_dual_pathology_batch(batch, teacher, path_loss_fn, optimizer, device, accelerator=None,
                                       aux_residual_alpha: float = 0.0):
    """engine.py:135-190."""
    _set_train_with_frozen_eval(teacher, accelerator)
    b = _move_lists(batch, device)
    out = teacher(b["x_ts"], b["x_static"], b["bin_ends"], b["pixel_values"])
    if not isinstance(out, dict):
        raise RuntimeError("dual_pathology mode but TeacherModel did not return a dict")
    losses = path_loss_fn(out["img_logits"], out["ts_logits"], out["fusion_logits"], b["y_multi"], b["y_multi_mask"])
    total = losses["total"]
    aux_residual_loss = torch.zeros((), device=device)
    if aux_residual_alpha > 0.0 and "scaled_correction" in out:
        aux_residual_loss = A.aux_residual_kl(out["img_logits"], out["scaled_correction"], b["y_multi"], b["y_multi_mask"], 0.05)
        total = A.add_scaled(total, aux_residual_loss, aux_residual_alpha)
    optimizer.zero_grad()
    _backward(total, accelerator)
    optimizer.step()
    return {"loss": total.detach().item(), "img_total": losses["img_total"].item(), "ts_total": losses["ts_total"].item(),
            "fus_total": losses["fus_total"].item(), "aux_residual": float(aux_residual_loss.detach().item()),
            "img_per": losses["img_per"].cpu(), "ts_per": losses["ts_per"].cpu(), "fus_per": losses["fus_per"].cpu(),
            "main_logit": out["main_logit"].detach(), "img_logits": out["img_logits"].detach(),
            "ts_logits": out["ts_logits"].detach(), "fusion_logits": out["fusion_logits"].detach(), "y": b["y"].detach(),
            "y_multi": b["y_multi"].detach(), "y_multi_mask": b["y_multi_mask"].detach()}


def train_student_batch(batch_stu, batch_tea, student, teacher, kd_loss_fn, optimizer, device, accelerator=None):
    """engine.py:270-301."""
    student.train()
    teacher.eval()
    b_s = _move_lists(batch_stu, device)
    b_t = _move_lists(batch_tea, device)
    with torch.no_grad():
        z_t = teacher(b_t["x_ts"], b_t["x_static"], b_t["bin_ends"], b_t["pixel_values"])["main_logit"]
    z_s = student(b_s["x_ts"], b_s["x_static"], b_s["bin_ends"])
    losses = kd_loss_fn(z_s, z_t, b_s["y"])
    optimizer.zero_grad()
    _backward(losses["total"], accelerator)
    optimizer.step()
    return {"loss": losses["total"].detach().item(), "bce": losses["bce"].item(), "kd": losses["kd"].item(),
            "logits": z_s.detach(), "y": b_s["y"].detach()}


@torch.no_grad()
def eval_teacher_batch(batch, teacher, loss_fn, device):
    """engine.py:76-88."""
    teacher.eval()
    b = _move_lists(batch, device)
    out = teacher(b["x_ts"], b["x_static"], b["bin_ends"], b["pixel_values"])
    main_logit = out["main_logit"] if isinstance(out, dict) else (out[0] if isinstance(out, tuple) else out)
    loss = loss_fn(main_logit, b["y"].float())
    return {"loss": loss.item(), "logits": main_logit, "y": b["y"]}


@torch.no_grad()
def eval_student_batch(batch, student, device):
    """engine.py:304-309."""
    student.eval()
    b = _move_lists(batch, device)
    z = student(b["x_ts"], b["x_static"], b["bin_ends"])
    return {"logits": z, "y": b["y"]}


def train_teacher_dual_pathology_lp_batch(batch, teacher, path_loss_fn, optimizer, device, accelerator=None, beta_l2: float = 0.0,
                                          corr_l2: float = 0.0, aux_residual_alpha: float = 0.0):
    """engine.py:196-264: everything in eval() except perceiver.correction_head; total = fusion losses + beta_l2*mean(beta^2)
    + corr_l2*mean(scaled_correction^2) (+ aux KL)."""
    teacher.eval()
    unwrapped = accelerator.unwrap_model(teacher) if accelerator is not None else getattr(teacher, "module", teacher)
    unwrapped.perceiver.correction_head.train()
    b = _move_lists(batch, device)
    out = teacher(b["x_ts"], b["x_static"], b["bin_ends"], b["pixel_values"])
    if not isinstance(out, dict):
        raise RuntimeError("dual_pathology LP mode but TeacherModel did not return a dict")
    losses = path_loss_fn(out["img_logits"], out["ts_logits"], out["fusion_logits"], b["y_multi"], b["y_multi_mask"])
    total = losses["total"]
    reg_beta = torch.zeros((), device=device)
    reg_corr = torch.zeros((), device=device)
    aux_residual_loss = torch.zeros((), device=device)
    if beta_l2 > 0.0:
        reg_beta = A.sq_mean(unwrapped.perceiver.beta, beta_l2)
        total = A.add_scaled(total, reg_beta, 1.0)
    if corr_l2 > 0.0:
        reg_corr = A.sq_mean(out["scaled_correction"], corr_l2)
        total = A.add_scaled(total, reg_corr, 1.0)
    if aux_residual_alpha > 0.0 and "scaled_correction" in out:
        aux_residual_loss = A.aux_residual_kl(out["img_logits"], out["scaled_correction"], b["y_multi"], b["y_multi_mask"], 0.05)
        total = A.add_scaled(total, aux_residual_loss, aux_residual_alpha)
    optimizer.zero_grad()
    _backward(total, accelerator)
    optimizer.step()
    return {"loss": total.detach().item(), "img_total": losses["img_total"].item(), "ts_total": losses["ts_total"].item(),
            "fus_total": losses["fus_total"].item(), "img_per": losses["img_per"].cpu(), "ts_per": losses["ts_per"].cpu(),
            "fus_per": losses["fus_per"].cpu(), "reg_beta_l2": float(reg_beta.detach().item()),
            "reg_corr_l2": float(reg_corr.detach().item()), "aux_residual": float(aux_residual_loss.detach().item()),
            "main_logit": out["main_logit"].detach(), "img_logits": out["img_logits"].detach(), "ts_logits": out["ts_logits"].detach(),
            "fusion_logits": out["fusion_logits"].detach(), "y": b["y"].detach(), "y_multi": b["y_multi"].detach(),
            "y_multi_mask": b["y_multi_mask"].detach()}
