"""Restatement of the optimiser side (reference
`training_duett/trainer.py:77-125,383,466`): name-pattern LR groups, AdamW update,
linear warm-up → cosine schedule (torch `SequentialLR(LinearLR, CosineAnnealingLR)`)."""
from __future__ import annotations

import math

import torch


def param_group_of(name: str) -> str:
    """trainer.py:88-102."""
    if name.startswith(("duett.", "cxr.")):
        return "backbone"
    if "correction_head" in name or name.endswith(".beta") or name == "beta":
        return "correction_head"
    if name.endswith("_queries"):
        return "pathology_queries"
    return "rest"


def group_lrs(lr, backbone_lr_mult=0.2, query_lr_mult=0.2, correction_lr_mult=1.0):
    """trainer.py:104-115."""
    return {"backbone": lr * backbone_lr_mult, "pathology_queries": lr * query_lr_mult,
            "correction_head": lr * correction_lr_mult, "rest": lr}


def lr_at(step: int, base_lr: float, total_steps: int, warmup_steps: int = 300, eta_min: float = 0.0) -> float:
    """LR in effect for optimiser step number `step` (0-based) under trainer.py:119-125:
    LinearLR(start 1e-4 → 1, total_iters=warmup) then CosineAnnealingLR(T_max, eta_min),
    the scheduler being stepped once after every optimiser step (trainer.py:466).
    NB eta_min = args.lr * min_lr_ratio for EVERY group (trainer.py:124)."""
    warmup = max(int(warmup_steps), 1)
    t_max = max(int(total_steps) - warmup, 1)
    if step < warmup:
        f = 1e-4 + (1.0 - 1e-4) * step / warmup
        return base_lr * f
    t = step - warmup
    return eta_min + (base_lr - eta_min) * (1 + math.cos(math.pi * t / t_max)) / 2


def adamw_step(p, g, m, v, step, lr, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=5e-2):
    """torch.optim.AdamW single-tensor update (decoupled decay), in place; `step` is 1-based."""
    p.mul_(1 - lr * weight_decay)
    m.mul_(beta1).add_(g, alpha=1 - beta1)
    v.mul_(beta2).addcmul_(g, g, value=1 - beta2)
    bc1 = 1 - beta1 ** step
    bc2 = 1 - beta2 ** step
    denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
    p.addcdiv_(m, denom, value=-lr / bc1)
