"""`best.pt` in the reference's on-disk layout (the checkpoint half): `{model, optimizer, epoch, metric, args}`
written by `_save_ckpt` (training_duett/trainer.py:63-71) whenever the validation macro-AUROC improves, and read back by the
linear-probe stage (:169-210), the student's teacher reconstruction (:770-822) and the analysis scripts.  The hot-path modules of
this package keep the reference's parameter names (`duett.* / cxr.* / perceiver.* / *_head.*`), so a file written here loads into
the reference's modules and the other way round.

Loading never unpickles arbitrary objects: `torch.load(..., weights_only=True)` — a checkpoint holds tensors, numbers, strings
and plain containers only (`args` is stored as a dict, as the reference does with `vars(args)`)."""
from __future__ import annotations

import os

import torch


def save_ckpt(path: str, model: torch.nn.Module, optimizer, epoch: int, metric: float, args) -> None:
    """trainer.py:63-71.  `args`: an argparse.Namespace (stored as `vars(args)`) or a mapping."""
    d = os.path.dirname(path)
    if d:
        os.makedirs(d, exist_ok=True)
    torch.save({
        "model": model.state_dict(),
        "optimizer": optimizer.state_dict(),
        "epoch": epoch,
        "metric": metric,
        "args": vars(args) if hasattr(args, "__dict__") else dict(args),
    }, path)


def load_ckpt(path: str, map_location="cpu") -> dict:
    """The whole dict, tensors on `map_location`; refuses anything but plain data."""
    state = torch.load(path, map_location=map_location, weights_only=True)
    missing = {"model", "optimizer", "epoch", "metric", "args"} - set(state)
    if missing:
        raise KeyError(f"{path}: not a trainer checkpoint, missing {sorted(missing)}")
    return state


def load_model_state(model: torch.nn.Module, state: dict, freeze: bool = False, strict: bool = True) -> torch.nn.Module:
    """`teacher.load_state_dict(teacher_state["model"])` (+ the freeze / eval the student's teacher gets, trainer.py:817-821)."""
    model.load_state_dict(state["model"], strict=strict)
    if freeze:
        for p in model.parameters():
            p.requires_grad = False
        model.eval()
    return model
