"""Metric side of the path: host-side mirror of the reference's `training_duett/evaluator.py` (`evaluate_binary`,
`make_*_forward`, `evaluate_dual_pathology`, `_bce_per_sample`, `_pearson`).  The forward passes run on the HIP path; the
metrics themselves are CPU/numpy in the reference too (sklearn) — here AUROC / AUPRC are restated in numpy (rank statistic
with tie averaging; step-wise precision–recall sum) and pinned against scikit-learn and against the reference evaluator's
own output table in the tests.  `gather=True` all-gathers logits over ranks first (the reference evaluates only rank 0's
shard, evaluator.py:19-37 — SURVEY.md §8e item 4)."""
from __future__ import annotations

import math

import numpy as np
import torch

from . import dp
from .engine import _move_lists


def auroc(y, score) -> float:
    y = np.asarray(y).astype(bool)
    s = np.asarray(score, dtype=np.float64)
    n_pos, n_neg = int(y.sum()), int((~y).sum())
    if n_pos == 0 or n_neg == 0:
        return float("nan")                       # sklearn raises ValueError -> the reference maps it to NaN (:29-32)
    order = np.argsort(s, kind="mergesort")
    ss = s[order]
    # average ranks over ties
    boundaries = np.r_[0, np.flatnonzero(np.diff(ss)) + 1, len(ss)]
    ranks_sorted = np.empty(len(ss))
    for a, b in zip(boundaries[:-1], boundaries[1:]):
        ranks_sorted[a:b] = 0.5 * (a + b - 1) + 1.0
    ranks = np.empty(len(ss))
    ranks[order] = ranks_sorted
    return float((ranks[y].sum() - n_pos * (n_pos + 1) / 2.0) / (n_pos * n_neg))


def average_precision(y, score) -> float:
    y = np.asarray(y, dtype=np.float64)
    s = np.asarray(score, dtype=np.float64)
    if y.size == 0:
        return float("nan")
    if y.sum() == 0:
        return 0.0                                # sklearn: warning "No positive class", AP 0.0
    order = np.argsort(-s, kind="mergesort")
    y, s = y[order], s[order]
    idx = np.r_[np.flatnonzero(np.diff(s)), len(y) - 1]
    tps = np.cumsum(y)[idx]
    precision = tps / (1.0 + idx)
    recall = tps / tps[-1]
    return float(np.sum(np.diff(np.r_[0.0, recall]) * precision))


def _bce_per_sample(logits: np.ndarray, y: np.ndarray) -> np.ndarray:
    """evaluator.py:181-183."""
    return np.maximum(logits, 0) - logits * y + np.log1p(np.exp(-np.abs(logits)))


def _pearson(a: np.ndarray, b: np.ndarray) -> float:
    """evaluator.py:186-194."""
    if a.size < 2:
        return float("nan")
    if a.std() == 0 or b.std() == 0:
        return float("nan")
    return float(np.corrcoef(a, b)[0, 1])


@torch.no_grad()
def evaluate_binary(model, loader, device, forward_fn, gather: bool = False):
    """evaluator.py:10-37."""
    model.eval()
    logits_all, y_all = [], []
    for batch in loader:
        out = forward_fn(model, batch, device)
        logits_all.append(out["logits"])
        y_all.append(out["y"])
    logits, y = torch.cat(logits_all).float(), torch.cat(y_all).float()
    if gather:
        logits, y = dp.gather_for_eval(logits, y)
    y = y.cpu().numpy()
    probs = torch.sigmoid(logits.cpu()).numpy()
    return {"auroc": auroc(y, probs), "auprc": average_precision(y, probs), "n": len(y), "pos_frac": float(y.mean())}


def make_teacher_forward():
    """evaluator.py:40-61."""
    @torch.no_grad()
    def _fwd(teacher, batch, device):
        b = _move_lists(batch, device)
        out = teacher(b["x_ts"], b["x_static"], b["bin_ends"], b["pixel_values"])
        z = out["main_logit"] if isinstance(out, dict) else (out[0] if isinstance(out, tuple) else out)
        return {"logits": z, "y": b["y"]}
    return _fwd


def make_student_forward():
    """evaluator.py:75-87."""
    @torch.no_grad()
    def _fwd(student, batch, device):
        b = _move_lists(batch, device)
        return {"logits": student(b["x_ts"], b["x_static"], b["bin_ends"]), "y": b["y"]}
    return _fwd


def dual_pathology_table(img, ts, fus, y, mk, corr, beta_vec, pathology_labels) -> dict:
    """evaluator.py:270-335 on gathered numpy arrays."""
    K = len(pathology_labels)
    per_label = []
    for k in range(K):
        m = mk[:, k].astype(bool)
        yk = y[m, k]
        li, lt, lf = img[m, k], ts[m, k], fus[m, k]
        pi, pt, pf = (1.0 / (1.0 + np.exp(-l)) for l in (li, lt, lf))
        ai, at, af = auroc(yk, pi), auroc(yk, pt), auroc(yk, pf)
        ri, rt, rf = average_precision(yk, pi), average_precision(yk, pt), average_precision(yk, pf)
        mean_bce = lambda l: float(_bce_per_sample(l, yk).mean()) if yk.size else float("nan")
        img_bce, ts_bce, fus_bce = mean_bce(li), mean_bce(lt), mean_bce(lf)
        if corr is not None and yk.size:
            ck = corr[m, k]
            mean_abs_corr, corr_r = float(np.abs(ck).mean()), _pearson(ck, yk - pi)
        else:
            mean_abs_corr, corr_r = float("nan"), float("nan")
        per_label.append({
            "name": pathology_labels[k], "n_valid": int(m.sum()), "pos_frac": float(yk.mean()) if len(yk) else float("nan"),
            "img_auroc": ai, "ts_auroc": at, "fus_auroc": af, "gap_i2f": af - ai, "gap_t2f": af - at,
            "img_auprc": ri, "ts_auprc": rt, "fus_auprc": rf, "gap_i2f_pr": rf - ri, "gap_t2f_pr": rf - rt,
            "img_bce": img_bce, "ts_bce": ts_bce, "fus_bce": fus_bce, "delta_bce": fus_bce - img_bce,
            "mean_abs_corr": mean_abs_corr, "corr_residual": corr_r,
            "beta": float(beta_vec[k]) if beta_vec is not None else float("nan")})

    def _macro(key):
        vals = [r[key] for r in per_label if not (isinstance(r[key], float) and math.isnan(r[key]))]
        return sum(vals) / len(vals) if vals else float("nan")

    return {"labels": list(pathology_labels), "n": int(len(y)), "main_auroc": _macro("fus_auroc"),
            "main_auprc": _macro("fus_auprc"), "per_label": per_label}


@torch.no_grad()
def evaluate_dual_pathology(model, loader, device, pathology_labels, *, query_ref=None, gather: bool = False) -> dict:
    """evaluator.py:197-335."""
    model.eval()
    cols = {k: [] for k in ("img", "ts", "fus", "y", "mask", "corr")}
    has_correction = None
    for batch in loader:
        b = _move_lists(batch, device)
        out = model(b["x_ts"], b["x_static"], b["bin_ends"], b["pixel_values"])
        if not isinstance(out, dict) or "fusion_logits" not in out:
            raise RuntimeError("evaluate_dual_pathology needs a dual_pathology_mode teacher")
        cols["img"].append(out["img_logits"]); cols["ts"].append(out["ts_logits"]); cols["fus"].append(out["fusion_logits"])
        cols["y"].append(b["y_multi"]); cols["mask"].append(b["y_multi_mask"])
        if has_correction is None:
            has_correction = "scaled_correction" in out
        if has_correction:
            cols["corr"].append(out["scaled_correction"])
    keys = ["img", "ts", "fus", "y", "mask"] + (["corr"] if has_correction else [])
    tens = [torch.cat(cols[k]).float() for k in keys]
    if gather:
        tens = list(dp.gather_for_eval(*tens))
    arr = {k: t.cpu().numpy() for k, t in zip(keys, tens)}
    unwrapped = model.module if hasattr(model, "module") else model
    perceiver = getattr(unwrapped, "perceiver", None)
    beta_vec = perceiver.beta.detach().float().cpu().numpy() if perceiver is not None and hasattr(perceiver, "beta") else None
    return dual_pathology_table(arr["img"], arr["ts"], arr["fus"], arr["y"], arr["mask"], arr.get("corr"), beta_vec, pathology_labels)
