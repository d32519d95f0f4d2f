"""numpy fp64 restatement of the metric side (reference
`training_duett/evaluator.py:10-37,181-194,276-335`).  The reference calls scikit-learn's
`roc_auc_score` / `average_precision_score`; these are restated from their definitions and
pinned against scikit-learn 1.7.2 in tests/test_oracle_metrics.py."""
from __future__ import annotations

import math

import numpy as np


def auroc(y, score) -> float:
    """Mann–Whitney U with average ranks for ties; NaN when only one class is present
    (the reference maps sklearn's ValueError to NaN, evaluator.py:29-32)."""
    y = np.asarray(y).astype(bool)
    s = np.asarray(score, dtype=np.float64)
    n_pos, n_neg = int(y.sum()), int((~y).sum())
    if n_pos == 0 or n_neg == 0:
        return float("nan")
    order = np.argsort(s, kind="mergesort")
    ss = s[order]
    ranks = np.empty(len(s), dtype=np.float64)
    i = 0
    while i < len(ss):
        j = i
        while j + 1 < len(ss) and ss[j + 1] == ss[i]:
            j += 1
        ranks[order[i:j + 1]] = 0.5 * (i + j) + 1.0
        i = j + 1
    return float((ranks[y].sum() - n_pos * (n_pos + 1) / 2.0) / (n_pos * n_neg))


def average_precision(y, score) -> float:
    """Σ_n (R_n − R_{n−1}) P_n over distinct thresholds (sklearn definition)."""
    y = np.asarray(y).astype(np.float64)
    s = np.asarray(score, dtype=np.float64)
    if y.size == 0:
        return float("nan")
    if y.sum() == 0:
        return 0.0            # sklearn 1.7: "No positive class found", recall := 1, precision 0 → AP 0.0 (a warning, no raise)
    order = np.argsort(-s, kind="mergesort")
    y, s = y[order], s[order]
    distinct = np.where(np.diff(s))[0]
    idx = np.r_[distinct, len(y) - 1]
    tps = np.cumsum(y)[idx]
    fps = 1 + idx - tps
    precision = tps / (tps + fps)
    recall = tps / tps[-1]
    return float(np.sum(np.diff(np.r_[0.0, recall]) * precision))


def bce_per_sample(logits, y):
    """evaluator.py:181-183."""
    return np.maximum(logits, 0) - logits * y + np.log1p(np.exp(-np.abs(logits)))


def pearson(a, b) -> float:
    """evaluator.py:186-194."""
    if a.size < 2 or a.std() == 0 or b.std() == 0:
        return float("nan")
    return float(np.corrcoef(a, b)[0, 1])


def evaluate_binary(logits, y) -> dict:
    """evaluator.py:23-37 on already-gathered arrays."""
    p = 1.0 / (1.0 + np.exp(-np.asarray(logits, dtype=np.float32)))
    y = np.asarray(y, dtype=np.float32)
    return {"auroc": auroc(y, p), "auprc": average_precision(y, p), "n": len(y), "pos_frac": float(y.mean())}


def evaluate_dual_pathology(img, ts, fus, y, mask, corr=None, beta=None, labels=None) -> dict:
    """evaluator.py:248-335 on already-gathered [N,K] arrays."""
    K = y.shape[1]
    labels = list(labels) if labels is not None else [f"label_{k}" for k in range(K)]
    sig = lambda l: 1.0 / (1.0 + np.exp(-l))
    per = []
    for k in range(K):
        m = mask[:, k].astype(bool)
        yk = y[m, k]
        li, lt, lf = img[m, k], ts[m, k], fus[m, k]
        pi, pt, pf = sig(li), sig(lt), sig(lf)
        ai, at, af = auroc(yk, pi), auroc(yk, pt), auroc(yk, pf)
        ri, rt, rf = average_precision(yk, pi), average_precision(yk, pt), average_precision(yk, pf)
        nanmean = lambda l: float(bce_per_sample(l, yk).mean()) if yk.size else float("nan")
        if corr is not None and yk.size:
            ck = corr[m, k]
            mac, cr = float(np.abs(ck).mean()), pearson(ck, yk - pi)
        else:
            mac, cr = float("nan"), float("nan")
        per.append({"name": labels[k], "n_valid": int(m.sum()),
                    "pos_frac": float(yk.mean()) if len(yk) else float("nan"),
                    "img_auroc": ai, "ts_auroc": at, "fus_auroc": af, "gap_i2f": af - ai, "gap_t2f": af - at,
                    "img_auprc": ri, "ts_auprc": rt, "fus_auprc": rf, "gap_i2f_pr": rf - ri, "gap_t2f_pr": rf - rt,
                    "img_bce": nanmean(li), "ts_bce": nanmean(lt), "fus_bce": nanmean(lf),
                    "delta_bce": nanmean(lf) - nanmean(li), "mean_abs_corr": mac, "corr_residual": cr,
                    "beta": float(beta[k]) if beta is not None else float("nan")})

    def macro(key):
        v = [r[key] for r in per if not (isinstance(r[key], float) and math.isnan(r[key]))]
        return sum(v) / len(v) if v else float("nan")

    return {"labels": labels, "n": int(len(y)), "main_auroc": macro("fus_auroc"),
            "main_auprc": macro("fus_auprc"), "per_label": per}
