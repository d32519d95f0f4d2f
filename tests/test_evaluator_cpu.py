"""The product's metric code (evaluator.py, numpy on the host — as in the reference) against the reference evaluator's own
output table (golden fixture) and against scikit-learn."""
import json
import os

import numpy as np
import pytest

from helpers import GOLDEN_DIR, load_npz
from multimodal_edema_prediction_amd import evaluator as ev


def test_dual_pathology_table_matches_reference_evaluator():
    gold = load_npz("evaluator_table.npz")
    keys = json.load(open(os.path.join(GOLDEN_DIR, "evaluator_keys.json")))
    K = gold["y"].shape[1]
    res = ev.dual_pathology_table(gold["img"], gold["ts"], gold["fus"], gold["y"], gold["mask"], gold["corr"],
                                  np.linspace(0.5, 1.5, K, dtype=np.float32), tuple(f"l{k}" for k in range(K)))
    table = np.array([[float(r[k]) for k in keys] for r in res["per_label"]])
    np.testing.assert_allclose(table, gold["per_label"], rtol=1e-6, atol=1e-7, equal_nan=True)
    np.testing.assert_allclose(res["main_auroc"], gold["main_auroc"], rtol=1e-9)
    np.testing.assert_allclose(res["main_auprc"], gold["main_auprc"], rtol=1e-9)


def test_auroc_ap_against_sklearn_with_ties_and_degenerate_labels():
    from sklearn.metrics import average_precision_score, roc_auc_score
    rng = np.random.default_rng(0)
    for n in (2, 17, 500):
        y = (rng.random(n) < 0.3).astype(float)
        s = np.round(rng.normal(size=n), 1)          # many ties
        if 0 < y.sum() < n:
            assert abs(ev.auroc(y, s) - roc_auc_score(y, s)) < 1e-12
            assert abs(ev.average_precision(y, s) - average_precision_score(y, s)) < 1e-12
    assert np.isnan(ev.auroc(np.zeros(5), np.arange(5.0))) and np.isnan(ev.auroc(np.ones(5), np.arange(5.0)))
    assert ev.average_precision(np.zeros(5), np.arange(5.0)) == 0.0
    assert np.isnan(ev._pearson(np.ones(4), np.arange(4.0)))
