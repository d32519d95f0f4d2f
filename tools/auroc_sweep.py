#!/usr/bin/env python3
"""Diagnostic: the 200-step AUROC parity experiment (tests/test_gpu_auroc_parity.run_parity) at several learning rates /
training-pool sizes — prints per-label AUROC of both runs, their differences, and where the loss trajectories part."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from test_gpu_auroc_parity import run_parity

for lr, nb in ((5e-4, 16), (2e-4, 16), (1e-4, 32), (5e-5, 32)):
    r = run_parity(lr=lr, n_steps=200, n_train_b=nb, n_eval_b=32)
    d = np.abs(r["a_hip"] - r["a_ref"])
    rel = np.abs(r["hip_losses"] - r["ref_losses"]) / np.maximum(np.abs(r["ref_losses"]), 1e-6)
    first_bad = int(np.argmax(rel > 0.03)) if (rel > 0.03).any() else -1
    print(f"lr {lr:g} pool {nb * 16}: AUROC ref {np.round(r['a_ref'], 4)} hip {np.round(r['a_hip'], 4)}  max|d| {d.max():.4f} macro d "
          f"{abs(r['a_hip'].mean() - r['a_ref'].mean()):.4f}  max|dlogit| {np.abs(r['hip_logits'] - r['ref_logits']).max():.3f}  "
          f"loss {r['ref_losses'][:2]} -> {r['ref_losses'][-4:]}  first step >3% apart: {first_bad}", flush=True)
