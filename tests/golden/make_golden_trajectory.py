#!/usr/bin/env python3
"""Golden fixture for `LocalTrajectoryEncoder` (models/main_architecture_duett.py:1242-1391): runs the
REFERENCE'S OWN CLASS (stubs for the absent lightning / torchmetrics / x_transformers as in make_golden.py; the class itself is
plain torch) on seeded synthetic weights and inputs and stores inputs, weights, tokens, padding mask and the gradient of a fixed
linear functional of the tokens with respect to every parameter.  Build container only.

Usage:  python tests/golden/make_golden_trajectory.py"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import REF, install_stubs, save  # noqa: E402


def synth_inputs(B, T, V, seed):
    g = torch.Generator().manual_seed(seed)
    vals = torch.randn(B, T, V, generator=g)
    counts = torch.poisson(torch.full((B, T, V), 0.7), generator=g)
    counts[:, :, 0] = 0.0                     # a variable that is never observed
    counts[0, :, 1] = 0.0
    counts[:, -6:, 2] = 0.0                   # nothing in the most recent window
    counts[1, 3, 3] = -1.0                    # negative count (clamped by the module)
    return [torch.cat([vals[b], counts[b]], dim=1) for b in range(B)]


def main():
    torch.set_num_threads(4)
    install_stubs()
    sys.path.insert(0, REF)
    from models.main_architecture_duett import LocalTrajectoryEncoder

    out = {}
    for tag, (B, T, V, d, windows) in {"a": (3, 24, 5, 32, (6, 12, 24)), "b": (2, 24, 7, 128, (6, 12, 24))}.items():
        torch.manual_seed(100 + ord(tag))
        m = LocalTrajectoryEncoder(n_vars=V, n_timesteps=T, d_model=d, n_layers=1, dropout=0.1, recency_windows=windows)
        with torch.no_grad():
            for k, p in m.named_parameters():             # livelier than the defaults (LayerNorm 1/0, small embeddings)
                if p.ndim == 1:
                    p.add_(0.1 * torch.randn_like(p))
        m.eval()                                          # dropout off: the parity form
        xs = synth_inputs(B, T, V, seed=7 + ord(tag))
        tokens, pad = m(tuple(xs), return_padding_mask=True)
        g = torch.Generator().manual_seed(55)
        wgt = torch.randn(tokens.shape, generator=g)
        (tokens * wgt).sum().backward()
        out[f"{tag}_x"] = torch.stack(xs)
        out[f"{tag}_tokens"] = tokens
        out[f"{tag}_pad"] = pad
        out[f"{tag}_wgt"] = wgt
        out[f"{tag}_cfg"] = np.array([B, T, V, d] + list(windows))
        for k, p in m.named_parameters():
            out[f"{tag}_p_{k}"] = p
            out[f"{tag}_g_{k}"] = p.grad
    save("trajectory.npz", **out)


if __name__ == "__main__":
    main()
