"""The CPU restatement of LocalTrajectoryEncoder (oracle/trajectory_ref.py) against the fixture the reference's own class
produced (tests/golden/trajectory.npz): tokens, padding mask and the gradient of every parameter."""
import os

import numpy as np
import pytest
import torch

from oracle import trajectory_ref as R

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "trajectory.npz")


def load(tag):
    z = np.load(GOLD)
    B, T, V, d, *windows = [int(v) for v in z[f"{tag}_cfg"]]
    sd = {k[len(tag) + 3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith(f"{tag}_p_")}
    grads = {k[len(tag) + 3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith(f"{tag}_g_")}
    return dict(B=B, T=T, V=V, d=d, windows=tuple(windows), sd=sd, grads=grads, x=torch.from_numpy(z[f"{tag}_x"]),
                tokens=torch.from_numpy(z[f"{tag}_tokens"]), pad=torch.from_numpy(z[f"{tag}_pad"]), wgt=torch.from_numpy(z[f"{tag}_wgt"]))


@pytest.mark.parametrize("tag", ["a", "b"])
def test_oracle_matches_the_reference_class(tag):
    g = load(tag)
    sd = {k: v.clone().requires_grad_(True) for k, v in g["sd"].items()}
    tokens, pad = R.forward(sd, g["x"], g["V"], g["windows"])
    assert tokens.shape == g["tokens"].shape == (g["B"], g["V"] * len(g["windows"]) + 1, g["d"])
    assert torch.equal(pad, g["pad"])
    assert (tokens - g["tokens"]).abs().max().item() <= 2e-5          # fp32, same arithmetic up to summation order
    (tokens * g["wgt"]).sum().backward()
    for k, ref in g["grads"].items():
        got = sd[k].grad
        assert got is not None, k
        tol = 2e-5 * max(1.0, ref.abs().max().item())
        assert (got - ref).abs().max().item() <= tol, (k, (got - ref).abs().max().item(), tol)


def test_time_since_last_observation_edge_cases():
    obs = torch.zeros(1, 5, 3, dtype=torch.bool)
    obs[0, 0, 1] = True; obs[0, 2, 1] = True; obs[0, 4, 2] = True
    out = R.time_since_last_observation(obs)[0]
    assert out[:, 0].tolist() == [1, 2, 3, 4, 5]          # never observed: keeps counting
    assert out[:, 1].tolist() == [1, 1, 2, 1, 2]          # resets AFTER an observed slot
    assert out[:, 2].tolist() == [1, 2, 3, 4, 5]
