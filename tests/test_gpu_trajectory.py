"""LocalTrajectoryEncoder on the HIP kernels (trajectory.py / csrc/trajectory.hip) against the oracle (oracle/trajectory_ref.py)
and against the fixture the reference's own class produced (tests/golden/trajectory.npz, case "b": d_model 128).
Tolerances: features fp32-exact up to libm (1e-6); everything behind a bf16 MFMA operand (both Linears, the GRU recurrence over
24 steps): tokens (LayerNorm output, O(1)) <= 3e-2 abs, parameter gradients cosine >= 0.995 and <= 5 % of their max."""
import os

import numpy as np
import pytest
import torch

from oracle import trajectory_ref as R

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "trajectory.npz")


def gold(tag):
    z = np.load(GOLD)
    B, T, V, d, *windows = [int(v) for v in z[f"{tag}_cfg"]]
    sd = {k[len(tag) + 3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith(f"{tag}_p_")}
    grads = {k[len(tag) + 3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith(f"{tag}_g_")}
    return dict(B=B, T=T, V=V, d=d, windows=tuple(windows), sd=sd, grads=grads, x=torch.from_numpy(z[f"{tag}_x"]),
                tokens=torch.from_numpy(z[f"{tag}_tokens"]), pad=torch.from_numpy(z[f"{tag}_pad"]), wgt=torch.from_numpy(z[f"{tag}_wgt"]))


def build(g, dev):
    from multimodal_edema_prediction_amd.main_architecture_duett import LocalTrajectoryEncoder
    m = LocalTrajectoryEncoder(n_vars=g["V"], n_timesteps=g["T"], d_model=g["d"], recency_windows=g["windows"])
    assert sorted(m.state_dict()) == sorted(g["sd"]), "state_dict keys differ from the reference module's"
    m.load_state_dict(g["sd"], strict=True)
    return m.to(dev).eval()


def test_features_kernel_matches_the_oracle():
    from multimodal_edema_prediction_amd.trajectory import traj_features
    g = gold("b")
    ref, _ = R.local_features(g["x"], g["V"])
    got = traj_features(g["x"].cuda(), g["V"]).cpu()
    assert got.shape == (g["B"] * g["V"], g["T"], 8)
    assert torch.equal(got[..., 5:], torch.zeros_like(got[..., 5:]))
    assert (got[..., :5] - ref).abs().max().item() <= 1e-6


def test_gru_kernels_match_the_oracle_forward_and_backward():
    from multimodal_edema_prediction_amd.trajectory import GruFn
    torch.manual_seed(3)
    S, T, d = 37, 24, 128                       # ragged against the 16-sequence workgroups
    sd = {"temporal.weight_ih_l0": torch.eye(d).repeat(3, 1), "temporal.bias_ih_l0": torch.zeros(3 * d),
          "temporal.weight_hh_l0": torch.randn(3 * d, d) * 0.08, "temporal.bias_hh_l0": torch.randn(3 * d) * 0.1}
    gi = torch.randn(S, T, 3 * d)
    wgt = torch.randn(S, T, d)
    # oracle: feed gi directly (W_ih = identity blocks would need x = gi; write the recurrence out with gi given)
    w, b = sd["temporal.weight_hh_l0"].clone().requires_grad_(True), sd["temporal.bias_hh_l0"].clone().requires_grad_(True)
    gi_r = gi.clone().requires_grad_(True)
    h = torch.zeros(S, d); outs = []
    for t in range(T):
        gh = h @ w.t() + b
        r = torch.sigmoid(gi_r[:, t, :d] + gh[:, :d]); z = torch.sigmoid(gi_r[:, t, d:2 * d] + gh[:, d:2 * d])
        n = torch.tanh(gi_r[:, t, 2 * d:] + r * gh[:, 2 * d:]); h = (1 - z) * n + z * h
        outs.append(h)
    ref = torch.stack(outs, 1)
    (ref * wgt).sum().backward()
    gi_g = gi.cuda().requires_grad_(True); w_g = sd["temporal.weight_hh_l0"].cuda().requires_grad_(True)
    b_g = sd["temporal.bias_hh_l0"].cuda().requires_grad_(True)
    got = GruFn.apply(gi_g, w_g, b_g)
    (got * wgt.cuda()).sum().backward()
    assert (got.cpu() - ref).abs().max().item() <= 2e-2
    for name, a, r_ in (("dgi", gi_g.grad, gi_r.grad), ("dW_hh", w_g.grad, w.grad), ("db_hh", b_g.grad, b.grad)):
        a = a.cpu()
        cos = torch.nn.functional.cosine_similarity(a.flatten(), r_.flatten(), dim=0).item()
        assert cos >= 0.999, (name, cos)
        assert (a - r_).abs().max().item() <= 3e-2 * r_.abs().max().item(), (name, (a - r_).abs().max().item(), r_.abs().max().item())


def test_module_matches_the_reference_fixture_forward_and_backward():
    g = gold("b")
    m = build(g, "cuda")
    xs = tuple(t.cuda() for t in g["x"])
    tokens, pad = m(xs, return_padding_mask=True)
    assert torch.equal(pad.cpu(), g["pad"])
    assert (tokens.cpu() - g["tokens"]).abs().max().item() <= 3e-2
    (tokens * g["wgt"].cuda()).sum().backward()
    for k, p in m.named_parameters():
        ref = g["grads"][k]
        got = p.grad.cpu()
        cos = torch.nn.functional.cosine_similarity(got.flatten(), ref.flatten(), dim=0).item()
        assert cos >= 0.995, (k, cos)
        assert (got - ref).abs().max().item() <= 5e-2 * ref.abs().max().item(), (k, (got - ref).abs().max().item(), ref.abs().max().item())
    assert m(xs).shape == tokens.shape                                  # default call returns the tokens only


def test_cohort_size_against_the_oracle():
    torch.manual_seed(11)
    from multimodal_edema_prediction_amd.main_architecture_duett import LocalTrajectoryEncoder
    B, T, V, d = 8, 24, 48, 128
    m = LocalTrajectoryEncoder(n_vars=V, n_timesteps=T, d_model=d).cuda().eval()
    x = torch.cat([torch.randn(B, T, V), torch.poisson(torch.full((B, T, V), 0.5))], dim=2)
    tokens, pad = m(tuple(x.cuda()), return_padding_mask=True)
    sd = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    ref, rpad = R.forward(sd, x, V, m.recency_windows)
    assert torch.equal(pad.cpu(), rpad)
    assert (tokens.detach().cpu() - ref).abs().max().item() <= 3e-2


def test_errors_mirror_the_reference():
    from multimodal_edema_prediction_amd.main_architecture_duett import LocalTrajectoryEncoder
    with pytest.raises(ValueError):
        LocalTrajectoryEncoder(n_vars=4, n_timesteps=24, recency_windows=(6, 12))          # must end at n_timesteps
    m = LocalTrajectoryEncoder(n_vars=4, n_timesteps=24).cuda()
    with pytest.raises(ValueError):
        m((torch.zeros(23, 8, device="cuda"),))
    m64 = LocalTrajectoryEncoder(n_vars=4, n_timesteps=24, d_model=64).cuda()
    with pytest.raises(ValueError, match="hidden size"):
        m64((torch.zeros(24, 8, device="cuda"),))
