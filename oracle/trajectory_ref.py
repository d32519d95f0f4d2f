"""fp32 CPU restatement of `LocalTrajectoryEncoder` (reference models/main_architecture_duett.py:1242-1391).
TEST INFRASTRUCTURE ONLY (tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg): the product path never imports it.
Functional form over a state_dict keyed like the reference module; the GRU is written out step by step (torch.nn.GRU's
equations: r, z, n gate order, `n = tanh(W_in x + b_in + r * (W_hn h + b_hn))`, `h' = (1 - z) n + z h`).  Dropout off (parity
form).  Pinned by tests/golden/trajectory.npz, which the reference's own class produced (tests/golden/make_golden_trajectory.py)."""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F


def time_since_last_observation(observed: torch.Tensor) -> torch.Tensor:
    """:1316-1330 — elapsed grid steps before each slot; resets after an observed slot.  observed [B,T,V] bool -> fp32."""
    B, T, V = observed.shape
    elapsed = torch.zeros((B, V), dtype=torch.float32)
    out = torch.empty((B, T, V), dtype=torch.float32)
    for t in range(T):
        elapsed = elapsed + 1.0
        out[:, t, :] = elapsed
        elapsed = torch.where(observed[:, t, :], torch.zeros_like(elapsed), elapsed)
    return out


def local_features(x: torch.Tensor, n_vars: int):
    """:1342-1358 — x [B,T,2V] -> local [B*V, T, 5] (value, observed, log count / log 16, time since last / T, time to CXR / T)
    and observed [B,T,V]."""
    B, T, _ = x.shape
    values = x[:, :, :n_vars]
    counts = x[:, :, n_vars:].clamp_min(0.0)
    observed = counts > 0
    values = torch.where(observed, values, torch.zeros_like(values))
    log_count = torch.log1p(counts) / math.log(16.0)
    delta = time_since_last_observation(observed) / float(T)
    ttc = (torch.arange(T, 0, -1, dtype=x.dtype).view(1, T, 1).expand(B, -1, n_vars)) / float(T)
    local = torch.stack([values, observed.to(x.dtype), log_count, delta, ttc], dim=-1)
    return local.permute(0, 2, 1, 3).reshape(B * n_vars, T, 5), observed


def gru(sd, x: torch.Tensor) -> torch.Tensor:
    """One-layer batch-first GRU, h0 = 0 (:1366): x [S,T,d] -> all hidden states [S,T,d]."""
    w_ih, w_hh, b_ih, b_hh = sd["temporal.weight_ih_l0"], sd["temporal.weight_hh_l0"], sd["temporal.bias_ih_l0"], sd["temporal.bias_hh_l0"]
    S, T, d = x.shape
    gi = x @ w_ih.t() + b_ih
    h = torch.zeros((S, d), dtype=x.dtype)
    out = []
    for t in range(T):
        gh = h @ w_hh.t() + b_hh
        r = torch.sigmoid(gi[:, t, :d] + gh[:, :d])
        z = torch.sigmoid(gi[:, t, d:2 * d] + gh[:, d:2 * d])
        n = torch.tanh(gi[:, t, 2 * d:] + r * gh[:, 2 * d:])
        h = (1.0 - z) * n + z * h
        out.append(h)
    return torch.stack(out, dim=1)


def forward(sd, x: torch.Tensor, n_vars: int, windows=(6, 12, 24), eps: float = 1e-5):
    """LocalTrajectoryEncoder.forward(x_ts_list, return_padding_mask=True) (:1332-1391) with x = stack(x_ts_list) [B,T,2V].
    Returns tokens [B, V*W+1, d] and the key-padding mask [B, V*W+1] (True = ignore)."""
    B, T, _ = x.shape
    d = sd["rep_token"].shape[-1]
    local, observed = local_features(x, n_vars)
    h = F.linear(local, sd["input_proj.0.weight"], sd["input_proj.0.bias"])
    h = F.layer_norm(F.gelu(h), (d,), sd["input_proj.2.weight"], sd["input_proj.2.bias"], eps)
    var_emb = sd["variable_embedding.weight"].unsqueeze(0).expand(B, -1, -1).reshape(B * n_vars, 1, d)
    h = h + var_emb + sd["hour_embedding.weight"][:T].unsqueeze(0)
    h = gru(sd, h)
    pooled, valid = [], []
    obs_v = observed.permute(0, 2, 1)
    prev = 0
    for wi, boundary in enumerate(windows):
        s, e = T - boundary, T - prev
        pooled.append(h[:, s:e, :].mean(dim=1) + sd["window_embedding.weight"][wi])
        valid.append(obs_v[:, :, s:e].any(dim=-1))
        prev = boundary
    tokens = torch.stack(pooled, dim=1).view(B, n_vars, len(windows), d)
    tokens = F.layer_norm(tokens, (d,), sd["output_norm.weight"], sd["output_norm.bias"], eps).reshape(B, -1, d)
    tokens = torch.cat([tokens, sd["rep_token"].expand(B, -1, -1)], dim=1)
    pad = ~torch.cat([torch.stack(valid, dim=2).reshape(B, -1), torch.ones((B, 1), dtype=torch.bool)], dim=1)
    return tokens, pad
