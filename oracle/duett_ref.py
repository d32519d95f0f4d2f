"""fp32 restatement of the DuETT side of the hot path (SURVEY.md §8a rows a1–a6, a11, a17).

Functional: every function takes `sd`, a dict of tensors keyed exactly like the
state_dict of the reference's `DuettFeatureExtractor` (reference `duett/duett.py:48-140`),
so fixtures produced by the reference and weights held by the product load unchanged.
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import numpy as np
import torch
import torch.nn.functional as F

from .xt_encoder import encoder_forward

BN_EPS = 1e-5
BN_MOMENTUM = 0.1


@dataclass
class DuettCfg:
    """Constructor defaults of reference `duett/duett.py:49-56` that shape the arithmetic."""
    d_static_num: int
    d_time_series_num: int          # V
    n_timesteps: int                # masked_transform_timesteps == max_len  (model file :106-118)
    d_embedding: int = 24
    n_heads: int = 2
    n_layers: int = 2
    d_feedforward: int = 512
    d_hidden_mlp_embedding: int = 64
    d_hidden_tab_encoder: int = 128
    transformer_dropout: float = 0.0

    @property
    def et_dim(self): return self.d_embedding * (self.n_timesteps + 1)      # duett.py:93
    @property
    def tt_dim(self): return self.d_embedding * (self.d_time_series_num + 1)  # duett.py:94
    @property
    def d_time_hidden(self): return int(np.sqrt(self.tt_dim))               # duett.py:154


# --------------------------------------------------------------------------- a3
def feats_to_input(x, max_len: int, *, training=False, aug_noise=0.0, aug_mask=0.0, pretrain=False,
                   device="cpu"):
    """reference `duett/duett.py:159-187`.  Augmentation (train only) uses torch's global RNG
    exactly where the reference draws from it; parity tests run with aug off."""
    xs_ts, xs_static, times = x
    xs_ts, times = list(xs_ts), list(times)
    for i, f in enumerate(xs_ts):
        n_vars = f.shape[1] // 2
        if f.shape[0] > max_len:                                   # :165-167 keep the LAST max_len steps
            f = f[-max_len:]
            times[i] = times[i][-max_len:]
        if training and aug_noise > 0 and not pretrain:            # :169-170
            f = f.clone()
            f[:, :n_vars] += aug_noise * torch.randn_like(f[:, :n_vars]) * f[:, n_vars:]
        f = torch.cat((f, torch.zeros_like(f[:, :1])), dim=1)      # :171 mask column
        if training and aug_mask > 0 and not pretrain:             # :172-175
            mask = torch.rand(f.shape[0]) < aug_mask
            f[mask, :] = 0.0
            f[mask, -1] = 1.0
        xs_ts[i] = f
    n_timesteps = [len(ts) for ts in times]
    pad_to = int(np.max(n_timesteps))
    xs_ts = torch.stack([F.pad(t, (0, 0, 0, pad_to - t.shape[0])) for t in xs_ts]).to(device)
    xs_times = torch.stack([F.pad(t, (0, pad_to - t.shape[0])) for t in times]).to(device)
    xs_static = torch.stack(list(xs_static)).to(device)
    if training and aug_noise > 0 and not pretrain:                # :184-185
        xs_static = xs_static + aug_noise * torch.randn_like(xs_static)
    return xs_static, xs_ts, xs_times, n_timesteps


# --------------------------------------------------------------------------- a2
def batchnorm_lastdim(x, sd, prefix, training=False, update_running=True):
    """`BatchNormLastDim` (duett.py:11-22): BatchNorm1d over the last dim, statistics over
    every other dim.  Train: biased batch variance normalises, unbiased updates running."""
    w, b = sd[prefix + "batch_norm.weight"], sd[prefix + "batch_norm.bias"]
    rm, rv = sd[prefix + "batch_norm.running_mean"], sd[prefix + "batch_norm.running_var"]
    if training:
        flat = x.reshape(-1, x.shape[-1])
        mean = flat.mean(0)
        var = flat.var(0, unbiased=False)
        if update_running:
            n = flat.shape[0]
            with torch.no_grad():
                rm.mul_(1 - BN_MOMENTUM).add_(BN_MOMENTUM * mean.detach())
                rv.mul_(1 - BN_MOMENTUM).add_(BN_MOMENTUM * var.detach() * n / max(n - 1, 1))
                if prefix + "batch_norm.num_batches_tracked" in sd:
                    sd[prefix + "batch_norm.num_batches_tracked"] += 1
    else:
        mean, var = rm, rv
    return (x - mean) / torch.sqrt(var + BN_EPS) * w + b


def simple_mlp_1hidden(x, sd, prefix, activation=F.relu, training=False):
    """`simple_mlp(..., n_hidden=1, hidden_batch_norm=True)` (duett.py:24-39):
    [0] Linear, [1] act, [2] Dropout(0), [3] BN(d_hidden), [4] Linear — BN sits AFTER the activation."""
    h = activation(F.linear(x, sd[prefix + "0.weight"], sd[prefix + "0.bias"]))
    h = batchnorm_lastdim(h, sd, prefix + "3.", training)
    return F.linear(h, sd[prefix + "4.weight"], sd[prefix + "4.bias"])


def cve_time_embedding(t, sd, prefix, training=False):
    """`cve(batch_norm=True)` (duett.py:151-157): Linear(1,h) → Tanh → BN → Linear(h, tt_dim)."""
    h = torch.tanh(F.linear(t, sd[prefix + "0.weight"], sd[prefix + "0.bias"]))
    h = batchnorm_lastdim(h, sd, prefix + "2.", training)
    return F.linear(h, sd[prefix + "3.weight"], sd[prefix + "3.bias"])


# --------------------------------------------------------------------------- a4
def build_psi(sd, cfg: DuettCfg, xs_static, xs_feats, xs_times, training=False, predict_events=True):
    """ψ build, model file `:31-69` (== duett.py:245-272).  Returns (ψ [B,T+1,V+1,E], time_emb [B,T+1,tt])."""
    B, T, _ = xs_feats.shape
    V, E = cfg.d_time_series_num, cfg.d_embedding
    values = xs_feats[:, :, :V]
    counts = xs_feats[:, :, V:2 * V]
    n_obs_inds = counts.to(torch.int64).clip(0, sd["n_obs_embedding.weight"].shape[0] - 1)     # :41
    n_obs = sd["n_obs_embedding.weight"][n_obs_inds].squeeze(-1)                             # :43
    psi = torch.zeros((B, T + 1, V + 1, E), dtype=xs_feats.dtype, device=xs_feats.device)
    for v in range(V):                                                                        # :53-55
        inp = torch.stack((values[:, :, v], n_obs[:, :, v]), dim=-1)
        psi[:, :-1, v, :] = simple_mlp_1hidden(inp, sd, f"embedding_layers.{v}.", training=training)
    psi[:, :-1, -1, :] = simple_mlp_1hidden(xs_static, sd, "tab_encoder.", training=training).unsqueeze(1)  # :57
    special = sd["special_embeddings.weight"]
    psi[:, -1, :, :] = special[1]                                                             # :58-60 REP key = 1
    mask_inds = torch.cat((xs_feats[:, :, -1] == 1,
                           torch.zeros((B, 1), dtype=torch.bool, device=xs_feats.device)), dim=1)  # :61-63
    psi[mask_inds] = special[0]                                                               # :64 MASKED key = 0
    if predict_events:                                                                        # :35-40, :65-66
        ev = counts == -1
        ev = torch.cat((ev, torch.zeros((B, T, 1), dtype=torch.bool, device=ev.device)), dim=2)
        ev = torch.cat((ev, ev[:, :1, :]), dim=1)
        psi[ev] = special[0]
    te = cve_time_embedding(xs_times.unsqueeze(2), sd, "full_time_embedding.", training)      # :67
    rep = sd["full_rep_embedding.weight"].T.unsqueeze(0).expand(B, -1, -1)                    # :68-69
    return psi, torch.cat((te, rep), dim=1)


# --------------------------------------------------------------------------- a5
def encode(sd, cfg: DuettCfg, duett_in, training=False, return_intermediates=False):
    """`DuettFeatureExtractor.encode`, model file `:31-94`."""
    xs_static, xs_feats, xs_times, _ = duett_in
    psi, time_emb = build_psi(sd, cfg, xs_static, xs_feats, xs_times, training)
    inter = {"psi0": psi, "time_emb": time_emb}
    drop = cfg.transformer_dropout
    for l in range(cfg.n_layers):
        B, T1, V1, E = psi.shape
        emb = psi.transpose(1, 2).flatten(2) + sd["full_event_embedding.weight"].unsqueeze(0)   # :80
        ev = encoder_forward(sd, f"event_transformers.{l}.", emb, cfg.n_heads, E // cfg.n_heads, drop, training)
        ev = ev.view(B, V1, T1, E).transpose(1, 2)                                              # :81
        emb = ev.flatten(2) + time_emb                                                          # :90
        psi = encoder_forward(sd, f"time_transformers.{l}.", emb, cfg.n_heads, E // cfg.n_heads, drop, training)
        psi = psi.view(B, T1, V1, E)                                                            # :91
        inter[f"psi{l + 1}"] = psi
    out = psi.flatten(2)                                                                        # :93
    return (out, inter) if return_intermediates else out


# --------------------------------------------------------------------------- a11
def student_forward(sd, cfg: DuettCfg, duett_in, pool="mean", training=False):
    """`StudentModel.forward`, model file `:1221-1235`.  `sd` keys: 'duett.*', 'head.{0,3}.*'."""
    dsd = {k[len("duett."):]: v for k, v in sd.items() if k.startswith("duett.")}
    tok = encode(dsd, cfg, duett_in, training)
    feat = tok[:, -1, :] if pool == "rep_token" else tok[:, :-1, :].mean(dim=1)
    h = F.gelu(F.linear(feat, sd["head.0.weight"], sd["head.0.bias"]))
    return F.linear(h, sd["head.3.weight"], sd["head.3.bias"]).squeeze(-1)


# --------------------------------------------------------------------------- a17 (config 1: DuETT-only SSL / supervised step)
def pretrain_prep_batch(x, rng, n_vars, max_len, pretrain_dropout=0.5, predict_events=True):
    """`Model.pretrain_prep_batch` (duett.py:189-237) for pretrain_masked_steps == 1: one masked timestep and one masked
    event per sample, drawn from the model's numpy Generator in the reference's call order, then variable dropout."""
    xs_static, xs_ts, xs_times, n_timesteps = feats_to_input(x, max_len)
    B = xs_ts.shape[0]
    y_ts, y_ts_n_obs, y_events, y_events_mask = [], [], [], []
    clipped = xs_ts.clone()
    for b, n in enumerate(n_timesteps):
        mask_i = n if n < 2 else rng.choice(np.arange(0, n))                    # :199-207
        y_ts.append(xs_ts[b, mask_i, :n_vars])
        y_ts_n_obs.append(xs_ts[b, mask_i, n_vars:2 * n_vars])
        clipped[b, mask_i, :] = 0.0
        clipped[b, mask_i, -1] = 1.0
        if predict_events:                                                       # :214-219
            ev = rng.choice(np.arange(0, n_vars))
            y_events.append(xs_ts[b, :, ev])
            y_events_mask.append(xs_ts[b, :, ev + n_vars].clip(0, 1))
            clipped[b, :, ev] = 0
            clipped[b, :, ev + n_vars] = -1
    y_ts, y_ts_masks = torch.stack(y_ts), torch.stack(y_ts_n_obs).clip(0, 1)
    if predict_events:
        y_events, y_events_mask = torch.stack(y_events), torch.stack(y_events_mask)
    if pretrain_dropout > 0:                                                     # :227-236
        keep = torch.tensor(rng.random((B, n_vars)) > pretrain_dropout)
        keep = torch.logical_or(1 - y_ts_masks, keep)
        keep = torch.cat((keep.tile(1, 2), torch.ones((B, 1))), dim=1)
        clipped = clipped * torch.logical_or(keep.unsqueeze(1), clipped == -1)
    return (xs_static, clipped, xs_times, n_timesteps), y_ts, y_ts_masks, y_events, y_events_mask


def model_forward(sd, cfg: DuettCfg, x, pretrain=False, fusion_method="masked_embed", training=False):
    """`Model.forward` (duett.py:239-323), pretrain_masked_steps == 1."""
    xs_static, xs_feats, xs_times, _ = x
    tok, inter = encode(sd, cfg, x, training, return_intermediates=True)
    psi = inter[f"psi{cfg.n_layers}"]
    B = tok.shape[0]
    if fusion_method == "rep_token":
        z = tok[:, -1, :]
    elif fusion_method == "masked_embed":
        idx = (xs_feats[:, :, -1] == 1).float().argmax(dim=1)
        z = tok[torch.arange(B), idx]
    else:
        z = tok[:, :-1, :].mean(dim=1)
    if pretrain:
        lin = lambda name, v: F.linear(v, sd[name + ".0.weight"], sd[name + ".0.bias"])
        V = cfg.d_time_series_num
        ev_idx = (xs_feats[:, 0, V:2 * V] == -1).float().argmax(dim=1)
        z_events = psi[torch.arange(B), :, ev_idx, :].flatten(1)                 # :311-313
        return (lin("pretrain_value_proj", z), lin("pretrain_presence_proj", z), lin("predict_events_proj", z_events),
                lin("predict_events_presence_proj", z_events))
    h = F.relu(F.linear(z, sd["head.0.weight"], sd["head.0.bias"]))
    h = batchnorm_lastdim(h, sd, "head.3.", training)
    return F.linear(h, sd["head.4.weight"], sd["head.4.bias"]).squeeze(1)


def ssl_loss(y_hat_value, y_hat_presence, y_hat_events, y_hat_events_presence, y, mask, y_events, y_events_mask, presence_weight=0.2):
    """`Model.training_step`, pretrain branch (duett.py:337-358)."""
    loss = F.mse_loss(y_hat_value * mask, y * mask)
    loss = loss + F.binary_cross_entropy_with_logits(y_hat_presence, mask) * presence_weight
    loss = loss + F.mse_loss(y_hat_events * y_events_mask, y_events * y_events_mask)
    return loss + F.binary_cross_entropy_with_logits(y_hat_events_presence, y_events_mask) * presence_weight
