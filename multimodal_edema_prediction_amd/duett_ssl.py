"""DuETT-only step (BASELINE.json configs[0]): host-side mirror of `Model.pretrain_prep_batch`, `Model.forward` and the
loss arithmetic of `Model.training_step` (reference duett/duett.py:189-372) for pretrain_masked_steps == 1.  The encoder
runs in its HIP training form (duett_train.encode_training); read-out gathers are index plumbing; heads and losses are HIP."""
from __future__ import annotations

import numpy as np
import torch

from . import autograd_ops as A
from .abi import check, lib, ptr, stream
from .duett_train import ActFn, GBatchNormFn, encode_training

F32 = torch.float32


def pretrain_prep_batch(model, x, batch_size):
    """duett.py:189-237.  The random choices are the reference's — the same numpy Generator calls in the same order (masked
    timestep and masked event per sample, then the [B, V] variable-dropout table) — drawn on the host; everything they drive
    (target gathers, index_put of the masked row / event columns, the keep multiply) is ONE launch of `medp_ssl_mask_batch`."""
    if model.pretrain_masked_steps != 1:
        raise NotImplementedError("pretrain_masked_steps > 1 is not built (the reference default is 1)")
    xs_static, xs_ts, xs_times, n_timesteps = model.feats_to_input(x, batch_size)
    B, T, Fd = xs_ts.shape
    n_vars = (Fd - 1) // 2
    mask_t, events = [], []
    for n in n_timesteps:
        mask_t.append(int(n if n < 2 else model.rng.choice(np.arange(0, n))))
        if model.predict_events:
            events.append(int(model.rng.choice(np.arange(0, model.d_time_series_num))))
    if max(mask_t) >= T:
        raise IndexError(f"index {max(mask_t)} is out of bounds for dimension 1 with size {T}")     # what the reference's indexing raises
    keep = model.rng.random((batch_size, n_vars)) > model.pretrain_dropout if model.pretrain_dropout > 0 else None
    dev = xs_ts.device
    mt = torch.tensor(mask_t, dtype=torch.int32, device=dev)
    ev = torch.tensor(events, dtype=torch.int32, device=dev) if events else None
    kp = torch.from_numpy(np.ascontiguousarray(keep).astype(np.uint8)).to(dev) if keep is not None else None
    clipped = torch.empty_like(xs_ts)
    y_ts = torch.empty((B, n_vars), dtype=F32, device=dev)
    y_masks = torch.empty((B, n_vars), dtype=F32, device=dev)
    y_events = torch.empty((B, T), dtype=F32, device=dev) if events else []
    y_events_mask = torch.empty((B, T), dtype=F32, device=dev) if events else []
    check(lib().medp_ssl_mask_batch(ptr(xs_ts), ptr(mt), ptr(ev), ptr(kp), ptr(clipped), ptr(y_ts), ptr(y_masks),
                                    ptr(y_events) if events else None, ptr(y_events_mask) if events else None, B, T, n_vars, stream()),
          "ssl_mask_batch")
    return (xs_static, clipped, xs_times, n_timesteps), y_ts, y_masks, y_events, y_events_mask


def model_forward(model, x, pretrain=False, representation=False):
    """duett.py:239-323."""
    xs_static, xs_feats, xs_times, _ = x
    tok = encode_training(model, x)
    B, T1, D = tok.shape
    V, E = model.d_time_series_num, model.d_embedding
    ar = torch.arange(B, device=tok.device)
    if model.fusion_method == "rep_token":
        z = tok[:, -1, :]
    elif model.fusion_method == "masked_embed":
        idx = (xs_feats[:, :, -1] == 1).float().argmax(dim=1)       # the single masked timestep of each sample
        z = tok[ar, idx]
    elif model.fusion_method == "averaging":
        z = A.MeanPoolFn.apply(tok, T1 - 1)
    else:
        raise ValueError(f"unknown fusion_method {model.fusion_method!r}")
    if representation:
        return z
    if pretrain:
        psi = tok.view(B, T1, V + 1, E)
        ev_idx = (xs_feats[:, 0, V:2 * V] == -1).float().argmax(dim=1)
        z_events = psi[ar, :, ev_idx, :].reshape(B, T1 * E)
        lin = lambda seq, v: A.linear(v.contiguous(), seq[0].weight, seq[0].bias)
        y_hat_presence = lin(model.pretrain_presence_proj, z) if model.pretrain_presence else None
        y_hat_value = lin(model.pretrain_value_proj, z) if model.pretrain_value else None
        y_hat_events = y_hat_events_presence = None
        if model.predict_events:
            y_hat_events = lin(model.predict_events_proj, z_events)
            y_hat_events_presence = lin(model.predict_events_presence_proj, z_events) if model.pretrain_presence else None
        return y_hat_value, y_hat_presence, y_hat_events, y_hat_events_presence
    hd = model.head                                                   # Linear -> ReLU -> Dropout(0) -> BN -> Linear(., d_target)
    h = A.linear(z.contiguous(), hd[0].weight, hd[0].bias)
    h = ActFn.apply(h, 0)
    bn = hd[3].batch_norm
    h = GBatchNormFn.apply(h.unsqueeze(0), bn.weight.unsqueeze(0), bn.bias.unsqueeze(0), bn.running_mean.unsqueeze(0),
                           bn.running_var.unsqueeze(0), bool(model.training))[0]
    if model.training:
        with torch.no_grad():
            bn.num_batches_tracked += 1
    if hd[4].out_features == 1:
        return A.rowdot(h, hd[4].weight, hd[4].bias)
    return A.linear(h, hd[4].weight, hd[4].bias)


def masked_mse(a, b, mask):
    """F.mse_loss(a*mask, b*mask)  (duett.py:344,356)"""
    bc, mc = b.detach().contiguous().to(F32), mask.detach().contiguous().to(F32)
    return A._ScalarLossFn.apply(a, lambda x, out, g: check(lib().medp_masked_mse(ptr(x), ptr(bc), ptr(mc), ptr(out), ptr(g), x.numel(), stream()), "masked_mse"))


def bce_mean(logits, y, weight=None):
    """F.binary_cross_entropy_with_logits(logits, y[, weight])  (duett.py:352,358,363-365)"""
    yc = y.detach().contiguous().to(F32)
    wc = weight.detach().contiguous().to(F32) if weight is not None else None
    return A._ScalarLossFn.apply(logits, lambda x, out, g: check(lib().medp_bce_mean(ptr(x), ptr(yc), ptr(wc), ptr(out), ptr(g), x.numel(), stream()), "bce_mean"))


def training_step_loss(model, batch):
    """Loss of `Model.training_step` (duett.py:329-372)."""
    x, y = batch
    y = torch.as_tensor(np.asarray(y), dtype=F32, device=model.device)
    B = y.shape[0]
    if model.pretrain:
        xp, yv, mask, y_events, y_events_mask = pretrain_prep_batch(model, x, B)
        hv, hp, he, hep = model_forward(model, xp, pretrain=True)
        loss = None
        add = lambda a, b, w=1.0: b if a is None and w == 1.0 else A.add_scaled(a if a is not None else torch.zeros((), device=b.device), b, w)
        if model.pretrain_value:
            loss = masked_mse(hv, yv, mask)
        if model.pretrain_presence:
            loss = add(loss, bce_mean(hp, mask), model.pretrain_presence_weight)
        if model.predict_events:
            if model.pretrain_value:
                loss = add(loss, masked_mse(he, y_events, y_events_mask))
            if model.pretrain_presence:
                loss = add(loss, bce_mean(hep, y_events_mask), model.pretrain_presence_weight)
        return loss
    y_hat = model_forward(model, model.feats_to_input(x, B))
    return bce_mean(y_hat, y)
