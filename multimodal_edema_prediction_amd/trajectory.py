"""`LocalTrajectoryEncoder` (reference models/main_architecture_duett.py:1242-1391) on the HIP kernels:
per-variable local features -> Linear(5, d) + GELU + LayerNorm + variable / hour embeddings -> a shared GRU over the B*V
sequences -> one token per (variable, recency window) + REP, with the key-padding mask of windows that saw no observation.

Same constructor, parameter names (`input_proj.0/2`, `variable_embedding`, `hour_embedding`, `temporal.weight_ih_l0` ...,
`window_embedding`, `output_norm`, `rep_token`) and outputs as the reference class, so a state_dict moves either way.
What runs where: features (sequential scan), GRU recurrence forward / backward = csrc/trajectory.hip; the two Linears = bf16
MFMA GEMMs (autograd_ops.linear, weight gradients by the transposed GEMM); GELU, LayerNorms = the kernels the fusion head
uses; embeddings, window means, concatenations and the mask = torch index / elementwise plumbing.  `nn.GRU` is only the
parameter container.  Hidden size 128 (the module's default `d_model`) is what the GRU kernels are built for.
Parity: tests/test_gpu_trajectory.py against oracle/trajectory_ref.py, itself pinned by the reference's own class."""
from __future__ import annotations

import torch
import torch.nn as nn

from . import autograd_ops as A
from . import functional as Fn
from .abi import check, lib, ptr, stream

F32, BF16 = torch.float32, torch.bfloat16


def traj_features(x: torch.Tensor, n_vars: int) -> torch.Tensor:
    """x [B,T,2V] fp32 -> [B*V, T, 8] fp32 (five features, zero padded)."""
    B, T, C = x.shape
    xc = x.detach().to(F32).contiguous()
    out = torch.empty((B * n_vars, T, 8), dtype=F32, device=x.device)
    check(lib().medp_traj_features(ptr(xc), ptr(out), B, T, n_vars, stream()), "traj_features")
    return out


class GruFn(torch.autograd.Function):
    """gi [S,T,3d] fp32 (x_t W_ih^T + b_ih), W_hh [3d,d], b_hh [3d] -> every hidden state [S,T,d]; h0 = 0."""

    @staticmethod
    def forward(ctx, gi, w_hh, b_hh):
        S, T, d3 = gi.shape
        d = d3 // 3
        gi = gi.contiguous()
        need = gi.requires_grad or w_hh.requires_grad or b_hh.requires_grad
        hseq = torch.empty((S, T, d), dtype=F32, device=gi.device)
        gates = torch.empty((S, T, d3), dtype=F32, device=gi.device) if need else None
        hn = torch.empty((S, T, d), dtype=F32, device=gi.device) if need else None
        check(lib().medp_gru_fwd(ptr(gi), ptr(A.weight_bf16(w_hh)), ptr(b_hh.detach().contiguous()), ptr(hseq), ptr(gates), ptr(hn),
                                 S, T, d, stream()), "gru_fwd")
        ctx.save_for_backward(gates, hn, hseq, w_hh)
        return hseq

    @staticmethod
    def backward(ctx, dh):
        gates, hn, hseq, w_hh = ctx.saved_tensors
        S, T, d = hseq.shape
        dh = dh.contiguous()
        dgi = torch.empty((S, T, 3 * d), dtype=F32, device=dh.device)
        dghn = torch.empty((S, T, d), dtype=F32, device=dh.device)
        dgh16 = torch.empty((S, T, 3 * d), dtype=BF16, device=dh.device)
        check(lib().medp_gru_bwd(ptr(dh), ptr(gates), ptr(hn), ptr(hseq), ptr(A.weight_t_bf16(w_hh)), ptr(dgi), ptr(dghn), ptr(dgh16),
                                 S, T, d, stream()), "gru_bwd")
        # dW_hh = sum over (sequence, step) of dgh^T h_{t-1}: the transposed GEMM over the stored rows; h_{-1} = 0
        hprev = torch.zeros_like(hseq)
        hprev[:, 1:] = hseq[:, :-1]
        dw = Fn.gemm_tn(dgh16.view(S * T, 3 * d), Fn.to_bf16(hprev.view(S * T, d)))
        db = torch.cat([Fn.colsum(dgi.view(S * T, 3 * d))[:2 * d], Fn.colsum(dghn.view(S * T, d))])
        return dgi, dw, db


class LocalTrajectoryEncoder(nn.Module):
    """Drop-in for the reference class (same arguments, defaults and errors, :1261-1309)."""

    def __init__(self, n_vars: int, n_timesteps: int = 24, d_model: int = 128, n_layers: int = 1, dropout: float = 0.1,
                 recency_windows: tuple = (6, 12, 24)):
        super().__init__()
        if n_vars <= 0 or n_timesteps <= 0 or d_model <= 0:
            raise ValueError("n_vars, n_timesteps, and d_model must be positive")
        windows = tuple(sorted(set(int(w) for w in recency_windows)))
        if not windows or windows[-1] != n_timesteps:
            raise ValueError(f"recency_windows must end at n_timesteps={n_timesteps}, got {windows}")
        if windows[0] <= 0 or windows[-1] > n_timesteps:
            raise ValueError(f"invalid recency_windows={windows}")
        if n_layers != 1:
            raise NotImplementedError("the GRU kernels implement the reference default, n_layers = 1")
        self.n_vars, self.n_timesteps, self.d_model, self.recency_windows = n_vars, n_timesteps, d_model, windows
        self.p_drop = float(dropout)
        self.input_proj = nn.Sequential(nn.Linear(5, d_model), nn.GELU(), nn.LayerNorm(d_model))
        self.variable_embedding = nn.Embedding(n_vars, d_model)
        self.hour_embedding = nn.Embedding(n_timesteps, d_model)
        self.temporal = nn.GRU(input_size=d_model, hidden_size=d_model, num_layers=1, batch_first=True)   # parameters only
        self.window_embedding = nn.Embedding(len(windows), d_model)
        self.output_norm = nn.LayerNorm(d_model)
        self.dropout = nn.Dropout(dropout)
        self.rep_token = nn.Parameter(torch.randn(1, 1, d_model) * 0.02)

    @property
    def d_representation(self) -> int:
        return self.d_model

    def forward(self, x_ts_list, return_padding_mask: bool = False):
        x = torch.stack(tuple(x_ts_list), dim=0)
        if x.ndim != 3:
            raise ValueError(f"x_ts must stack to [B,T,2V], got {tuple(x.shape)}")
        B, T, C = x.shape
        V, d = self.n_vars, self.d_model
        if T != self.n_timesteps or C != 2 * V:
            raise ValueError(f"expected [B,{self.n_timesteps},{2 * V}], got {tuple(x.shape)}")
        if not x.is_cuda:
            raise RuntimeError("LocalTrajectoryEncoder runs on the GPU only (there is no CPU fallback)")
        local = traj_features(x, V)                                               # [B*V, T, 8]
        w0 = torch.nn.functional.pad(self.input_proj[0].weight, (0, 3))           # Linear(5, d) as a K = 8 GEMM
        h = A.linear(local.view(B * V * T, 8), w0, self.input_proj[0].bias)
        h = A.gelu_dropout(h, 0.0, 0, 0)
        h = A.layer_norm(h, self.input_proj[2].weight, self.input_proj[2].bias, self.input_proj[2].eps).view(B, V, T, d)
        h = h + self.variable_embedding.weight.view(1, V, 1, d) + self.hour_embedding.weight[:T].view(1, 1, T, d)
        if self.training and self.p_drop > 0:
            h = A.DropoutFn.apply(h.contiguous().view(-1, d), self.p_drop, A.next_seed(), 0).view(B, V, T, d)
        gi = A.linear(h.reshape(B * V * T, d), self.temporal.weight_ih_l0, self.temporal.bias_ih_l0).view(B * V, T, 3 * d)
        hs = GruFn.apply(gi, self.temporal.weight_hh_l0, self.temporal.bias_hh_l0)                    # [B*V, T, d]
        # non-overlapping windows measured backwards from the CXR anchor (:1370-1381)
        obs = (x[:, :, V:] > 0).permute(0, 2, 1)                                  # [B,V,T]
        pooled, valid, prev = [], [], 0
        for wi, boundary in enumerate(self.recency_windows):
            s, e = T - boundary, T - prev
            pooled.append(hs[:, s:e, :].mean(dim=1) + self.window_embedding.weight[wi])
            valid.append(obs[:, :, s:e].any(dim=-1))
            prev = boundary
        W = len(self.recency_windows)
        tokens = torch.stack(pooled, dim=1)                                       # [B*V, W, d]
        tokens = A.layer_norm(tokens.reshape(B * V * W, d), self.output_norm.weight, self.output_norm.bias, self.output_norm.eps)
        tokens = torch.cat([tokens.view(B, V * W, d), self.rep_token.expand(B, -1, -1)], dim=1)
        if not return_padding_mask:
            return tokens
        pad = ~torch.cat([torch.stack(valid, dim=2).reshape(B, -1), torch.ones((B, 1), dtype=torch.bool, device=x.device)], dim=1)
        return tokens, pad
