"""Restatement of `x_transformers.Encoder(dim, depth=1, heads=2, pre_norm=True,
use_scalenorm=True, attn_dim_head=12, ff_glu=False, ff_mult=512/dim, ...)`.

PARITY UNPINNED: x_transformers is not in /root/reference, not version-pinned by
it and not installed here.  Constructed at reference `duett/duett.py:95-99`
(event axis) and `:101-105` (time axis); invoked with a single positional tensor
and no mask at `models/main_architecture_duett.py:81,91`.

Semantics restated (x_transformers 1.x/2.x `AttentionLayers`, depth=1 encoder):
    layer order ('a', 'f'); every branch is  x = x + f(ScaleNorm(x))
    ScaleNorm(x) = x / max(||x||_2, eps) * sqrt(dim) * g      (g scalar, init 1)
    attention: to_q/to_k/to_v = Linear(dim, heads*dim_head, bias=False)
               heads split 'b n (h d) -> b h n d'; scale = dim_head**-0.5
               softmax over keys in fp32; dropout; merge 'b h n d -> b n (h d)'
               to_out = Linear(heads*dim_head, dim, bias=False)
    feed-forward: Linear(dim, inner)+bias -> GELU(erf) -> Dropout -> Linear(inner, dim)+bias
                  inner = int(dim * ff_mult)
    final ScaleNorm after the last layer when pre_norm=True (switch FINAL_NORM).

State-dict key names follow x_transformers' module tree so that a checkpoint
written by the reference would load:  layers.{0,1}.0.0.g, layers.0.1.to_{q,k,v,out}.weight,
layers.1.1.ff.0.0.{weight,bias}, layers.1.1.ff.2.{weight,bias}, final_norm.g
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn
import torch.nn.functional as F

FINAL_NORM = True          # 1.x/2.x: present when pre_norm=True; 0.x: absent
SCALENORM_EPS = 1e-12      # F.normalize default (2.x); 1.x clamps ||x||*dim^-.5 at 1e-5


def ff_inner_dim(dim: int, d_feedforward: int = 512) -> int:
    """inner = int(dim * mult), mult = d_feedforward / dim  (duett.py:98,104)."""
    return int(dim * (d_feedforward / dim))


def scalenorm(x: torch.Tensor, g: torch.Tensor, eps: float = SCALENORM_EPS) -> torch.Tensor:
    dim = x.shape[-1]
    n = x.float().norm(dim=-1, keepdim=True).clamp_min(eps)
    return x / n * math.sqrt(dim) * g


def encoder_forward(sd: dict, prefix: str, x: torch.Tensor, heads: int = 2, dim_head: int = 12,
                    dropout: float = 0.0, training: bool = False) -> torch.Tensor:
    """x: [B, N, D] -> [B, N, D].  `sd` maps '<prefix>layers...' -> tensors."""
    g = lambda k: sd[prefix + k]
    B, N, D = x.shape
    # --- attention branch ---
    h = scalenorm(x, g("layers.0.0.0.g"))
    q = F.linear(h, g("layers.0.1.to_q.weight"))
    k = F.linear(h, g("layers.0.1.to_k.weight"))
    v = F.linear(h, g("layers.0.1.to_v.weight"))
    sp = lambda t: t.view(B, N, heads, dim_head).transpose(1, 2)
    q, k, v = sp(q), sp(k), sp(v)
    sim = torch.matmul(q * dim_head ** -0.5, k.transpose(-1, -2))
    attn = sim.float().softmax(dim=-1).to(x.dtype)
    attn = F.dropout(attn, dropout, training)
    o = torch.matmul(attn, v).transpose(1, 2).reshape(B, N, heads * dim_head)
    x = x + F.linear(o, g("layers.0.1.to_out.weight"))
    # --- feed-forward branch ---
    h = scalenorm(x, g("layers.1.0.0.g"))
    h = F.gelu(F.linear(h, g("layers.1.1.ff.0.0.weight"), g("layers.1.1.ff.0.0.bias")))
    h = F.dropout(h, dropout, training)
    x = x + F.linear(h, g("layers.1.1.ff.2.weight"), g("layers.1.1.ff.2.bias"))
    if FINAL_NORM:
        x = scalenorm(x, g("final_norm.g"))
    return x


class Encoder(nn.Module):
    """nn.Module form of the same arithmetic with x_transformers' parameter tree.
    Used (a) as the `x_transformers.Encoder` stub when the reference is imported to
    generate golden vectors and (b) to hold seeded weights in tests."""

    def __init__(self, dim, depth=1, heads=2, pre_norm=True, use_scalenorm=True, attn_dim_head=12,
                 ff_glu=False, ff_mult=4, attn_dropout=0.0, ff_dropout=0.0, **kw):
        super().__init__()
        assert depth == 1 and pre_norm and use_scalenorm and not ff_glu, "only the DuETT configuration is restated"
        self.dim, self.heads, self.dim_head = dim, heads, attn_dim_head
        self.dropout = float(attn_dropout)
        assert float(ff_dropout) == self.dropout
        inner = int(dim * ff_mult)
        hd = heads * attn_dim_head

        class _G(nn.Module):
            def __init__(s):
                super().__init__()
                s.g = nn.Parameter(torch.ones(1))

        class _Attn(nn.Module):
            def __init__(s):
                super().__init__()
                s.to_q = nn.Linear(dim, hd, bias=False)
                s.to_k = nn.Linear(dim, hd, bias=False)
                s.to_v = nn.Linear(dim, hd, bias=False)
                s.to_out = nn.Linear(hd, dim, bias=False)

        class _FF(nn.Module):
            def __init__(s):
                super().__init__()
                s.ff = nn.Sequential(nn.Sequential(nn.Linear(dim, inner), nn.GELU()),
                                     nn.Dropout(ff_dropout), nn.Linear(inner, dim))

        mk_norms = lambda: nn.ModuleList([_G(), nn.Identity(), nn.Identity()])
        self.layers = nn.ModuleList([
            nn.ModuleList([mk_norms(), _Attn(), nn.Identity()]),
            nn.ModuleList([mk_norms(), _FF(), nn.Identity()]),
        ])
        self.final_norm = _G()

    def forward(self, x):
        return encoder_forward(dict(self.state_dict(keep_vars=True)), "", x, self.heads, self.dim_head,
                               self.dropout, self.training)
