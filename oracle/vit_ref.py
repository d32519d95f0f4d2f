"""fp32 restatement of `CXREncoder.forward` → `transformers.Dinov2Model.forward`
(model file `:152-158`; transformers 5.15.0
`models/dinov2/modeling_dinov2.py:38-118,182-236,272-300,342-381`).
`sd` is keyed like `Dinov2Model.state_dict()`."""
from __future__ import annotations

from dataclasses import dataclass

import torch
import torch.nn.functional as F


@dataclass
class VitCfg:
    hidden: int = 768
    layers: int = 12
    heads: int = 12
    mlp_ratio: int = 4
    patch: int = 14
    image_size: int = 518          # size of the stored position grid (37x37)
    ln_eps: float = 1e-6


def interpolate_pos_embed(pos, cfg: VitCfg, height: int, width: int):
    """modeling_dinov2.py:57-95 — bicubic resize of the patch position grid (fp32, align_corners=False)."""
    n_pos = pos.shape[1] - 1
    gh, gw = height // cfg.patch, width // cfg.patch
    if gh * gw == n_pos and height == width:
        return pos
    cls_pos, patch_pos = pos[:, :1], pos[:, 1:]
    s = int(n_pos ** 0.5)
    dim = pos.shape[-1]
    grid = patch_pos.reshape(1, s, s, dim).permute(0, 3, 1, 2).float()
    grid = F.interpolate(grid, size=(gh, gw), mode="bicubic", align_corners=False)
    return torch.cat((cls_pos, grid.permute(0, 2, 3, 1).reshape(1, -1, dim)), dim=1)


def vit_forward(sd, cfg: VitCfg, pixel_values, return_hidden=False):
    B, _, H, W = pixel_values.shape
    x = F.conv2d(pixel_values, sd["embeddings.patch_embeddings.projection.weight"],
                 sd["embeddings.patch_embeddings.projection.bias"], stride=cfg.patch)
    x = x.flatten(2).transpose(1, 2)                                        # :148
    x = torch.cat((sd["embeddings.cls_token"].expand(B, -1, -1), x), dim=1)  # :108-109
    x = x + interpolate_pos_embed(sd["embeddings.position_embeddings"], cfg, H, W)  # :112
    dh = cfg.hidden // cfg.heads
    hidden = [x]
    for l in range(cfg.layers):
        p = f"encoder.layer.{l}."
        h = F.layer_norm(x, (cfg.hidden,), sd[p + "norm1.weight"], sd[p + "norm1.bias"], cfg.ln_eps)
        sp = lambda t: t.view(B, -1, cfg.heads, dh).transpose(1, 2)
        q = sp(F.linear(h, sd[p + "attention.attention.query.weight"], sd[p + "attention.attention.query.bias"]))
        k = sp(F.linear(h, sd[p + "attention.attention.key.weight"], sd[p + "attention.attention.key.bias"]))
        v = sp(F.linear(h, sd[p + "attention.attention.value.weight"], sd[p + "attention.attention.value.bias"]))
        a = torch.softmax(torch.matmul(q, k.transpose(2, 3)) * dh ** -0.5, dim=-1)   # :167-172
        a = torch.matmul(a, v).transpose(1, 2).reshape(B, -1, cfg.hidden)
        a = F.linear(a, sd[p + "attention.output.dense.weight"], sd[p + "attention.output.dense.bias"])
        x = a * sd[p + "layer_scale1.lambda1"] + x                                   # :367-370
        h = F.layer_norm(x, (cfg.hidden,), sd[p + "norm2.weight"], sd[p + "norm2.bias"], cfg.ln_eps)
        h = F.gelu(F.linear(h, sd[p + "mlp.fc1.weight"], sd[p + "mlp.fc1.bias"]))
        h = F.linear(h, sd[p + "mlp.fc2.weight"], sd[p + "mlp.fc2.bias"])
        x = h * sd[p + "layer_scale2.lambda1"] + x                                   # :373-378
        hidden.append(x)
    x = F.layer_norm(x, (cfg.hidden,), sd["layernorm.weight"], sd["layernorm.bias"], cfg.ln_eps)
    cls, patches = x[:, 0], x[:, 1:]                                                 # model file :155-158
    return (cls, patches, hidden) if return_hidden else (cls, patches)
