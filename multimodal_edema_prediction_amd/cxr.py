"""CXR image encoder: host-side mirror of the reference's `CXREncoder` (models/main_architecture_duett.py:129-158)
around a ViT-B/14 whose parameter tree is key-compatible with `transformers.Dinov2Model.state_dict()`
(RAD-DINO), executed by ONE C call into libmedp_hip (`medp_vit_forward`).

The backbone is frozen in the reference's live configuration (trainer.py:287-289); an unfrozen backbone
(`--unfreeze_cxr`) needs the ViT backward, which is SURVEY.md §8f-1 "next" and raises here.
"""
from __future__ import annotations

import ctypes
from dataclasses import dataclass

import torch
import torch.nn as nn

from . import abi
from . import functional as Fn
from .abi import MedpVitLayer, MedpVitWeights, check, lib, ptr, stream


@dataclass
class Dinov2Cfg:
    """RAD-DINO's published ViT-B/14 configuration."""
    hidden_size: int = 768
    num_hidden_layers: int = 12
    num_attention_heads: int = 12
    mlp_ratio: int = 4
    patch_size: int = 14
    image_size: int = 518
    layer_norm_eps: float = 1e-6
    layerscale_value: float = 1.0
    num_channels: int = 3


class _Node(nn.Module):
    """Plain container: lets us lay out nn.Parameters under exactly the reference's state_dict keys."""


def _set_param(root: nn.Module, key: str, value: nn.Parameter) -> None:
    parts = key.split(".")
    m = root
    for p in parts[:-1]:
        if not hasattr(m, p):
            m.add_module(p, _Node())
        m = getattr(m, p)
    m.register_parameter(parts[-1], value)


def dinov2_param_shapes(cfg: Dinov2Cfg) -> dict:
    D, ps, C = cfg.hidden_size, cfg.patch_size, cfg.num_channels
    n_pos = (cfg.image_size // ps) ** 2 + 1
    mlp = D * cfg.mlp_ratio
    sh = {"embeddings.cls_token": (1, 1, D), "embeddings.mask_token": (1, D),
          "embeddings.position_embeddings": (1, n_pos, D),
          "embeddings.patch_embeddings.projection.weight": (D, C, ps, ps),
          "embeddings.patch_embeddings.projection.bias": (D,)}
    for l in range(cfg.num_hidden_layers):
        p = f"encoder.layer.{l}."
        sh[p + "norm1.weight"] = (D,); sh[p + "norm1.bias"] = (D,)
        for n in ("query", "key", "value"):
            sh[p + f"attention.attention.{n}.weight"] = (D, D); sh[p + f"attention.attention.{n}.bias"] = (D,)
        sh[p + "attention.output.dense.weight"] = (D, D); sh[p + "attention.output.dense.bias"] = (D,)
        sh[p + "layer_scale1.lambda1"] = (D,)
        sh[p + "norm2.weight"] = (D,); sh[p + "norm2.bias"] = (D,)
        sh[p + "mlp.fc1.weight"] = (mlp, D); sh[p + "mlp.fc1.bias"] = (mlp,)
        sh[p + "mlp.fc2.weight"] = (D, mlp); sh[p + "mlp.fc2.bias"] = (D,)
        sh[p + "layer_scale2.lambda1"] = (D,)
    sh["layernorm.weight"] = (D,); sh["layernorm.bias"] = (D,)
    return sh


class Dinov2Backbone(nn.Module):
    """Parameter holder with Dinov2Model's key layout; `forward` is the HIP whole-module call."""

    def __init__(self, cfg: Dinov2Cfg | None = None):
        super().__init__()
        self.cfg = cfg or Dinov2Cfg()
        g = torch.Generator().manual_seed(0)
        for k, shape in dinov2_param_shapes(self.cfg).items():
            leaf = k.split(".")[-1]
            if leaf == "bias" or "mask_token" in k:
                v = torch.zeros(shape)
            elif leaf == "lambda1":
                v = torch.full(shape, float(self.cfg.layerscale_value))
            elif leaf == "weight" and len(shape) == 1:
                v = torch.ones(shape)
            else:
                v = torch.randn(shape, generator=g) * 0.02      # HF initializer_range
            _set_param(self, k, nn.Parameter(v))
        self._prep = None
        self._prep_key = None
        self._ws = {}

    @property
    def config(self):
        return self.cfg

    # ---- weight preparation: bf16 GEMM operands, fused QKV, C structs (rebuilt only if a parameter changed) ----
    def _prepare(self):
        sd = dict(self.named_parameters())
        key = tuple((p.data_ptr(), p._version) for p in sd.values())
        if self._prep is not None and self._prep_key == key:
            return self._prep
        c = self.cfg
        D = c.hidden_size
        dev = sd["layernorm.weight"].device
        keep = []          # keep every device buffer alive as long as the structs point at them

        def f32(t):
            t = t.detach().to(torch.float32).contiguous()
            keep.append(t)
            return t

        def bf(t2d):
            t = Fn.to_bf16(t2d.detach().to(torch.float32).contiguous())
            keep.append(t)
            return t

        K = c.num_channels * c.patch_size ** 2
        kpad = (K + 7) // 8 * 8
        pw = torch.zeros((D, kpad), dtype=torch.float32, device=dev)
        pw[:, :K] = sd["embeddings.patch_embeddings.projection.weight"].detach().reshape(D, K)
        layers = (MedpVitLayer * c.num_hidden_layers)()
        for l in range(c.num_hidden_layers):
            p = f"encoder.layer.{l}."
            L = layers[l]
            qkv_w = torch.cat([sd[p + f"attention.attention.{n}.weight"].detach() for n in ("query", "key", "value")], 0)
            qkv_b = torch.cat([sd[p + f"attention.attention.{n}.bias"].detach() for n in ("query", "key", "value")], 0)
            L.ln1_w, L.ln1_b = ptr(f32(sd[p + "norm1.weight"])), ptr(f32(sd[p + "norm1.bias"]))
            L.qkv_w, L.qkv_b = ptr(bf(qkv_w)), ptr(f32(qkv_b))
            L.proj_w, L.proj_b = ptr(bf(sd[p + "attention.output.dense.weight"])), ptr(f32(sd[p + "attention.output.dense.bias"]))
            L.ls1 = ptr(f32(sd[p + "layer_scale1.lambda1"]))
            L.ln2_w, L.ln2_b = ptr(f32(sd[p + "norm2.weight"])), ptr(f32(sd[p + "norm2.bias"]))
            L.fc1_w, L.fc1_b = ptr(bf(sd[p + "mlp.fc1.weight"])), ptr(f32(sd[p + "mlp.fc1.bias"]))
            L.fc2_w, L.fc2_b = ptr(bf(sd[p + "mlp.fc2.weight"])), ptr(f32(sd[p + "mlp.fc2.bias"]))
            L.ls2 = ptr(f32(sd[p + "layer_scale2.lambda1"]))
            # LayerNorm fold operands (include/medp_hip.h): W g in bf16, the row sums of THAT rounded matrix, b + W beta
            for name, W, b, g, beta in (("qkv", qkv_w, qkv_b, sd[p + "norm1.weight"], sd[p + "norm1.bias"]),
                                        ("fc1", sd[p + "mlp.fc1.weight"], sd[p + "mlp.fc1.bias"], sd[p + "norm2.weight"], sd[p + "norm2.bias"])):
                W32 = W.detach().to(torch.float32)
                wg = bf(W32 * g.detach().to(torch.float32)[None, :])
                setattr(L, name + "_wg", ptr(wg))
                setattr(L, name + "_cs", ptr(f32(wg.to(torch.float32).sum(dim=1))))
                setattr(L, name + "_b2", ptr(f32(b.detach().to(torch.float32) + W32 @ beta.detach().to(torch.float32))))
        w = MedpVitWeights()
        w.hidden, w.n_layers, w.n_heads, w.mlp_hidden = D, c.num_hidden_layers, c.num_attention_heads, D * c.mlp_ratio
        w.patch, w.pos_side, w.patch_kpad, w.ln_eps = c.patch_size, c.image_size // c.patch_size, kpad, c.layer_norm_eps
        w.patch_w = ptr(bf(pw))
        w.patch_b = ptr(f32(sd["embeddings.patch_embeddings.projection.bias"]))
        w.cls = ptr(f32(sd["embeddings.cls_token"].reshape(-1)))
        w.pos = ptr(f32(sd["embeddings.position_embeddings"].reshape(-1, D)))
        w.final_ln_w, w.final_ln_b = ptr(f32(sd["layernorm.weight"])), ptr(f32(sd["layernorm.bias"]))
        w.layers = ctypes.cast(layers, ctypes.POINTER(MedpVitLayer))
        self._prep, self._prep_key = (w, layers, keep), key
        return self._prep

    def forward(self, pixel_values: torch.Tensor, want_f32: bool = True, want_bf16: bool = False, layers=None, slot: int = 0, out16=None):
        """pixel_values fp32 [B,3,H,W] -> last_hidden_state after the final LayerNorm ([B, P+1, hidden]).
        `layers=(first, last)` (frozen path only): run that piece of the encoder (medp_vit_forward_part); the outputs exist
        only for the piece that ends at the last layer, the token stream waits in this module's workspace in between.
        `slot`: which of this module's workspaces to use — concurrent forwards on different streams (sub-batches of one step,
        graph_step.GraphedTeacherStep) need one each; `out16`: a preallocated bf16 [B, P+1, hidden] destination."""
        abi.require_gpu()
        if pixel_values.dim() != 4 or pixel_values.shape[1] != self.cfg.num_channels:
            raise ValueError("Make sure that the channel dimension of the pixel values match with the one set in the "
                             f"configuration. Expected {self.cfg.num_channels} but got {tuple(pixel_values.shape)}.")
        if any(p.requires_grad for p in self.parameters()) and torch.is_grad_enabled():
            # --unfreeze_cxr: the forward is composed of autograd nodes over the same kernels (cxr_train.py)
            from .cxr_train import forward_training
            if out16 is not None:
                raise ValueError("out16 is a frozen-path argument (the training path returns fp32 tokens with a gradient)")
            tok = forward_training(self, pixel_values)
            return (tok if want_f32 else None), (tok if want_bf16 else None)       # fp32 either way: it carries the gradient
        if Fn.precision() == "fp32" and layers is None:
            # the fp32 kernel mode reaches the frozen encoder too (cxr_train.forward_fp32: a parity instrument, not a fast path)
            from .cxr_train import forward_fp32
            tok = forward_fp32(self, pixel_values)
            t16 = None
            if want_bf16:
                t16 = tok.to(torch.bfloat16) if out16 is None else out16.copy_(tok)
            return (tok if want_f32 else None), t16
        w, _, _ = self._prepare()
        px = pixel_values.detach().to(torch.float32).contiguous()
        B, _, H, W = px.shape
        P = (H // self.cfg.patch_size) * (W // self.cfg.patch_size)
        need = lib().medp_vit_workspace_bytes(ctypes.byref(w), B, H, W)
        ws = self._ws.get(slot)
        if ws is None or ws.numel() < need or ws.device != px.device:
            ws = self._ws[slot] = torch.empty(need, dtype=torch.uint8, device=px.device)
        D = self.cfg.hidden_size
        first, last = (0, self.cfg.num_hidden_layers) if layers is None else layers
        fin = last == self.cfg.num_hidden_layers
        out32 = torch.empty((B, P + 1, D), dtype=torch.float32, device=px.device) if (want_f32 and fin) else None
        if out16 is not None:
            if out16.shape != (B, P + 1, D) or out16.dtype != torch.bfloat16 or not out16.is_contiguous() or out16.device != px.device:
                raise ValueError("out16 must be a contiguous bf16 [B, P+1, hidden] tensor on the input's device")
        else:
            out16 = torch.empty((B, P + 1, D), dtype=torch.bfloat16, device=px.device) if (want_bf16 and fin) else None
        check(lib().medp_vit_forward_part(ctypes.byref(w), ptr(px), B, H, W, ptr(out32), ptr(out16), ptr(ws), need, first, last,
                                          stream()), "vit_forward")
        return out32, out16


class CXREncoder(nn.Module):
    """Mirror of the reference's `CXREncoder` (model file :129-158): same constructor, attributes
    (`backbone`, `d_out`, `return_patches`, `_frozen`), `train()` override and return convention."""

    def __init__(self, model_name: str = "microsoft/rad-dino", freeze: bool = True, return_patches: bool = True, *,
                 config: Dinov2Cfg | None = None):
        super().__init__()
        self.backbone = Dinov2Backbone(config)
        if config is None and model_name and model_name != "synthetic":
            self._try_load_pretrained(model_name)
        self.d_out = self.backbone.cfg.hidden_size
        self.return_patches = return_patches
        if freeze:
            for p in self.backbone.parameters():
                p.requires_grad = False
            self.backbone.eval()
        self._frozen = freeze

    def _try_load_pretrained(self, model_name: str) -> None:
        """`AutoModel.from_pretrained(model_name)` (model file :137).  Like the reference this RAISES when the weights cannot
        be had (offline: no local path / cache) — a frozen random ViT would silently feed noise features to the fusion head.
        Random initialisation is only ever taken for `model_name == "synthetic"` or an explicit `config=` (the benchmark)."""
        from transformers import AutoModel
        try:
            hf = AutoModel.from_pretrained(model_name, local_files_only=True)
        except Exception as e:
            raise RuntimeError(f"CXREncoder: pretrained weights for {model_name!r} are not available locally "
                               f"({type(e).__name__}: {e}); pass a local checkpoint directory, or model_name='synthetic' "
                               "for seeded random weights (synthetic benchmark only)") from e
        self.backbone.load_state_dict(hf.state_dict(), strict=True)      # key / shape mismatches fail loudly too

    def train(self, mode: bool = True):
        super().train(mode)
        if self._frozen:
            self.backbone.eval()
        return self

    def forward(self, pixel_values: torch.Tensor):
        tokens, _ = self.backbone(pixel_values)
        cls = tokens[:, 0]
        if self.return_patches:
            return cls, tokens[:, 1:]
        return cls

    def forward_bf16(self, pixel_values: torch.Tensor, slot: int = 0, out=None):
        """Build-internal fast path: tokens as bf16 [B, P+1, D], directly consumable by the img_proj GEMM (fp32 tokens with a
        gradient when the encoder is being trained).  `slot` / `out`: see Dinov2Backbone.forward."""
        if Fn.precision() == "fp32" and out is None and not (any(p.requires_grad for p in self.backbone.parameters()) and torch.is_grad_enabled()):
            return self.backbone(pixel_values, want_f32=True, want_bf16=False)[0]      # fp32 kernel mode: the tokens stay fp32 all the way
        _, t16 = self.backbone(pixel_values, want_f32=False, want_bf16=True, slot=slot, out16=out)
        return t16

    def forward_bf16_part(self, pixel_values: torch.Tensor, first: int, last: int):
        """A piece of the frozen encoder (layers [first, last)); returns the bf16 tokens from the piece that finishes it, else None."""
        if any(p.requires_grad for p in self.backbone.parameters()):
            raise RuntimeError("the encoder can only be run in pieces when it is frozen")
        _, t16 = self.backbone(pixel_values, want_f32=False, want_bf16=True, layers=(first, last))
        return t16
