"""Trainable form of the CXR encoder (`--unfreeze_cxr`, run.py:184-187; SURVEY.md §8(f1), second half).

The frozen encoder is one C call (`medp_vit_forward`); when any of its parameters requires a gradient the forward is
composed here from the autograd nodes of `autograd_ops` instead — the same HIP kernels (bf16 MFMA GEMMs incl. the transposed
weight-gradient GEMM, LayerNorm forward/backward, flash attention forward) plus two nodes of its own:

* `PatchEmbedFn`: conv14/stride14 as im2col + GEMM; backward is the weight-gradient GEMM on the saved bf16 columns;
* `AttnDh64Fn`: forward = the head-dim-64 MFMA kernel (also writing the logsumexp); backward = the MFMA flash backward of
  attention_dh64_bwd.hip (dQ and dK/dV launches of one templated kernel).

LayerScale and the residual add ride in the epilogue of the projection / fc2 GEMM (autograd_ops.LinearScaleResidualFn); the
position-grid resize has its own forward / backward kernels (PosBicubicFn).
Parity: tests/test_gpu_unfrozen_cxr.py against the oracle's autograd.
"""
from __future__ import annotations

import torch

from . import autograd_ops as A
from . import functional as Fn
from .abi import check, lib, ptr, stream

BF16, F32 = torch.bfloat16, torch.float32
_SMALL_BWD = __import__("os").environ.get("MEDP_ATTN_BWD", "") == "small"


class PatchEmbedFn(torch.autograd.Function):
    """pixels [B,3,H,W] fp32, weight [D,3,p,p], bias [D] -> patch tokens [B, P, D] fp32."""

    @staticmethod
    def forward(ctx, pixels, weight, bias, patch):
        B, C, H, W = pixels.shape
        gh, gw = H // patch, W // patch
        D, K = weight.shape[0], C * patch * patch
        kpad = (K + 7) // 8 * 8
        cols = torch.empty((B * gh * gw, kpad), dtype=BF16, device=pixels.device)
        px = pixels.detach().to(F32).contiguous()
        check(lib().medp_im2col_patch(ptr(px), ptr(cols), B, C, H, W, patch, kpad, stream()), "im2col_patch")
        wpad = torch.zeros((D, kpad), dtype=F32, device=pixels.device)
        wpad[:, :K] = weight.detach().reshape(D, K)
        y = Fn.gemm(cols, Fn.to_bf16(wpad), bias=bias.detach(), out_dtype=F32, k=kpad)
        ctx.save_for_backward(cols)
        ctx.wshape, ctx.K = weight.shape, K
        return y.view(B, gh * gw, D)

    @staticmethod
    def backward(ctx, dy):
        (cols,) = ctx.saved_tensors
        D = ctx.wshape[0]
        dy2 = dy.reshape(-1, D).contiguous()
        dw = Fn.gemm_tn(Fn.to_bf16(dy2), cols)[:, :ctx.K].reshape(ctx.wshape)
        return None, dw, Fn.colsum(dy2), None


class PosBicubicFn(torch.autograd.Function):
    """position_embeddings [1, 1 + s*s, D] -> [1, 1 + gh*gw, D]: bicubic resize of the patch grid (modeling_dinov2.py:57-95,
    align_corners=False), class position copied.  Forward = the kernel the frozen path uses; backward = its transpose written as a
    gather (deterministic).  torch's upsample_bicubic2d took 1.7 ms forward + 0.6 ms backward per step here."""

    @staticmethod
    def forward(ctx, pos, side, gh, gw):
        D = pos.shape[-1]
        p2 = pos.detach().reshape(-1, D).to(F32).contiguous()
        out = torch.empty((1 + gh * gw, D), dtype=F32, device=pos.device)
        check(lib().medp_pos_embed_bicubic(ptr(p2), ptr(out), side, gh, gw, D, stream()), "pos_embed_bicubic")
        ctx.dims = (side, gh, gw, D, pos.shape)
        return out.view(1, 1 + gh * gw, D)

    @staticmethod
    def backward(ctx, dout):
        side, gh, gw, D, shape = ctx.dims
        d2 = dout.reshape(-1, D).contiguous()
        dpos = torch.empty((1 + side * side, D), dtype=F32, device=dout.device)
        check(lib().medp_pos_embed_bicubic_bwd(ptr(d2), ptr(dpos), side, gh, gw, D, stream()), "pos_embed_bicubic_bwd")
        return dpos.view(shape), None, None, None


class AttnDh64Fn(torch.autograd.Function):
    """qkv fp32 [B*S, 3*H*64] (q | k | v column blocks) -> o fp32 [B*S, H*64].  Forward and backward are the head-dim-64 MFMA
    flash kernels (attention_dh64.hip / attention_dh64_bwd.hip); `MEDP_ATTN_BWD=small` selects the fp32 small-attention
    backward instead (the first, slow implementation — kept as a cross-check)."""

    @staticmethod
    def forward(ctx, qkv, B, S, H):
        qkv16 = Fn.to_bf16(qkv.contiguous())
        o, lse = Fn.attn_dh64_lse(qkv16, B, S, H, 0.125)
        ctx.save_for_backward(qkv, qkv16, o, lse)
        ctx.dims = (B, S, H)
        return o.float()

    @staticmethod
    def backward(ctx, do):
        qkv, qkv16, o, lse = ctx.saved_tensors
        B, S, H = ctx.dims
        D = H * 64
        if _SMALL_BWD:
            q3 = qkv.view(B, S, 3 * D)
            q, k, v = q3[..., :D], q3[..., D:2 * D], q3[..., 2 * D:]
            dq, dk, dv = Fn.attn_small_bwd(do.contiguous().view(B, S, D), q, k, v, B, S, S, H, 64, 0.125,
                                           q_batch_stride=S * 3 * D, kv_batch_stride=S * 3 * D)
            return torch.cat([dq, dk, dv], dim=-1).view(B * S, 3 * D), None, None, None
        return Fn.attn_dh64_bwd(do, qkv16, o, lse, B, S, H, 0.125), None, None, None


def forward_training(backbone, pixel_values: torch.Tensor) -> torch.Tensor:
    """Dinov2Model.forward(...).last_hidden_state with autograd through every parameter: [B, P+1, hidden] fp32."""
    c = backbone.cfg
    sd = dict(backbone.named_parameters())
    B, _, Hh, Ww = pixel_values.shape
    D, H = c.hidden_size, c.num_attention_heads
    if D != H * 64:
        raise ValueError("the attention kernel needs head dim 64")
    gh, gw = Hh // c.patch_size, Ww // c.patch_size
    patch = PatchEmbedFn.apply(pixel_values, sd["embeddings.patch_embeddings.projection.weight"],
                               sd["embeddings.patch_embeddings.projection.bias"], c.patch_size)
    cls = sd["embeddings.cls_token"].reshape(1, 1, D).expand(B, 1, D)
    pos = sd["embeddings.position_embeddings"].reshape(1, -1, D)
    side = c.image_size // c.patch_size
    if not (gh == side and gw == side):
        # modeling_dinov2.py:57-95: bicubic resize of the stored patch grid (align_corners=False)
        pos = PosBicubicFn.apply(sd["embeddings.position_embeddings"], side, gh, gw)
    x = torch.cat([cls, patch], dim=1) + pos
    S = x.shape[1]
    x = x.reshape(B * S, D)
    for l in range(c.num_hidden_layers):
        p = f"encoder.layer.{l}."
        h = A.layer_norm(x, sd[p + "norm1.weight"], sd[p + "norm1.bias"], c.layer_norm_eps)
        wqkv = torch.cat([sd[p + f"attention.attention.{n}.weight"] for n in ("query", "key", "value")], 0)
        bqkv = torch.cat([sd[p + f"attention.attention.{n}.bias"] for n in ("query", "key", "value")], 0)
        qkv = A.linear(h, wqkv, bqkv)
        att = AttnDh64Fn.apply(qkv, B, S, H)
        x = A.linear_scale_residual(att, sd[p + "attention.output.dense.weight"], sd[p + "attention.output.dense.bias"],
                                    sd[p + "layer_scale1.lambda1"], x)
        h = A.layer_norm(x, sd[p + "norm2.weight"], sd[p + "norm2.bias"], c.layer_norm_eps)
        f = A.gelu_dropout(A.linear(h, sd[p + "mlp.fc1.weight"], sd[p + "mlp.fc1.bias"]), 0.0, 0, 0)
        x = A.linear_scale_residual(f, sd[p + "mlp.fc2.weight"], sd[p + "mlp.fc2.bias"], sd[p + "layer_scale2.lambda1"], x)
    x = A.layer_norm(x, sd["layernorm.weight"], sd["layernorm.bias"], c.layer_norm_eps)
    return x.view(B, S, D)
