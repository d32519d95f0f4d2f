"""Trainable form of the CXR encoder (`--unfreeze_cxr`, run.py:184-187; SURVEY.md §8(f1), second half).

The frozen encoder is one C call (`medp_vit_forward`); when any of its parameters requires a gradient the forward is composed
here from autograd nodes over the same HIP kernels:

* `PatchEmbedFn`: conv14/stride14 as im2col + GEMM; backward is the weight-gradient GEMM on the saved bf16 columns;
* `PosBicubicFn`: the position-grid resize, forward and (gather-form, deterministic) backward kernels;
* `AttnHalfFn` / `MlpHalfFn`: the two halves of a block, each ONE node with a hand-chained backward — LayerNorm, qkv GEMM, flash
  attention (forward with logsumexp / MFMA flash backward of attention_dh64_bwd.hip) and the fc1 pre-activation all hand bf16
  to the next kernel, LayerScale + residual ride in the epilogue of the projection / fc2 GEMM, and their gradients (dW, db,
  dlambda) come from the unscaled transposed GEMM, so no activation-sized elementwise or cast kernel runs between the stages;
  only the residual stream is fp32.  (`AttnDh64Fn` is the stand-alone attention node of the first, op-by-op composition.)

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


def _ls_linear_backward(dy2, dyb, x16, weight, bias, lam, need_dx=True):
    """Gradients of out = (x W^T + b) * lam + res given dY (fp32) and its bf16 copy: (dX fp32 | None, dW, db, dlam) — no
    activation-sized elementwise kernel (see autograd_ops.LinearScaleResidualFn)."""
    G = Fn.gemm_tn(dyb, x16)
    s = Fn.colsum(dy2)
    lam_d, w_d = lam.detach(), weight.detach()
    dx = Fn.gemm(dyb, Fn.transpose_to_bf16((lam_d[:, None] * w_d).contiguous()), out_dtype=F32, k=weight.shape[0]) if need_dx else None
    return dx, lam_d[:, None] * G, lam_d * s, (w_d * G).sum(dim=1) + bias.detach() * s


class AttnHalfFn(torch.autograd.Function):
    """x + lam * (dense(attention(qkv(LayerNorm(x)))) + b): the attention half of a Dinov2 block (modeling_dinov2.py:342-381) as ONE
    autograd node.  Inside, every GEMM operand is produced in bf16 by the kernel before it (LayerNorm -> bf16, qkv GEMM -> bf16,
    attention -> bf16) and the backward is chained by hand, so no fp32 copy of qkv / attention output exists and no cast kernel runs
    between the stages; only the residual stream x is fp32.  The three projection weights stay separate parameters."""

    @staticmethod
    def forward(ctx, x, ln_w, ln_b, wq, wk, wv, bq, bk, bv, wo, bo, lam, eps, B, S, H):
        x = x.contiguous()
        h16, mean, rstd = Fn.layernorm(x, ln_w.detach(), ln_b.detach(), eps, out_dtype=BF16, save_stats=True)
        wqkv = torch.cat([wq.detach(), wk.detach(), wv.detach()], 0)
        bqkv = torch.cat([bq.detach(), bk.detach(), bv.detach()], 0)
        qkv16 = Fn.gemm(h16, Fn.to_bf16(wqkv), bias=bqkv, out_dtype=BF16)
        o16, lse = Fn.attn_dh64_lse(qkv16, B, S, H, 0.125)
        out = Fn.gemm(o16, A.weight_bf16(wo), bias=bo.detach(), scale=lam.detach().contiguous(), residual=x, out_dtype=F32)
        ctx.save_for_backward(x, mean, rstd, h16, qkv16, o16, lse, ln_w, wqkv, wo, bo, lam)
        ctx.dims = (B, S, H)
        return out

    @staticmethod
    def backward(ctx, dy):
        x, mean, rstd, h16, qkv16, o16, lse, ln_w, wqkv, wo, bo, lam = ctx.saved_tensors
        B, S, H = ctx.dims
        D = H * 64
        dy2 = dy.contiguous()
        dyb = Fn.to_bf16(dy2)
        do, dwo, dbo, dlam = _ls_linear_backward(dy2, dyb, o16, wo, bo, lam)
        dqkv = Fn.attn_dh64_bwd(do, qkv16, o16, lse, B, S, H, 0.125)                  # fp32 [M, 3D]
        dqkvb = Fn.to_bf16(dqkv)
        dwqkv = Fn.gemm_tn(dqkvb, h16)
        dbqkv = Fn.colsum(dqkv)
        dh = Fn.gemm(dqkvb, Fn.transpose_to_bf16(wqkv), out_dtype=F32, k=3 * D)
        dx_ln, dlnw, dlnb = Fn.layernorm_bwd(dh, x, ln_w.detach(), mean, rstd)
        dx = dy2 + dx_ln
        return (dx, dlnw, dlnb, dwqkv[:D], dwqkv[D:2 * D], dwqkv[2 * D:], dbqkv[:D], dbqkv[D:2 * D], dbqkv[2 * D:], dwo, dbo, dlam,
                None, None, None, None)


class MlpHalfFn(torch.autograd.Function):
    """x + lam * (fc2(gelu(fc1(LayerNorm(x)))) + b2): the MLP half of a Dinov2 block as one autograd node.  fc1 writes its
    pre-activation in bf16 (the only copy kept for the backward), gelu runs bf16 -> bf16, fc2 carries LayerScale + residual in its
    epilogue; backward: d(gelu) in bf16 straight into both fc1 gradient GEMMs, db1 from the transposed GEMM against a ones column."""

    @staticmethod
    def forward(ctx, x, ln_w, ln_b, w1, b1, w2, b2, lam, eps):
        x = x.contiguous()
        h16, mean, rstd = Fn.layernorm(x, ln_w.detach(), ln_b.detach(), eps, out_dtype=BF16, save_stats=True)
        pre16 = Fn.gemm(h16, A.weight_bf16(w1), bias=b1.detach(), out_dtype=BF16)
        f16 = torch.empty_like(pre16)
        check(lib().medp_gelu_bf16_fwd(ptr(pre16), ptr(f16), pre16.numel(), stream()), "gelu_bf16_fwd")
        out = Fn.gemm(f16, A.weight_bf16(w2), bias=b2.detach(), scale=lam.detach().contiguous(), residual=x, out_dtype=F32)
        ctx.save_for_backward(x, mean, rstd, h16, pre16, f16, ln_w, w1, w2, b2, lam)
        return out

    @staticmethod
    def backward(ctx, dy):
        x, mean, rstd, h16, pre16, f16, ln_w, w1, w2, b2, lam = ctx.saved_tensors
        dy2 = dy.contiguous()
        dyb = Fn.to_bf16(dy2)
        G2 = Fn.gemm_tn(dyb, f16)
        s = Fn.colsum(dy2)
        lam_d, w2_d = lam.detach(), w2.detach()
        dw2, db2, dlam = lam_d[:, None] * G2, lam_d * s, (w2_d * G2).sum(dim=1) + b2.detach() * s
        df16 = Fn.gemm(dyb, Fn.transpose_to_bf16((lam_d[:, None] * w2_d).contiguous()), out_dtype=BF16, k=w2.shape[0])
        dpre16 = torch.empty_like(df16)
        check(lib().medp_gelu_bf16_bwd(ptr(df16), ptr(pre16), ptr(dpre16), df16.numel(), stream()), "gelu_bf16_bwd")
        dw1 = Fn.gemm_tn(dpre16, h16)
        ones = torch.ones((dpre16.shape[0], 8), dtype=BF16, device=dy.device)
        db1 = Fn.gemm_tn(dpre16, ones)[:, 0].contiguous()                       # column sums of a bf16 matrix, fp32 accumulation
        dh = Fn.gemm(dpre16, A.weight_t_bf16(w1), out_dtype=F32, k=w1.shape[0])
        dx_ln, dlnw, dlnb = Fn.layernorm_bwd(dh, x, ln_w.detach(), mean, rstd)
        return dy2 + dx_ln, dlnw, dlnb, dw1, db1, dw2, db2, dlam, None


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
        a = p + "attention.attention."
        x = AttnHalfFn.apply(x, sd[p + "norm1.weight"], sd[p + "norm1.bias"], sd[a + "query.weight"], sd[a + "key.weight"],
                             sd[a + "value.weight"], sd[a + "query.bias"], sd[a + "key.bias"], sd[a + "value.bias"],
                             sd[p + "attention.output.dense.weight"], sd[p + "attention.output.dense.bias"],
                             sd[p + "layer_scale1.lambda1"], c.layer_norm_eps, B, S, H)
        x = MlpHalfFn.apply(x, sd[p + "norm2.weight"], sd[p + "norm2.bias"], sd[p + "mlp.fc1.weight"], sd[p + "mlp.fc1.bias"],
                            sd[p + "mlp.fc2.weight"], sd[p + "mlp.fc2.bias"], sd[p + "layer_scale2.lambda1"], c.layer_norm_eps)
    x = A.layer_norm(x, sd["layernorm.weight"], sd["layernorm.bias"], c.layer_norm_eps)
    return x.view(B, S, D)


@torch.no_grad()
def forward_fp32(backbone, pixel_values: torch.Tensor) -> torch.Tensor:
    """The FROZEN encoder under the fp32 kernel mode (functional.set_precision("fp32")): every GEMM with fp32 operands
    (`medp_gemm_f32_nt`: bias / exact-erf GELU / LayerScale + residual in its epilogue), LayerNorm in fp32, attention by the fp32
    small-attention kernel (head dim 64, <= 1536 tokens).  A parity instrument (logits <= 1e-4 against the CPU restatement down to the
    pixels), ~50x slower than the bf16 path; the patch gather and the position add are data movement done by torch."""
    c = backbone.cfg
    sd = {k: v.detach().to(F32) for k, v in backbone.named_parameters()}
    px = pixel_values.detach().to(F32)
    B, C, Hh, Ww = px.shape
    D, H, P = c.hidden_size, c.num_attention_heads, c.patch_size
    if D != H * 64:
        raise ValueError("the attention kernel needs head dim 64")
    gh, gw = Hh // P, Ww // P
    if 1 + gh * gw > 1536:
        raise ValueError("fp32 mode: the small-attention kernel takes at most 1536 tokens")
    cols = px[:, :, :gh * P, :gw * P].unfold(2, P, P).unfold(3, P, P).permute(0, 2, 3, 1, 4, 5).reshape(B * gh * gw, C * P * P).contiguous()
    patch = Fn.gemm(cols, sd["embeddings.patch_embeddings.projection.weight"].reshape(D, C * P * P).contiguous(),
                    bias=sd["embeddings.patch_embeddings.projection.bias"]).view(B, gh * gw, D)
    side = c.image_size // P
    pos = sd["embeddings.position_embeddings"].reshape(1, -1, D)
    if not (gh == side and gw == side):
        pos = PosBicubicFn.apply(sd["embeddings.position_embeddings"], side, gh, gw)
    x = (torch.cat([sd["embeddings.cls_token"].reshape(1, 1, D).expand(B, 1, D), patch], dim=1) + pos).contiguous()
    S = x.shape[1]
    x = x.view(B * S, D)
    for l in range(c.num_hidden_layers):
        p = f"encoder.layer.{l}."
        a = p + "attention.attention."
        h = Fn.layernorm(x, sd[p + "norm1.weight"], sd[p + "norm1.bias"], c.layer_norm_eps, out_dtype=F32)
        wqkv = torch.cat([sd[a + "query.weight"], sd[a + "key.weight"], sd[a + "value.weight"]], 0).contiguous()
        bqkv = torch.cat([sd[a + "query.bias"], sd[a + "key.bias"], sd[a + "value.bias"]], 0).contiguous()
        qkv = Fn.gemm(h, wqkv, bias=bqkv).view(B, S, 3 * D)
        o = Fn.attn_small_fwd(qkv[..., :D], qkv[..., D:2 * D], qkv[..., 2 * D:], B, S, S, H, 64, 0.125, q_batch_stride=S * 3 * D,
                              kv_batch_stride=S * 3 * D)
        x = Fn.gemm(o.reshape(B * S, D), sd[p + "attention.output.dense.weight"].contiguous(), bias=sd[p + "attention.output.dense.bias"],
                    scale=sd[p + "layer_scale1.lambda1"].contiguous(), residual=x)
        h = Fn.layernorm(x, sd[p + "norm2.weight"], sd[p + "norm2.bias"], c.layer_norm_eps, out_dtype=F32)
        f = Fn.gemm(h, sd[p + "mlp.fc1.weight"].contiguous(), bias=sd[p + "mlp.fc1.bias"], act=1)
        x = Fn.gemm(f, sd[p + "mlp.fc2.weight"].contiguous(), bias=sd[p + "mlp.fc2.bias"], scale=sd[p + "layer_scale2.lambda1"].contiguous(),
                    residual=x)
    x = Fn.layernorm(x, sd["layernorm.weight"], sd["layernorm.bias"], c.layer_norm_eps, out_dtype=F32)
    return x.view(B, S, D)
