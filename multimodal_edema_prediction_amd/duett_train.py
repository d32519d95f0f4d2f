"""Training form of `DuettFeatureExtractor.encode` (model file :31-94): BatchNorm with batch statistics when the module is
in train(), dropout inside the encoders, and gradients to every DuETT parameter — the student-KD path
(`StudentModel`, engine.py:270-301) and a teacher with an unfrozen DuETT.  Composition of autograd Functions whose
forward/backward are HIP kernels (C ABI); torch is used for stacking the V per-variable parameter sets into grouped
operands and for autograd bookkeeping only.
"""
from __future__ import annotations

import torch

from . import autograd_ops as A
from . import functional as Fn
from .abi import check, lib, ptr, stream

F32, BF16 = torch.float32, torch.bfloat16
BN_EPS, BN_MOMENTUM = 1e-5, 0.1


# ------------------------------------------------------------------------------------------------ grouped tiny layers
class GLinearFn(torch.autograd.Function):
    """x [G,R,K], W [G,N,K], b [G,N] -> [G,R,N]"""

    @staticmethod
    def forward(ctx, x, W, b):
        x, W, b = x.contiguous(), W.contiguous(), b.contiguous()
        G, R, K = x.shape
        N = W.shape[1]
        y = torch.empty((G, R, N), dtype=F32, device=x.device)
        check(lib().medp_glinear_fwd(ptr(x), ptr(W), ptr(b), ptr(y), G, R, K, N, stream()), "glinear_fwd")
        ctx.save_for_backward(x, W)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, W = ctx.saved_tensors
        G, R, K = x.shape
        N = W.shape[1]
        dy = dy.contiguous()
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        dW, db = torch.empty_like(W), torch.empty((G, N), dtype=F32, device=x.device)
        ws = torch.empty(lib().medp_glinear_bwd_workspace_bytes(G, R, K, N) // 4, dtype=F32, device=x.device)
        check(lib().medp_glinear_bwd(ptr(dy), ptr(x), ptr(W), ptr(dx), ptr(dW), ptr(db), ptr(ws), G, R, K, N, stream()), "glinear_bwd")
        return dx, dW, db


class ActFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, mode):
        xc = x.contiguous()
        y = torch.empty_like(xc)
        check(lib().medp_act_fwd(ptr(xc), ptr(y), xc.numel(), mode, stream()), "act_fwd")
        ctx.save_for_backward(y)
        ctx.mode = mode
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        dyc = dy.contiguous()
        dx = torch.empty_like(y)
        check(lib().medp_act_bwd(ptr(dyc), ptr(y), ptr(dx), y.numel(), ctx.mode, stream()), "act_bwd")
        return dx, None


class GBatchNormFn(torch.autograd.Function):
    """BatchNorm over the R rows of each group.  x [G,R,C]; w, b, running_* [G,C] (running stats updated in place in train)."""

    @staticmethod
    def forward(ctx, x, w, b, rmean, rvar, batch_stats):
        x, w, b = x.contiguous(), w.contiguous(), b.contiguous()
        G, R, C = x.shape
        y = torch.empty_like(x)
        sm = torch.empty((G, C), dtype=F32, device=x.device)
        sv = torch.empty((G, C), dtype=F32, device=x.device)
        ws = torch.empty(lib().medp_gbn_workspace_bytes(G, R, C) // 4, dtype=F32, device=x.device) if batch_stats else None
        check(lib().medp_gbn_fwd(ptr(x), ptr(w), ptr(b), ptr(rmean), ptr(rvar), ptr(y), ptr(sm), ptr(sv), G, R, C, BN_EPS, BN_MOMENTUM,
                                 int(batch_stats), ptr(ws), stream()), "gbn_fwd")
        ctx.save_for_backward(x, w, sm, sv)
        ctx.batch_stats = batch_stats
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, sm, sv = ctx.saved_tensors
        G, R, C = x.shape
        dyc = dy.contiguous()
        dx, dw, db = torch.empty_like(x), torch.empty_like(w), torch.empty_like(w)
        ws = torch.empty(lib().medp_gbn_workspace_bytes(G, R, C) // 4, dtype=F32, device=x.device)
        check(lib().medp_gbn_bwd(ptr(dyc), ptr(x), ptr(w), ptr(sm), ptr(sv), ptr(dx), ptr(dw), ptr(db), G, R, C, BN_EPS,
                                 int(ctx.batch_stats), ptr(ws), stream()), "gbn_bwd")
        return dx, dw, db, None, None, None


class GroupMlpFn(torch.autograd.Function):
    """Linear(KIN, C) -> ReLU -> BatchNorm(C) -> Linear(C, E) of every group (variable) as fused kernels that recompute the hidden row
    instead of storing it (csrc/duett_embed_train.hip): x [G,R,KIN] -> [G,R,E].  Replaces GLinearFn -> ActFn -> GBatchNormFn -> GLinearFn
    (7 forward and ~20 backward launches over two 75-MB activations at cfg3) by 3 + 4 launches over x and dout."""

    @staticmethod
    def forward(ctx, x, W0, b0, bn_w, bn_b, rmean, rvar, W1, b1, batch_stats):
        x, W0, b0, bn_w, bn_b, W1, b1 = (t.contiguous() for t in (x, W0, b0, bn_w, bn_b, W1, b1))
        G, R, KIN = x.shape
        Ch, E = W0.shape[1], W1.shape[1]
        out = torch.empty((G, R, E), dtype=F32, device=x.device)
        sm = torch.empty((G, Ch), dtype=F32, device=x.device)
        sv = torch.empty((G, Ch), dtype=F32, device=x.device)
        ws = torch.empty(lib().medp_gmlp_workspace_bytes(G, R, KIN, Ch, E) // 4, dtype=F32, device=x.device)
        check(lib().medp_gmlp_fwd(ptr(x), ptr(W0), ptr(b0), ptr(bn_w), ptr(bn_b), ptr(rmean), ptr(rvar), ptr(W1), ptr(b1), ptr(out), ptr(sm),
                                  ptr(sv), G, R, KIN, Ch, E, BN_EPS, BN_MOMENTUM, int(batch_stats), ptr(ws), stream()), "gmlp_fwd")
        ctx.save_for_backward(x, W0, b0, bn_w, bn_b, sm, sv, W1)
        ctx.batch_stats = batch_stats
        return out

    @staticmethod
    def backward(ctx, dout):
        x, W0, b0, bn_w, bn_b, sm, sv, W1 = ctx.saved_tensors
        G, R, KIN = x.shape
        Ch, E = W0.shape[1], W1.shape[1]
        d = dout.contiguous()
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        dW0, db0, dbw, dbb = torch.empty_like(W0), torch.empty_like(b0), torch.empty_like(bn_w), torch.empty_like(bn_b)
        dW1, db1 = torch.empty_like(W1), torch.empty((G, E), dtype=F32, device=x.device)
        ws = torch.empty(lib().medp_gmlp_workspace_bytes(G, R, KIN, Ch, E) // 4, dtype=F32, device=x.device)
        check(lib().medp_gmlp_bwd(ptr(d), ptr(x), ptr(W0), ptr(b0), ptr(bn_w), ptr(bn_b), ptr(sm), ptr(sv), ptr(W1), ptr(dx), ptr(dW0), ptr(db0),
                                  ptr(dbw), ptr(dbb), ptr(dW1), ptr(db1), G, R, KIN, Ch, E, BN_EPS, int(ctx.batch_stats), ptr(ws), stream()), "gmlp_bwd")
        return dx, dW0, db0, dbw, dbb, None, None, dW1, db1, None


_FUSED_GMLP = __import__("os").environ.get("MEDP_DUETT_FUSED_GMLP", "1") == "1"


class EmbedInputsFn(torch.autograd.Function):
    """(value, n_obs_embedding[clip(int(count))]) per variable: xs_feats [B,T,2V+1], table [16,1] -> [V, B*T, 2]"""

    @staticmethod
    def forward(ctx, xs, table):
        xs = xs.contiguous()
        B, T, Fd = xs.shape
        V = (Fd - 1) // 2
        tab = table.reshape(-1).contiguous()
        xin = torch.empty((V, B * T, 2), dtype=F32, device=xs.device)
        check(lib().medp_embed_inputs_fwd(ptr(xs), ptr(tab), tab.numel(), ptr(xin), B, T, V, 2, stream()), "embed_inputs_fwd")
        ctx.save_for_backward(xs)
        ctx.cfg = (B, T, V, tab.numel(), tuple(table.shape))
        return xin

    @staticmethod
    def backward(ctx, dxin):
        (xs,) = ctx.saved_tensors
        B, T, V, rows, tshape = ctx.cfg
        nb = lib().medp_embed_inputs_bwd_blocks(B, T, V)
        part = torch.empty((nb, rows), dtype=F32, device=xs.device)
        d = dxin.contiguous()
        check(lib().medp_embed_inputs_bwd(ptr(xs), ptr(d), ptr(part), rows, B, T, V, 2, stream()), "embed_inputs_bwd")
        return None, Fn.colsum(part).view(tshape)


class PsiAssembleFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xs, var_out, tab_out, special):
        xs, var_out, tab_out, special = xs.contiguous(), var_out.contiguous(), tab_out.contiguous(), special.contiguous()
        B, T, Fd = xs.shape
        V = (Fd - 1) // 2
        E = special.shape[1]
        psi = torch.empty((B, T + 1, V + 1, E), dtype=F32, device=xs.device)
        check(lib().medp_psi_assemble_fwd(ptr(xs), ptr(var_out), ptr(tab_out), ptr(special), ptr(psi), B, T, V, E, stream()), "psi_assemble_fwd")
        ctx.save_for_backward(xs)
        ctx.cfg = (B, T, V, E, special.shape[0])
        return psi

    @staticmethod
    def backward(ctx, dpsi):
        (xs,) = ctx.saved_tensors
        B, T, V, E, n_special = ctx.cfg
        d = dpsi.contiguous()
        d_var = torch.empty((V, B * T, E), dtype=F32, device=xs.device)
        S = lib().medp_psi_assemble_bwd_slices(B, T, V)                     # cell slices per batch element (one workgroup each)
        tab_part = torch.empty((B, S, E), dtype=F32, device=xs.device)
        part = torch.empty((B * S, 2 * E), dtype=F32, device=xs.device)
        check(lib().medp_psi_assemble_bwd(ptr(xs), ptr(d), ptr(d_var), ptr(tab_part), ptr(part), B, T, V, E, stream()), "psi_assemble_bwd")
        d_special = torch.zeros((n_special, E), dtype=F32, device=xs.device)
        d_special[:2] = Fn.colsum(part).view(2, E)
        d_tab = tab_part[:, 0].contiguous() if S == 1 else Fn.colsum(tab_part.transpose(0, 1).reshape(S, B * E)).view(B, E)
        return None, d_var, d_tab, d_special


class AxisSwapFn(torch.autograd.Function):
    """[B, A1, A2, E] -> [B, A2, A1, E] (whole E-float cells)."""

    @staticmethod
    def forward(ctx, x):
        x = x.contiguous()
        B, A1, A2, E = x.shape
        y = torch.empty((B, A2, A1, E), dtype=F32, device=x.device)
        check(lib().medp_axis_swap(ptr(x), ptr(y), B, A1, A2, E, stream()), "axis_swap")
        return y

    @staticmethod
    def backward(ctx, dy):
        dy = dy.contiguous()
        B, A2, A1, E = dy.shape
        dx = torch.empty((B, A1, A2, E), dtype=F32, device=dy.device)
        check(lib().medp_axis_swap(ptr(dy), ptr(dx), B, A2, A1, E, stream()), "axis_swap(bwd)")
        return dx


class AddBcastFn(torch.autograd.Function):
    """a [B, ...] + b, where b is either [B, ...] or [...] broadcast over the batch."""

    @staticmethod
    def forward(ctx, a, b):
        a, b = a.contiguous(), b.contiguous()
        B = a.shape[0]
        per = a.numel() // B
        bcast = b.numel() == per
        out = torch.empty_like(a)
        check(lib().medp_add_bcast(ptr(a), ptr(b), ptr(out), per, B, int(bcast), stream()), "add_bcast")
        ctx.cfg = (B, per, bcast, tuple(b.shape))
        return out

    @staticmethod
    def backward(ctx, d):
        B, per, bcast, bshape = ctx.cfg
        db = Fn.colsum(d.contiguous().view(B, per)).view(bshape) if bcast else d
        return d, db


class SwapAddFn(torch.autograd.Function):
    """AxisSwapFn + AddBcastFn as one pass: x [B, A1, A2, E] -> [B, A2, A1*E] + positional embedding.  `add_last is None`: `add` [A2, A1*E] is
    shared by the batch (full_event_embedding, model :80-81).  Otherwise `add` [B, A2-1, A1*E] holds the per-sample rows and `add_last`
    [A1*E] the last row (the time embedding and the REP embedding, model :90): the [B, T+1, tt] concatenation of the reference is not built,
    and its backward hands the Linear behind `add` a view instead of a reduced / re-copied tensor."""

    @staticmethod
    def forward(ctx, x, add, add_last):
        x, add = x.contiguous(), add.contiguous()
        B, A1, A2, E = x.shape
        y = torch.empty((B, A2, A1 * E), dtype=F32, device=x.device)
        last = add_last.contiguous() if add_last is not None else None
        check(lib().medp_axis_swap_add(ptr(x), ptr(add), ptr(last), ptr(y), B, A1, A2, E, 1 if last is None else 3, stream()), "axis_swap_add")
        ctx.cfg = (B, A1, A2, E, tuple(add.shape), None if last is None else tuple(add_last.shape))
        return y

    @staticmethod
    def backward(ctx, d):
        B, A1, A2, E, ashape, lshape = ctx.cfg
        d = d.contiguous()
        dx = torch.empty((B, A1, A2, E), dtype=F32, device=d.device)
        check(lib().medp_axis_swap(ptr(d), ptr(dx), B, A2, A1, E, stream()), "axis_swap(bwd)")
        if lshape is None:
            return dx, Fn.colsum(d.view(B, A2 * A1 * E)).view(ashape), None
        d3 = d.view(B, A2, A1 * E)
        d_add = d3[:, :A2 - 1]                                                                # a view: the per-sample rows [B, A2-1, A1*E]
        d_last = Fn.colsum(d3[:, A2 - 1]).view(lshape)                                        # REP row: summed over the batch
        return dx, d_add, d_last


class ScaleNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, g, eps):
        xc = x.contiguous()
        y, rn = Fn.scalenorm(xc, g, eps, out_dtype=F32, save_rnorm=True)
        ctx.save_for_backward(xc, g, rn)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, g, rn = ctx.saved_tensors
        dx, dg = Fn.scalenorm_bwd(dy.contiguous(), x, g, rn, need_dg=True)
        return dx, dg, None


class ResidualScaleNormFn(torch.autograd.Function):
    """(x, ScaleNorm(x)) with the residual join folded into the backward: an encoder's x feeds BOTH the residual of the next Linear and
    the ScaleNorm in front of it, so autograd used to sum the two gradient branches with a torch `add` (12 launches of an 88-MB pass per
    student step, profiles/r02_kerneltrace_bench_student.txt).  Here the residual branch's gradient arrives as the gradient of the first
    output and `medp_scalenorm_bwd_add` writes  d_pass + (norm branch)  in the same pass that computes the norm branch: one pass less per
    join, no torch kernel, and the incoming gradient is only read."""

    @staticmethod
    def forward(ctx, x, g, eps):
        xc = x.contiguous()
        y, rn = Fn.scalenorm(xc, g, eps, out_dtype=F32, save_rnorm=True)
        ctx.save_for_backward(xc, g, rn)
        return xc.view_as(xc), y

    @staticmethod
    def backward(ctx, d_pass, dy):
        x, g, rn = ctx.saved_tensors
        if dy is None:
            return d_pass, None, None
        if d_pass is None or d_pass.dtype != F32:
            dx, dg = Fn.scalenorm_bwd(dy.contiguous(), x, g, rn, need_dg=True)
            return (dx if d_pass is None else dx + d_pass), dg, None
        D = x.shape[-1]
        dx, dg = _scalenorm_bwd_join(dy.contiguous().view(-1, D), x, g, rn, d_pass.contiguous().view(-1, D))
        return dx.view_as(x), dg, None


_FOLD_RESIDUAL_ADD = __import__("os").environ.get("MEDP_DUETT_FOLD_RESIDUAL_ADD", "1") == "1"
_MFMA_ATTN = __import__("os").environ.get("MEDP_DUETT_TRAIN_MFMA_ATTN", "1") == "1"


class SelfAttnQKVFn(torch.autograd.Function):
    """qkv [B, N, 3*H*dh] (q | k | v column blocks) -> [B, N, H*dh]; dense softmax, dropout on the probabilities.
    bf16 mode: the MFMA kernels of csrc/attention_dh16_train.hip (head dim <= 16, N <= 272); fp32 kernel mode or other shapes: the
    fp32 VALU kernels of csrc/attention_small.hip.  Both draw the same dropout mask."""

    @staticmethod
    def forward(ctx, qkv, H, p, seed, sid):
        qkv = qkv.contiguous()
        B, N, D3 = qkv.shape
        D = D3 // 3
        dh = D // H
        ctx.cfg = (H, p, seed, sid)
        if _MFMA_ATTN and Fn.precision() != "fp32":
            o = torch.empty((B, N, D), dtype=torch.float32, device=qkv.device)
            lse = torch.empty((B * H * N,), dtype=torch.float32, device=qkv.device)
            rc = lib().medp_attn_dh16_train_fwd(ptr(qkv), D3, ptr(o), D, ptr(lse), 0, B, N, H, dh, dh ** -0.5, p, seed, sid, stream())
            if rc != -2:
                check(rc, "attn_dh16_train_fwd")
                ctx.save_for_backward(qkv, lse)
                return o
        o = Fn.attn_small_fwd(qkv[..., :D], qkv[..., D:2 * D], qkv[..., 2 * D:], B, N, N, H, dh, dh ** -0.5, q_batch_stride=N * D3,
                              kv_batch_stride=N * D3, dropout_p=p, seed=seed, stream_id=sid)
        ctx.save_for_backward(qkv)
        return o

    @staticmethod
    def backward(ctx, do):
        qkv = ctx.saved_tensors[0]
        H, p, seed, sid = ctx.cfg
        B, N, D3 = qkv.shape
        D = D3 // 3
        dh = D // H
        dqkv = torch.empty_like(qkv)
        do2 = do.contiguous().view(B * N, D)
        if len(ctx.saved_tensors) == 2:                       # the forward ran on the matrix cores: so does the backward
            lse = ctx.saved_tensors[1]
            delta = torch.empty_like(lse)
            check(lib().medp_attn_dh16_train_bwd(ptr(do2), D, ptr(qkv), D3, ptr(lse), ptr(delta), ptr(dqkv), D3, 0, B, N, H, dh, dh ** -0.5, p, seed,
                                                 sid, stream()), "attn_dh16_train_bwd")
            return dqkv, None, None, None, None
        base = dqkv.data_ptr()
        check(lib().medp_attn_small_bwd(ptr(do2), D, ptr(qkv), D3, N * D3, qkv.data_ptr() + 4 * D, qkv.data_ptr() + 8 * D, D3, N * D3, base, D3,
                                        base + 4 * D, D3, base + 8 * D, 0, N * D3, B, N, N, H, dh, dh ** -0.5, p, seed, sid, stream()),
              "attn_small_bwd(qkv)")
        return dqkv, None, None, None, None


# ------------------------------------------------------------------------------------------------ fused halves of an encoder block
_FUSED_NODES = __import__("os").environ.get("MEDP_DUETT_FUSED_NODES", "1") == "1"


def _scalenorm_bwd_join(dh, x, g, rn, d_pass):
    """(d_pass + d ScaleNorm(x) / dx applied to dh, dg [1]): the residual join of a pre-norm half in the backward as ONE pass, out of place —
    `d_pass` (the gradient of the half's output, which the residual hands straight through) is only read."""
    x2 = x.view(-1, x.shape[-1])
    rows, D = x2.shape
    dx = torch.empty((rows, D), dtype=F32, device=x.device)
    dg = torch.empty(1, dtype=F32, device=x.device)
    ws = torch.empty(rows, dtype=F32, device=x.device)
    check(lib().medp_scalenorm_bwd_add(ptr(dh), D, ptr(x2), D, ptr(g), ptr(rn), ptr(d_pass), D, ptr(dx), D, ptr(dg), ptr(ws), rows, D, stream()),
          "scalenorm_bwd_add")
    return dx, dg


class AttnHalfFn(torch.autograd.Function):
    """x + to_out(Attention(ScaleNorm(x)))  — the attention half of an x_transformers pre-norm block (duett/duett.py:95-105) as ONE autograd
    node with 16-bit hand-overs inside: ScaleNorm writes the qkv GEMM's bf16 operand, that GEMM writes bf16 q | k | v, the MFMA attention
    (csrc/attention_dh16_train.hip, io_bf16 = 1) reads them and writes bf16 o, the out-projection adds the residual in its epilogue.
    Backward: one cast of dY, then dO, dQ | dK | dV in bf16 between the kernels, and the ScaleNorm backward joined with dY (the residual's
    gradient) in one out-of-place pass.
    Every value is rounded to bf16 exactly where the separate nodes (ScaleNormFn -> LinearFn -> SelfAttnQKVFn -> LinearFn) round it, so
    the results are bit-identical; 13 cast / transpose / concatenation launches per block and step fewer."""

    @staticmethod
    def forward(ctx, x, g, eps, wq, wk, wv, wo, H, p, seed, sid):
        xc = x.contiguous()
        B, N, D = xc.shape
        ws = (wq, wk, wv)
        Dv = wq.shape[0]
        dh = Dv // H
        h16, rn = Fn.scalenorm(xc, g, eps, out_dtype=BF16, save_rnorm=True)
        qkv16 = Fn.gemm(h16.view(B * N, D), A.weights_cat_bf16(ws), out_dtype=BF16, k=D)             # [B*N, 3 Dv]
        o16 = torch.empty((B * N, Dv), dtype=BF16, device=x.device)
        lse = torch.empty((B * H * N,), dtype=F32, device=x.device)
        check(lib().medp_attn_dh16_train_fwd(ptr(qkv16), 3 * Dv, ptr(o16), Dv, ptr(lse), 1, B, N, H, dh, dh ** -0.5, p, seed, sid, stream()),
              "attn_dh16_train_fwd(bf16)")
        y = Fn.gemm(o16, A.weight_bf16(wo), residual=xc.view(B * N, D), out_dtype=F32, k=Dv)
        ctx.save_for_backward(xc, g, rn, h16, qkv16, lse, o16, wq, wk, wv, wo)
        ctx.cfg = (H, p, seed, sid)
        return y.view(B, N, D)

    @staticmethod
    def backward(ctx, dy):
        xc, g, rn, h16, qkv16, lse, o16, wq, wk, wv, wo = ctx.saved_tensors
        H, p, seed, sid = ctx.cfg
        B, N, D = xc.shape
        ws = (wq, wk, wv)
        Dv = wq.shape[0]
        dh = Dv // H
        dy2 = dy.contiguous().view(B * N, D)
        dy16 = Fn.operand(dy2)
        do16 = Fn.gemm(dy16, A.weight_t_bf16(wo), out_dtype=BF16, k=D)                                # [B*N, Dv]
        dwo = Fn.gemm_tn(dy16, o16)
        dqkv16 = torch.empty_like(qkv16)
        delta = torch.empty_like(lse)
        check(lib().medp_attn_dh16_train_bwd(ptr(do16), do16.stride(0), ptr(qkv16), 3 * Dv, ptr(lse), ptr(delta), ptr(dqkv16), 3 * Dv, 1, B, N, H,
                                             dh, dh ** -0.5, p, seed, sid, stream()), "attn_dh16_train_bwd(bf16)")
        dhid = Fn.gemm(dqkv16, A.weights_cat_t_bf16(ws), out_dtype=F32, k=3 * Dv)                     # [B*N, D]
        dwq, dwk, dwv = Fn.gemm_tn(dqkv16, h16.view(B * N, D)).split([w.shape[0] for w in ws], 0)
        dx, dg = _scalenorm_bwd_join(dhid, xc, g, rn, dy2)
        return dx.view(B, N, D), dg, None, dwq, dwk, dwv, dwo, None, None, None, None


class FeedForwardHalfFn(torch.autograd.Function):
    """x + W2 dropout(gelu(W1 ScaleNorm(x) + b1)) + b2  — the feed-forward half as ONE node: ScaleNorm writes W1's bf16 operand, GELU + dropout
    write W2's, the residual rides in W2's epilogue; the backward casts dY once and `medp_gelu_dropout_bwd_bf16` writes d(pre-activation) in
    fp32 (bias gradient) and bf16 (operand of both W1 gradient GEMMs).  Bit-identical to the separate nodes."""

    @staticmethod
    def forward(ctx, x, g, eps, w1, b1, w2, b2, p, seed, sid):
        xc = x.contiguous()
        B, N, D = xc.shape
        h16, rn = Fn.scalenorm(xc, g, eps, out_dtype=BF16, save_rnorm=True)
        f = Fn.gemm(h16.view(B * N, D), A.weight_bf16(w1), bias=b1, out_dtype=F32, k=D)               # pre-activation, kept for the backward
        a16 = torch.empty(f.shape, dtype=BF16, device=x.device)
        check(lib().medp_gelu_dropout_fwd_bf16(ptr(f), ptr(a16), f.numel(), p, seed, sid, stream()), "gelu_dropout_fwd_bf16")
        y = Fn.gemm(a16, A.weight_bf16(w2), bias=b2, residual=xc.view(B * N, D), out_dtype=F32, k=w2.shape[1])
        ctx.save_for_backward(xc, g, rn, h16, f, a16, w1, w2)
        ctx.cfg = (p, seed, sid)
        return y.view(B, N, D)

    @staticmethod
    def backward(ctx, dy):
        xc, g, rn, h16, f, a16, w1, w2 = ctx.saved_tensors
        p, seed, sid = ctx.cfg
        B, N, D = xc.shape
        dy2 = dy.contiguous().view(B * N, D)
        dy16 = Fn.operand(dy2)
        da = Fn.gemm(dy16, A.weight_t_bf16(w2), out_dtype=F32, k=D)                                   # [B*N, hidden]
        dw2 = Fn.gemm_tn(dy16, a16)
        db2 = Fn.colsum(dy2)
        df = torch.empty_like(f)
        df16 = torch.empty(f.shape, dtype=BF16, device=f.device)
        check(lib().medp_gelu_dropout_bwd_bf16(ptr(da), ptr(f), ptr(df), ptr(df16), f.numel(), p, seed, sid, stream()), "gelu_dropout_bwd_bf16")
        dhid = Fn.gemm(df16, A.weight_t_bf16(w1), out_dtype=F32, k=w1.shape[0])                       # [B*N, D]
        dw1 = Fn.gemm_tn(df16, h16.view(B * N, D))
        db1 = Fn.colsum(df)
        dx, dg = _scalenorm_bwd_join(dhid, xc, g, rn, dy2)
        return dx.view(B, N, D), dg, None, dw1, db1, dw2, db2, None, None, None


def _fused_nodes_ok(m, x):
    """The fused halves take the shapes the MFMA attention and the transposing weight-gradient GEMM take, in bf16 kernel mode."""
    if not (_FUSED_NODES and _MFMA_ATTN and _FOLD_RESIDUAL_ADD) or Fn.precision() == "fp32" or not x.requires_grad:
        return False
    a, ff = m.layers[0][1], m.layers[1][1].ff
    B, N, D = x.shape
    Dv = a.to_q.weight.shape[0]
    dh = Dv // m.heads
    hid = ff[0][0].weight.shape[0]
    return (D % 8 == 0 and Dv % 8 == 0 and hid % 8 == 0 and a.to_out.bias is None and ff[0][0].bias is not None and ff[2].bias is not None
            and lib().medp_attn_dh16_train_supported(B, N, m.heads, dh, 3 * Dv, Dv) == 1)


# ------------------------------------------------------------------------------------------------ the composition
def _bind_stacked_bn(model):
    """Make the V per-variable BatchNorm running statistics (and batch counters) views of stacked buffers so the grouped kernel
    updates them in place and ONE add advances all V counters (state_dict keys and values are unchanged; re-bound if
    .to(device) replaced the buffers)."""
    bns = [m[3].batch_norm for m in model.embedding_layers]
    st = getattr(model, "_bn_stack", None)
    ok = st is not None and all(bn.running_mean.data_ptr() == st[0][i].data_ptr() and bn.running_var.data_ptr() == st[1][i].data_ptr()
                                and bn.num_batches_tracked.data_ptr() == st[2][i].data_ptr() for i, bn in enumerate(bns))
    if not ok:
        rm = torch.stack([bn.running_mean.detach() for bn in bns]).contiguous()
        rv = torch.stack([bn.running_var.detach() for bn in bns]).contiguous()
        nb = torch.stack([bn.num_batches_tracked.detach() for bn in bns]).contiguous()
        for i, bn in enumerate(bns):
            bn.running_mean = rm[i]
            bn.running_var = rv[i]
            bn.num_batches_tracked = nb[i]
        model._bn_stack = (rm, rv, nb)
    return model._bn_stack


def _mlp_bn(x, lin0_w, lin0_b, bn_w, bn_b, rm, rv, lin1_w, lin1_b, act_mode, batch_stats):
    if (_FUSED_GMLP and lin1_w is not None and act_mode == 0
            and lib().medp_gmlp_supported(x.shape[-1], lin0_w.shape[1], lin1_w.shape[1]) == 1):
        return GroupMlpFn.apply(x, lin0_w, lin0_b, bn_w, bn_b, rm, rv, lin1_w, lin1_b, batch_stats)
    h = GLinearFn.apply(x, lin0_w, lin0_b)
    a = ActFn.apply(h, act_mode)
    hb = GBatchNormFn.apply(a, bn_w, bn_b, rm, rv, batch_stats)
    return hb if lin1_w is None else GLinearFn.apply(hb, lin1_w, lin1_b)


def encoder_training(m, x, eps, final_norm, training, seed, sid):
    """One x_transformers-style encoder block (see oracle/xt_encoder.py) on x [B, N, D]."""
    a, ff = m.layers[0][1], m.layers[1][1].ff
    p = float(m.dropout) if training else 0.0
    if _fused_nodes_ok(m, x):
        x = AttnHalfFn.apply(x, m.layers[0][0][0].g, eps, a.to_q.weight, a.to_k.weight, a.to_v.weight, a.to_out.weight, m.heads, p, seed, sid)
        x = FeedForwardHalfFn.apply(x, m.layers[1][0][0].g, eps, ff[0][0].weight, ff[0][0].bias, ff[2].weight, ff[2].bias, p, seed, sid + 1)
        return ScaleNormFn.apply(x, m.final_norm.g, eps) if final_norm else x
    norm = (lambda t, g: ResidualScaleNormFn.apply(t, g, eps)) if (_FOLD_RESIDUAL_ADD and x.requires_grad) else \
        (lambda t, g: (t, ScaleNormFn.apply(t, g, eps)))
    x, h = norm(x, m.layers[0][0][0].g)
    qkv = A.linear_cat(h, (a.to_q.weight, a.to_k.weight, a.to_v.weight))
    o = SelfAttnQKVFn.apply(qkv, m.heads, p, seed, sid)
    x = A.linear(o, a.to_out.weight, None, residual=x)
    x, h = norm(x, m.layers[1][0][0].g)
    f = A.linear(h, ff[0][0].weight, ff[0][0].bias)
    f = A.gelu_dropout(f, p, seed, sid + 1)
    x = A.linear(f, ff[2].weight, ff[2].bias, residual=x)
    if final_norm:
        x = ScaleNormFn.apply(x, m.final_norm.g, eps)
    return x


def encode_training(model, x):
    from .duett import FINAL_NORM, SCALENORM_EPS
    xs_static, xs_feats, xs_times, _ = x
    xs_static = xs_static.detach().to(F32).contiguous()
    xs_feats = xs_feats.detach().to(F32).contiguous()
    xs_times = xs_times.detach().to(F32).contiguous()
    B, T, Fd = xs_feats.shape
    V, E = model.d_time_series_num, model.d_embedding
    if Fd != 2 * V + 1:
        raise ValueError(f"xs_feats must be [B, T, 2V+1] with V={V}, got {tuple(xs_feats.shape)}")
    if T != model.masked_transform_timesteps:
        raise ValueError(f"this backbone was built for n_timesteps={model.masked_transform_timesteps}, got T={T}")
    bs = bool(model.training)                                           # BatchNorm: batch statistics in train(), running in eval()
    el = model.embedding_layers
    rm, rv, nbt = _bind_stacked_bn(model)
    # per-variable MLPs as ONE grouped pass (group = variable)                                  (model :45-55)
    xin = EmbedInputsFn.apply(xs_feats, model.n_obs_embedding.weight)
    var_out = _mlp_bn(xin, torch.stack([m[0].weight for m in el]), torch.stack([m[0].bias for m in el]),
                      torch.stack([m[3].batch_norm.weight for m in el]), torch.stack([m[3].batch_norm.bias for m in el]), rm, rv,
                      torch.stack([m[4].weight for m in el]), torch.stack([m[4].bias for m in el]), 0, bs)
    te = model.tab_encoder                                                                     # (model :57)
    tbn = te[3].batch_norm
    tab_out = _mlp_bn(xs_static.unsqueeze(0), te[0].weight.unsqueeze(0), te[0].bias.unsqueeze(0), tbn.weight.unsqueeze(0),
                      tbn.bias.unsqueeze(0), tbn.running_mean.unsqueeze(0), tbn.running_var.unsqueeze(0), te[4].weight.unsqueeze(0),
                      te[4].bias.unsqueeze(0), 0, bs)[0]
    psi = PsiAssembleFn.apply(xs_feats, var_out, tab_out, model.special_embeddings.weight)      # (model :53-66)
    # time embedding: Linear(1,h) -> tanh -> BN -> Linear(h, tt) ; hidden padded to a multiple of 8 for the MFMA GEMM   (model :67-69)
    tm = model.full_time_embedding
    mbn = tm[2].batch_norm
    hb = _mlp_bn(xs_times.reshape(1, B * T, 1), tm[0].weight.unsqueeze(0), tm[0].bias.unsqueeze(0), mbn.weight.unsqueeze(0),
                 mbn.bias.unsqueeze(0), mbn.running_mean.unsqueeze(0), mbn.running_var.unsqueeze(0), None, None, 1, bs)[0]   # [B*T, h]
    Hd = hb.shape[1]
    pad = (-Hd) % 8
    hb_p = torch.cat([hb, hb.new_zeros(hb.shape[0], pad)], 1) if pad else hb
    w3_p = torch.cat([tm[3].weight, tm[3].weight.new_zeros(tm[3].weight.shape[0], pad)], 1) if pad else tm[3].weight
    tt = E * (V + 1)
    temb = A.linear(hb_p, w3_p, tm[3].bias).view(B, T, tt)
    fused_swap = __import__("os").environ.get("MEDP_DUETT_FUSED_SWAP", "1") == "1"
    if not fused_swap:
        time_emb = torch.cat([temb, model.full_rep_embedding.weight.view(1, 1, -1).expand(B, -1, -1)], 1)     # [B, T+1, tt]
    if bs:
        with torch.no_grad():
            nbt += 1                                      # all V per-variable counters at once (views of one stacked buffer)
            tbn.num_batches_tracked += 1
            mbn.num_batches_tracked += 1
    seed = A.next_seed() if (model.training and model.transformer_dropout > 0) else 0
    T1, V1 = T + 1, V + 1
    for l, (ev, tv) in enumerate(zip(model.event_transformers, model.time_transformers)):
        if fused_swap:
            xe = SwapAddFn.apply(psi, model.full_event_embedding.weight, None)                        # (model :80-81) [B, V1, T1*E]
        else:
            xe = AxisSwapFn.apply(psi).view(B, V1, T1 * E)
            xe = AddBcastFn.apply(xe, model.full_event_embedding.weight)
        xe = encoder_training(ev, xe, SCALENORM_EPS, FINAL_NORM, model.training, seed, 100 + 10 * l)    # (model :81)
        if fused_swap:
            xt = SwapAddFn.apply(xe.view(B, V1, T1, E), temb, model.full_rep_embedding.weight.view(-1))   # (model :90) [B, T1, V1*E]
        else:
            xt = AxisSwapFn.apply(xe.view(B, V1, T1, E)).view(B, T1, V1 * E)
            xt = AddBcastFn.apply(xt, time_emb)
        psi = encoder_training(tv, xt, SCALENORM_EPS, FINAL_NORM, model.training, seed, 105 + 10 * l).view(B, T1, V1, E)   # (model :91)
    return psi.flatten(2)
