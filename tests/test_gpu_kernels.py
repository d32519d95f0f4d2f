"""GPU parity tests of the individual HIP kernels, called through the C ABI (functional.py -> abi.py -> libmedp_hip.so).
Reference = torch fp32/fp64 on the CPU of the same op on the same (bf16-representable) inputs; tolerances stated inline."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from multimodal_edema_prediction_amd import functional as Fn  # noqa: E402
from multimodal_edema_prediction_amd.abi import lib  # noqa: E402

DEV = "cuda"


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def bf_round(x):
    return x.to(torch.bfloat16).to(torch.float32)


def assert_close(got, want, rtol, atol, what=""):
    got = got.detach().float().cpu().double()
    want = want.detach().double()
    err = (got - want).abs()
    tol = atol + rtol * want.abs()
    bad = err > tol
    assert not bad.any(), f"{what}: {int(bad.sum())}/{bad.numel()} off, max err {float(err.max()):.3e} (tol {float(tol.min()):.1e})"


# ------------------------------------------------------------------------------------------------ GEMM
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (257, 768, 768), (16448 // 8, 2304, 768), (300, 72, 2328), (448, 256, 1024),
                                   (3136 // 4, 2328, 24), (130, 64, 1176), (64, 24, 408), (1000, 512, 792), (256, 3072, 768)])
def test_gemm_plain(M, N, K):
    a = bf_round(rnd(M, K, seed=1))
    w = bf_round(rnd(N, K, seed=2) / math.sqrt(K))
    want = a.double() @ w.double().T
    got = Fn.gemm(a.to(DEV).bfloat16(), w.to(DEV).bfloat16())
    # fp32 accumulation of exact bf16 products: error ~ 1e-6 * sqrt(K) relative to |a||w| row norms
    assert_close(got, want, 1e-4, 2e-4, f"gemm {M}x{N}x{K}")


def test_gemm_identity_asymmetric():
    """A = I against an ASYMMETRIC W catches a swapped accumulator row/col map (cdna guide §3)."""
    K = 128
    a = torch.eye(K)
    w = torch.arange(256 * K, dtype=torch.float32).reshape(256, K) % 251 - 125.0   # exact in bf16
    got = Fn.gemm(a.to(DEV).bfloat16(), w.to(DEV).bfloat16())
    assert torch.equal(got.cpu(), w.T.contiguous())


def test_gemm_epilogue_and_strides():
    M, N, K = 515, 772, 200
    a_full = bf_round(rnd(M, K + 56, seed=3))          # lda > K
    w = bf_round(rnd(N, K, seed=4) / math.sqrt(K))
    bias, scale, res = rnd(N, seed=5), 1 + 0.1 * rnd(N, seed=6), rnd(M, N, seed=7)
    a = a_full[:, :K]
    pre = a.double() @ w.double().T + bias.double()
    want = torch.nn.functional.gelu(pre) * scale.double() + res.double()
    ad = a_full.to(DEV).bfloat16()[:, :K]
    got = Fn.gemm(ad, w.to(DEV).bfloat16(), bias=bias.to(DEV), scale=scale.to(DEV), residual=res.to(DEV), act=1)
    assert_close(got, want, 2e-4, 3e-4, "gemm epilogue fp32")
    got16 = Fn.gemm(ad, w.to(DEV).bfloat16(), bias=bias.to(DEV), scale=scale.to(DEV), residual=res.to(DEV), act=1,
                    out_dtype=torch.bfloat16)
    assert_close(got16, want, 8e-3, 8e-3, "gemm epilogue bf16")     # one bf16 rounding of the result: 2^-8 relative
    # in-place residual (C aliases the residual), as the ViT block uses it
    x = res.to(DEV).clone()
    Fn.gemm(ad, w.to(DEV).bfloat16(), bias=bias.to(DEV), scale=scale.to(DEV), residual=x, out=x)
    assert_close(x, (a.double() @ w.double().T + bias.double()) * scale.double() + res.double(), 2e-4, 3e-4, "gemm in-place residual")


@pytest.mark.parametrize("M,N,K", [(3136, 512, 2328), (6208, 72, 1176), (3136, 72, 2328), (777, 100, 1160)])
def test_gemm_split_k_small_grids(M, N, K):
    """DuETT's skinny GEMMs fill 25-100 of 256 CUs with one tile each: they are split along K (medp_gemm_bf16_nt_ws) — partial
    sums in a caller workspace, summed in a fixed order by a second pass that applies the epilogue.  Same values as the fp64
    reference, every epilogue form, in-place residual, and bit-identical from run to run (no atomics)."""
    from multimodal_edema_prediction_amd.abi import lib
    assert lib().medp_gemm_nt_workspace_bytes(M, N, K) > 0                      # these shapes do take the split path
    assert lib().medp_gemm_nt_workspace_bytes(16448, 768, 768) == 0              # a full grid does not
    a = bf_round(rnd(M, K, seed=11))
    w = bf_round(rnd(N, K, seed=12) / math.sqrt(K))
    bias, res = rnd(N, seed=13), rnd(M, N, seed=14)
    ad, wd = a.to(DEV).bfloat16(), w.to(DEV).bfloat16()
    want = a.double() @ w.double().T
    assert_close(Fn.gemm(ad, wd), want, 1e-4, 2e-4, "split-K plain")
    g1 = Fn.gemm(ad, wd, bias=bias.to(DEV), act=1, out_dtype=torch.bfloat16)
    g2 = Fn.gemm(ad, wd, bias=bias.to(DEV), act=1, out_dtype=torch.bfloat16)
    assert torch.equal(g1, g2)
    assert_close(g1, torch.nn.functional.gelu(want + bias.double()), 8e-3, 8e-3, "split-K bias+gelu -> bf16")
    x = res.to(DEV).clone()
    Fn.gemm(ad, wd, residual=x, out=x)                                           # in place: C = A W^T + C
    assert_close(x, want + res.double(), 1e-4, 2e-4, "split-K in-place residual")


def test_gemm_rejects_bad_arguments():
    a = torch.zeros(8, 12, device=DEV, dtype=torch.bfloat16)
    w = torch.zeros(8, 12, device=DEV, dtype=torch.bfloat16)
    with pytest.raises(ValueError):
        Fn.gemm(a, w)             # K = 12 not a multiple of 8
    with pytest.raises(RuntimeError):
        Fn.gemm(torch.zeros(8, 16, dtype=torch.bfloat16), torch.zeros(8, 16, dtype=torch.bfloat16))   # CPU tensors: no fallback


# ------------------------------------------------------------------------------------------------ norms
@pytest.mark.parametrize("rows,D", [(5, 768), (1030, 256), (64 * 7, 256), (9, 2328)])
def test_layernorm_fwd_bwd(rows, D):
    x = rnd(rows, D, seed=1) * 2 + 0.3
    w, b = 1 + 0.1 * rnd(D, seed=2), 0.1 * rnd(D, seed=3)
    dy = rnd(rows, D, seed=4)
    xr = x.clone().double().requires_grad_(True)
    wr, br = w.clone().double().requires_grad_(True), b.clone().double().requires_grad_(True)
    yr = torch.nn.functional.layer_norm(xr, (D,), wr, br, 1e-5)
    yr.backward(dy.double())
    y, mean, rstd = Fn.layernorm(x.to(DEV), w.to(DEV), b.to(DEV), 1e-5, out_dtype=torch.float32, save_stats=True)
    assert_close(y, yr, 1e-5, 1e-5, "ln fwd fp32")
    y16 = Fn.layernorm(x.to(DEV), w.to(DEV), b.to(DEV), 1e-5, out_dtype=torch.bfloat16)
    assert_close(y16, yr, 8e-3, 8e-3, "ln fwd bf16")
    dx, dw, db = Fn.layernorm_bwd(dy.to(DEV), x.to(DEV), w.to(DEV), mean, rstd)
    assert_close(dx, xr.grad, 1e-4, 1e-5, "ln dx")
    assert_close(dw, wr.grad, 1e-4, 1e-4, "ln dw")
    assert_close(db, br.grad, 1e-4, 1e-4, "ln db")


@pytest.mark.parametrize("rows,D", [(7, 408), (3136 // 8, 2328), (100, 1176), (9, 2564), (5, 6216)])      # 6216: the two-pass backward
def test_scalenorm_fwd_bwd(rows, D):
    x = rnd(rows, D, seed=1)
    g = torch.tensor([1.13])
    dy = rnd(rows, D, seed=2)
    xr, gr = x.clone().double().requires_grad_(True), g.clone().double().requires_grad_(True)
    yr = xr / xr.norm(dim=-1, keepdim=True).clamp_min(1e-12) * math.sqrt(D) * gr
    yr.backward(dy.double())
    y, rn = Fn.scalenorm(x.to(DEV), g.to(DEV), out_dtype=torch.float32, save_rnorm=True)
    assert_close(y, yr, 1e-5, 1e-5, "scalenorm fwd")
    dx, dg = Fn.scalenorm_bwd(dy.to(DEV), x.to(DEV), g.to(DEV), rn)
    assert_close(dx, xr.grad, 1e-4, 1e-5, "scalenorm dx")
    assert_close(dg, gr.grad, 1e-4, 1e-3, "scalenorm dg")
    # the residual join, out of place: dx = add + (norm branch); `add` untouched
    from multimodal_edema_prediction_amd.abi import check, lib, ptr, stream
    add = rnd(rows, D, seed=3).to(DEV)
    add0 = add.clone()
    xd, dyd, gd = x.to(DEV), dy.to(DEV), g.to(DEV)
    dx2, dg2, ws = torch.empty_like(xd), torch.empty(1, device=DEV), torch.empty(rows, device=DEV)
    check(lib().medp_scalenorm_bwd_add(ptr(dyd), D, ptr(xd), D, ptr(gd), ptr(rn), ptr(add), D, ptr(dx2), D, ptr(dg2), ptr(ws), rows, D, stream()), "bwd_add")
    assert torch.equal(add, add0)
    assert torch.equal(dx2, dx + add0) and torch.equal(dg2, dg)


def test_colsum_and_cast_transpose():
    x = rnd(1000, 300, seed=1)
    assert_close(Fn.colsum(x.to(DEV)), x.double().sum(0), 1e-5, 1e-4, "colsum")
    x4 = rnd(37, 64, seed=2)
    assert torch.equal(Fn.to_bf16(x4.to(DEV)).cpu(), x4.bfloat16())
    xt = Fn.transpose_to_bf16(x.to(DEV))
    assert xt.shape == (300, 1000) and torch.equal(xt.cpu(), x.bfloat16().T.contiguous())
    xt2 = Fn.transpose_to_bf16(x[:, :99].bfloat16().to(DEV).contiguous())
    assert torch.equal(xt2.cpu(), x[:, :99].bfloat16().T.contiguous())
    y = rnd(33, 7, seed=3)         # rows padded to a multiple of 8 with zeros
    yt = Fn.transpose_to_bf16(y.to(DEV))
    assert yt.shape == (7, 40) and torch.equal(yt[:, :33].cpu(), y.bfloat16().T.contiguous()) and float(yt[:, 33:].abs().sum()) == 0


# ------------------------------------------------------------------------------------------------ attention
@pytest.mark.parametrize("B,S,H", [(2, 257, 12), (1, 1297, 2), (3, 64, 1), (1, 130, 3), (2, 321, 2)])
def test_attn_dh64(B, S, H):
    D = H * 64
    qkv = bf_round(rnd(B * S, 3 * D, seed=S))
    q, k, v = [t.reshape(B, S, H, 64).transpose(1, 2).double() for t in qkv.split(D, dim=1)]
    want = torch.softmax(q @ k.transpose(-1, -2) * 0.125, dim=-1) @ v
    want = want.transpose(1, 2).reshape(B * S, D)
    got = Fn.attn_dh64(qkv.to(DEV).bfloat16(), B, S, H, 0.125)
    # P is rounded to bf16 before the PV product (2^-9 relative per term) and the output once more
    assert_close(got, want, 1e-2, 1e-2, f"attn_dh64 B{B} S{S} H{H}")


def test_attn_dh64_forced_rescale():
    """A key whose score dominates arrives late: the online-softmax rescale branch must fire (cdna guide rule 26)."""
    B, S, H = 1, 320, 1
    qkv = bf_round(rnd(B * S, 192, seed=11) * 0.5)
    qkv[300, 64:128] = bf_round(qkv[5, 0:64] * 40)          # key 300 aligned with query 5, large
    q, k, v = [t.reshape(B, S, H, 64).transpose(1, 2).double() for t in qkv.split(64, dim=1)]
    want = (torch.softmax(q @ k.transpose(-1, -2) * 0.125, dim=-1) @ v).transpose(1, 2).reshape(S, 64)
    got = Fn.attn_dh64(qkv.to(DEV).bfloat16(), B, S, H, 0.125)
    assert_close(got, want, 1e-2, 1e-2, "attn_dh64 rescale")


@pytest.mark.parametrize("arrangement", ["whole_units", "whole_and_half_units"])
@pytest.mark.parametrize("case", ["late_key", "key256", "cls_query", "cls_key", "ramp"])
def test_attn_dh64_s257_rescale_paths(case, arrangement, request):
    """The S = 257 kernel (one workgroup per (batch, head), online softmax initialised from key 256, deferred rescale, class query
    split over the waves): inputs that FORCE each data-dependent path (cdna guide rule 26) against fp64, every row checked, plus the
    log-sum-exp the backward consumes.
      late_key : a dominant key in the LAST block for some patch queries -> the rescale branch fires late
      key256   : key 256 (the initial state) dominates -> no block ever moves the maximum (deferred path throughout)
      cls_query: the class query has a dominant key inside ONE wave's key slice -> the 8-way partial combine must rescale
      cls_key  : key 0 dominates every query
      ramp     : scores grow by a few units per block: growth both below and above the defer threshold"""
    B, S, H = 2, 257, 3
    D = H * 64
    # "whole_units" (the default arrangement): one workgroup per (b, h), two subtiles per wave; "whole_and_half_units" (MEDP_ATTN_S257_BALANCE=1,
    # measured slower): on a pretended 4 resident slots the 6 (b, h) run as 4 whole units + 2 x 2 half units (one subtile per wave).
    prev = lib().medp_dbg_attn_s257_slots(4 if arrangement == "whole_and_half_units" else 0)
    request.addfinalizer(lambda: lib().medp_dbg_attn_s257_slots(prev))
    qkv = bf_round(rnd(B * S, 3 * D, seed=23) * 0.5)
    x = qkv.view(B, S, 3, H, 64)
    if case == "late_key":
        x[0, 250, 1, 0] = bf_round(x[0, 37, 0, 0] * 30)
        x[1, 200, 1, 1] = bf_round(x[1, 256, 0, 1] * 30)
        x[1, 255, 1, 2] = bf_round(x[1, 140, 0, 2] * 30)            # (b 1, h 2) runs as half units
    elif case == "key256":
        x[:, 256, 1] = bf_round(x[:, 5, 0] * 12)
    elif case == "cls_query":
        x[0, 100, 1, 0] = bf_round(x[0, 0, 0, 0] * 25)
        x[1, 256, 1, 1] = bf_round(x[1, 0, 0, 1] * 25)
        x[1, 77, 1, 2] = bf_round(x[1, 0, 0, 2] * 25)
    elif case == "cls_key":
        x[:, 0, 1] = bf_round(x[:, 0, 1] * 6)
    else:
        ramp = torch.linspace(0.2, 3.0, S).view(1, S, 1, 1)
        x[:, :, 1] = bf_round(x[:, :, 1] * ramp)
    qkv = x.reshape(B * S, 3 * D)
    q, k, v = [t.reshape(B, S, H, 64).transpose(1, 2).double() for t in qkv.split(D, dim=1)]
    sc = q @ k.transpose(-1, -2) * 0.125
    want = (torch.softmax(sc, dim=-1) @ v).transpose(1, 2).reshape(B * S, D)
    got, lse = Fn.attn_dh64_lse(qkv.to(DEV).bfloat16(), B, S, H, 0.125)
    assert_close(got, want, 1e-2, 1e-2, f"attn_dh64 S=257 {case}")
    assert_close(Fn.attn_dh64(qkv.to(DEV).bfloat16(), B, S, H, 0.125), want, 1e-2, 1e-2, f"attn_dh64 S=257 {case} (no lse)")
    want_lse = torch.logsumexp(sc, dim=-1) * 1.4426950408889634          # log2 domain, [B, H, S]
    assert_close(lse, want_lse, 1e-4, 2e-3, f"attn_dh64 S=257 lse {case}")


@pytest.mark.parametrize("B,Lq,Lk,H,dh,p", [(3, 7, 256, 4, 64, 0.0), (2, 49, 49, 2, 12, 0.0), (2, 97, 97, 2, 12, 0.0),
                                            (2, 7, 7, 4, 64, 0.0), (1, 7, 1296, 4, 64, 0.0), (2, 33, 17, 2, 12, 0.0)])
def test_attn_small_fwd_bwd(B, Lq, Lk, H, dh, p):
    D = H * dh
    q, k, v, do = rnd(B, Lq, D, seed=1), rnd(B, Lk, D, seed=2), rnd(B, Lk, D, seed=3), rnd(B, Lq, D, seed=4)
    scale = dh ** -0.5
    qr, kr, vr = [t.clone().double().requires_grad_(True) for t in (q, k, v)]
    sp = lambda t, L: t.view(B, L, H, dh).transpose(1, 2)
    w = torch.softmax(sp(qr, Lq) @ sp(kr, Lk).transpose(-1, -2) * scale, dim=-1)
    o = (w @ sp(vr, Lk)).transpose(1, 2).reshape(B, Lq, D)
    o.backward(do.double())
    avg = torch.zeros(B, Lq, Lk, device=DEV)
    got = Fn.attn_small_fwd(q.to(DEV), k.to(DEV), v.to(DEV), B, Lq, Lk, H, dh, scale, attn_avg=avg)
    assert_close(got, o, 1e-4, 1e-5, "attn_small fwd")
    assert_close(avg, w.mean(dim=1), 1e-4, 1e-6, "attn_small averaged weights")
    dq, dk, dv = Fn.attn_small_bwd(do.to(DEV), q.to(DEV), k.to(DEV), v.to(DEV), B, Lq, Lk, H, dh, scale)
    assert_close(dq, qr.grad, 1e-4, 1e-5, "attn_small dq")
    assert_close(dk, kr.grad, 1e-4, 1e-5, "attn_small dk")
    assert_close(dv, vr.grad, 1e-4, 1e-5, "attn_small dv")


@pytest.mark.parametrize("B,Lq,Lk", [(3, 7, 256), (2, 7, 97), (2, 7, 7), (2, 8, 1000), (2, 5, 300), (1, 1, 64)])
def test_attn_few_query_kernels(B, Lq, Lk):
    """<= 8 queries, head dim 64, <= 1024 keys (the perceiver blocks): the thread-per-key kernels of attention_small.hip, forward
    and backward against fp64, and — with dropout on — against the wave-per-query forward, which draws the same mask."""
    H, dh = 4, 64
    D = H * dh
    q, k, v, do = rnd(B, Lq, D, seed=1), rnd(B, Lk, D, seed=2), rnd(B, Lk, D, seed=3), rnd(B, Lq, D, seed=4)
    scale = dh ** -0.5
    qr, kr, vr = [t.clone().double().requires_grad_(True) for t in (q, k, v)]
    sp = lambda t, L: t.view(B, L, H, dh).transpose(1, 2)
    w = torch.softmax(sp(qr, Lq) @ sp(kr, Lk).transpose(-1, -2) * scale, dim=-1)
    o = (w @ sp(vr, Lk)).transpose(1, 2).reshape(B, Lq, D)
    o.backward(do.double())
    qd, kd, vd = q.to(DEV), k.to(DEV), v.to(DEV)
    got = Fn.attn_small_fwd(qd, kd, vd, B, Lq, Lk, H, dh, scale)                          # no attn_avg: the few-query forward
    assert_close(got, o, 1e-4, 1e-5, "few-query fwd")
    dq, dk, dv = Fn.attn_small_bwd(do.to(DEV), qd, kd, vd, B, Lq, Lk, H, dh, scale)
    assert_close(dq, qr.grad, 1e-4, 1e-5, "few-query dq")
    assert_close(dk, kr.grad, 1e-4, 1e-5, "few-query dk")
    assert_close(dv, vr.grad, 1e-4, 1e-5, "few-query dv")
    args = dict(dropout_p=0.25, seed=77, stream_id=3)
    o_fq = Fn.attn_small_fwd(qd, kd, vd, B, Lq, Lk, H, dh, scale, **args)
    o_gen = Fn.attn_small_fwd(qd, kd, vd, B, Lq, Lk, H, dh, scale, attn_avg=torch.zeros(B, Lq, Lk, device=DEV), **args)
    assert_close(o_fq, o_gen.cpu(), 1e-4, 1e-5, "few-query fwd vs wave-per-query fwd under the same dropout mask")
    _, _, dv_d = Fn.attn_small_bwd(do.to(DEV), qd, kd, vd, B, Lq, Lk, H, dh, scale, **args)
    lhs, rhs = float((dv_d.double() * vd.double()).sum()), float((do.to(DEV).double() * o_fq.double()).sum())
    assert abs(lhs - rhs) <= 1e-4 * max(1.0, abs(rhs))                                     # backward regenerates the forward's mask


def test_attn_small_shared_query_and_strided_kv():
    """Perceiver cross blocks: one [7,256] query shared by the batch (batch stride 0) over keys that skip the CLS row."""
    B, Lq, Lk, H, dh = 3, 7, 16, 4, 64
    D = H * dh
    q = rnd(Lq, D, seed=1)
    kv = rnd(B, Lk + 1, 2 * D, seed=2)                    # [k | v] fused rows, first row of each batch (CLS) skipped
    k, v = kv[:, 1:, :D], kv[:, 1:, D:]
    sp = lambda t, L: t.reshape(-1, L, H, dh).transpose(1, 2).double()
    w = torch.softmax(sp(q.expand(B, -1, -1), Lq) @ sp(k, Lk).transpose(-1, -2) * dh ** -0.5, dim=-1)
    want = (w @ sp(v, Lk)).transpose(1, 2).reshape(B, Lq, D)
    kvd = kv.to(DEV)
    got = Fn.attn_small_fwd(q.to(DEV), kvd[:, 1:, :D], kvd[:, 1:, D:], B, Lq, Lk, H, dh, dh ** -0.5, q_batch_stride=0,
                            kv_batch_stride=(Lk + 1) * 2 * D)
    assert_close(got, want, 1e-4, 1e-5, "attn_small shared q / strided kv")


def test_attn_small_dropout_consistent_between_fwd_and_bwd():
    """With dropout on, backward must regenerate the forward's mask: check dV = P_dropped^T dO via linearity in V."""
    B, Lq, Lk, H, dh, p = 2, 7, 64, 4, 64, 0.25
    D = H * dh
    q, k, v, do = rnd(B, Lq, D, seed=1), rnd(B, Lk, D, seed=2), rnd(B, Lk, D, seed=3), rnd(B, Lq, D, seed=4)
    args = dict(dropout_p=p, seed=123, stream_id=5)
    o1 = Fn.attn_small_fwd(q.to(DEV), k.to(DEV), v.to(DEV), B, Lq, Lk, H, dh, 0.125, **args)
    o2 = Fn.attn_small_fwd(q.to(DEV), k.to(DEV), v.to(DEV), B, Lq, Lk, H, dh, 0.125, **args)
    assert torch.equal(o1, o2)                                           # deterministic in (seed, stream)
    o0 = Fn.attn_small_fwd(q.to(DEV), k.to(DEV), v.to(DEV), B, Lq, Lk, H, dh, 0.125)
    assert not torch.allclose(o1, o0)
    _, _, dv = Fn.attn_small_bwd(do.to(DEV), q.to(DEV), k.to(DEV), v.to(DEV), B, Lq, Lk, H, dh, 0.125, **args)
    # <dO, o(V)> is linear in V with gradient dV: <dV, V> must equal <dO, o1>
    lhs = float((dv.double() * v.to(DEV).double()).sum())
    rhs = float((do.to(DEV).double() * o1.double()).sum())
    assert abs(lhs - rhs) <= 1e-4 * max(1.0, abs(rhs))
    avg = torch.zeros(B, Lq, Lk, device=DEV)
    Fn.attn_small_fwd(q.to(DEV), k.to(DEV), v.to(DEV), B, Lq, Lk, H, dh, 0.125, attn_avg=avg, **args)
    frac_zero = float((avg == 0).float().mean())
    assert 0.0 < frac_zero < 0.05            # a weight is zero in the head-average only if all 4 heads dropped it (p^4 = 0.4 %)


def test_gelu_bwd():
    pre, dy = rnd(64, 100, seed=1) * 2, rnd(64, 100, seed=2)
    pr = pre.clone().double().requires_grad_(True)
    torch.nn.functional.gelu(pr).backward(dy.double())
    assert_close(Fn.gelu_bwd(dy.to(DEV), pre.to(DEV)), pr.grad, 1e-5, 1e-6, "gelu bwd")


@pytest.mark.parametrize("M,N,K", [(16448 // 4, 256, 768), (448, 1024, 256), (6144 // 2, 512, 256), (100, 64, 256), (3136, 72, 2328 // 3 // 8 * 8),
                                   (64, 128, 408), (2056, 256, 1176)])
def test_gemm_tn_weight_gradient(M, N, K):
    """dW = dY^T X with transposing LDS reads and split-m slabs, vs fp64 on bf16-representable operands."""
    dy = bf_round(rnd(M, N, seed=1))
    x = bf_round(rnd(M, K, seed=2))
    want = dy.double().T @ x.double()
    got = Fn.gemm_tn(dy.to(DEV).bfloat16(), x.to(DEV).bfloat16())
    assert got.shape == (N, K)
    assert_close(got, want, 1e-4, 2e-4 * math.sqrt(M), f"gemm_tn {M}x{N}x{K}")
    # asymmetric identity check: dY = one-hot rows picks rows of X
    dy1 = torch.zeros(M, N)
    dy1[torch.arange(min(M, N)), torch.arange(min(M, N))] = 1.0
    got1 = Fn.gemm_tn(dy1.to(DEV).bfloat16(), x.to(DEV).bfloat16())
    assert torch.equal(got1[:min(M, N)].cpu(), x[:min(M, N)])
