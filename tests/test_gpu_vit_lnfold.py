"""LayerNorm folded into the CXR encoder's block GEMMs (csrc/vit.hip, gemm_variants.h): LN(x) W^T + b = rstd (x (W g)^T) - rstd mean
colsum(W g) + (b + W beta).  Reference: the Dinov2 block of the reference's CXREncoder (model :152-158 -> transformers Dinov2Layer:
norm1 -> attention -> layer_scale1 -> residual -> norm2 -> mlp -> layer_scale2 -> residual).
  * GEMM level: the producer epilogue (fp32 residual result + bf16 copy + per-tile row sums) and the consumer epilogue (row-affine
    correction), through both tile kernels (one-tile-per-workgroup and persistent), against fp64 on the same bf16 operands;
  * encoder level, at the size where the fold is used (B 64, 224 x 224): tokens with the fold against tokens with the LayerNorm
    launches (same weights, same pixels), and both against the transformers reference fixture via test_gpu_vit's tolerance."""
import ctypes
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

from multimodal_edema_prediction_amd.abi import check, lib, ptr, stream  # noqa: E402

DEV = "cuda"


def _fold(A, W, C, M, N, K, bias=None, scale=None, residual=None, act=0, out_bf16=0, c2=None, stats_out=None, stats_in=None, tiles=0,
          colsum=None, eps=0.0, dim=0):
    L = lib()
    L.medp_dbg_gemm_fold.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int] * 3 + [ctypes.c_void_p] * 3 + [ctypes.c_int] * 2 + \
        [ctypes.c_void_p] * 3 + [ctypes.c_int, ctypes.c_void_p, ctypes.c_float, ctypes.c_int, ctypes.c_void_p]
    check(L.medp_dbg_gemm_fold(ptr(A), ptr(W), ptr(C), M, N, K, ptr(bias), ptr(scale), ptr(residual), act, out_bf16, ptr(c2), ptr(stats_out),
                               ptr(stats_in), tiles, ptr(colsum), eps, dim, stream()), "dbg_gemm_fold")


@pytest.mark.parametrize("M", [16448, 14080])
def test_producer_epilogue_writes_bf16_copy_and_row_sums(M):
    """proj-shaped GEMM (N = K = 768, fp32 in-place residual, LayerScale): x exactly as without the fold, c2 = bf16(x) bit for bit, and per
    row and 256-column tile the (sum, sum of squares) of the fp32 x; rows past M up to the tile boundary are zeros."""
    torch.manual_seed(1)
    N = K = 768
    a = (torch.randn(M, K, device=DEV) * 0.5).bfloat16()
    w = (torch.randn(N, K, device=DEV) * 0.05).bfloat16()
    bias, ls = torch.randn(N, device=DEV) * 0.1, torch.rand(N, device=DEV) + 0.5
    x0 = torch.randn(M, N, device=DEV)
    from multimodal_edema_prediction_amd import functional as Fn
    x_plain = x0.clone()
    Fn.gemm(a, w, bias=bias, scale=ls, residual=x_plain, out=x_plain)
    Mpad = (M + 255) // 256 * 256
    x = x0.clone()
    c2 = torch.zeros(M, N, device=DEV, dtype=torch.bfloat16)
    stats = torch.full((Mpad, 3, 2), float("nan"), device=DEV)
    _fold(a, w, x, M, N, K, bias=bias, scale=ls, residual=x, c2=c2, stats_out=stats)
    torch.cuda.synchronize()
    assert torch.equal(x, x_plain)
    assert torch.equal(c2, x.bfloat16())
    xt = x.double().view(M, 3, 256)
    s1, s2 = xt.sum(-1), (xt * xt).sum(-1)
    assert float((stats[:M, :, 0].double() - s1).abs().max()) <= 1e-5 * float(s1.abs().max() + 256)
    assert float((stats[:M, :, 1].double() - s2).abs().max()) <= 1e-5 * float(s2.abs().max())
    assert float(stats[M:].abs().max() if Mpad > M else 0.0) == 0.0


@pytest.mark.parametrize("N,act", [(2304, 0), (3072, 1), (768, 0)])
def test_consumer_epilogue_applies_layernorm(N, act):
    """qkv- / fc1-shaped GEMMs (persistent kernel: 585 / 780 tiles) and a 195-tile shape (one tile per workgroup): the folded form
    against fp64 LayerNorm -> Linear (-> GELU) on the same bf16 operands; and against the unfolded product path within bf16 rounding."""
    torch.manual_seed(2)
    M, K = 16448, 768
    x = torch.randn(M, K, device=DEV) * (0.5 + 2.0 * torch.rand(M, 1, device=DEV)) + 0.7 * torch.randn(M, 1, device=DEV)     # rows of different mean / scale
    x[:, 5] *= 30.0                                                                                                          # an outlier channel
    g, beta = torch.rand(K, device=DEV) + 0.5, 0.2 * torch.randn(K, device=DEV)
    Wm = torch.randn(N, K, device=DEV) * 0.05
    b = 0.1 * torch.randn(N, device=DEV)
    eps = 1e-6
    xb = x.bfloat16()
    wg = (Wm * g[None, :]).bfloat16()
    cs = wg.float().sum(1)
    b2 = b + Wm @ beta
    Mpad = (M + 255) // 256 * 256
    stats = torch.zeros(Mpad, 3, 2, device=DEV)
    xt = x.view(M, 3, 256)
    stats[:M, :, 0], stats[:M, :, 1] = xt.sum(-1), (xt * xt).sum(-1)
    out = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
    _fold(xb, wg, out, M, N, K, bias=b2, act=act, out_bf16=1, stats_in=stats, tiles=3, colsum=cs, eps=eps, dim=K)
    torch.cuda.synchronize()
    # fp64 reference of the FOLDED arithmetic on the same bf16 operands (what the kernel computes) ...
    mean = x.double().mean(1, keepdim=True)
    rstd = 1.0 / torch.sqrt(x.double().var(1, unbiased=False, keepdim=True) + eps)
    ref = rstd * (xb.double() @ wg.double().T) - rstd * mean * cs.double()[None, :] + b2.double()[None, :]
    if act:
        ref = torch.nn.functional.gelu(ref)
    err = (out.double() - ref).abs()
    tol = 1e-2 * ref.abs() + 2e-2
    assert not bool((err > tol).any()), f"{int((err > tol).sum())} off, max {float(err.max()):.3e}"
    # ... and the exact LayerNorm -> Linear in fp64: the fold changes which bf16 rounding the operand carries, not the result class
    exact = torch.nn.functional.layer_norm(x.double(), (K,), g.double(), beta.double(), eps) @ Wm.double().T + b.double()
    if act:
        exact = torch.nn.functional.gelu(exact)
    rel = float((out.double() - exact).norm() / exact.norm())
    assert rel < 8e-3, rel


def test_encoder_with_fold_matches_encoder_with_layernorm_launches(request):
    from multimodal_edema_prediction_amd.main_architecture_duett import CXREncoder
    torch.manual_seed(0)
    enc = CXREncoder("synthetic", freeze=True).to(DEV)
    px = torch.randn(64, 3, 224, 224, device=DEV)
    L = lib()
    prev = L.medp_dbg_vit_lnfold(1)
    request.addfinalizer(lambda: L.medp_dbg_vit_lnfold(prev))
    with torch.no_grad():
        t_fold = enc.backbone.forward(px, want_f32=True)[0].clone()
        L.medp_dbg_vit_lnfold(0)
        t_ln = enc.backbone.forward(px, want_f32=True)[0].clone()
    torch.cuda.synchronize()
    assert not torch.isnan(t_fold).any()
    rel = float((t_fold - t_ln).norm() / t_ln.norm())
    assert rel < 1e-2, rel                                   # two bf16 pipelines of 12 blocks: a few 1e-3 of each other
    assert float((t_fold - t_ln).abs().max()) < 0.15 * float(t_ln.abs().max())
    # the fold really ran: with it on, a second forward is bit-identical to the first (deterministic) but differs from the unfolded one
    L.medp_dbg_vit_lnfold(1)
    with torch.no_grad():
        t_fold2 = enc.backbone.forward(px, want_f32=True)[0]
    assert torch.equal(t_fold, t_fold2) and not torch.equal(t_fold, t_ln)
