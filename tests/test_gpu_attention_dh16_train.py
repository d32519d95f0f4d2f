"""DuETT's attention in training form on the matrix cores (csrc/attention_dh16_train.hip: 2 heads of dim 12 over 49 / 97 tokens,
dropout on the probabilities): forward and all three gradients against fp64 on the bf16-rounded operands, the dropout mask
against the fp32 VALU kernels of attention_small.hip (same counter hash), and the autograd function the student path uses."""
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pytestmark = pytest.mark.gpu
DEV = "cuda"


def _ref(qkv, do, H):
    B, N, D3 = qkv.shape
    D = D3 // 3
    dh = D // H
    x = qkv.bfloat16().double().requires_grad_(True)             # the kernels round q, k, v to bf16
    q, k, v = [x[..., i * D:(i + 1) * D].reshape(B, N, H, dh).transpose(1, 2) for i in range(3)]
    w = torch.softmax(q @ k.transpose(-1, -2) * dh ** -0.5, dim=-1)
    o = (w @ v).transpose(1, 2).reshape(B, N, D)
    o.backward(do.double())
    return o.detach(), x.grad


def _call(qkv, do, H, p=0.0, seed=0, sid=0, io16=False):
    from multimodal_edema_prediction_amd.abi import check, lib, ptr, stream
    B, N, D3 = qkv.shape
    D = D3 // 3
    dh = D // H
    if io16:
        qkv, do = qkv.to(torch.bfloat16), do.to(torch.bfloat16)
    o = torch.empty(B, N, D, device=DEV, dtype=qkv.dtype)
    lse, delta = torch.empty(B * H * N, device=DEV), torch.empty(B * H * N, device=DEV)
    dqkv = torch.full_like(qkv, float("nan"))
    check(lib().medp_attn_dh16_train_fwd(ptr(qkv), D3, ptr(o), D, ptr(lse), int(io16), B, N, H, dh, dh ** -0.5, p, seed, sid, stream()), "fwd")
    check(lib().medp_attn_dh16_train_bwd(ptr(do), D, ptr(qkv), D3, ptr(lse), ptr(delta), ptr(dqkv), D3, int(io16), B, N, H, dh, dh ** -0.5, p,
                                         seed, sid, stream()), "bwd")
    return o, dqkv


@pytest.mark.parametrize("B,N,H,dh,p", [(5, 97, 2, 12, 0.0), (7, 49, 2, 12, 0.3), (2, 130, 1, 16, 0.1), (1, 5, 3, 4, 0.0)])
def test_bf16_hand_over_form_has_the_bits_of_the_fp32_form(B, N, H, dh, p):
    """io_bf16 = 1 reads bf16 q | k | v / dO and writes bf16 o / dQ | dK | dV: the fp32 form rounds the same fp32 values to the same MFMA
    operands, so on inputs that are exactly representable in bf16 both compute the same numbers and differ only by the final rounding."""
    torch.manual_seed(3)
    qkv = (torch.randn(B, N, 3 * H * dh) * 0.8).to(DEV).to(torch.bfloat16).float()
    do = torch.randn(B, N, H * dh).to(DEV).to(torch.bfloat16).float()
    o32, g32 = _call(qkv, do, H, p, seed=5, sid=3)
    o16, g16 = _call(qkv, do, H, p, seed=5, sid=3, io16=True)
    assert o16.dtype == torch.bfloat16 and g16.dtype == torch.bfloat16 and not torch.isnan(g16.float()).any()
    assert torch.equal(o16, o32.to(torch.bfloat16)) and torch.equal(g16, g32.to(torch.bfloat16))


@pytest.mark.parametrize("B,N,H,dh", [(5, 97, 2, 12), (7, 49, 2, 12), (3, 17, 2, 12), (2, 130, 1, 16), (2, 257, 2, 8), (1, 5, 3, 4)])
def test_forward_and_gradients_against_fp64(B, N, H, dh):
    torch.manual_seed(0)
    qkv = (torch.randn(B, N, 3 * H * dh) * 0.8).to(DEV)
    do = torch.randn(B, N, H * dh).to(DEV)
    o, dqkv = _call(qkv, do, H)
    o_ref, g_ref = _ref(qkv.cpu(), do.cpu(), H)
    assert float((o.cpu().double() - o_ref).abs().max()) <= 1e-2 * float(o_ref.abs().max())          # bf16 probabilities / V
    assert not torch.isnan(dqkv).any()                                                                 # every element written
    err = float((dqkv.cpu().double() - g_ref).abs().max())
    assert err <= 2e-2 * float(g_ref.abs().max()), err
    a, b = dqkv.cpu().double().flatten(), g_ref.flatten()
    assert float(torch.dot(a, b) / (a.norm() * b.norm())) > 0.9995
    o2, dqkv2 = _call(qkv, do, H)
    assert torch.equal(o, o2) and torch.equal(dqkv, dqkv2)                                             # nothing is accumulated in memory


def test_dropout_mask_is_the_one_of_the_fp32_kernels_and_backward_regenerates_it():
    from multimodal_edema_prediction_amd import functional as Fn
    torch.manual_seed(1)
    B, N, H, dh, p = 4, 97, 2, 12, 0.3
    D = H * dh
    qkv = (torch.randn(B, N, 3 * D) * 0.8).to(DEV)
    do = torch.randn(B, N, D).to(DEV)
    o, dqkv = _call(qkv, do, H, p, seed=99, sid=7)
    o_valu = Fn.attn_small_fwd(qkv[..., :D], qkv[..., D:2 * D], qkv[..., 2 * D:], B, N, N, H, dh, dh ** -0.5, q_batch_stride=N * 3 * D,
                               kv_batch_stride=N * 3 * D, dropout_p=p, seed=99, stream_id=7)
    assert float((o - o_valu).abs().max()) <= 2e-2 * float(o_valu.abs().max())                        # same mask: only bf16 rounding apart
    o0, _ = _call(qkv, do, H)
    assert float((o - o0).abs().max()) > 0.1 * float(o0.abs().max())                                   # the mask does something
    # <dO, o(V)> is linear in V with gradient dV: <dV, V> = <dO, o> (V rounded to bf16 as the kernel sees it)
    v = qkv[..., 2 * D:].bfloat16().double()
    lhs = float((dqkv[..., 2 * D:].double() * v).sum())
    rhs = float((do.double() * o.double()).sum())
    assert abs(lhs - rhs) <= 2e-2 * max(1.0, abs(rhs))


def test_autograd_function_takes_the_mfma_path_and_matches_the_fp32_mode():
    from multimodal_edema_prediction_amd import functional as Fn
    from multimodal_edema_prediction_amd.duett_train import SelfAttnQKVFn
    torch.manual_seed(2)
    B, N, H, dh = 6, 49, 2, 12
    qkv = (torch.randn(B, N, 3 * H * dh) * 0.8).to(DEV)
    do = torch.randn(B, N, H * dh).to(DEV)

    def run(mode):
        x = qkv.clone().requires_grad_(True)
        with Fn.precision_mode(mode):
            o = SelfAttnQKVFn.apply(x, H, 0.0, 0, 0)
            o.backward(do)
        return o.detach(), x.grad

    o16, g16 = run("bf16")
    o32, g32 = run("fp32")
    assert float((o16 - o32).abs().max()) <= 1e-2 * float(o32.abs().max())
    assert float((g16 - g32).abs().max()) <= 3e-2 * float(g32.abs().max())
    assert not torch.equal(o16, o32)                                                                   # (two different kernels did run)
