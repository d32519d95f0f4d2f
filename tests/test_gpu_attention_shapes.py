"""CXR-encoder attention (head dim 64) over the sequence lengths that exercise every wave specialisation of the kernel:
1 / 2 / 3 query subtiles per wave, idle waves, ragged key tails, the multi-chunk path (S > 320) and the 8-wave kernel of long sequences (S >= 512).  Reference: the same
softmax(QK^T/8)V in fp32 torch on the bf16 operands (Dinov2SelfAttention as called from the reference model :152-158)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("B,S,H", [(2, 256, 2), (2, 257, 3), (2, 272, 2), (1, 320, 2), (1, 64, 2), (1, 100, 2), (1, 17, 1),
                                   (1, 1, 2), (1, 400, 2), (1, 1297, 2),
                                   # S >= 512: the 8-wave kernel (<= 2 subtiles per wave; idle waves in the last workgroup; 2 .. 7 key chunks)
                                   (2, 512, 2), (1, 513, 3), (2, 577, 2), (1, 1024, 2), (2, 1370, 3), (1, 2049, 1)])
def test_attn_dh64_shapes(B, S, H):
    from multimodal_edema_prediction_amd import functional as Fn
    torch.manual_seed(S)
    D = H * 64
    qkv = torch.randn(B * S, 3 * D, device="cuda").bfloat16()
    o = Fn.attn_dh64(qkv, B, S, H, 0.125).float().view(B, S, H, 64)
    q, k, v = [t.float().view(B, S, H, 64).permute(0, 2, 1, 3) for t in qkv.split(D, dim=1)]
    ref = (torch.softmax(q @ k.transpose(-1, -2) * 0.125, -1) @ v).permute(0, 2, 1, 3)
    err = (o - ref).abs()
    assert not torch.isnan(o).any()
    # bf16 P and bf16 output: 1e-2 absolute on O(1) values
    assert float(err.max()) < 1.5e-2, f"S={S}: max err {float(err.max()):.3e}"


@pytest.mark.parametrize("B,S,H", [(2, 257, 3), (1, 64, 2), (1, 100, 1), (1, 17, 2), (1, 400, 2), (1, 1370, 1)])
def test_attn_dh64_backward(B, S, H):
    """MFMA flash backward (attention_dh64_bwd.hip) against torch autograd of the same softmax(QK^T/8)V on the bf16 operands."""
    from multimodal_edema_prediction_amd import functional as Fn
    torch.manual_seed(100 + S)
    D = H * 64
    qkv = (torch.randn(B * S, 3 * D, device="cuda") * 0.7).bfloat16()
    dout = torch.randn(B * S, D, device="cuda")
    o, lse = Fn.attn_dh64_lse(qkv, B, S, H, 0.125)
    dqkv = Fn.attn_dh64_bwd(dout, qkv, o, lse, B, S, H, 0.125)
    ref_in = qkv.float().requires_grad_(True)
    q, k, v = [t.view(B, S, H, 64).permute(0, 2, 1, 3) for t in ref_in.split(D, dim=1)]
    oref = (torch.softmax(q @ k.transpose(-1, -2) * 0.125, -1) @ v).permute(0, 2, 1, 3).reshape(B * S, D)
    oref.backward(dout)
    # forward by-products: output and logsumexp (log2 domain)
    assert float((o.float() - oref.detach()).abs().max()) < 1.5e-2
    lse_ref = torch.logsumexp(q.detach() @ k.detach().transpose(-1, -2) * 0.125, -1) * 1.4426950408889634      # [B, H, S]
    assert float((lse - lse_ref).abs().max()) < 2e-3
    want = ref_in.grad
    assert not torch.isnan(dqkv).any()
    for name, lo in (("dq", 0), ("dk", D), ("dv", 2 * D)):
        g, w = dqkv[:, lo:lo + D], want[:, lo:lo + D]
        rel = float((g - w).norm() / (w.norm() + 1e-30))
        assert rel < 2e-2, (name, rel)                           # bf16 P / dS / dO operands, fp32 accumulation
        assert float((g - w).abs().max()) < 3e-2 * float(w.abs().max()) + 1e-3, name
