"""SURVEY §8(f1), second half: `--unfreeze_cxr` — the CXR encoder trains inside the teacher step.  The product path composes
the ViT forward from autograd nodes over the HIP kernels (cxr_train.py); the check is the CPU oracle with autograd through
its ViT restatement on the same seeded inputs: tokens, loss and the gradient of every ViT tensor."""
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pytestmark = pytest.mark.gpu


def test_trainable_cxr_encoder_matches_oracle_autograd():
    from multimodal_edema_prediction_amd.main_architecture_duett import CXREncoder
    from oracle import vit_ref
    dev = torch.device("cuda")
    torch.manual_seed(0)
    enc = CXREncoder("synthetic", freeze=False).to(dev).train()
    # non-trivial LayerScale / biases so every gradient path carries signal
    g = torch.Generator().manual_seed(5)
    with torch.no_grad():
        for k, p in enc.backbone.named_parameters():
            if k.endswith("lambda1"):
                p.copy_(0.5 + torch.rand(p.shape, generator=g))
            elif k.endswith(".bias"):
                p.copy_(0.05 * torch.randn(p.shape, generator=g))
    sd = {k: v.detach().float().cpu().clone() for k, v in enc.backbone.state_dict().items()}
    B = 2
    px = torch.randn(B, 3, 224, 224, generator=g)
    wsel = torch.randn(257, 768, generator=g) / 50.0              # a fixed linear read-out as the loss

    tok = enc.forward_bf16(px.to(dev))
    assert tok.dtype == torch.float32 and tok.requires_grad
    loss = (tok * wsel.to(dev)).sum()
    enc.zero_grad()
    loss.backward()

    for v in sd.values():
        if v.is_floating_point():
            v.requires_grad_(True)
    cls_ref, patches_ref = vit_ref.vit_forward(sd, vit_ref.VitCfg(), px)
    tok_ref = torch.cat([cls_ref.unsqueeze(1), patches_ref], 1)
    loss_ref = (tok_ref * wsel).sum()
    loss_ref.backward()

    err = (tok.detach().float().cpu() - tok_ref.detach()).abs()
    assert float(err.max()) < 6e-2 and float(err.mean()) < 6e-3, (float(err.max()), float(err.mean()))
    assert abs(float(loss.detach()) - float(loss_ref.detach())) <= 2e-2 * abs(float(loss_ref.detach())) + 1e-2
    named = dict(enc.backbone.named_parameters())
    n = 0
    for k, v in sd.items():
        if k not in named or v.grad is None or float(v.grad.norm()) == 0.0:
            continue
        if k.endswith("attention.key.bias"):
            # softmax is invariant to a constant added to every key (q.(k+b) shifts all scores of a query alike): the exact
            # gradient is zero and both sides hold rounding noise only
            assert float(named[k].grad.norm()) < 1e-3 * float(named[k.replace("key.bias", "value.bias")].grad.norm()) + 1e-4
            continue
        gk = named[k].grad
        assert gk is not None, k
        gk, want = gk.float().cpu(), v.grad
        cos = float((gk * want).sum() / (gk.norm() * want.norm() + 1e-30))
        rel = float((gk - want).norm() / (want.norm() + 1e-30))
        assert cos > 0.98 and rel < 0.25, (k, cos, rel)
        n += 1
    assert n >= 12 * 13 + 4, n
