"""CXR encoder (ViT-B/14) through `medp_vit_forward` against (a) the golden fixture made by transformers' Dinov2Model
on CPU and (b) the CPU oracle, same seeded weights and pixels.  bf16 GEMM inputs / fp32 accumulation, fp32 residual
stream and norms: tolerance 3e-2 absolute on the final-LayerNorm tokens (unit scale), mean error far below."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from helpers import load_npz, load_shapes, synth_state_dict  # noqa: E402
from multimodal_edema_prediction_amd.cxr import CXREncoder  # noqa: E402


@pytest.fixture(scope="module")
def encoder():
    enc = CXREncoder("synthetic", freeze=True, return_patches=True)
    enc.backbone.load_state_dict(synth_state_dict(load_shapes("shapes.json")["vit"], seed=3), strict=True)
    return enc.to("cuda")


def test_state_dict_keys_match_dinov2(encoder):
    assert sorted(encoder.backbone.state_dict()) == sorted(load_shapes("shapes.json")["vit"])


def test_vit_against_golden(encoder):
    gold = load_npz("vit_b14.npz")
    g = torch.Generator().manual_seed(99)
    px224 = torch.randn(2, 3, 224, 224, generator=g)
    px512 = torch.randn(1, 3, 512, 512, generator=g)
    cls, patches = encoder(px224.cuda())
    o = torch.cat((cls.unsqueeze(1), patches), 1).cpu()
    assert o.shape == (2, 257, 768)
    want = torch.from_numpy(gold["out224_rows"])
    err = (o[:, list(gold["rows224"])] - want).abs()
    assert float(err.max()) < 3e-2 and float(err.mean()) < 4e-3, (float(err.max()), float(err.mean()))
    serr = (o.sum(-1) - torch.from_numpy(gold["out224_sum"])).abs()
    assert float(serr.max()) < 0.5, float(serr.max())          # sum of 768 unit-scale values, each within ~4e-3 on average
    cls, patches = encoder(px512.cuda())                       # 512² → 1297 tokens, bicubic position grid, 5 key chunks
    o = torch.cat((cls.unsqueeze(1), patches), 1).cpu()
    assert o.shape == (1, 1297, 768)
    err = (o[:, list(gold["rows512"])] - torch.from_numpy(gold["out512_rows"])).abs()
    assert float(err.max()) < 3e-2 and float(err.mean()) < 4e-3, (float(err.max()), float(err.mean()))


def test_vit_against_oracle_full_tensor(encoder):
    from oracle import vit_ref
    sd = synth_state_dict(load_shapes("shapes.json")["vit"], seed=3)
    g = torch.Generator().manual_seed(7)
    px = torch.randn(1, 3, 224, 224, generator=g)
    with torch.no_grad():
        c, p = vit_ref.vit_forward(sd, vit_ref.VitCfg(), px)
    cls, patches = encoder(px.cuda())
    err = torch.cat(((cls.cpu() - c).abs().flatten(), (patches.cpu() - p).abs().flatten()))
    assert float(err.max()) < 3e-2 and float(err.mean()) < 4e-3, (float(err.max()), float(err.mean()))
    t16 = encoder.forward_bf16(px.cuda())
    assert t16.dtype == torch.bfloat16 and float((t16[:, 1:].float().cpu() - p).abs().max()) < 5e-2
