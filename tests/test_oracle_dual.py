"""the oracle's restatement of the `DualPathologyPerceiver` teacher (oracle/fusion_ref.dual_perceiver_forward,
teacher_dual_forward) against the fixture that the reference's OWN code produced — its commented-out class text executed inside
its imported model module, around its live `TeacherModel(dual_pathology_mode=True)` branch (tests/golden/make_golden_dual.py)."""
import numpy as np
import torch

from helpers import load_npz, load_shapes, synth_state_dict, t
from multimodal_edema_prediction_amd.cohort import CohortCfg, make_batch
from oracle import duett_ref, fusion_ref, losses_ref, vit_ref
from oracle.step_ref import split_teacher_sd

B, T, V, DS, K = 8, 32, 16, 8, 7


def _ctx():
    gold = load_npz("teacher_dual_cfg1.npz")
    shapes = load_shapes("shapes.json")
    sd = synth_state_dict(shapes["teacher_dual"], seed=5)
    for k, v in synth_state_dict(shapes["vit"], seed=3).items():
        sd["cxr.backbone." + k] = v
    from tests_dual_common import cxr_head_state
    hs = cxr_head_state()["classifier_state_dict"]
    sd["pretrained_cxr_head.weight"], sd["pretrained_cxr_head.bias"] = hs["1.weight"], hs["1.bias"]
    sd["cxr_head_keep_idx"] = t(gold["keep_idx"]).long()
    tb = make_batch(CohortCfg(n_timesteps=T, n_vars=V, d_static=DS, image_size=224, seed=1234), 100, B, mode="teacher")
    return gold, sd, tb


def _forward(sd, tb, **kw):
    dsd, vsd = split_teacher_sd(sd)
    dcfg = duett_ref.DuettCfg(d_static_num=DS, d_time_series_num=V, n_timesteps=T)
    xin = duett_ref.feats_to_input((tb["x_ts"], tb["x_static"], list(tb["bin_ends"])), max_len=T)
    with torch.no_grad():
        ts_tokens = duett_ref.encode(dsd, dcfg, xin)
        cls, _ = vit_ref.vit_forward(vsd, vit_ref.VitCfg(), tb["pixel_values"])
    return fusion_ref.teacher_dual_forward(sd, ts_tokens, cls, 4, **kw)


def test_dual_teacher_forward_loss_and_gradients_match_the_reference():
    gold, sd, tb = _ctx()
    assert list(gold["keep_idx"]) == [1, 3, 5, 7, 8, 0, 4]
    with torch.no_grad():
        out = _forward(sd, tb, return_attn=True)
    assert {"fwd:" + k for k in out} == {k for k in gold if k.startswith("fwd:")}
    for k, v in out.items():
        np.testing.assert_allclose(v.numpy(), gold["fwd:" + k], rtol=5e-4, atol=5e-5, err_msg=k)
    train = {k: v for k, v in sd.items() if v.is_floating_point() and not k.startswith(("duett.", "cxr.", "pretrained_cxr_head."))}
    for v in train.values():
        v.requires_grad_(True)
    out = _forward(sd, tb)
    L = losses_ref.dual_pathology_loss(out["img_logits"], out["ts_logits"], out["fusion_logits"], tb["y_multi"], tb["y_multi_mask"],
                                       torch.ones(K))
    for k in ("total", "img_total", "ts_total", "fus_total"):
        np.testing.assert_allclose(float(L[k]), float(gold["loss:" + k]), rtol=2e-5)
    L["total"].backward()
    unused = sorted(k for k, v in train.items() if v.grad is None)
    assert unused == sorted(str(s) for s in gold["unused_parameters"]) == ["img_proj.bias", "img_proj.weight"]
    for key in gold:
        if key.startswith("grad:"):
            np.testing.assert_allclose(sd[key[5:]].grad.numpy(), gold[key], rtol=2e-3, atol=2e-6, err_msg=key)
        elif key.startswith("gsum:"):
            g = sd[key[5:]].grad.double()
            np.testing.assert_allclose([float(g.sum()), float(g.abs().sum())], gold[key], rtol=2e-3, atol=1e-5 + 1e-5 * float(gold[key][1]))
