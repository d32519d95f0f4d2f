"""Config 1 (DuETT-only SSL pre-training step and supervised step) on the HIP path vs the fixtures produced by the
reference's own `Model.training_step` (duett/duett.py:329-372)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from helpers import load_npz, load_shapes, synth_state_dict, t  # noqa: E402
from multimodal_edema_prediction_amd import duett as D  # noqa: E402
from multimodal_edema_prediction_amd.cohort import CohortCfg, collate, make_item  # noqa: E402

B, T, V, DS = 8, 32, 16, 8
DEV = "cuda"


def batch():
    ccfg = CohortCfg(n_timesteps=T, n_vars=V, d_static=DS, seed=1234)
    b = collate([make_item(ccfg, 200 + i, with_image=False) for i in range(B)], "student")
    x = (tuple(v.to(DEV) for v in b["x_ts"]), tuple(v.to(DEV) for v in b["x_static"]), [v.to(DEV) for v in b["bin_ends"]])
    return x, [float(v) for v in b["y"]]


def cosrel(g, w):
    g, w = g.float().cpu(), torch.as_tensor(w)
    return float((g * w).sum() / (g.norm() * w.norm() + 1e-30)), float((g - w).norm() / (w.norm() + 1e-30))


def test_ssl_step_matches_reference():
    gold = load_npz("duett_ssl_cfg1.npz")
    m = D.pretrain_model(DS, V, 1, masked_transform_timesteps=T, max_len=T, seed=42)
    assert sorted(m.state_dict()) == sorted(load_shapes("shapes.json")["duett_model"])
    m.load_state_dict(synth_state_dict(load_shapes("shapes.json")["duett_model"], seed=11), strict=True)
    m = m.to(DEV).train()
    x, y = batch()
    xp, yv, mask, yev, yevm = m.pretrain_prep_batch(x, B)
    assert torch.equal(xp[1].cpu(), t(gold["xs_ts_clipped"])) and torch.equal(yv.cpu(), t(gold["y_value"]))     # bit-exact masks
    hv, hp, he, hep = m.forward(xp, pretrain=True)
    for got, key in ((hv, "hat_value"), (hp, "hat_presence"), (he, "hat_events"), (hep, "hat_events_presence")):
        err = (got.detach().cpu() - t(gold[key])).abs()
        # regression read-outs of unit-scale tokens through a 408/792-wide Linear: bf16 operand rounding gives ~1e-2 typical error
        assert float(err.max()) < 0.15 and float(err.mean()) < 3e-2, (key, float(err.max()), float(err.mean()))
    m2 = D.pretrain_model(DS, V, 1, masked_transform_timesteps=T, max_len=T, seed=42)
    m2.load_state_dict(synth_state_dict(load_shapes("shapes.json")["duett_model"], seed=11), strict=True)
    m2 = m2.to(DEV).train()
    loss = m2.training_step((x, y))
    assert abs(float(loss) - float(gold["ssl_loss"])) <= 1e-2 * abs(float(gold["ssl_loss"])), (float(loss), float(gold["ssl_loss"]))
    loss.backward()
    named = dict(m2.named_parameters())
    for key in gold:
        if key.startswith("grad:"):
            cos, rel = cosrel(named[key[5:]].grad, gold[key])
            assert cos > 0.99 and rel < 0.15, (key, cos, rel)


def test_supervised_step_matches_reference():
    gold = load_npz("duett_ssl_cfg1.npz")
    m = D.Model(DS, V, 1, pretrain=False, fusion_method="rep_token", masked_transform_timesteps=T, max_len=T, aug_mask=0.0)
    m.load_state_dict(synth_state_dict(load_shapes("shapes.json")["duett_model"], seed=11), strict=True)
    m = m.to(DEV).train()
    x, y = batch()
    loss = m.training_step((x, y))
    assert abs(float(loss) - float(gold["sup_loss"])) <= 1e-2 * abs(float(gold["sup_loss"]))
    loss.backward()
    named = dict(m.named_parameters())
    for key in ("head.0.weight", "head.3.batch_norm.weight"):
        cos, rel = cosrel(named[key].grad, gold["sup_grad:" + key])
        assert cos > 0.99 and rel < 0.15, (key, cos, rel)
