"""Config 1 (DuETT-only SSL / supervised step): the CPU oracle against fixtures produced by the reference's own
`Model.training_step` (tests/golden/make_golden_cfg1.py)."""
import numpy as np
import torch

from helpers import load_npz, load_shapes, synth_state_dict, t
from multimodal_edema_prediction_amd.cohort import CohortCfg, collate, make_item
from oracle import duett_ref

B, T, V, DS = 8, 32, 16, 8


def batch():
    ccfg = CohortCfg(n_timesteps=T, n_vars=V, d_static=DS, seed=1234)
    b = collate([make_item(ccfg, 200 + i, with_image=False) for i in range(B)], "student")
    return (b["x_ts"], b["x_static"], list(b["bin_ends"])), b["y"]


def test_ssl_prep_forward_loss():
    gold = load_npz("duett_ssl_cfg1.npz")
    sd = synth_state_dict(load_shapes("shapes.json")["duett_model"], seed=11)
    x, y = batch()
    xp, yv, mask, yev, yevm = duett_ref.pretrain_prep_batch(x, np.random.default_rng(42), V, T)
    np.testing.assert_array_equal(xp[1].numpy(), gold["xs_ts_clipped"])
    np.testing.assert_array_equal(yv.numpy(), gold["y_value"])
    np.testing.assert_array_equal(yevm.numpy(), gold["y_events_mask"])
    cfg = duett_ref.DuettCfg(d_static_num=DS, d_time_series_num=V, n_timesteps=T)
    sd2 = {k: v.clone() for k, v in sd.items()}
    with torch.no_grad():
        hv, hp, he, hep = duett_ref.model_forward(sd2, cfg, xp, pretrain=True, fusion_method="masked_embed", training=True)
    for got, key in ((hv, "hat_value"), (hp, "hat_presence"), (he, "hat_events"), (hep, "hat_events_presence")):
        np.testing.assert_allclose(got.numpy(), gold[key], rtol=2e-4, atol=2e-5)
    loss = duett_ref.ssl_loss(hv, hp, he, hep, yv, mask, yev, yevm)
    np.testing.assert_allclose(float(loss), float(gold["ssl_loss"]), rtol=2e-5)


def test_supervised_step_loss():
    gold = load_npz("duett_ssl_cfg1.npz")
    sd = synth_state_dict(load_shapes("shapes.json")["duett_model"], seed=11)
    x, y = batch()
    cfg = duett_ref.DuettCfg(d_static_num=DS, d_time_series_num=V, n_timesteps=T)
    xin = duett_ref.feats_to_input(x, T)
    with torch.no_grad():
        z = duett_ref.model_forward(sd, cfg, xin, pretrain=False, fusion_method="rep_token", training=True)
    loss = torch.nn.functional.binary_cross_entropy_with_logits(z, y)
    np.testing.assert_allclose(float(loss), float(gold["sup_loss"]), rtol=2e-5)
