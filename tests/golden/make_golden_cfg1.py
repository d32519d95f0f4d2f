#!/usr/bin/env python3
"""Config-1 fixtures (DuETT-only step, reference duett/duett.py `Model.training_step`): SSL pre-training step (masked
timestep + masked event read-outs, variable dropout from the model's numpy Generator) and the supervised fine-tune step,
by running the reference's own Python with the stubs of make_golden.py.  Writes duett_ssl_cfg1.npz."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import make_golden as mg  # noqa: E402
from helpers import shapes_of, synth_state_dict  # noqa: E402
from multimodal_edema_prediction_amd.cohort import CohortCfg, collate, make_item  # noqa: E402


def main():
    torch.set_num_threads(8)
    mg.install_stubs()
    sys.path.insert(0, mg.REF)
    from duett import duett as ref_duett
    B, T, V, DS = 8, 32, 16, 8
    ccfg = CohortCfg(n_timesteps=T, n_vars=V, d_static=DS, seed=1234)
    items = [make_item(ccfg, 200 + i, with_image=False) for i in range(B)]
    b = collate(items, "student")
    x = (tuple(t.clone() for t in b["x_ts"]), tuple(t.clone() for t in b["x_static"]), [t.clone() for t in b["bin_ends"]])
    y = [float(v) for v in b["y"]]
    # --- SSL pre-training step: pretrain_model(...) defaults except the cfg-1 sizes
    m = ref_duett.pretrain_model(DS, V, 1, masked_transform_timesteps=T, max_len=T, seed=42)
    shapes = shapes_of(m.state_dict())
    m.load_state_dict(synth_state_dict(shapes, seed=11), strict=True)
    m.train()
    logged = {}
    m.log = lambda k, v, **kw: logged.__setitem__(k, float(v))
    xp, yv, mask, yev, yevm = m.pretrain_prep_batch((tuple(t.clone() for t in x[0]), x[1], [t.clone() for t in x[2]]), B)
    m.rng = np.random.default_rng(42)          # replay the same draws inside training_step
    loss = m.training_step(((tuple(t.clone() for t in x[0]), x[1], [t.clone() for t in x[2]]), y), 0)
    m.zero_grad(); loss.backward()
    named = dict(m.named_parameters())
    gk = ["pretrain_value_proj.0.weight", "predict_events_proj.0.weight", "embedding_layers.2.0.weight", "special_embeddings.weight",
          "event_transformers.0.layers.1.1.ff.0.0.weight", "full_time_embedding.3.weight"]
    out = {"xs_ts_clipped": xp[1], "y_value": yv, "y_mask": mask, "y_events": yev, "y_events_mask": yevm, "ssl_loss": loss.detach()}
    out.update({"grad:" + k: named[k].grad for k in gk})
    # heads for the same prepared batch in eval-free train mode (BatchNorm batch statistics), second forward
    m2 = ref_duett.pretrain_model(DS, V, 1, masked_transform_timesteps=T, max_len=T, seed=42)
    m2.load_state_dict(synth_state_dict(shapes, seed=11), strict=True); m2.train()
    with torch.no_grad():
        hv, hp, he, hep = m2.forward((xp[0], xp[1].clone(), xp[2], xp[3]), pretrain=True)
    out.update({"hat_value": hv, "hat_presence": hp, "hat_events": he, "hat_events_presence": hep})
    # --- supervised step (fine_tune_model settings: pretrain False, fusion rep_token; aug off for parity)
    ms = ref_duett.Model(DS, V, 1, pretrain=False, fusion_method="rep_token", masked_transform_timesteps=T, max_len=T, aug_mask=0.0)
    ms.load_state_dict(synth_state_dict(shapes, seed=11), strict=True); ms.train()
    ms.log = lambda *a, **k: None
    ls = ms.training_step(((tuple(t.clone() for t in x[0]), x[1], [t.clone() for t in x[2]]), y), 0)
    ms.zero_grad(); ls.backward()
    out.update({"sup_loss": ls.detach(), "sup_grad:head.0.weight": dict(ms.named_parameters())["head.0.weight"].grad,
                "sup_grad:head.3.batch_norm.weight": dict(ms.named_parameters())["head.3.batch_norm.weight"].grad})
    np.savez_compressed(os.path.join(HERE, "duett_ssl_cfg1.npz"), **{k: (v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)) for k, v in out.items()})
    import json
    allsh = json.load(open(os.path.join(HERE, "shapes.json")))
    allsh["duett_model"] = shapes
    json.dump(allsh, open(os.path.join(HERE, "shapes.json"), "w"))
    print("wrote duett_ssl_cfg1.npz", float(loss), float(ls))


if __name__ == "__main__":
    main()
