"""best.pt round trip in the reference's layout (trainer.py:63-71): keys, values, optimizer state, safe loading."""
import argparse
import os

import pytest
import torch

from multimodal_edema_prediction_amd import checkpoint as C


def _toy():
    torch.manual_seed(0)
    m = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.GELU(), torch.nn.Linear(5, 3))
    opt = torch.optim.AdamW(m.parameters(), lr=1e-3, weight_decay=0.1)
    m(torch.randn(4, 6)).sum().backward()
    opt.step()
    return m, opt


def test_round_trip_matches_the_reference_layout(tmp_path):
    m, opt = _toy()
    args = argparse.Namespace(perceiver_type="dual_patch", d_latent=256, pathology_labels="a,b,c", lr=3e-4)
    path = os.path.join(tmp_path, "ckpt", "best.pt")
    C.save_ckpt(path, m, opt, epoch=7, metric=0.8125, args=args)
    st = C.load_ckpt(path)
    assert sorted(st) == ["args", "epoch", "metric", "model", "optimizer"]
    assert st["epoch"] == 7 and st["metric"] == 0.8125 and st["args"] == vars(args)
    m2, opt2 = _toy()
    for p in m2.parameters():
        p.data.zero_()
    C.load_model_state(m2, st, freeze=True)
    for (k, a), (_, b) in zip(m.state_dict().items(), m2.state_dict().items()):
        assert torch.equal(a, b), k
    assert not any(p.requires_grad for p in m2.parameters()) and not m2.training
    opt2.load_state_dict(st["optimizer"])
    s1, s2 = opt.state_dict()["state"], opt2.state_dict()["state"]
    assert all(torch.equal(s1[i]["exp_avg"], s2[i]["exp_avg"]) for i in s1)


def test_refuses_files_that_are_not_trainer_checkpoints(tmp_path):
    path = os.path.join(tmp_path, "x.pt")
    torch.save({"model": {}}, path)
    with pytest.raises(KeyError):
        C.load_ckpt(path)


def test_loader_executes_nothing_from_the_file(tmp_path):
    class Evil:
        def __reduce__(self):
            return (os.system, ("true",))
    path = os.path.join(tmp_path, "evil.pt")
    torch.save({"model": {}, "optimizer": {}, "epoch": 0, "metric": 0.0, "args": {"x": Evil()}}, path)
    with pytest.raises(Exception):
        C.load_ckpt(path)
