"""The epoch-level driver (multimodal_edema_prediction_amd/train_synthetic.py — the counterpart of the reference's train_teacher /
train_student, training_duett/trainer.py:216-764, 828-989) end to end on a tiny synthetic cohort: loader -> step -> gathered eval ->
best.pt -> early-stop flag -> reload best -> test; teacher (live `dual_patch` type, eager and captured-graph step), then the
student entry point's own chain: a `dual` teacher trained -> its best.pt -> student KD from that checkpoint."""
import math
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

TINY = ["--n_timesteps", "32", "--n_vars", "16", "--image_size", "112", "--n_train", "64", "--n_val", "48", "--n_test", "48",
        "--batch_size", "16", "--learnable_labels", "--lr", "5e-4", "--warmup_steps", "2"]


@pytest.mark.parametrize("graph", [False, True])
def test_teacher_driver_trains_saves_and_reloads(tmp_path, graph):
    from multimodal_edema_prediction_amd import checkpoint, train_synthetic
    d = str(tmp_path / "t")
    out = train_synthetic.main(["teacher", "--ckpt_dir", d, "--epochs", "2", "--freeze_duett"] + TINY + ([] if graph else ["--eager"]))
    assert len(out["history"]) == 2 and out["history"][0]["improved"]
    assert math.isfinite(out["history"][-1]["train_loss"]) and 0.0 <= out["best_val_auroc"] <= 1.0
    assert out["test"]["n"] == 48 and len(out["test"]["per_label"]) == 7
    st = checkpoint.load_ckpt(os.path.join(d, "best.pt"))
    assert st["args"]["perceiver_type"] == "dual_patch" and st["epoch"] in (1, 2)
    assert any(k.startswith("perceiver.shared_queries") for k in st["model"]) and "state" in st["optimizer"]


def test_early_stop_and_limit_batches(tmp_path):
    from multimodal_edema_prediction_amd import train_synthetic
    out = train_synthetic.main(["teacher", "--ckpt_dir", str(tmp_path / "e"), "--epochs", "6", "--freeze_duett", "--patience", "1",
                                "--limit_batches", "1", "--lr", "0.0"] + [a for a in TINY if a not in ("--lr", "5e-4")])
    # lr 0: the model never changes, so epoch 2 cannot improve on epoch 1 and patience 1 stops the run there
    assert len(out["history"]) == 2 and not out["history"][1]["improved"]


def test_dual_teacher_then_student_from_its_checkpoint(tmp_path):
    from multimodal_edema_prediction_amd import train_synthetic
    from tests_dual_common import cxr_head_state
    head = str(tmp_path / "cxr_head.pt")
    torch.save(cxr_head_state(), head)
    t = train_synthetic.main(["teacher", "--ckpt_dir", str(tmp_path / "dual"), "--epochs", "1", "--freeze_duett", "--perceiver_type", "dual",
                              "--pretrained_cxr_head_ckpt", head] + TINY)
    assert os.path.exists(t["ckpt"])
    s = train_synthetic.main(["student", "--teacher_ckpt", t["ckpt"], "--ckpt_dir", str(tmp_path / "stu"), "--epochs", "2"] + TINY)
    assert len(s["history"]) == 2 and math.isfinite(s["history"][-1]["train_loss"])
    assert s["test"]["n"] == 48
    if not math.isnan(s["history"][0]["val_auroc"]):          # 48 validation items at 8 % prevalence may hold a single class
        assert os.path.exists(s["ckpt"])
