#!/usr/bin/env python3
"""Golden fixture for SURVEY.md §8(f2): the `DualPathologyPerceiver` teacher that the reference's student entry point asks for
(`training_duett/trainer.py:770-822`), whose class exists in the reference only as COMMENTED-OUT source
(`models/main_architecture_duett.py:656-741`).  This script — build container only — reads those lines from the reference
file, strips the comment markers, executes the text inside the reference's own (imported) model module, and runs the
reference's live `TeacherModel(dual_pathology_mode=True)` branch (`:1047-1071, :1132-1150`) around it: the fixture holds
numbers the reference's own code produced, never its text.

Also written: `cxr_head_dual.pt`-style linear-probe checkpoint CONTENT is synthetic (label list + Linear(768, 9) weights from
tests/helpers.synth_tensor), so the GPU box can rebuild it without this script.

Usage:  python tests/golden/make_golden_dual.py
"""
from __future__ import annotations

import os
import re
import sys
import tempfile

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, HERE)

import make_golden as mg  # noqa: E402
from helpers import shapes_of, synth_state_dict, synth_tensor  # noqa: E402
from multimodal_edema_prediction_amd.cohort import PATHOLOGY_LABELS, CohortCfg, make_batch  # noqa: E402

PRETRAINED_LABELS = ["label_opacity", "label_edema", "label_fracture", "label_cardiomegaly", "label_consolidation",
                     "label_effusion", "label_pneumothorax", "label_pneumonia", "label_atelectasis"]


def uncommented_class(path: str, name: str) -> str:
    """The commented-out `class <name>` block of the reference file with its `# ` markers removed."""
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if re.match(rf"#\s*class {name}\(", l))
    out = []
    for l in lines[start:]:
        if l.startswith("####") or (l.strip() and not l.startswith("#")):
            break
        out.append(l[2:] if l.startswith("# ") else l[1:])
    return "\n".join(out)


def cxr_head_state():
    """What cxr_linear_training.ipynb :827-845 saves for the linear probe (plain data: loadable with weights_only=True)."""
    return {"num_classes": len(PRETRAINED_LABELS), "label_cols": list(PRETRAINED_LABELS),
            "classifier_state_dict": {"1.weight": synth_tensor("cxr_head.1.weight", (len(PRETRAINED_LABELS), 768), seed=6),
                                      "1.bias": synth_tensor("cxr_head.1.bias", (len(PRETRAINED_LABELS),), seed=6)}}


def main():
    torch.set_num_threads(8)
    mg.install_stubs()
    sys.path.insert(0, REF)
    import models.main_architecture_duett as ref
    from loss.losses_duett import DualPathologyLoss
    from training_duett import engine as ref_engine
    from transformers import Dinov2Config, Dinov2Model

    src = uncommented_class(os.path.join(REF, "models", "main_architecture_duett.py"), "DualPathologyPerceiver")
    exec(compile(src, "<reference main_architecture_duett.py:659-741, un-commented>", "exec"), ref.__dict__)
    DualPathologyPerceiver = ref.DualPathologyPerceiver

    B, T, V, DS, K = 8, 32, 16, 8, 7
    ccfg = CohortCfg(n_timesteps=T, n_vars=V, d_static=DS, image_size=224, seed=1234)
    vcfg = Dinov2Config(hidden_size=768, num_hidden_layers=12, num_attention_heads=12, mlp_ratio=4, patch_size=14,
                        image_size=518, layerscale_value=1.0, qkv_bias=True, use_swiglu_ffn=False)
    vit = Dinov2Model(vcfg)
    mg.load_synth(vit, seed=3)
    vit.eval()
    cxr = ref.CXREncoder.__new__(ref.CXREncoder)
    nn.Module.__init__(cxr)
    cxr.backbone, cxr.d_out, cxr.return_patches, cxr._frozen = vit, 768, False, True
    for p in cxr.backbone.parameters():
        p.requires_grad = False
    backbone = ref.DuettFeatureExtractor(d_static_num=DS, d_time_series_num=V, d_target=1, pretrain=False, masked_transform_timesteps=T,
                                         max_len=T, aug_noise=0.0, aug_mask=0.0, transformer_dropout=0.0)
    for p in backbone.parameters():
        p.requires_grad = False
    backbone.eval()
    perceiver = DualPathologyPerceiver(n_pathologies=K, d_ts=backbone.d_representation, d_latent=256, n_heads=4, dropout=0.0,
                                       head_dropout=0.0)
    with tempfile.TemporaryDirectory() as td:
        head_ckpt = os.path.join(td, "cxr_head.pt")
        torch.save(cxr_head_state(), head_ckpt)
        teacher = ref.TeacherModel(backbone, cxr, perceiver, head_hidden=128, head_dropout=0.0, cxr_return_patches=False, d_img=768,
                                   use_aux_cxr=False, dual_pathology_mode=True, pretrained_cxr_head_ckpt=head_ckpt,
                                   pathology_labels=tuple(PATHOLOGY_LABELS))
    sd_shapes = shapes_of(teacher.state_dict())
    sd_syn = synth_state_dict(sd_shapes, seed=5)
    for k, v in vit.state_dict().items():
        sd_syn["cxr.backbone." + k] = v.clone()
    hs = cxr_head_state()["classifier_state_dict"]
    sd_syn["pretrained_cxr_head.weight"], sd_syn["pretrained_cxr_head.bias"] = hs["1.weight"], hs["1.bias"]
    sd_syn["cxr_head_keep_idx"] = teacher.cxr_head_keep_idx.clone()
    teacher.load_state_dict(sd_syn, strict=True)

    tb = make_batch(ccfg, 100, B, mode="teacher")
    teacher.eval()
    with torch.no_grad():
        out_eval = teacher(tb["x_ts"], tb["x_static"], list(tb["bin_ends"]), tb["pixel_values"], return_attn=True)
    arrays = {"fwd:" + k: v for k, v in out_eval.items()}
    arrays["keep_idx"] = teacher.cxr_head_keep_idx

    loss_fn = DualPathologyLoss(torch.ones(K), None, 0.5, 0.5, 1.0)
    teacher.train()
    ref_engine._set_train_with_frozen_eval(teacher)
    out = teacher(tb["x_ts"], tb["x_static"], list(tb["bin_ends"]), tb["pixel_values"])
    losses = loss_fn(out["img_logits"], out["ts_logits"], out["fusion_logits"], tb["y_multi"], tb["y_multi_mask"])
    teacher.zero_grad()
    losses["total"].backward()
    for k in ("total", "img_total", "ts_total", "fus_total"):
        arrays["loss:" + k] = losses[k]
    named = dict(teacher.named_parameters())
    full = ["perceiver.temporal_queries", "perceiver.ts_proj.weight", "perceiver.temporal_heads.0.0.weight", "perceiver.temporal_heads.6.3.weight",
            "perceiver.residual_heads.3.0.weight", "perceiver.residual_heads.3.3.bias", "perceiver.ts_cross.attn.in_proj_weight",
            "perceiver.ts_self.ff.0.weight"]
    for k in full:
        arrays["grad:" + k] = named[k].grad
    unused = []
    for k, p in named.items():
        if not p.requires_grad:
            continue
        if p.grad is None:
            unused.append(k)
            continue
        g = p.grad.double()
        arrays["gsum:" + k] = np.array([float(g.sum()), float(g.abs().sum())])
    arrays["unused_parameters"] = np.array(unused)

    # one full reference engine step (train_teacher_dual_pathology_batch, engine.py:135-190) with plain AdamW
    teacher.load_state_dict(sd_syn, strict=True)
    opt = torch.optim.AdamW([p for p in teacher.parameters() if p.requires_grad], lr=8e-5, weight_decay=5e-2)
    step = ref_engine.train_teacher_dual_pathology_batch(tb, teacher, loss_fn, opt, torch.device("cpu"))
    arrays["step:loss"] = np.array(step["loss"])
    for k, p in teacher.named_parameters():
        if p.requires_grad and p.grad is not None:
            arrays["post:" + k] = np.array([float(p.detach().double().sum()), float(p.detach().double().abs().sum())])
    mg.save("teacher_dual_cfg1.npz", **arrays)

    import json
    shapes = json.load(open(os.path.join(HERE, "shapes.json")))
    shapes["teacher_dual"] = sd_shapes
    json.dump(shapes, open(os.path.join(HERE, "shapes.json"), "w"), indent=0)
    print("unused trainable parameters in the dual teacher:", unused)


if __name__ == "__main__":
    main()
