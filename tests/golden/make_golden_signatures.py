#!/usr/bin/env python3
"""Pin the module-protocol boundary as DATA: run the reference's own Python (this container only) and write
tests/golden/signatures.json with

  * `inspect.signature` of the constructor and `forward` (and the other methods callers use) of every class / function the
    reference's trainer imports from models/main_architecture_duett.py, loss/losses_duett.py and duett/duett.py
    (/root/reference/training_duett/trainer.py:27-45);
  * the `named_parameters()` / `named_buffers()` name lists of cfg1-sized instances (the name-pattern contract of
    trainer.py:88-102, engine.py:14-20, evaluator.py:249-253);
  * the key / dtype / shape table of `duett_kd_collate` (training_duett/data_processing.py:394-411) for both modes;
  * the keys of the dicts the step functions and the model forwards return.

tests/test_signatures_cpu.py compares the product classes against it (keyword-only `_private` extras allowed).
Encoder-internal parameter names (`event_transformers.*`, `time_transformers.*`) belong to x_transformers (stubbed by
oracle/xt_encoder.py, parity unpinned): the test compares only their shapes as a multiset.

Usage:  python tests/golden/make_golden_signatures.py
"""
from __future__ import annotations

import inspect
import json
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from make_golden import install_stubs  # noqa: E402
from multimodal_edema_prediction_amd.cohort import CohortCfg, make_item  # noqa: E402


def sig(fn) -> list:
    """[[name, kind, default-repr or None], ...] without `self`."""
    out = []
    for name, p in inspect.signature(fn).parameters.items():
        if name == "self":
            continue
        out.append([name, p.kind.name, None if p.default is inspect.Parameter.empty else repr(p.default)])
    return out


def table(d: dict) -> dict:
    out = {}
    for k, v in d.items():
        if isinstance(v, torch.Tensor):
            out[k] = {"type": "tensor", "dtype": str(v.dtype), "shape": list(v.shape)}
        elif isinstance(v, (tuple, list)):
            e = v[0]
            out[k] = {"type": type(v).__name__, "len": len(v), "elem_dtype": str(e.dtype), "elem_shape": list(e.shape)}
        else:
            out[k] = {"type": type(v).__name__}
    return out


def main():
    install_stubs()
    sys.path.insert(0, REF)
    import models.main_architecture_duett as M
    import loss.losses_duett as Lo
    from duett import duett as D
    from training_duett import data_processing as DP
    from transformers import Dinov2Config, Dinov2Model

    T, V, DS, K = 32, 16, 8, 7
    doc = {"classes": {}, "functions": {}, "parameters": {}, "buffers": {}, "collate": {}, "outputs": {}}

    methods = {
        "DuettFeatureExtractor": ["__init__", "encode", "feats_to_input"],
        "CXREncoder": ["__init__", "forward", "train"],
        "PatchDualPathologyPerceiver": ["__init__", "forward"],
        "_PerceiverBlock": ["__init__", "forward"],
        "TeacherModel": ["__init__", "forward"],
        "StudentModel": ["__init__", "forward"],
        "LocalTrajectoryEncoder": ["__init__", "forward"],
    }
    for cname, ms in methods.items():
        cls = getattr(M, cname)
        doc["classes"][cname] = {m: sig(getattr(cls, m)) for m in ms}
    doc["classes"]["Model"] = {m: sig(getattr(D.Model, m)) for m in ["__init__", "forward", "feats_to_input"]}
    for cname in ["StudentKDLoss", "PathologyMultiLabelLoss", "DualPathologyLoss", "VanillaKLKD"]:
        cls = getattr(Lo, cname)
        doc["classes"][cname] = {m: sig(getattr(cls, m)) for m in ["__init__", "forward"]}
    doc["functions"]["load_duett_backbone"] = sig(M.load_duett_backbone)
    doc["functions"]["duett_kd_collate"] = sig(DP.duett_kd_collate)

    # ---- name lists of cfg1-sized instances
    duett = M.DuettFeatureExtractor(d_static_num=DS, d_time_series_num=V, d_target=1, pretrain=False,
                                    masked_transform_timesteps=T, max_len=T, aug_noise=0.0, aug_mask=0.0, transformer_dropout=0.0)
    cxr = M.CXREncoder.__new__(M.CXREncoder)
    torch.nn.Module.__init__(cxr)
    cxr.backbone = Dinov2Model(Dinov2Config(hidden_size=768, num_hidden_layers=12, num_attention_heads=12, mlp_ratio=4, patch_size=14,
                                            image_size=518, layerscale_value=1.0, qkv_bias=True, use_swiglu_ffn=False))
    cxr.d_out, cxr.return_patches, cxr._frozen = 768, True, True
    for p in cxr.backbone.parameters():
        p.requires_grad_(False)
    per = M.PatchDualPathologyPerceiver(K, duett.d_representation)
    teacher = M.TeacherModel(duett, cxr, per, cxr_return_patches=True, d_img=768, use_aux_cxr=False, patch_dual_pathology_mode=True)
    student = M.StudentModel(M.DuettFeatureExtractor(d_static_num=DS, d_time_series_num=V, d_target=1, pretrain=False,
                                                     masked_transform_timesteps=T, max_len=T, aug_noise=0.0, aug_mask=0.0,
                                                     transformer_dropout=0.0))
    traj = M.LocalTrajectoryEncoder(n_vars=V, n_timesteps=24)

    def names(mod, what):
        it = mod.named_parameters() if what == "parameters" else mod.named_buffers()
        return [[n, list(t.shape), bool(getattr(t, "requires_grad", False))] for n, t in it]

    for label, mod in (("TeacherModel", teacher), ("StudentModel", student), ("PatchDualPathologyPerceiver", per),
                       ("DuettFeatureExtractor", duett), ("LocalTrajectoryEncoder", traj)):
        for what in ("parameters", "buffers"):
            rows = names(mod, what)
            if label == "TeacherModel":      # the Dinov2 backbone's own names are transformers', pinned by tests/golden/vit_b14.npz
                rows = [r for r in rows if not r[0].startswith("cxr.backbone.")]
            doc[what][label] = rows
    doc["attrs"] = {
        "TeacherModel": [a for a in ("duett", "cxr", "perceiver", "img_proj", "pretrained_cxr_head") if hasattr(teacher, a)],
        "PatchDualPathologyPerceiver": [a for a in ("correction_head", "beta", "d_latent", "shared_queries") if hasattr(per, a)],
        "DuettFeatureExtractor": [a for a in ("d_representation", "d_embedding", "feats_to_input", "encode") if hasattr(duett, a)],
        "CXREncoder": [a for a in ("d_out", "return_patches", "backbone") if hasattr(cxr, a)],
    }

    # ---- collate table
    ccfg = CohortCfg(n_timesteps=T, n_vars=V, d_static=DS, image_size=224, seed=1234)
    for mode in ("teacher", "student"):
        items = [make_item(ccfg, i, with_image=(mode == "teacher")) for i in range(3)]
        doc["collate"][mode] = table(DP.duett_kd_collate(items, mode))

    # ---- output dict keys
    teacher.eval()
    items = [make_item(ccfg, i, with_image=True) for i in range(2)]
    b = DP.duett_kd_collate(items, "teacher")
    with torch.no_grad():
        out = teacher(b["x_ts"], b["x_static"], list(b["bin_ends"]), b["pixel_values"])
        out_attn = teacher(b["x_ts"], b["x_static"], list(b["bin_ends"]), b["pixel_values"], return_attn=True)
        z = student.eval()(b["x_ts"], b["x_static"], list(b["bin_ends"]))
    doc["outputs"]["TeacherModel.forward"] = table(out)
    doc["outputs"]["TeacherModel.forward(return_attn=True)"] = table(out_attn)
    doc["outputs"]["StudentModel.forward"] = {"type": "tensor", "dtype": str(z.dtype), "shape": list(z.shape)}
    dl = Lo.DualPathologyLoss(torch.ones(K))
    L = dl(out["img_logits"], out["ts_logits"], out["fusion_logits"], b["y_multi"], b["y_multi_mask"])
    doc["outputs"]["DualPathologyLoss.forward"] = sorted(L.keys())
    kd = Lo.StudentKDLoss()
    doc["outputs"]["StudentKDLoss.forward"] = sorted(kd(z, out["main_logit"], b["y"]).keys())

    path = os.path.join(HERE, "signatures.json")
    with open(path, "w") as f:
        json.dump(doc, f, indent=1, sort_keys=True)
    print(f"wrote {path}: {os.path.getsize(path) / 1e3:.1f} kB")


if __name__ == "__main__":
    main()
