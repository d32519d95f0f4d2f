#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by running the REFERENCE'S OWN PYTHON
(/root/reference) on seeded synthetic inputs and weights.  Runs in the build container only
(the reference never travels to the GPU box; the fixtures — data only — do).

Three packages the reference imports are absent here and are stubbed in sys.modules:
  lightning.pytorch.LightningModule -> nn.Module + `.device`
  torchmetrics.AUROC / AveragePrecision -> inert
  x_transformers.Encoder -> oracle.xt_encoder.Encoder  (OUR restatement: that boundary is
      "parity unpinned"; everything around it is the reference's code)
RAD-DINO weights are remote: `CXREncoder` is built with `__new__` around a local
`transformers.Dinov2Model(Dinov2Config(...))`.

Usage:  python tests/golden/make_golden.py
"""
from __future__ import annotations

import json
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from helpers import shapes_of, synth_state_dict  # noqa: E402
from multimodal_edema_prediction_amd.cohort import CohortCfg, make_batch, make_item, collate  # noqa: E402
from oracle import xt_encoder  # noqa: E402


def install_stubs():
    pl = types.ModuleType("lightning.pytorch")

    class LightningModule(nn.Module):
        @property
        def device(self):
            return next(self.parameters()).device

        def log(self, *a, **k):
            pass

    pl.LightningModule = LightningModule
    lightning = types.ModuleType("lightning")
    lightning.pytorch = pl
    sys.modules["lightning"] = lightning
    sys.modules["lightning.pytorch"] = pl
    tm = types.ModuleType("torchmetrics")

    class _Metric:
        def __init__(self, *a, **k): pass
        def update(self, *a, **k): pass
        def compute(self): return float("nan")

    tm.AUROC = tm.AveragePrecision = _Metric
    sys.modules["torchmetrics"] = tm
    xt = types.ModuleType("x_transformers")
    xt.Encoder = xt_encoder.Encoder
    sys.modules["x_transformers"] = xt


def save(name, **arrays):
    out = {}
    for k, v in arrays.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    np.savez_compressed(os.path.join(HERE, name), **out)
    print(f"wrote {name}: {sum(a.nbytes for a in out.values()) / 1e6:.2f} MB raw")


def load_synth(module, seed):
    sd = module.state_dict()
    shapes = shapes_of(sd)
    module.load_state_dict(synth_state_dict(shapes, seed), strict=True)
    return shapes


def main():
    torch.set_num_threads(8)
    install_stubs()
    sys.path.insert(0, REF)
    from models.main_architecture_duett import (DuettFeatureExtractor, CXREncoder, PatchDualPathologyPerceiver,
                                                TeacherModel, StudentModel)
    from loss.losses_duett import DualPathologyLoss, StudentKDLoss, PathologyMultiLabelLoss
    from training_duett import engine as ref_engine
    from training_duett import evaluator as ref_eval
    from transformers import Dinov2Config, Dinov2Model

    B, T, V, DS, K = 8, 32, 16, 8, 7
    ccfg = CohortCfg(n_timesteps=T, n_vars=V, d_static=DS, image_size=224, seed=1234)
    shapes_all = {}

    # ---------------------------------------------------------------- DuETT backbone (cfg1 shapes)
    def new_backbone():
        m = DuettFeatureExtractor(d_static_num=DS, d_time_series_num=V, d_target=1, pretrain=False,
                                  masked_transform_timesteps=T, max_len=T, aug_noise=0.0, aug_mask=0.0,
                                  transformer_dropout=0.0)
        return m

    duett = new_backbone()
    shapes_all["duett"] = load_synth(duett, seed=1)
    duett.eval()

    # (1) feats_to_input incl. one over-length and one shorter series
    items = [make_item(ccfg, i, with_image=False) for i in range(B)]
    items[2] = make_item(ccfg, 2, with_image=False, n_steps=40)     # over-length → truncated to last 32
    items[5] = make_item(ccfg, 5, with_image=False, n_steps=20)     # shorter (and every other is 32 → padded)
    batch = collate(items, "student")
    x = (tuple(t.clone() for t in batch["x_ts"]), tuple(t.clone() for t in batch["x_static"]),
         [t.clone() for t in batch["bin_ends"]])
    xs_static, xs_ts, xs_times, n_timesteps = duett.feats_to_input(x, B)
    # (2)/(3) ψ + encode in eval mode.  ψ after embed is captured with a hook on the first event encoder.
    cap = {}
    h = duett.event_transformers[0].register_forward_pre_hook(lambda m, a: cap.__setitem__("emb0", a[0].detach().clone()))
    with torch.no_grad():
        enc_eval = duett.encode((xs_static, xs_ts.clone(), xs_times, n_timesteps))
    h.remove()
    # emb0 = ψ0.transpose(1,2).flatten(2) + E_event  → recover ψ0
    psi0_eval = (cap["emb0"] - duett.full_event_embedding.weight.unsqueeze(0)).view(B, V + 1, T + 1, 24).transpose(1, 2)
    # train-mode BN (batch statistics) — fresh copy so running stats of `duett` stay put
    duett_tr = new_backbone(); duett_tr.load_state_dict(duett.state_dict()); duett_tr.train()
    cap2 = {}
    h = duett_tr.event_transformers[0].register_forward_pre_hook(lambda m, a: cap2.__setitem__("emb0", a[0].detach().clone()))
    with torch.no_grad():
        enc_train = duett_tr.encode((xs_static, xs_ts.clone(), xs_times, n_timesteps))
    h.remove()
    psi0_train = (cap2["emb0"] - duett_tr.full_event_embedding.weight.unsqueeze(0)).view(B, V + 1, T + 1, 24).transpose(1, 2)
    sd_tr = duett_tr.state_dict()
    save("duett_cfg1.npz", xs_static=xs_static, xs_ts=xs_ts, xs_times=xs_times, n_timesteps=np.array(n_timesteps),
         psi0_eval=psi0_eval, enc_eval=enc_eval, psi0_train=psi0_train, enc_train=enc_train,
         bn_rm_after=sd_tr["embedding_layers.3.3.batch_norm.running_mean"],
         bn_rv_after=sd_tr["embedding_layers.3.3.batch_norm.running_var"],
         tbn_rm_after=sd_tr["full_time_embedding.2.batch_norm.running_mean"],
         tbn_rv_after=sd_tr["full_time_embedding.2.batch_norm.running_var"])

    # (4) student logits (+ grads through all of DuETT, train-mode BN, dropout 0)
    student = StudentModel(new_backbone(), pool="mean", head_hidden=128, head_dropout=0.0)
    shapes_all["student"] = load_synth(student, seed=2)
    student.eval()
    with torch.no_grad():
        z_eval = student(batch["x_ts"], batch["x_static"], list(batch["bin_ends"]))
        student.pool = "rep_token"
        z_rep = student(batch["x_ts"], batch["x_static"], list(batch["bin_ends"]))
        student.pool = "mean"
    save("student_cfg1.npz", z_eval=z_eval, z_rep=z_rep)

    # ---------------------------------------------------------------- Dinov2 (9): random weights, 224² and 512²
    vcfg = Dinov2Config(hidden_size=768, num_hidden_layers=12, num_attention_heads=12, mlp_ratio=4, patch_size=14,
                        image_size=518, layerscale_value=1.0, qkv_bias=True, use_swiglu_ffn=False)
    vit = Dinov2Model(vcfg)
    shapes_all["vit"] = load_synth(vit, seed=3)
    vit.eval()
    gpx = torch.Generator().manual_seed(99)
    px224 = torch.randn(2, 3, 224, 224, generator=gpx)
    px512 = torch.randn(1, 3, 512, 512, generator=gpx)
    with torch.no_grad():
        o224 = vit(pixel_values=px224).last_hidden_state
        o512 = vit(pixel_values=px512).last_hidden_state
    rows = [0, 1, 2, 17, 128, 256]
    rows5 = [0, 1, 2, 37, 640, 1296]
    save("vit_b14.npz", out224_rows=o224[:, rows], out224_sum=o224.sum(-1), out224_abs=o224.abs().sum(-1),
         out512_rows=o512[:, rows5], out512_sum=o512.sum(-1), out512_abs=o512.abs().sum(-1),
         rows224=np.array(rows), rows512=np.array(rows5))

    # ---------------------------------------------------------------- teacher (5)(6)(7): cfg1 TS shapes + ViT-B/14 @224
    cxr = CXREncoder.__new__(CXREncoder)
    nn.Module.__init__(cxr)
    cxr.backbone = vit
    cxr.d_out = 768
    cxr.return_patches = True
    cxr._frozen = True
    for p in cxr.backbone.parameters():
        p.requires_grad = False
    backbone_t = new_backbone()
    for p in backbone_t.parameters():
        p.requires_grad = False
    backbone_t.eval()
    perceiver = PatchDualPathologyPerceiver(n_pathologies=K, d_ts=backbone_t.d_representation, d_latent=256, n_heads=4,
                                            dropout=0.0, head_dropout=0.0)
    teacher = TeacherModel(backbone_t, cxr, perceiver, head_hidden=128, head_dropout=0.0, cxr_return_patches=True,
                           d_img=768, use_aux_cxr=False, patch_dual_pathology_mode=True)
    sd_shapes = shapes_of(teacher.state_dict())
    sd_syn = synth_state_dict(sd_shapes, seed=4)
    # the ViT inside keeps the seed-3 weights of fixture (9): keys 'cxr.backbone.*'
    for k, v in vit.state_dict().items():
        sd_syn["cxr.backbone." + k] = v.clone()
    teacher.load_state_dict(sd_syn, strict=True)
    shapes_all["teacher"] = sd_shapes
    tb = make_batch(ccfg, 100, B, mode="teacher")
    teacher.eval()
    with torch.no_grad():
        out_eval = teacher(tb["x_ts"], tb["x_static"], list(tb["bin_ends"]), tb["pixel_values"], return_attn=True)
    save("teacher_fwd_cfg1.npz", **{k: v for k, v in out_eval.items()})

    # (6) losses + selected grads (dropout 0 ⇒ train()==eval() arithmetic for the perceiver)
    loss_fn = DualPathologyLoss(torch.ones(K), None, 0.5, 0.5, 1.0)
    teacher.train(); ref_engine._set_train_with_frozen_eval(teacher)
    out = teacher(tb["x_ts"], tb["x_static"], list(tb["bin_ends"]), tb["pixel_values"])
    losses = loss_fn(out["img_logits"], out["ts_logits"], out["fusion_logits"], tb["y_multi"], tb["y_multi_mask"])
    teacher.zero_grad()
    losses["total"].backward()
    named = dict(teacher.named_parameters())
    grad_keys = ["perceiver.shared_queries", "perceiver.ts_proj.weight", "perceiver.correction_head.1.weight",
                 "perceiver.correction_head.4.weight", "perceiver.beta", "img_proj.weight", "img_proj.bias",
                 "perceiver.img_cross.attn.in_proj_weight", "perceiver.img_cross.attn.in_proj_bias",
                 "perceiver.ts_self.ff.3.weight", "perceiver.image_label_bias", "perceiver.img_cross.norm_kv.weight",
                 "perceiver.ts_cross.attn.out_proj.weight", "perceiver.temporal_head.0.weight"]
    g = {"grad:" + k: named[k].grad for k in grad_keys}
    gs = {"gsum:" + k: np.array([float(p.grad.double().sum()), float(p.grad.double().abs().sum())])
          for k, p in named.items() if p.grad is not None}
    # aux KL + LP regularisers (engine.py:149-165, 217-223) on the same forward
    eps = 0.05
    y = tb["y_multi"].float(); ys = y * (1 - eps) + (1 - y) * eps
    pc = torch.sigmoid(out["img_logits"].detach() + out["scaled_correction"]).clamp(min=1e-6, max=1 - 1e-6)
    kl = ys * (torch.log(ys) - torch.log(pc)) + (1 - ys) * (torch.log(1 - ys) - torch.log(1 - pc))
    m = tb["y_multi_mask"].float()
    aux_kl = (kl * m).sum() / m.sum().clamp(min=1.0)
    reg_beta = 1e-3 * (teacher.perceiver.beta ** 2).mean()
    reg_corr = 1e-2 * (out["scaled_correction"] ** 2).mean()
    pml = PathologyMultiLabelLoss(torch.ones(K), None, 1.0, 0.5)(out["img_logits"].detach(), out["fusion_logits"].detach(),
                                                                 tb["y_multi"], tb["y_multi_mask"])
    save("teacher_loss_cfg1.npz", total=losses["total"], img_total=losses["img_total"], ts_total=losses["ts_total"],
         fus_total=losses["fus_total"], img_per=losses["img_per"], ts_per=losses["ts_per"], fus_per=losses["fus_per"],
         aux_kl=aux_kl, reg_beta=reg_beta, reg_corr=reg_corr, pml_total=pml["total"], pml_s2=pml["stage2_per"],
         y_multi=tb["y_multi"], y_multi_mask=tb["y_multi_mask"], **g, **gs)

    # (7a) one full reference engine step (train_teacher_dual_pathology_batch) with plain AdamW
    teacher.load_state_dict(sd_syn, strict=True)
    opt = torch.optim.AdamW([p for p in teacher.parameters() if p.requires_grad], lr=8e-5, weight_decay=5e-2)
    step_out = ref_engine.train_teacher_dual_pathology_batch(tb, teacher, loss_fn, opt, torch.device("cpu"))
    post = {"post:" + k: np.array([float(p.detach().double().sum()), float(p.detach().double().abs().sum())])
            for k, p in teacher.named_parameters() if p.requires_grad}
    save("teacher_step_cfg1.npz", loss=np.array(step_out["loss"]), img_total=np.array(step_out["img_total"]),
         fus_logits=step_out["fusion_logits"], beta_after=teacher.perceiver.beta.detach(),
         queries_after=teacher.perceiver.shared_queries.detach(), **post)

    # (7b) one full reference student KD step (train_student_batch): BN in train mode, grads through all of DuETT
    teacher.load_state_dict(sd_syn, strict=True)
    student.load_state_dict(synth_state_dict(shapes_all["student"], seed=2), strict=True)
    kd = StudentKDLoss("vanilla_kl", 4.0, 0.5, None)
    sb = {k: v for k, v in tb.items() if k != "pixel_values"}
    with torch.no_grad():
        teacher.eval()
        z_t = teacher(tb["x_ts"], tb["x_static"], list(tb["bin_ends"]), tb["pixel_values"])["main_logit"]
    student.train()
    z_s = student(sb["x_ts"], sb["x_static"], list(sb["bin_ends"]))
    l = kd(z_s, z_t, sb["y"])
    student.zero_grad(); l["total"].backward()
    sn = dict(student.named_parameters())
    sgrad_keys = ["head.0.weight", "head.3.bias", "duett.embedding_layers.0.0.weight", "duett.embedding_layers.5.4.bias",
                  "duett.embedding_layers.3.3.batch_norm.weight", "duett.tab_encoder.0.weight",
                  "duett.special_embeddings.weight", "duett.n_obs_embedding.weight", "duett.full_event_embedding.weight",
                  "duett.full_rep_embedding.weight", "duett.full_time_embedding.0.weight",
                  "duett.full_time_embedding.3.bias", "duett.event_transformers.0.layers.0.1.to_q.weight",
                  "duett.event_transformers.1.layers.1.1.ff.2.weight", "duett.time_transformers.0.layers.0.0.0.g",
                  "duett.time_transformers.1.final_norm.g", "duett.time_transformers.0.layers.0.1.to_out.weight"]
    sg = {"grad:" + k: sn[k].grad for k in sgrad_keys}
    sgs = {"gsum:" + k: np.array([float(p.grad.double().sum()), float(p.grad.double().abs().sum())])
           for k, p in sn.items() if p.grad is not None}
    student.load_state_dict(synth_state_dict(shapes_all["student"], seed=2), strict=True)
    sopt = torch.optim.AdamW([p for p in student.parameters() if p.requires_grad], lr=8e-5, weight_decay=5e-2)
    sstep = ref_engine.train_student_batch(sb, tb, student, teacher, kd, sopt, torch.device("cpu"))
    spost = {"post:" + k: np.array([float(p.detach().double().sum()), float(p.detach().double().abs().sum())])
             for k, p in student.named_parameters() if p.requires_grad}
    save("student_step_cfg1.npz", z_t=z_t, z_s_train=z_s, total=l["total"], bce=l["bce"], kd=l["kd"],
         step_loss=np.array(sstep["loss"]), y=sb["y"], **sg, **sgs, **spost)

    # ---------------------------------------------------------------- (8) evaluator on a fixed logit table
    ge = torch.Generator().manual_seed(5)
    N = 512
    tab = {"img": torch.randn(N, K, generator=ge), "ts": torch.randn(N, K, generator=ge),
           "corr": 0.3 * torch.randn(N, K, generator=ge)}
    tab["fus"] = tab["img"] + tab["corr"]
    ym = (torch.rand(N, K, generator=ge) < 0.25).float()
    mk = (torch.rand(N, K, generator=ge) < 0.9).float()
    mk[:, 6] = 0.0; mk[:5, 6] = 1.0; ym[:5, 6] = torch.tensor([1., 0., 0., 1., 0.])   # nearly-all-masked label
    ym[:, 5] = 0.0                                 # a single-class label → AUROC NaN
    tab["img"][:50, 0] = tab["img"][50:100, 0]     # ties

    class _Fake(nn.Module):
        def __init__(s):
            super().__init__(); s.i = 0
            s.perceiver = types.SimpleNamespace(beta=torch.linspace(0.5, 1.5, K))
        def forward(s, x_ts, x_static, bin_ends, pixel_values):
            sl = slice(s.i, s.i + 64); s.i += 64
            return {"img_logits": tab["img"][sl], "ts_logits": tab["ts"][sl], "fusion_logits": tab["fus"][sl],
                    "scaled_correction": tab["corr"][sl], "main_logit": tab["fus"][sl, 0]}

    loader = []
    for s0 in range(0, N, 64):
        loader.append({"x_ts": (), "x_static": (), "bin_ends": (), "y": ym[s0:s0 + 64, 0],
                       "pixel_values": torch.zeros(1), "y_multi": ym[s0:s0 + 64], "y_multi_mask": mk[s0:s0 + 64]})
    labels = tuple(f"l{k}" for k in range(K))
    ev = ref_eval.evaluate_dual_pathology(_Fake(), loader, torch.device("cpu"), labels)
    keys = [k for k in ev["per_label"][0] if k != "name"]
    table = np.array([[float(r[k]) for k in keys] for r in ev["per_label"]], dtype=np.float64)
    fake = _Fake()
    evb = ref_eval.evaluate_binary(fake, loader, torch.device("cpu"),
                                   lambda m, b, d: {"logits": m(None, None, None, None)["main_logit"], "y": b["y"]})
    save("evaluator_table.npz", img=tab["img"], ts=tab["ts"], fus=tab["fus"], corr=tab["corr"], y=ym, mask=mk,
         per_label=table, main_auroc=np.array(ev["main_auroc"]), main_auprc=np.array(ev["main_auprc"]),
         bin_auroc=np.array(evb["auroc"]), bin_auprc=np.array(evb["auprc"]), bin_pos=np.array(evb["pos_frac"]))
    with open(os.path.join(HERE, "evaluator_keys.json"), "w") as f:
        json.dump(keys, f)

    with open(os.path.join(HERE, "shapes.json"), "w") as f:
        json.dump(shapes_all, f)
    meta = {"B": B, "T": T, "V": V, "DS": DS, "K": K, "cohort_seed": ccfg.seed, "teacher_batch_start": 100,
            "torch": torch.__version__, "xt_final_norm": xt_encoder.FINAL_NORM, "xt_eps": xt_encoder.SCALENORM_EPS,
            "note": "x_transformers.Encoder was the build's restatement (parity unpinned at that boundary)"}
    with open(os.path.join(HERE, "meta.json"), "w") as f:
        json.dump(meta, f, indent=1)
    print("done")


if __name__ == "__main__":
    main()
