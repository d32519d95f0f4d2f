"""End-to-end parity of the product modules (HIP path, through the C ABI) against the golden fixtures produced by the
reference's own Python on CPU: DuETT encode, teacher forward dict, DualPathologyLoss + gradients, one full
engine step with AdamW.  bf16 GEMM operands / fp32 accumulate, fp32 norms/softmax/losses.
Tolerances (SURVEY.md §8d, bf16 mode): logits <= 3e-2 abs, loss <= 1e-2 rel."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from helpers import GOLDEN_DIR, load_npz, load_shapes, synth_state_dict, t  # noqa: E402
from multimodal_edema_prediction_amd import engine  # noqa: E402
from multimodal_edema_prediction_amd.cohort import CohortCfg, collate, make_batch, make_item  # noqa: E402
from multimodal_edema_prediction_amd.losses_duett import DualPathologyLoss  # noqa: E402
from multimodal_edema_prediction_amd.main_architecture_duett import (CXREncoder, DuettFeatureExtractor,  # noqa: E402
                                                                       PatchDualPathologyPerceiver, StudentModel, TeacherModel)
from multimodal_edema_prediction_amd.optim import FusedAdamW  # noqa: E402

META = json.load(open(os.path.join(GOLDEN_DIR, "meta.json")))
SHAPES = load_shapes("shapes.json")
B, T, V, DS, K = META["B"], META["T"], META["V"], META["DS"], META["K"]
CCFG = CohortCfg(n_timesteps=T, n_vars=V, d_static=DS, image_size=224, seed=META["cohort_seed"])
DEV = "cuda"


def new_backbone():
    return DuettFeatureExtractor(d_static_num=DS, d_time_series_num=V, d_target=1, pretrain=False, masked_transform_timesteps=T,
                                 max_len=T, aug_noise=0.0, aug_mask=0.0, transformer_dropout=0.0)


def maxerr(a, b):
    return float((a.detach().float().cpu() - torch.as_tensor(b)).abs().max())


def test_duett_state_dict_keys_match_reference():
    assert sorted(new_backbone().state_dict()) == sorted(SHAPES["duett"])


def test_duett_encode_against_golden():
    gold = load_npz("duett_cfg1.npz")
    m = new_backbone()
    m.load_state_dict(synth_state_dict(SHAPES["duett"], seed=1), strict=True)
    m = m.to(DEV).eval()
    items = [make_item(CCFG, i, with_image=False) for i in range(B)]
    items[2] = make_item(CCFG, 2, with_image=False, n_steps=40)      # over-length
    items[5] = make_item(CCFG, 5, with_image=False, n_steps=20)      # shorter -> padded
    b = collate(items, "student")
    xin = m.feats_to_input((tuple(x.to(DEV) for x in b["x_ts"]), tuple(x.to(DEV) for x in b["x_static"]),
                            [x.to(DEV) for x in b["bin_ends"]]), B)
    assert torch.equal(xin[1].cpu(), t(gold["xs_ts"])) and torch.equal(xin[2].cpu(), t(gold["xs_times"]))
    assert xin[3] == list(gold["n_timesteps"])
    with torch.no_grad():
        tok, _, psi0 = m._encode_inference(xin, want_psi0=True)
    assert maxerr(psi0, gold["psi0_eval"]) < 2e-5                     # fp32 embedding MLPs: same arithmetic, different sum order
    err = (tok.cpu() - t(gold["enc_eval"])).abs()
    assert float(err.max()) < 3e-2 and float(err.mean()) < 3e-3, (float(err.max()), float(err.mean()))
    with torch.no_grad():
        assert torch.equal(m.encode(xin), tok)


def build_teacher():
    cxr = CXREncoder("synthetic", freeze=True, return_patches=True)
    backbone = new_backbone()
    for p in backbone.parameters():
        p.requires_grad = False
    backbone.eval()
    perceiver = PatchDualPathologyPerceiver(n_pathologies=K, d_ts=backbone.d_representation, d_latent=256, n_heads=4, dropout=0.0,
                                            head_dropout=0.0)
    teacher = TeacherModel(backbone, cxr, perceiver, head_hidden=128, head_dropout=0.0, cxr_return_patches=True, d_img=768,
                           use_aux_cxr=False, patch_dual_pathology_mode=True)
    sd = synth_state_dict(SHAPES["teacher"], seed=4)
    for k, v in synth_state_dict(SHAPES["vit"], seed=3).items():
        sd["cxr.backbone." + k] = v
    assert sorted(teacher.state_dict()) == sorted(SHAPES["teacher"])
    teacher.load_state_dict(sd, strict=True)
    return teacher.to(DEV)


@pytest.fixture(scope="module")
def teacher():
    return build_teacher()


@pytest.fixture(scope="module")
def tbatch():
    return make_batch(CCFG, META["teacher_batch_start"], B, mode="teacher")


def fwd(teacher, tb, **kw):
    b = engine._move_lists(tb, DEV)
    return teacher(b["x_ts"], b["x_static"], b["bin_ends"], b["pixel_values"], **kw), b


def test_teacher_forward_dict(teacher, tbatch):
    gold = load_npz("teacher_fwd_cfg1.npz")
    teacher.eval()
    with torch.no_grad():
        out, _ = fwd(teacher, tbatch, return_attn=True)
    assert set(out) == set(gold)
    for k in ("main_logit", "img_logits", "ts_logits", "fusion_logits", "ts_correction", "scaled_correction"):
        assert maxerr(out[k], gold[k]) < 3e-2, (k, maxerr(out[k], gold[k]))
    for k in ("img_tokens", "ts_tokens", "fusion_tokens"):
        assert maxerr(out[k], gold[k]) < 6e-2, (k, maxerr(out[k], gold[k]))
    for k in ("img_attn", "ts_attn"):
        assert out[k].shape == gold[k].shape and maxerr(out[k], gold[k]) < 5e-3, (k, maxerr(out[k], gold[k]))


def test_teacher_loss_and_grads(teacher, tbatch):
    gold = load_npz("teacher_loss_cfg1.npz")
    loss_fn = DualPathologyLoss(torch.ones(K), None, 0.5, 0.5, 1.0).to(DEV)
    engine._set_train_with_frozen_eval(teacher)
    out, b = fwd(teacher, tbatch)
    losses = loss_fn(out["img_logits"], out["ts_logits"], out["fusion_logits"], b["y_multi"], b["y_multi_mask"])
    for k in ("total", "img_total", "ts_total", "fus_total"):
        assert abs(float(losses[k]) - float(gold[k])) <= 1e-2 * abs(float(gold[k])), (k, float(losses[k]), float(gold[k]))
    for k in ("img_per", "ts_per", "fus_per"):
        np.testing.assert_allclose(losses[k].cpu().numpy(), gold[k], rtol=2e-2, atol=1e-3)
    teacher.zero_grad()
    losses["total"].backward()
    named = dict(teacher.named_parameters())
    checked = 0
    for key in gold:
        if not key.startswith("grad:"):
            continue
        g, want = named[key[5:]].grad.float().cpu(), torch.as_tensor(gold[key])
        # bf16 GEMM operands in forward and backward: compare direction and scale of every gradient tensor
        cos = float((g * want).sum() / (g.norm() * want.norm() + 1e-30))
        rel = float((g - want).norm() / (want.norm() + 1e-30))
        assert cos > 0.995 and rel < 0.1, (key, cos, rel)
        checked += 1
    assert checked >= 12
    # every trainable parameter received a gradient of the right magnitude
    for key in gold:
        if key.startswith("gsum:"):
            g = named[key[5:]].grad
            assert g is not None, key
            np.testing.assert_allclose(float(g.double().abs().sum()), gold[key][1], rtol=0.1, atol=1e-6)
    # frozen parts got none
    assert all(p.grad is None for n, p in named.items() if n.startswith(("duett.", "cxr.")))


def test_teacher_engine_step_with_fused_adamw(tbatch):
    gold = load_npz("teacher_step_cfg1.npz")
    teacher = build_teacher()
    loss_fn = DualPathologyLoss(torch.ones(K), None, 0.5, 0.5, 1.0).to(DEV)
    opt = FusedAdamW([p for p in teacher.parameters() if p.requires_grad], lr=8e-5, weight_decay=5e-2)
    before = {k: p.detach().clone() for k, p in teacher.named_parameters() if p.requires_grad}
    out = engine.train_teacher_dual_pathology_batch(tbatch, teacher, loss_fn, opt, torch.device(DEV))
    assert abs(out["loss"] - float(gold["loss"])) <= 1e-2 * abs(float(gold["loss"]))
    assert maxerr(out["fusion_logits"], gold["fus_logits"]) < 3e-2
    # Adam's first step moves every element by ~lr*sign(g) (+ decoupled decay): check against the reference's post-step checksums
    n = 0
    for k, p in teacher.named_parameters():
        if not p.requires_grad:
            continue
        step = (p.detach() - before[k] * (1 - 8e-5 * 5e-2)).abs()
        assert float(step.max()) <= 8e-5 + 5e-7, k      # fp32 rounding of the subtraction on O(1) values
        s = float(p.detach().double().abs().sum())
        np.testing.assert_allclose(s, gold["post:" + k][1], rtol=2e-4, atol=2e-3)
        n += 1
    assert n > 50
    assert maxerr(teacher.perceiver.beta, gold["beta_after"]) < 2 * 8e-5 + 1e-6


def test_fused_adamw_matches_torch_adamw():
    torch.manual_seed(0)
    ps = [torch.randn(s, device=DEV) for s in [(300, 257), (7,), (64, 64), (1,), (4099,)]]
    gs = [[torch.randn_like(p) for p in ps] for _ in range(3)]
    a = [torch.nn.Parameter(p.clone()) for p in ps]
    b = [torch.nn.Parameter(p.clone()) for p in ps]
    oa = torch.optim.AdamW([{"params": a[:2], "lr": 1e-3}, {"params": a[2:], "lr": 3e-4}], weight_decay=5e-2)
    ob = FusedAdamW([{"params": b[:2], "lr": 1e-3}, {"params": b[2:], "lr": 3e-4}], weight_decay=5e-2)
    for step in range(3):
        for x, y, g in zip(a, b, gs[step]):
            x.grad, y.grad = g.clone(), g.clone()
        oa.step(); ob.step()
    for x, y in zip(a, b):
        assert float((x - y).abs().max()) < 2e-6


def test_engine_extras_aux_kl_and_lp_regularisers(teacher, tbatch):
    """engine.py:149-165, 217-223 against the reference values (golden) and their analytic gradients (torch on CPU)."""
    from multimodal_edema_prediction_amd import autograd_ops as A
    gold = load_npz("teacher_loss_cfg1.npz")
    teacher.eval()
    out, b = fwd(teacher, tbatch)
    aux = A.aux_residual_kl(out["img_logits"], out["scaled_correction"], b["y_multi"], b["y_multi_mask"])
    assert abs(float(aux) - float(gold["aux_kl"])) <= 1e-2 * abs(float(gold["aux_kl"]))
    rb = A.sq_mean(teacher.perceiver.beta, 1e-3)
    rc = A.sq_mean(out["scaled_correction"], 1e-2)
    assert abs(float(rb) - float(gold["reg_beta"])) <= 1e-5 * abs(float(gold["reg_beta"])) + 1e-9
    assert abs(float(rc) - float(gold["reg_corr"])) <= 5e-2 * abs(float(gold["reg_corr"])) + 1e-9
    # gradient of the aux KL w.r.t. scaled_correction vs torch autograd on the same (GPU-produced) inputs
    sc = out["scaled_correction"].detach().cpu().double().requires_grad_(True)
    img = out["img_logits"].detach().cpu().double()
    y, m = tbatch["y_multi"].double(), tbatch["y_multi_mask"].double()
    ys = y * 0.95 + (1 - y) * 0.05
    p = torch.sigmoid(img + sc).clamp(1e-6, 1 - 1e-6)
    ref = ((ys * (ys.log() - p.log()) + (1 - ys) * ((1 - ys).log() - (1 - p).log())) * m).sum() / m.sum().clamp(min=1.0)
    ref.backward()
    x = out["scaled_correction"].detach().clone().requires_grad_(True)
    A.aux_residual_kl(out["img_logits"], x, b["y_multi"], b["y_multi_mask"]).backward()
    assert maxerr(x.grad, sc.grad.float()) < 1e-6
    tot = A.add_scaled(aux, rb, 0.5)
    assert abs(float(tot) - (float(aux) + 0.5 * float(rb))) < 1e-6


def test_lp_step_only_moves_correction_head_and_beta(tbatch):
    teacher = build_teacher()
    for p in teacher.parameters():
        p.requires_grad = False
    for p in teacher.perceiver.correction_head.parameters():
        p.requires_grad = True
    teacher.perceiver.beta.requires_grad = True
    before = {k: v.detach().clone() for k, v in teacher.named_parameters()}
    loss_fn = DualPathologyLoss(torch.ones(K), None, 0.5, 0.5, 1.0).to(DEV)
    opt = FusedAdamW([p for p in teacher.parameters() if p.requires_grad], lr=8e-5, weight_decay=5e-2)
    out = engine.train_teacher_dual_pathology_lp_batch(tbatch, teacher, loss_fn, opt, torch.device(DEV), beta_l2=1e-3, corr_l2=1e-2,
                                                        aux_residual_alpha=0.1)
    assert out["reg_beta_l2"] > 0 and out["reg_corr_l2"] > 0 and out["aux_residual"] > 0
    moved = {k for k, v in teacher.named_parameters() if not torch.equal(v.detach(), before[k])}
    assert moved and all(("correction_head" in k or k.endswith(".beta")) for k in moved), moved


def test_linear_probe_cfg2():
    """RadDinoClassifier mirror: logits = Linear(768,7)(CLS) and the global masked BCE, vs the CPU oracle."""
    from multimodal_edema_prediction_amd.linear_probe import RadDinoClassifier, masked_bce_with_logits_loss
    from oracle import losses_ref, vit_ref
    torch.manual_seed(0)
    m = RadDinoClassifier("synthetic", num_classes=7, dropout=0.0)
    m.encoder.backbone.load_state_dict(synth_state_dict(SHAPES["vit"], seed=3), strict=True)
    m = m.to(DEV).train()
    g = torch.Generator().manual_seed(5)
    px = torch.randn(2, 3, 224, 224, generator=g)
    y, mk = (torch.rand(2, 7, generator=g) < 0.3).float(), (torch.rand(2, 7, generator=g) < 0.8).float()
    logits = m(px.to(DEV))
    loss = masked_bce_with_logits_loss(logits, y.to(DEV), mk.to(DEV))
    loss.backward()
    with torch.no_grad():
        cls, _ = vit_ref.vit_forward(synth_state_dict(SHAPES["vit"], seed=3), vit_ref.VitCfg(), px)
    W = m.classifier[1].weight.detach().cpu().clone().requires_grad_(True)
    bb = m.classifier[1].bias.detach().cpu().clone().requires_grad_(True)
    ref_logits = torch.nn.functional.linear(cls, W, bb)
    ref = losses_ref.masked_bce_global(ref_logits, y, mk)
    ref.backward()
    assert maxerr(logits, ref_logits.detach()) < 3e-2
    assert abs(float(loss) - float(ref)) <= 1e-2 * abs(float(ref))
    gw = m.classifier[1].weight.grad.cpu()
    assert float((gw - W.grad).norm() / W.grad.norm()) < 0.05
    assert maxerr(m.classifier[1].bias.grad, bb.grad) < 1e-3


def test_graphed_step_equals_eager_step(tbatch):
    """The captured HIP-graph step (graph_step.py) must do exactly the eager engine step's arithmetic (dropout off)."""
    from multimodal_edema_prediction_amd.graph_step import GraphedTeacherStep
    from multimodal_edema_prediction_amd.optim import make_param_groups
    loss_fn = DualPathologyLoss(torch.ones(K), None, 0.5, 0.5, 1.0).to(DEV)
    te = build_teacher()
    oe = FusedAdamW(make_param_groups(te, 8e-5), weight_decay=5e-2)
    eager_losses = [engine.train_teacher_dual_pathology_batch(tbatch, te, loss_fn, oe, torch.device(DEV))["loss"] for _ in range(3)]
    tg = build_teacher()
    og = FusedAdamW(make_param_groups(tg, 8e-5), weight_decay=5e-2)
    gs = GraphedTeacherStep(tg, loss_fn, og, tbatch, torch.device(DEV), warmup=3)       # the 3 warm-up steps are undone before the capture
    graph_losses = [float(gs.step(tbatch)["loss"].item()) for _ in range(3)]
    # eager steps 1..3 vs the three replays: the same initial state, no warm-up offset
    np.testing.assert_allclose(graph_losses, eager_losses, rtol=1e-5, atol=1e-6)
    for (k, a), (_, b) in zip(te.named_parameters(), tg.named_parameters()):
        if a.requires_grad:
            assert float((a - b).abs().max()) <= 1e-6, k
    assert og._step == oe._step == 3


def test_pathology_multilabel_loss_value_and_gradient():
    """a14 (`PathologyMultiLabelLoss`, reference loss/losses_duett.py:63-125) on the HIP path: totals and per-label terms against
    the fixture the reference's own class produced (`pml_total`, `pml_s2`, same logits: teacher_fwd_cfg1.npz), gradients of
    both logit sets element-wise against the oracle's autograd (fp32 in, fp32 math: 1e-6), with and without pos_weight."""
    from multimodal_edema_prediction_amd.losses_duett import PathologyMultiLabelLoss
    from oracle import losses_ref
    gold_f, gold_l = load_npz("teacher_fwd_cfg1.npz"), load_npz("teacher_loss_cfg1.npz")
    y, mk = t(gold_l["y_multi"]), t(gold_l["y_multi_mask"])
    s2 = t(gold_f["img_logits"]).to(DEV).requires_grad_(True)
    s4 = t(gold_f["fusion_logits"]).to(DEV).requires_grad_(True)
    out = PathologyMultiLabelLoss(torch.ones(K), None, 1.0, 0.5).to(DEV)(s2, s4, y.to(DEV), mk.to(DEV))
    assert set(out) == {"total", "stage2_total", "stage4_total", "stage2_per", "stage4_per"}
    assert abs(float(out["total"]) - float(gold_l["pml_total"])) <= 2e-6 * abs(float(gold_l["pml_total"])) + 1e-7
    assert maxerr(out["stage2_per"], gold_l["pml_s2"]) < 1e-6
    out["total"].backward()
    for pw in (None, torch.linspace(0.5, 3.0, K)):
        w = torch.linspace(0.5, 1.5, K)
        a2, a4 = 0.7, 1.3
        h2 = t(gold_f["img_logits"]).to(DEV).requires_grad_(True)
        h4 = t(gold_f["fusion_logits"]).to(DEV).requires_grad_(True)
        o = PathologyMultiLabelLoss(w, pw, a2, a4).to(DEV)(h2, h4, y.to(DEV), mk.to(DEV))
        o["total"].backward()
        r2 = t(gold_f["img_logits"]).clone().requires_grad_(True)
        r4 = t(gold_f["fusion_logits"]).clone().requires_grad_(True)
        ref = losses_ref.pathology_multilabel_loss(r2, r4, y, mk, w, pw, a2, a4)
        ref["total"].backward()
        assert abs(float(o["total"]) - float(ref["total"])) <= 2e-6 * abs(float(ref["total"]))
        for k in ("stage2_total", "stage4_total", "stage2_per", "stage4_per"):
            assert maxerr(o[k], ref[k]) < 2e-6, k
        assert maxerr(h2.grad, r2.grad) < 1e-6 and maxerr(h4.grad, r4.grad) < 1e-6
        assert float(r4.grad.abs().max()) > 1e-4          # the check is not vacuous


def test_pixel_prefetcher_hands_over_the_batches_in_order():
    """linear_probe.PixelPrefetcher: the next batch's pixels staged on a copy stream beside the current step, two device buffers."""
    from multimodal_edema_prediction_amd.linear_probe import PixelPrefetcher
    host = [torch.full((4, 3, 56, 56), float(i)).pin_memory() for i in range(7)]
    pre = PixelPrefetcher("cuda", host[0])
    pre.stage(host[0])
    sums = []
    for i in range(7):
        px = pre.take()
        if i + 1 < 7:
            pre.stage(host[i + 1])
        sums.append(px.double().mean())              # (device work on the taken buffer, enqueued before the next-but-one stage may overwrite it)
    assert [float(s) for s in sums] == [float(i) for i in range(7)]
    with pytest.raises(RuntimeError):
        pre.take()
    pre.stage(host[0]); pre.stage(host[1])
    with pytest.raises(RuntimeError):
        pre.stage(host[2])


def test_graphed_probe_step_equals_the_eager_probe_step():
    """graph_step.GraphedProbeStep (configs[1] as one captured graph) against the eager loop of cxr_linear_training.ipynb: same
    parameters after the same batches."""
    from multimodal_edema_prediction_amd.cohort import CohortCfg, make_batch
    from multimodal_edema_prediction_amd.graph_step import GraphedProbeStep
    from multimodal_edema_prediction_amd.linear_probe import RadDinoClassifier, masked_bce_with_logits_loss
    from multimodal_edema_prediction_amd.optim import FusedAdamW
    dev = torch.device("cuda")
    ccfg = CohortCfg(n_timesteps=8, n_vars=4, d_static=8, image_size=112, n_labels=7, seed=5)
    batches = [make_batch(ccfg, 13 * i, 4, mode="teacher") for i in range(3)]

    def build():
        torch.manual_seed(0)
        m = RadDinoClassifier("synthetic", num_classes=7, dropout=0.0).to(dev)
        m.train()
        return m, FusedAdamW([p for p in m.parameters() if p.requires_grad], lr=1e-3, weight_decay=1e-4)

    me, oe = build()
    for k in (0, 1, 2, 1):                           # (the captured step's warm-up step on the example batch is undone before the capture)
        b = batches[k]
        oe.zero_grad()
        masked_bce_with_logits_loss(me(b["pixel_values"].to(dev)), b["y_multi"].to(dev).float(), b["y_multi_mask"].to(dev).float()).backward()
        oe.step()
    mg, og = build()
    gs = GraphedProbeStep(mg, masked_bce_with_logits_loss, og, batches[0]["pixel_values"], batches[0]["y_multi"], batches[0]["y_multi_mask"], dev,
                          warmup=1)
    for k in (0, 1, 2, 1):
        b = batches[k]
        out = gs.step(b["pixel_values"], b["y_multi"].float(), b["y_multi_mask"].float())
    assert np.isfinite(float(out["loss"]))
    assert og._step == oe._step == 4
    for (k, a), (_, b2) in zip(me.named_parameters(), mg.named_parameters()):
        if a.requires_grad:
            assert float((a - b2).abs().max()) <= 1e-6, k
