"""Student KD path (DuETT trained end-to-end, BatchNorm in train mode) on the HIP path against the golden fixtures produced
by the reference's own `train_student_batch` on CPU (tests/golden/student_step_cfg1.npz, duett_cfg1.npz)."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from helpers import GOLDEN_DIR, load_npz, load_shapes, synth_state_dict, t  # noqa: E402
from multimodal_edema_prediction_amd import engine  # noqa: E402
from multimodal_edema_prediction_amd.cohort import CohortCfg, make_batch  # noqa: E402
from multimodal_edema_prediction_amd.losses_duett import StudentKDLoss  # noqa: E402
from multimodal_edema_prediction_amd.main_architecture_duett import DuettFeatureExtractor, StudentModel  # noqa: E402
from multimodal_edema_prediction_amd.optim import FusedAdamW  # noqa: E402

META = json.load(open(os.path.join(GOLDEN_DIR, "meta.json")))
SHAPES = load_shapes("shapes.json")
B, T, V, DS, K = META["B"], META["T"], META["V"], META["DS"], META["K"]
CCFG = CohortCfg(n_timesteps=T, n_vars=V, d_static=DS, image_size=224, seed=META["cohort_seed"])
DEV = "cuda"


def new_backbone():
    return DuettFeatureExtractor(d_static_num=DS, d_time_series_num=V, d_target=1, pretrain=False, masked_transform_timesteps=T,
                                 max_len=T, aug_noise=0.0, aug_mask=0.0, transformer_dropout=0.0)


def build_student():
    s = StudentModel(new_backbone(), pool="mean", head_hidden=128, head_dropout=0.0)
    assert sorted(s.state_dict()) == sorted(SHAPES["student"])
    s.load_state_dict(synth_state_dict(SHAPES["student"], seed=2), strict=True)
    return s.to(DEV)


class _FixedTeacher(torch.nn.Module):
    """Stands in for the frozen teacher: returns the reference's own teacher logits (fixture), so this test isolates the student."""

    def __init__(self, z):
        super().__init__()
        self.z = z

    def forward(self, *a, **k):
        return {"main_logit": self.z}


def maxerr(a, b):
    return float((a.detach().float().cpu() - torch.as_tensor(b)).abs().max())


def test_train_mode_encode_matches_reference_batchnorm():
    gold = load_npz("duett_cfg1.npz")
    m = new_backbone()
    m.load_state_dict(synth_state_dict(SHAPES["duett"], seed=1), strict=True)
    m = m.to(DEV).train()
    xin = (t(gold["xs_static"]).to(DEV), t(gold["xs_ts"]).to(DEV), t(gold["xs_times"]).to(DEV), list(gold["n_timesteps"]))
    tok = m.encode(xin)
    assert tok.requires_grad
    err = (tok.detach().cpu() - t(gold["enc_train"])).abs()
    assert float(err.max()) < 3e-2 and float(err.mean()) < 3e-3, (float(err.max()), float(err.mean()))
    sd = m.state_dict()
    assert maxerr(sd["embedding_layers.3.3.batch_norm.running_mean"], gold["bn_rm_after"]) < 1e-5
    assert maxerr(sd["embedding_layers.3.3.batch_norm.running_var"], gold["bn_rv_after"]) < 1e-5
    assert maxerr(sd["full_time_embedding.2.batch_norm.running_var"], gold["tbn_rv_after"]) < 1e-5
    assert int(sd["embedding_layers.0.3.batch_norm.num_batches_tracked"]) == 1
    # eval-mode training path (running statistics, autograd on) == the inference megacall
    gold_eval = t(gold["enc_eval"])
    m2 = new_backbone()
    m2.load_state_dict(synth_state_dict(SHAPES["duett"], seed=1), strict=True)
    m2 = m2.to(DEV).eval()
    tok2 = m2.encode(xin)
    assert tok2.requires_grad and float((tok2.detach().cpu() - gold_eval).abs().max()) < 3e-2


def test_student_kd_forward_loss_and_gradients():
    gold = load_npz("student_step_cfg1.npz")
    student = build_student().train()
    tb = make_batch(CCFG, META["teacher_batch_start"], B, mode="student")
    b = engine._move_lists(tb, DEV)
    z_s = student(b["x_ts"], b["x_static"], b["bin_ends"])
    assert maxerr(z_s, gold["z_s_train"]) < 3e-2, maxerr(z_s, gold["z_s_train"])
    L = StudentKDLoss("vanilla_kl", 4.0, 0.5, None)(z_s, t(gold["z_t"]).to(DEV), b["y"])
    for k in ("total", "bce", "kd"):
        assert abs(float(L[k]) - float(gold[k])) <= 1e-2 * abs(float(gold[k])) + 1e-5, (k, float(L[k]), float(gold[k]))
    student.zero_grad()
    L["total"].backward()
    named = dict(student.named_parameters())
    n = 0
    for key in gold:
        if not key.startswith("grad:"):
            continue
        g, want = named[key[5:]].grad, torch.as_tensor(gold[key])
        assert g is not None, key
        g = g.float().cpu()
        cos = float((g * want).sum() / (g.norm() * want.norm() + 1e-30))
        rel = float((g - want).norm() / (want.norm() + 1e-30))
        assert cos > 0.99 and rel < 0.15, (key, cos, rel)
        n += 1
    assert n >= 15
    # same set of parameters receives gradients as in the reference (SSL heads stay None -> find_unused_parameters)
    got = {k for k, p in named.items() if p.grad is not None}
    want = {k[5:] for k in gold if k.startswith("gsum:")}
    assert got == want, (sorted(got - want)[:5], sorted(want - got)[:5])
    for key in gold:
        if key.startswith("gsum:"):
            np.testing.assert_allclose(float(named[key[5:]].grad.double().abs().sum()), gold[key][1], rtol=0.15, atol=1e-5)


def test_student_engine_step():
    gold = load_npz("student_step_cfg1.npz")
    student = build_student()
    teacher = _FixedTeacher(t(gold["z_t"]).to(DEV))
    tb = make_batch(CCFG, META["teacher_batch_start"], B, mode="student")
    tb_t = dict(tb, pixel_values=torch.zeros(1))
    opt = FusedAdamW([p for p in student.parameters() if p.requires_grad], lr=8e-5, weight_decay=5e-2)
    before = {k: p.detach().clone() for k, p in student.named_parameters()}
    out = engine.train_student_batch(tb, tb_t, student, teacher, StudentKDLoss("vanilla_kl", 4.0, 0.5, None), opt, torch.device(DEV))
    assert abs(out["loss"] - float(gold["step_loss"])) <= 1e-2 * abs(float(gold["step_loss"]))
    n = 0
    for k, p in student.named_parameters():
        if ("post:" + k) not in gold or p.grad is None:
            assert torch.equal(p.detach(), before[k]) or p.grad is not None
            continue
        np.testing.assert_allclose(float(p.detach().double().abs().sum()), gold["post:" + k][1], rtol=3e-4, atol=3e-3)
        n += 1
    assert n > 50


def _student_grads(dropout, fused, pool):
    """One forward / backward of the student at a fixed dropout seed; (logits, {name: grad})."""
    from multimodal_edema_prediction_amd import autograd_ops as A, duett_train as DT
    torch.manual_seed(11)
    bb = DuettFeatureExtractor(d_static_num=DS, d_time_series_num=V, d_target=1, pretrain=False, masked_transform_timesteps=T, max_len=T,
                               aug_noise=0.0, aug_mask=0.0, transformer_dropout=dropout)
    s = StudentModel(bb, pool="mean", head_hidden=128, head_dropout=0.0)
    s.load_state_dict(synth_state_dict(SHAPES["student"], seed=2), strict=True)
    s = s.to(DEV).train()
    b = engine._move_lists(make_batch(CCFG, META["teacher_batch_start"], B, mode="student"), DEV)
    prev = DT._FUSED_NODES
    DT._FUSED_NODES = fused
    try:
        if pool:
            with A.record_operands() as log:                       # a first pass tells which operands the step asks for
                s(b["x_ts"], b["x_static"], b["bin_ends"]).sum().backward()
            s.zero_grad()
            wp = A.WeightOperandPool(log, torch.device(DEV))
            assert wp.n_jobs >= 16
            for prm in s.parameters():                             # as after an optimiser update: every cached operand is stale
                torch.autograd.graph.increment_version(prm)
            wp.refresh()
        torch.manual_seed(5)                                       # A.next_seed() draws the dropout seed from the CPU generator
        z = s(b["x_ts"], b["x_static"], b["bin_ends"])
        (z * torch.linspace(-1, 1, z.numel(), device=DEV).view_as(z)).sum().backward()
    finally:
        DT._FUSED_NODES = prev
    return z.detach().clone(), {k: p.grad.detach().clone() for k, p in s.named_parameters() if p.grad is not None}


@pytest.mark.parametrize("dropout", [0.0, 0.2])
def test_fused_encoder_halves_and_operand_pool_are_bit_identical_to_the_separate_nodes(dropout):
    """AttnHalfFn / FeedForwardHalfFn (16-bit hand-overs inside one autograd node) and WeightOperandPool (all weight operands from one
    launch) round every value where the separate nodes round it: logits and every gradient are the same bits."""
    z0, g0 = _student_grads(dropout, fused=False, pool=False)
    for fused, pool in ((True, False), (False, True), (True, True)):
        z1, g1 = _student_grads(dropout, fused=fused, pool=pool)
        assert torch.equal(z0, z1), (fused, pool, float((z0 - z1).abs().max()))
        assert g0.keys() == g1.keys()
        for k in g0:
            assert torch.equal(g0[k], g1[k]), (fused, pool, k, float((g0[k] - g1[k]).abs().max()))


def test_swap_with_embedding_added_on_the_way_matches_the_separate_kernels():
    """SwapAddFn (axis swap + positional embedding in one pass, the [B, T+1, tt] time-embedding concatenation never built) against
    AxisSwapFn -> AddBcastFn -> torch.cat: same logits bit for bit; gradients equal except where a sum over the batch changed its order."""
    res = {}
    for flag in ("0", "1"):
        os.environ["MEDP_DUETT_FUSED_SWAP"] = flag
        try:
            res[flag] = _student_grads(0.2, fused=True, pool=False)
        finally:
            os.environ.pop("MEDP_DUETT_FUSED_SWAP", None)
    (z0, g0), (z1, g1) = res["0"], res["1"]
    assert torch.equal(z0, z1)
    assert g0.keys() == g1.keys()
    same = 0
    for k in g0:
        a, b = g0[k].double(), g1[k].double()
        assert float((a - b).abs().max()) <= 1e-5 * float(a.abs().max()) + 1e-9, k
        same += int(torch.equal(g0[k], g1[k]))
    assert same >= len(g0) - 4, (same, len(g0))
