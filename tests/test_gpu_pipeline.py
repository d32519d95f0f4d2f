"""Software pipelining of the frozen CXR encoder across steps (graph_step.GraphedTeacherStep(pipeline_cxr=True)): the encoder
forward of batch k+1 runs beside the training step of batch k.  It must change nothing numerically: same losses and the
same parameters, bit for bit, as the unpipelined captured step over a sequence of DISTINCT batches, including when the
caller breaks the announced order."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("split", [False, True])
def test_pipelined_step_is_bit_identical(split):
    # split=True is the N > 1 arrangement: forward/backward graph -> (all-reduce) -> optimiser graph on the main stream, the
    # frozen encoder's graph for the next batch on its own stream beside them
    import test_gpu_model as T
    from multimodal_edema_prediction_amd.graph_step import GraphedTeacherStep
    from multimodal_edema_prediction_amd.losses_duett import DualPathologyLoss
    from multimodal_edema_prediction_amd.optim import FusedAdamW, make_param_groups
    dev = torch.device("cuda")
    loss_fn = DualPathologyLoss(torch.ones(T.K), None, 0.5, 0.5, 1.0).to(dev)
    start = T.META["teacher_batch_start"]
    batches = [T.make_batch(T.CCFG, start + 97 * i, T.B, mode="teacher") for i in range(4)]
    assert not torch.equal(batches[0]["pixel_values"], batches[1]["pixel_values"])
    order = [0, 1, 2, 3, 0, 2, 1]                     # position 5 breaks the announced order on purpose (announced 1, got 2)
    announce = [1, 2, 3, 0, 1, 1, 0]

    def run(pipeline):
        te = T.build_teacher()
        opt = FusedAdamW(make_param_groups(te, 8e-5), weight_decay=5e-2)
        gs = GraphedTeacherStep(te, loss_fn, opt, batches[0], dev, warmup=2, pipeline_cxr=pipeline, split=split)
        losses = [float(gs.step(batches[k], batches[n])["loss"].item()) for k, n in zip(order, announce)]
        return losses, {k: p.detach().clone() for k, p in te.named_parameters() if p.requires_grad}

    l0, p0 = run(False)
    l1, p1 = run(True)
    np.testing.assert_array_equal(np.array(l1), np.array(l0))
    assert len(set(l0)) > 3                            # the batches really differ
    for k in p0:
        assert torch.equal(p0[k], p1[k]), k


def test_staged_host_batches_give_the_same_steps():
    """Host batches staged one call ahead on the copy stream (step(batch, next, after_next)) against device-resident batches:
    same losses and parameters, bit for bit, including a broken announcement."""
    import test_gpu_model as T
    from multimodal_edema_prediction_amd import engine
    from multimodal_edema_prediction_amd.graph_step import GraphedTeacherStep
    from multimodal_edema_prediction_amd.losses_duett import DualPathologyLoss
    from multimodal_edema_prediction_amd.optim import FusedAdamW, make_param_groups
    dev = torch.device("cuda")
    loss_fn = DualPathologyLoss(torch.ones(T.K), None, 0.5, 0.5, 1.0).to(dev)
    start = T.META["teacher_batch_start"]
    host = [T.make_batch(T.CCFG, start + 97 * i, T.B, mode="teacher") for i in range(4)]
    host = [dict(b, pixel_values=b["pixel_values"].pin_memory()) for b in host]
    order = [0, 1, 2, 3, 0, 2, 1, 3]
    nxt = [1, 2, 3, 0, 1, 1, 3, 0]                  # position 4 announces batch 1 but batch 2 arrives
    aft = [2, 3, 0, 1, 1, 3, 0, 1]

    def run(staged):
        te = T.build_teacher()
        opt = FusedAdamW(make_param_groups(te, 8e-5), weight_decay=5e-2)
        pool = host if staged else [engine._move_lists(b, dev) for b in host]
        gs = GraphedTeacherStep(te, loss_fn, opt, host[0], dev, warmup=2, pipeline_cxr=True)
        losses = [float(gs.step(pool[k], pool[n], pool[a] if staged else None)["loss"].item()) for k, n, a in zip(order, nxt, aft)]
        return losses, {k: p.detach().clone() for k, p in te.named_parameters() if p.requires_grad}

    l0, p0 = run(False)
    l1, p1 = run(True)
    np.testing.assert_array_equal(np.array(l1), np.array(l0))
    for k in p0:
        assert torch.equal(p0[k], p1[k]), k


def test_graph_step_with_scheduler_matches_eager_without_host_sync():
    """the learning-rate table must be the one of ITS step even when the host runs replays ahead of the GPU.
    LinearLR warm-up changes the rate every step; the graph run enqueues all steps with no host sync in between and is compared
    with the eager engine step (same arithmetic): per-step learning rates, losses and final parameters."""
    import test_gpu_model as T
    from multimodal_edema_prediction_amd import engine
    from multimodal_edema_prediction_amd.graph_step import GraphedTeacherStep
    from multimodal_edema_prediction_amd.losses_duett import DualPathologyLoss
    from multimodal_edema_prediction_amd.optim import FusedAdamW, make_param_groups, make_scheduler
    import warnings
    warnings.filterwarnings("ignore", message=".*lr_scheduler.step.*")
    dev = torch.device("cuda")
    loss_fn = DualPathologyLoss(torch.ones(T.K), None, 0.5, 0.5, 1.0).to(dev)
    batch = T.make_batch(T.CCFG, T.META["teacher_batch_start"], T.B, mode="teacher")
    N, W = 12, 2
    # the graph class's W warm-up iterations are undone before the capture (graph_step._TrainSnapshot): BOTH runs start from the
    # initial state, no warm-up offset on the eager side
    te = T.build_teacher()
    oe = FusedAdamW(make_param_groups(te, 8e-3), weight_decay=5e-2)
    se = make_scheduler(oe, total_steps=100, lr=8e-3, warmup_steps=10)
    eager_loss, eager_lr = [], []
    for _ in range(N):
        eager_lr.append([g["lr"] for g in oe.param_groups])
        eager_loss.append(engine.train_teacher_dual_pathology_batch(batch, te, loss_fn, oe, dev)["loss"])
        se.step()
    tg = T.build_teacher()
    og = FusedAdamW(make_param_groups(tg, 8e-3), weight_decay=5e-2)
    gs = GraphedTeacherStep(tg, loss_fn, og, batch, dev, warmup=W)
    sg = make_scheduler(og, total_steps=100, lr=8e-3, warmup_steps=10)
    graph_lr = []
    stream_losses = torch.zeros(N, device=dev)
    for i in range(N):                 # no .item(), no synchronize: the host runs ahead of the GPU
        graph_lr.append([g["lr"] for g in og.param_groups])
        stream_losses[i].copy_(gs.step(batch)["loss"])
        sg.step()
    torch.cuda.synchronize()
    np.testing.assert_allclose(np.array(graph_lr), np.array(eager_lr), rtol=1e-12)
    assert len({lr[-1] for lr in graph_lr}) == N                       # the rate really changed every step
    np.testing.assert_allclose(stream_losses.cpu().numpy(), np.array(eager_loss), rtol=2e-5, atol=1e-6)
    for (k, a), (_, b) in zip(te.named_parameters(), tg.named_parameters()):
        if a.requires_grad:
            assert float((a - b).abs().max()) <= 2e-6, k
    assert og._step == oe._step == N


def test_split_step_leaves_unused_parameters_alone():
    """in the N > 1 arrangement (flat gradient arena) parameters that never receive a gradient — DuETT's SSL heads
    when the backbone is trained inside the teacher — must stay untouched (no weight decay, no Adam state), exactly as in the
    one-graph step; and both arrangements must train the used parameters identically."""
    from multimodal_edema_prediction_amd.cohort import CohortCfg, make_batch
    from multimodal_edema_prediction_amd.graph_step import GraphedTeacherStep
    from multimodal_edema_prediction_amd.losses_duett import DualPathologyLoss
    from multimodal_edema_prediction_amd.main_architecture_duett import (CXREncoder, PatchDualPathologyPerceiver, TeacherModel,
                                                                           load_duett_backbone)
    from multimodal_edema_prediction_amd.optim import FusedAdamW, make_param_groups
    dev = torch.device("cuda")
    Tn, V, DS, K, B = 32, 16, 8, 7, 4
    batch = make_batch(CohortCfg(n_timesteps=Tn, n_vars=V, d_static=DS, image_size=224, n_labels=K), 0, B, mode="teacher")
    loss_fn = DualPathologyLoss(torch.ones(K)).to(dev)

    def run(split):
        torch.manual_seed(0)
        backbone = load_duett_backbone("synthetic", d_static_num=DS, d_time_series_num=V, n_timesteps=Tn, freeze=False)
        cxr = CXREncoder("synthetic", freeze=True)
        per = PatchDualPathologyPerceiver(K, backbone.d_representation, dropout=0.0, head_dropout=0.0)
        te = TeacherModel(backbone, cxr, per, cxr_return_patches=True, d_img=768, use_aux_cxr=False, patch_dual_pathology_mode=True).to(dev)
        before = {k: p.detach().clone() for k, p in te.named_parameters()}
        opt = FusedAdamW(make_param_groups(te, 1e-3), weight_decay=5e-2)
        gs = GraphedTeacherStep(te, loss_fn, opt, batch, dev, warmup=2, split=split, pipeline_cxr=True)
        losses = [float(gs.step(batch, batch)["loss"].item()) for _ in range(3)]
        return te, opt, before, losses, gs

    te0, opt0, before0, l0, _ = run(False)
    te1, opt1, before1, l1, gs1 = run(True)
    unused = [k for k, p in te1.named_parameters() if p.requires_grad and (k.startswith("duett.head") or "pretrain_" in k or "predict_events" in k)]
    assert unused and len(gs1.arena.unused) == len(unused)
    named1 = dict(te1.named_parameters())
    for k in unused:
        assert torch.equal(named1[k], before1[k]), k            # untouched: no decay
        assert named1[k].grad is None and len(opt1.state.get(named1[k], {})) == 0, k
    np.testing.assert_allclose(l1, l0, rtol=1e-6)
    for (k, a), (_, b) in zip(te0.named_parameters(), te1.named_parameters()):
        assert float((a - b).abs().max()) <= 1e-6, k


@pytest.mark.parametrize("kind", ["teacher", "student"])
def test_phase_check_leaves_the_training_state_untouched(kind, monkeypatch):
    """MEDP_PHASE_CHECK=1 (bench.py): after the capture the step runs a few times with staged host batches and with resident ones to
    see which hardware-queue phase it sits in — on a snapshot of everything a step writes.  The steps that follow must be the ones
    that follow without the check, bit for bit (losses, parameters, buffers, optimiser step count)."""
    import test_gpu_model as T
    from multimodal_edema_prediction_amd.graph_step import GraphedStudentStep, GraphedTeacherStep
    from multimodal_edema_prediction_amd.losses_duett import DualPathologyLoss, StudentKDLoss
    from multimodal_edema_prediction_amd.optim import FusedAdamW, make_param_groups
    dev = torch.device("cuda")
    start = T.META["teacher_batch_start"]
    batches = [T.make_batch(T.CCFG, start + 97 * i, T.B, mode="teacher") for i in range(3)]

    def run(check):
        monkeypatch.setenv("MEDP_PHASE_CHECK", "1" if check else "0")
        if kind == "teacher":
            model = T.build_teacher()
            opt = FusedAdamW(make_param_groups(model, 8e-5), weight_decay=5e-2)
            gs = GraphedTeacherStep(model, DualPathologyLoss(torch.ones(T.K), None, 0.5, 0.5, 1.0).to(dev), opt, batches[0], dev, warmup=2,
                                    pipeline_cxr=True)
        else:
            import test_gpu_student_graph as S
            sb = S._batches(3)
            model, teacher = S._build(dev, dropout=0.1)
            opt = FusedAdamW([p for p in model.parameters() if p.requires_grad], lr=1e-3, weight_decay=5e-2)
            gs = GraphedStudentStep(model, teacher, StudentKDLoss("vanilla_kl", 4.0, 0.5), opt, sb[0], dev, warmup=2)
        assert (gs.phase_log is not None) == check
        pool = batches if kind == "teacher" else sb
        losses = [float(gs.step(pool[i % 3], pool[(i + 1) % 3])["loss"].item()) for i in range(4)]
        return (losses, {k: p.detach().clone() for k, p in model.named_parameters()}, {k: b.detach().clone() for k, b in model.named_buffers()},
                opt._step)

    l0, p0, b0, s0 = run(False)
    l1, p1, b1, s1 = run(True)
    np.testing.assert_array_equal(np.array(l1), np.array(l0))
    assert s0 == s1
    for k in p0:
        assert torch.equal(p0[k], p1[k]), k
    for k in b0:
        assert torch.equal(b0[k], b1[k]), k


def test_replay_refuses_frozen_weights_changed_after_capture():
    """The captured graph reads the frozen encoders' PREPARED weights by address: writing a frozen parameter and calling the module
    eagerly rebuilds them.  The step keeps the captured ones alive and raises instead of training on stale weights."""
    import test_gpu_model as T
    from multimodal_edema_prediction_amd.graph_step import GraphedTeacherStep
    from multimodal_edema_prediction_amd.losses_duett import DualPathologyLoss
    from multimodal_edema_prediction_amd.optim import FusedAdamW, make_param_groups
    dev = torch.device("cuda")
    batch = T.make_batch(T.CCFG, T.META["teacher_batch_start"], T.B, mode="teacher")
    te = T.build_teacher()
    opt = FusedAdamW(make_param_groups(te, 8e-5), weight_decay=5e-2)
    gs = GraphedTeacherStep(te, DualPathologyLoss(torch.ones(T.K), None, 0.5, 0.5, 1.0).to(dev), opt, batch, dev, warmup=2, pipeline_cxr=True)
    l0 = float(gs.step(batch, batch)["loss"].item())
    assert np.isfinite(l0)
    with torch.no_grad():
        te.cxr.backbone.layernorm.weight.mul_(1.5)          # a frozen parameter changes ...
        te.cxr.forward_bf16(batch["pixel_values"].to(dev))  # ... and an eager call rebuilds the prepared weights
    with pytest.raises(RuntimeError, match="frozen weights were modified"):
        gs.step(batch, batch)
