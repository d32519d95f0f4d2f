"""The drop-in claim under the wrapper the reference trainer really uses: `accelerate.prepare` wraps the teacher in
`torch.nn.parallel.DistributedDataParallel(find_unused_parameters=True)` (training_duett/trainer.py:217-218, 418-421).  Two
processes share the box's one GPU (gloo carries the gradient buckets, so no second device is needed), each wraps this package's
`TeacherModel` in torch DDP and runs the engine step on its own shard; afterwards both ranks must hold identical parameters, and
they must equal a single-process run that averages the two shards' gradients by hand — DDP's reducer hooks, bucket views and
unused-parameter detection all work on modules whose arithmetic is HIP behind `torch.autograd.Function`s on two streams."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pytestmark = pytest.mark.gpu

T, V, DS, K, B = 32, 16, 8, 7, 2


def _build():
    from multimodal_edema_prediction_amd.main_architecture_duett import (CXREncoder, PatchDualPathologyPerceiver, TeacherModel,
                                                                           load_duett_backbone)
    torch.manual_seed(0)
    backbone = load_duett_backbone("synthetic", d_static_num=DS, d_time_series_num=V, n_timesteps=T, freeze=False)   # trainable DuETT: its SSL heads are the unused parameters
    cxr = CXREncoder("synthetic", freeze=True)
    per = PatchDualPathologyPerceiver(K, backbone.d_representation, dropout=0.0, head_dropout=0.0)
    torch.nn.init.normal_(per.correction_head[-1].weight, std=0.05)
    return TeacherModel(backbone, cxr, per, cxr_return_patches=True, d_img=768, use_aux_cxr=False, patch_dual_pathology_mode=True).to("cuda")


def _batch(i):
    from multimodal_edema_prediction_amd.cohort import CohortCfg, make_batch
    return make_batch(CohortCfg(n_timesteps=T, n_vars=V, d_static=DS, image_size=112, n_labels=K), 10 * i, B, mode="teacher")


def _worker(rank, world, port, q):
    import torch.distributed as dist
    from multimodal_edema_prediction_amd import engine
    from multimodal_edema_prediction_amd.losses_duett import DualPathologyLoss
    from multimodal_edema_prediction_amd.optim import FusedAdamW, make_param_groups
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    teacher = _build()
    opt = FusedAdamW(make_param_groups(teacher, 1e-3), weight_decay=5e-2)            # groups by parameter NAME, before the wrap (trainer.py:382)
    ddp = torch.nn.parallel.DistributedDataParallel(teacher, device_ids=[0], find_unused_parameters=True)
    loss_fn = DualPathologyLoss(torch.ones(K)).to("cuda")
    losses = [engine.train_teacher_dual_pathology_batch(_batch(2 * s + rank), ddp, loss_fn, opt, torch.device("cuda"))["loss"] for s in range(2)]
    unused = [k for k, p in teacher.named_parameters() if p.requires_grad and p.grad is None]
    q.put((rank, losses, {k: p.detach().cpu().numpy().copy() for k, p in teacher.named_parameters() if p.requires_grad}, unused))   # numpy: pickled by value (a torch tensor would travel as a shared-memory handle that dies with the worker)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_teacher_under_torch_ddp_matches_hand_averaged_gradients():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(2):
        rank, losses, params, unused = q.get(timeout=240)
        res[rank] = (losses, params, unused)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for k in res[0][1]:
        assert (res[0][1][k] == res[1][1][k]).all(), k                       # replicas stay in lock-step
    assert res[0][2] == res[1][2] and any("pretrain_" in k or k.startswith("duett.head") for k in res[0][2])   # DDP tolerated the unused heads

    # single process, the same two shards per step, gradients averaged by hand
    from multimodal_edema_prediction_amd import engine
    from multimodal_edema_prediction_amd.losses_duett import DualPathologyLoss
    from multimodal_edema_prediction_amd.optim import FusedAdamW, make_param_groups
    teacher = _build()
    opt = FusedAdamW(make_param_groups(teacher, 1e-3), weight_decay=5e-2)
    loss_fn = DualPathologyLoss(torch.ones(K)).to("cuda")
    engine._set_train_with_frozen_eval(teacher)
    train = [p for p in teacher.parameters() if p.requires_grad]
    for step in range(2):
        acc = {}
        for r in range(2):
            b = engine._move_lists(_batch(2 * step + r), "cuda")
            out = teacher(b["x_ts"], b["x_static"], b["bin_ends"], b["pixel_values"])
            L = loss_fn(out["img_logits"], out["ts_logits"], out["fusion_logits"], b["y_multi"], b["y_multi_mask"])
            for p in train:
                p.grad = None
            L["total"].backward()
            for p in train:
                if p.grad is not None:
                    acc[p] = acc.get(p, 0) + 0.5 * p.grad
        for p in train:
            p.grad = acc.get(p)
        opt.step()
    for k, p in teacher.named_parameters():
        if p.requires_grad:
            assert float((p.detach().cpu() - torch.from_numpy(res[0][1][k])).abs().max()) <= 2e-6, k
