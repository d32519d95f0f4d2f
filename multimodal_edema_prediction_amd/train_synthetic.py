"""Epoch-level driver of the hot path on the synthetic cohort: the host-side counterpart of the reference's `train_teacher`
(training_duett/trainer.py:216-764) and `train_student` (:828-989) for everything that is not data ETL — DataLoader ->
step function -> per-epoch evaluation (gathered over ranks) -> best.pt -> early-stop decision broadcast -> reload best -> test.
MIMIC loaders, wandb, tqdm and the in-loop gradient diagnostics (always "skipped" at the reference's HEAD) are
out of scope; the cohort comes from cohort.py (the reference's own smoke-test recipe).

    python -m multimodal_edema_prediction_amd.train_synthetic teacher --ckpt_dir runs/t0 --epochs 3
    python -m multimodal_edema_prediction_amd.train_synthetic student --teacher_ckpt runs/t0/best.pt --ckpt_dir runs/s0
    torchrun --nproc-per-node N -m multimodal_edema_prediction_amd.train_synthetic teacher ...      (one rank per GPU, RCCL)

Defaults follow training_duett/run.py (lr 8e-5, weight decay 5e-2, warm-up 300 steps, backbone / query LR x0.2, patience 5,
perceiver dropout 0.2, d_latent 256, 4 heads, head_hidden 128).  The step is the captured-graph step (graph_step.py) by default —
the eager engine step is host-bound at ~1/3 of its rate; `--eager` runs `engine.py` from Python instead.  Both run the same
arithmetic from the same initial state (the graph class undoes its warm-up steps; tests/test_gpu_pipeline.py).  The loop hands the
graph step the NEXT batch too (one-batch look-ahead), so the frozen part of batch k+1 runs beside the step of batch k and no
encoder forward is run twice.
"""
from __future__ import annotations

import argparse
import os
import sys

import torch

from . import checkpoint, dp, engine, evaluator
from .cohort import PATHOLOGY_LABELS, CohortCfg, SyntheticCohort, collate
from .losses_duett import DualPathologyLoss, StudentKDLoss
from .main_architecture_duett import (CXREncoder, DualPathologyPerceiver, PatchDualPathologyPerceiver, StudentModel, TeacherModel,
                                      load_duett_backbone)
from .optim import FusedAdamW, make_param_groups, make_scheduler


# ------------------------------------------------------------------------------------------------------------------ arguments
def _common(ap: argparse.ArgumentParser) -> None:
    ap.add_argument("--ckpt_dir", required=True)
    ap.add_argument("--epochs", type=int, default=3)
    ap.add_argument("--batch_size", type=int, default=64, help="per rank (accelerate semantics, run.py:100)")
    ap.add_argument("--lr", type=float, default=8e-5)
    ap.add_argument("--weight_decay", type=float, default=5e-2)
    ap.add_argument("--warmup_steps", type=int, default=300)
    ap.add_argument("--backbone_lr_mult", type=float, default=0.2)
    ap.add_argument("--query_lr_mult", type=float, default=0.2)
    ap.add_argument("--correction_lr_mult", type=float, default=1.0)
    ap.add_argument("--patience", type=int, default=5)
    ap.add_argument("--limit_batches", type=int, default=0, help="dry-run knob of the reference (run.py:106-107): stop an epoch early")
    ap.add_argument("--n_timesteps", type=int, default=96)
    ap.add_argument("--n_vars", type=int, default=48)
    ap.add_argument("--d_static", type=int, default=8)
    ap.add_argument("--image_size", type=int, default=224)
    ap.add_argument("--n_train", type=int, default=4096)
    ap.add_argument("--n_val", type=int, default=1024)
    ap.add_argument("--n_test", type=int, default=1024)
    ap.add_argument("--num_workers", type=int, default=0)
    ap.add_argument("--seed", type=int, default=1234)
    ap.add_argument("--duett_ckpt", default="synthetic")
    ap.add_argument("--cxr_model_name", default="synthetic")
    ap.add_argument("--head_hidden", type=int, default=128)
    ap.add_argument("--head_dropout", type=float, default=0.2)
    ap.add_argument("--transformer_dropout", type=float, default=0.0)
    ap.add_argument("--aug_noise", type=float, default=0.0)
    ap.add_argument("--aug_mask", type=float, default=0.0)
    ap.add_argument("--graph", action="store_true", help="(default) replay the captured HIP-graph step; kept for old command lines")
    ap.add_argument("--eager", action="store_true", help="run the eager engine step (engine.py from Python) instead of the captured-graph step")
    ap.add_argument("--learnable_labels", action="store_true", help="labels depend on the inputs (so that AUROC moves)")


def parse_args(argv=None) -> argparse.Namespace:
    ap = argparse.ArgumentParser(prog="train_synthetic")
    sub = ap.add_subparsers(dest="stage", required=True)
    t = sub.add_parser("teacher")
    _common(t)
    t.add_argument("--perceiver_type", choices=("dual_patch", "dual"), default="dual_patch")
    t.add_argument("--d_latent", type=int, default=256)
    t.add_argument("--n_perceiver_heads", type=int, default=4)
    t.add_argument("--perceiver_dropout", type=float, default=0.2)
    t.add_argument("--freeze_duett", action="store_true")
    t.add_argument("--unfreeze_cxr", action="store_true")
    t.add_argument("--pathology_labels", default=",".join(PATHOLOGY_LABELS))
    t.add_argument("--pretrained_cxr_head_ckpt", default=None, help="linear-probe checkpoint (perceiver_type dual)")
    t.add_argument("--alpha_img", type=float, default=0.5)
    t.add_argument("--alpha_ts", type=float, default=0.5)
    t.add_argument("--alpha_fus", type=float, default=1.0)
    t.add_argument("--aux_residual_alpha", type=float, default=0.0)
    s = sub.add_parser("student")
    _common(s)
    s.add_argument("--teacher_ckpt", required=True)
    s.add_argument("--student_pool", default="mean")
    s.add_argument("--kd_name", default="vanilla_kl")
    s.add_argument("--kd_T", type=float, default=4.0)
    s.add_argument("--kd_alpha", type=float, default=0.5)
    args = ap.parse_args(argv)
    # graph mode is the default; the eager engine step stays for --eager and for what the captured step does not carry
    # (the aux residual KL of engine.py:149-165 is an engine extra, off by default)
    args.graph = not args.eager and float(getattr(args, "aux_residual_alpha", 0.0)) == 0.0
    return args


def _with_next(loader, limit: int = 0):
    """(batch, next batch or None) pairs: the graph step runs the frozen part of the NEXT batch beside this batch's step."""
    it = iter(loader)
    cur = next(it, None)
    n = 0
    while cur is not None:
        n += 1
        nxt = None if (limit and n >= limit) else next(it, None)
        yield cur, nxt
        cur = nxt


def _same_size(a: dict, b: dict | None) -> dict | None:
    """The captured step has static shapes: a ragged last batch is neither trained on by the graph nor announced to it."""
    return b if (b is not None and b["y"].shape[0] == a["y"].shape[0]) else None


# ------------------------------------------------------------------------------------------------------------------ plumbing
def _device(local: int) -> torch.device:
    from . import abi
    abi.require_gpu()
    torch.cuda.set_device(local)
    return torch.device("cuda", local)


def make_loader(dataset, batch_size: int, shuffle: bool, num_workers: int, mode: str, rank: int, world: int, epoch_seed: int = 0):
    """trainer.py:54-60 (+ accelerate's sharding: rank r takes items r, r+N, ... of every global batch; `drop_last` with shuffle)."""
    n = len(dataset)
    g = torch.Generator().manual_seed(epoch_seed)
    order = torch.randperm(n, generator=g).tolist() if shuffle else list(range(n))
    mine = order[rank::world]
    if shuffle:
        mine = mine[:len(mine) // batch_size * batch_size]
    sub = torch.utils.data.Subset(dataset, mine)
    return torch.utils.data.DataLoader(sub, batch_size=batch_size, shuffle=False, num_workers=num_workers, pin_memory=True,
                                       collate_fn=lambda items: collate(items, mode), drop_last=False)


def _datasets(args, mode="teacher"):
    ccfg = CohortCfg(n_timesteps=args.n_timesteps, n_vars=args.n_vars, d_static=args.d_static, image_size=args.image_size,
                     n_labels=len(PATHOLOGY_LABELS), seed=args.seed, learnable=bool(args.learnable_labels))
    return (SyntheticCohort(ccfg, args.n_train, mode, 0), SyntheticCohort(ccfg, args.n_val, mode, 10_000_000),
            SyntheticCohort(ccfg, args.n_test, mode, 20_000_000))


def _steps_per_epoch(n_items: int, args, world: int) -> int:
    per_rank = (n_items + world - 1) // world // args.batch_size
    return min(per_rank, args.limit_batches) if args.limit_batches else per_rank


def _log(rank, *a):
    if rank == 0:
        print(*a, flush=True)


# ------------------------------------------------------------------------------------------------------------------ teacher
def build_teacher(args, device) -> TeacherModel:
    """trainer.py:274-355."""
    backbone = load_duett_backbone(args.duett_ckpt, d_static_num=args.d_static, d_time_series_num=args.n_vars, n_timesteps=args.n_timesteps,
                                   freeze=bool(args.freeze_duett), aug_noise=args.aug_noise, aug_mask=args.aug_mask,
                                   transformer_dropout=args.transformer_dropout)
    dual = args.perceiver_type == "dual"
    cxr = CXREncoder(args.cxr_model_name, freeze=not args.unfreeze_cxr, return_patches=not dual)
    labels = tuple(s.strip() for s in args.pathology_labels.split(","))
    cls = DualPathologyPerceiver if dual else PatchDualPathologyPerceiver
    perceiver = cls(n_pathologies=len(labels), d_ts=backbone.d_representation, d_latent=args.d_latent, n_heads=args.n_perceiver_heads,
                    dropout=args.perceiver_dropout)
    teacher = TeacherModel(backbone, cxr, perceiver, head_hidden=args.head_hidden, head_dropout=args.head_dropout,
                           cxr_return_patches=not dual, d_img=cxr.d_out, use_aux_cxr=False, dual_pathology_mode=dual,
                           patch_dual_pathology_mode=not dual, pretrained_cxr_head_ckpt=args.pretrained_cxr_head_ckpt if dual else None,
                           pathology_labels=labels if dual else None)
    return teacher.to(device)


def train_teacher(args) -> dict:
    rank, local, world = dp.init_distributed()
    device = _device(local)
    os.makedirs(args.ckpt_dir, exist_ok=True)
    torch.manual_seed(0)
    teacher = build_teacher(args, device)
    dp.broadcast_parameters(teacher)
    labels = tuple(s.strip() for s in args.pathology_labels.split(","))
    train_ds, val_ds, test_ds = _datasets(args)
    opt = FusedAdamW(make_param_groups(teacher, args.lr, args.backbone_lr_mult, args.query_lr_mult, args.correction_lr_mult),
                     weight_decay=args.weight_decay)
    loss_fn = DualPathologyLoss(torch.ones(len(labels)), None, args.alpha_img, args.alpha_ts, args.alpha_fus).to(device)
    spe = max(_steps_per_epoch(len(train_ds), args, world), 1)
    sched = make_scheduler(opt, spe * args.epochs, args.lr, args.warmup_steps)
    reducer = dp.GradAllReducer([p for p in teacher.parameters() if p.requires_grad]).attach(opt) if (world > 1 and not args.graph) else None
    gstep = None
    best, best_path, since_best, history = -1.0, os.path.join(args.ckpt_dir, "best.pt"), 0, []
    val_loader = make_loader(val_ds, args.batch_size, False, args.num_workers, "teacher", rank, world)
    for epoch in range(1, args.epochs + 1):
        loader = make_loader(train_ds, args.batch_size, True, args.num_workers, "teacher", rank, world, epoch_seed=args.seed + epoch)
        run = {"loss": 0.0, "img": 0.0, "ts": 0.0, "fus": 0.0, "n": 0}
        for step, (batch, nxt) in enumerate(_with_next(loader, args.limit_batches)):
            if args.graph and gstep is None:
                from .graph_step import GraphedTeacherStep
                gstep = GraphedTeacherStep(teacher, loss_fn, opt, batch, device, world=world,
                                           pipeline_cxr=not args.unfreeze_cxr and args.perceiver_type != "dual")
                graph_bs = batch["y"].shape[0]
            if args.graph and batch["y"].shape[0] == graph_bs:
                out = gstep.step(batch, _same_size(batch, nxt))
                out = {"loss": out["loss"], "img_total": out["img_total"], "ts_total": out["ts_total"], "fus_total": out["fus_total"]}
            else:
                if args.graph:         # (the training loader drops the ragged last batch, trainer.py:54-60: not reached from it)
                    raise RuntimeError("captured-graph step: the loader produced a batch of a different size than the captured one; "
                                       "the graph's optimiser table must not be rebuilt by an eager step in between — use --eager")
                out = engine.train_teacher_dual_pathology_batch(batch, teacher, loss_fn, opt, device, aux_residual_alpha=args.aux_residual_alpha)
            sched.step()
            bs = batch["y"].shape[0]
            run["n"] += bs
            for k, src in (("loss", "loss"), ("img", "img_total"), ("ts", "ts_total"), ("fus", "fus_total")):
                run[k] += float(out[src]) * bs                 # sample-weighted running means (trainer.py:467-479)
        n = max(run["n"], 1)
        val = evaluator.evaluate_dual_pathology(teacher, val_loader, device, labels, gather=True)
        improved = val["main_auroc"] > best
        if improved:
            best = val["main_auroc"]
            if rank == 0:
                checkpoint.save_ckpt(best_path, teacher, opt, epoch, best, args)
        improved = dp.broadcast_flag(improved, src=0, device=device)     # rank 0's decision, so ranks don't hang (trainer.py:708-711)
        since_best = 0 if improved else since_best + 1
        history.append({"epoch": epoch, "train_loss": run["loss"] / n, "val_macro_auroc": val["main_auroc"], "improved": bool(improved)})
        _log(rank, f"[teacher ep{epoch}] loss {run['loss'] / n:.4f} (img {run['img'] / n:.4f} ts {run['ts'] / n:.4f} fus {run['fus'] / n:.4f})  "
                   f"val macro-AUROC {val['main_auroc']:.4f}  best {best:.4f}  lr {opt.param_groups[-1]['lr']:.2e}")
        if args.patience > 0 and since_best >= args.patience:            # trainer.py:712-716
            _log(rank, f"[teacher] early stop at epoch {epoch}: no val AUROC improvement for {args.patience} epochs (best {best:.4f})")
            break
    if world > 1:
        torch.distributed.barrier()
    state = {"epoch": None}
    if os.path.exists(best_path):                                # reload best (the reference does it on rank 0 only, :719-721)
        state = checkpoint.load_ckpt(best_path)
        checkpoint.load_model_state(teacher, state)
    test = evaluator.evaluate_dual_pathology(teacher, make_loader(test_ds, args.batch_size, False, args.num_workers, "teacher", rank, world),
                                             device, labels, gather=True)
    _log(rank, f"[teacher] test macro-AUROC {test['main_auroc']:.4f} macro-AUPRC {test['main_auprc']:.4f} (best epoch {state['epoch']})")
    if reducer is not None:
        reducer.detach()
    return {"best_val_auroc": best, "test": test, "history": history, "ckpt": best_path}


# ------------------------------------------------------------------------------------------------------------------ student
def build_teacher_from_ckpt(teacher_state: dict, d_static: int, n_vars: int, duett_ckpt: str, n_timesteps: int,
                            cxr_model_name_fallback: str) -> TeacherModel:
    """trainer.py:770-822: rebuild the frozen dual-mode teacher from the hyper-parameters stored in its own checkpoint."""
    t_args = teacher_state["args"]
    if t_args.get("perceiver_type") != "dual":
        raise NotImplementedError("student KD supports a `dual` perceiver teacher only (as the reference does); the checkpoint has "
                                  f"perceiver_type={t_args.get('perceiver_type')!r}")
    labels = tuple(s.strip() for s in t_args["pathology_labels"].split(","))
    backbone = load_duett_backbone(duett_ckpt, d_static_num=d_static, d_time_series_num=n_vars, n_timesteps=n_timesteps, freeze=True,
                                   aug_noise=0.0, aug_mask=0.0, transformer_dropout=0.0)
    cxr = CXREncoder(t_args.get("cxr_model_name", cxr_model_name_fallback), freeze=True, return_patches=False)
    perceiver = DualPathologyPerceiver(n_pathologies=len(labels), d_ts=backbone.d_representation, d_latent=int(t_args["d_latent"]),
                                       n_heads=int(t_args["n_perceiver_heads"]), dropout=float(t_args["perceiver_dropout"]))
    teacher = TeacherModel(backbone, cxr, perceiver, head_hidden=int(t_args["head_hidden"]), head_dropout=float(t_args["head_dropout"]),
                           cxr_return_patches=False, d_img=cxr.d_out, use_aux_cxr=False, pathology_mode=False, dual_pathology_mode=True,
                           pretrained_cxr_head_ckpt=t_args["pretrained_cxr_head_ckpt"], pathology_labels=labels)
    return checkpoint.load_model_state(teacher, teacher_state, freeze=True)


def train_student(args) -> dict:
    rank, local, world = dp.init_distributed()
    device = _device(local)
    os.makedirs(args.ckpt_dir, exist_ok=True)
    teacher = build_teacher_from_ckpt(checkpoint.load_ckpt(args.teacher_ckpt), args.d_static, args.n_vars, args.duett_ckpt,
                                      args.n_timesteps, args.cxr_model_name).to(device)
    torch.manual_seed(1)
    backbone = load_duett_backbone(args.duett_ckpt, d_static_num=args.d_static, d_time_series_num=args.n_vars, n_timesteps=args.n_timesteps,
                                   freeze=False, aug_noise=args.aug_noise, aug_mask=args.aug_mask, transformer_dropout=args.transformer_dropout)
    student = StudentModel(backbone, pool=args.student_pool, head_hidden=args.head_hidden, head_dropout=args.head_dropout).to(device)
    dp.broadcast_parameters(student)
    train_ds, val_ds, test_ds = _datasets(args)                   # every loader in "teacher" mode: one batch feeds both (:890-895)
    kd = StudentKDLoss(args.kd_name, args.kd_T, args.kd_alpha, None)
    opt = FusedAdamW(make_param_groups(student, args.lr, args.backbone_lr_mult, args.query_lr_mult, args.correction_lr_mult),
                     weight_decay=args.weight_decay)
    spe = max(_steps_per_epoch(len(train_ds), args, world), 1)
    sched = make_scheduler(opt, spe * args.epochs, args.lr, args.warmup_steps)
    reducer = dp.GradAllReducer([p for p in student.parameters() if p.requires_grad]).attach(opt) if (world > 1 and not args.graph) else None
    gstep = None
    fwd = evaluator.make_student_forward()
    best, best_path, since_best, history = -1.0, os.path.join(args.ckpt_dir, "best.pt"), 0, []
    val_loader = make_loader(val_ds, args.batch_size, False, args.num_workers, "teacher", rank, world)
    for epoch in range(1, args.epochs + 1):
        loader = make_loader(train_ds, args.batch_size, True, args.num_workers, "teacher", rank, world, epoch_seed=args.seed + epoch)
        run = {"loss": 0.0, "bce": 0.0, "kd": 0.0, "n": 0}
        for step, (batch, nxt) in enumerate(_with_next(loader, args.limit_batches)):
            if args.graph and gstep is None:
                from .graph_step import GraphedStudentStep
                gstep = GraphedStudentStep(student, teacher, kd, opt, batch, device, world=world, pipeline_teacher=True)
                graph_bs = batch["y"].shape[0]
            if args.graph and batch["y"].shape[0] == graph_bs:
                out = gstep.step(batch, _same_size(batch, nxt))
            else:
                if args.graph:         # (the training loader drops the ragged last batch, trainer.py:54-60: not reached from it)
                    raise RuntimeError("captured-graph step: the loader produced a batch of a different size than the captured one; "
                                       "the graph's optimiser table must not be rebuilt by an eager step in between — use --eager")
                out = engine.train_student_batch(batch, batch, student, teacher, kd, opt, device)
            sched.step()
            bs = batch["y"].shape[0]
            run["n"] += bs
            for k in ("loss", "bce", "kd"):
                run[k] += float(out[k]) * bs
        n = max(run["n"], 1)
        val = evaluator.evaluate_binary(student, val_loader, device, fwd, gather=True)
        improved = val["auroc"] > best
        if improved:
            best = val["auroc"]
            if rank == 0:
                checkpoint.save_ckpt(best_path, student, opt, epoch, best, args)
        improved = dp.broadcast_flag(improved, src=0, device=device)     # trainer.py:969-973
        since_best = 0 if improved else since_best + 1
        history.append({"epoch": epoch, "train_loss": run["loss"] / n, "val_auroc": val["auroc"], "improved": bool(improved)})
        _log(rank, f"[student ep{epoch}] loss {run['loss'] / n:.4f} (bce {run['bce'] / n:.4f} kd {run['kd'] / n:.4f})  val AUROC {val['auroc']:.4f}  "
                   f"best {best:.4f}")
        if args.patience > 0 and since_best >= args.patience:
            _log(rank, f"[student] early stop at epoch {epoch}: no val AUROC improvement for {args.patience} epochs (best {best:.4f})")
            break
    if world > 1:
        torch.distributed.barrier()
    if os.path.exists(best_path):          # a validation AUROC that is never defined (one class only) saves nothing, as in the reference
        checkpoint.load_model_state(student, checkpoint.load_ckpt(best_path))
    test = evaluator.evaluate_binary(student, make_loader(test_ds, args.batch_size, False, args.num_workers, "teacher", rank, world), device,
                                     fwd, gather=True)
    _log(rank, f"[student] test AUROC {test['auroc']:.4f} AUPRC {test['auprc']:.4f}")
    if reducer is not None:
        reducer.detach()
    return {"best_val_auroc": best, "test": test, "history": history, "ckpt": best_path}


def main(argv=None) -> dict:
    args = parse_args(argv)
    out = train_teacher(args) if args.stage == "teacher" else train_student(args)
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()
    return out


if __name__ == "__main__":
    main(sys.argv[1:])
