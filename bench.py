#!/usr/bin/env python3
"""Headline benchmark: multimodal TEACHER training-step throughput (BASELINE.json configs[2]: full `dual_patch` teacher,
CXR 224x224 through a frozen ViT-B/14 + frozen DuETT over T=96 / V=48 vitals + trainable pathology-query fusion head,
batch 64 per GPU, DualPathologyLoss, AdamW with the reference's LR groups and warm-up/cosine schedule).

    python bench.py --gpus N --steps K --warmup W          (N>1: launched by torch.distributed.run, one rank per GPU)

One step = `train_teacher_dual_pathology_batch` (forward of both encoders and the fusion head, loss, backward of every
trainable parameter, gradient all-reduce when N>1, optimiser step, scheduler step) on one batch already resident in HBM.
Prints ONE JSON line (rank 0) with the driver's contract plus `roofline` (live HIP-event timing of the dominant kernel,
the CXR-encoder block GEMMs) and `cpu_baseline` (the CPU oracle timed on the host cores, N=1 only).
"""
from __future__ import annotations

import argparse
import ctypes
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0          # dense bf16 MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md
GFLOP_PER_SAMPLE = 48.25           # ViT 46.32 + DuETT fwd 1.04 + 3 x fusion 0.294


def build_teacher(T, V, DS, K, device, seed=0, freeze_cxr=True):
    from multimodal_edema_prediction_amd.main_architecture_duett import (CXREncoder, PatchDualPathologyPerceiver, TeacherModel,
                                                                           load_duett_backbone)
    torch.manual_seed(seed)
    backbone = load_duett_backbone("synthetic", d_static_num=DS, d_time_series_num=V, n_timesteps=T, freeze=True)   # --freeze_duett
    cxr = CXREncoder("synthetic", freeze=freeze_cxr, return_patches=True)
    perceiver = PatchDualPathologyPerceiver(n_pathologies=K, d_ts=backbone.d_representation, d_latent=256, n_heads=4, dropout=0.2)
    teacher = TeacherModel(backbone, cxr, perceiver, head_hidden=128, head_dropout=0.2, cxr_return_patches=True, d_img=cxr.d_out,
                           use_aux_cxr=False, patch_dual_pathology_mode=True)
    return teacher.to(device)


def usable_cores() -> int:
    """Cores this process may really use: the affinity mask capped by the cgroup CPU quota (the GPU box hands a 1-GPU job a
    16-core share of a 256-thread host; running 256 torch threads on it oversubscribes ~16x)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period) + 0.5)))
    except Exception:
        pass
    return max(1, min(n, int(os.environ.get("MEDP_CPU_CORES", "16"))))


def cpu_baseline(teacher, ccfg, K, batch_cpu, target_seconds=15.0):
    """The CPU oracle (oracle/step_ref.py) on the host cores, same step, bounded sample."""
    from oracle import duett_ref, step_ref, vit_ref, optim_ref
    cores = usable_cores()
    torch.set_num_threads(cores)
    sd = {k: v.detach().float().cpu().clone() for k, v in teacher.state_dict().items()}
    dcfg = duett_ref.DuettCfg(d_static_num=ccfg.d_static, d_time_series_num=ccfg.n_vars, n_timesteps=ccfg.n_timesteps)
    vcfg = vit_ref.VitCfg()
    lrs = optim_ref.group_lrs(8e-5)
    lr_of = lambda name: lrs[optim_ref.param_group_of(name)] * 1e-4
    state = {"step": 0, "m": {}, "v": {}}
    n = batch_cpu["y"].shape[0]
    t0 = time.perf_counter()
    step_ref.teacher_step(sd, dcfg, vcfg, batch_cpu, state, lr_of)          # warm-up (thread pools, allocator)
    warm = time.perf_counter() - t0
    steps = max(1, min(8, int((target_seconds - warm) / max(warm, 1e-3))))
    t0 = time.perf_counter()
    for _ in range(steps):
        step_ref.teacher_step(sd, dcfg, vcfg, batch_cpu, state, lr_of)
    dt = time.perf_counter() - t0
    return {"value": n * steps / dt, "unit": "samples/s", "cores": cores, "kind": "port",
            "sample": f"{steps} teacher steps of batch {n} (same shapes: 224x224 CXR, T={ccfg.n_timesteps}, V={ccfg.n_vars}) "
                      f"through the fp32 CPU oracle, torch.set_num_threads({cores}), {dt:.1f} s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=64, help="per-GPU batch (weak scaling, accelerate semantics)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pipeline", action="store_true",
                    help="graph mode: run the frozen CXR encoder inside its own batch's step instead of one batch ahead")
    ap.add_argument("--host-batch", action="store_true", help="keep batches on the host: PCIe-inclusive rate (never `value`)")
    ap.add_argument("--eager", action="store_true", help="run the step eagerly from Python (engine.py) instead of replaying the captured HIP graph")
    ap.add_argument("--unfreeze-cxr", action="store_true",
                    help="train the CXR encoder too (run.py --unfreeze_cxr; SURVEY 8f1): a side measurement, never the contract line")
    ap.add_argument("--stress", action="store_true",
                    help="BASELINE.json configs[4] shapes per GPU (CXR 512x512, T=256, F=96, batch 32 = 256 / 8 GPUs) instead of the "
                         "metric's configs[2]; a side measurement, never the contract line")
    args = ap.parse_args()

    from multimodal_edema_prediction_amd import abi, dp, engine
    from multimodal_edema_prediction_amd.build import build
    from multimodal_edema_prediction_amd.cohort import CohortCfg, make_batch
    from multimodal_edema_prediction_amd.losses_duett import DualPathologyLoss
    from multimodal_edema_prediction_amd.optim import FusedAdamW, make_param_groups, make_scheduler

    rank, local, world = dp.init_distributed()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if rank == 0:
        build(verbose=False)
    if world > 1:
        torch.distributed.barrier()
    abi.require_gpu()
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)

    T, V, DS, K, B = 96, 48, 8, 7, args.batch
    img = 224
    if args.stress:
        T, V, img = 256, 96, 512
        B = 32 if args.batch == 64 else args.batch
    ccfg = CohortCfg(n_timesteps=T, n_vars=V, d_static=DS, image_size=img, n_labels=K, seed=1234)
    teacher = build_teacher(T, V, DS, K, device, freeze_cxr=not args.unfreeze_cxr)
    dp.broadcast_parameters(teacher)
    loss_fn = DualPathologyLoss(torch.ones(K), None, 0.5, 0.5, 1.0).to(device)
    opt = FusedAdamW(make_param_groups(teacher, 8e-5), weight_decay=5e-2)
    sched = make_scheduler(opt, total_steps=max(args.steps + args.warmup, 1000), lr=8e-5)
    reducer = dp.GradAllReducer([p for p in teacher.parameters() if p.requires_grad]).attach(opt) if (world > 1 and args.eager) else None

    # a small pool of distinct synthetic batches; rank r takes items r, r+N, ... of each global batch (§8e)
    n_pool = 4
    pool = []
    for i in range(n_pool):
        bt = make_batch(ccfg, start=i * B * world + rank, batch_size=B, mode="teacher", stride=world)
        pool.append(bt if args.host_batch else engine._move_lists(bt, device))
    torch.cuda.synchronize()

    if args.eager:
        def step(i):
            out = engine.train_teacher_dual_pathology_batch(pool[i % n_pool], teacher, loss_fn, opt, device)
            sched.step()
            return out
    else:
        from multimodal_edema_prediction_amd.graph_step import GraphedTeacherStep
        force_pg = os.environ.get("MEDP_FORCE_PG") == "1"       # rehearsal of the N>1 code path (RCCL group + split graphs) on one GPU
        if force_pg and not torch.distributed.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
            torch.distributed.init_process_group("nccl", rank=0, world_size=1)
        if not args.host_batch:       # resident batches: stack the per-sample tuples once instead of on every step
            pool_g = [dict(b, x_ts=torch.stack(tuple(b["x_ts"])), x_static=torch.stack(tuple(b["x_static"])),
                           bin_ends=torch.stack(tuple(b["bin_ends"]))) for b in pool]
        else:
            # host batches as a DataLoader with pin_memory=True hands them over: pinned pages, per-sample tuples stacked by the collate
            pool_g = [dict(b, x_ts=torch.stack(tuple(b["x_ts"])).pin_memory(), x_static=torch.stack(tuple(b["x_static"])).pin_memory(),
                           bin_ends=torch.stack(tuple(b["bin_ends"])).pin_memory(), pixel_values=b["pixel_values"].pin_memory(),
                           y_multi=b["y_multi"].float().pin_memory(), y_multi_mask=b["y_multi_mask"].float().pin_memory()) for b in pool]
        gstep = GraphedTeacherStep(teacher, loss_fn, opt, pool[0], device, world=2 if force_pg else world,
                                   pipeline_cxr=not (args.no_pipeline or args.unfreeze_cxr))
        if force_pg:
            gstep.world = 1
            _ar = gstep._allreduce
            gstep._allreduce = lambda: torch.distributed.all_reduce(gstep.flat_grad, op=torch.distributed.ReduceOp.AVG)

        # One host read of the loss per step, like the reference's per-step logging — of the PREVIOUS step: the loss is copied
        # to pinned memory behind an event, so the host enqueues replay k+1 while the GPU still runs replay k instead of
        # idling the GPU for a launch latency every step (0.14 ms of 5.7).  Every step still runs to completion inside the
        # timed region (the final torch.cuda.synchronize()).
        _pin = [torch.empty((), dtype=torch.float32, pin_memory=True) for _ in range(2)]
        _ev = [torch.cuda.Event() for _ in range(2)]
        _state = {"n": 0, "loss": float("nan")}

        def step(i):
            out = gstep.step(pool_g[i % n_pool], pool_g[(i + 1) % n_pool],      # (batch to train on, batch the next call will bring,
                             pool_g[(i + 2) % n_pool] if args.host_batch else None)   #  host batches: the one after, staged ahead)
            sched.step()
            k = _state["n"]
            _pin[k % 2].copy_(out["loss"], non_blocking=True)
            _ev[k % 2].record()
            if k > 0:
                _ev[(k - 1) % 2].synchronize()
                _state["loss"] = float(_pin[(k - 1) % 2])
            _state["n"] = k + 1
            return {"loss": _state["loss"]}

    import warnings
    warnings.filterwarnings("ignore", message=".*lr_scheduler.step.*")
    last = None
    for i in range(args.warmup):
        last = step(i)
    L = abi.lib()
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    if args.eager:
        L.medp_gemm_profile_enable(1)
    t0 = time.perf_counter()
    for i in range(args.steps):
        last = step(args.warmup + i)
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    dt = time.perf_counter() - t0
    if not args.eager:
        last = {"loss": float(gstep.out["loss"].item())}          # the final step's loss (steps report the previous one)
    if not args.eager:
        # Graph replay: HIP events cannot be read back out of a replayed hipGraph on this ROCm (hipEventElapsedTime ->
        # "invalid resource handle"), so the dominant kernel is timed right after the timed region, same process, same
        # stream, same buffers and clocks: 5 eager passes of the CXR encoder = 240 launches of exactly the replayed kernels.
        L.medp_gemm_profile_enable(1)
        with torch.no_grad():
            for i in range(5):
                teacher.cxr.forward_bf16(gstep.pixels)
        torch.cuda.synchronize()
    ms, n_l, fl = ctypes.c_double(), ctypes.c_longlong(), ctypes.c_double()
    abi.check(L.medp_gemm_profile_collect(ctypes.byref(ms), ctypes.byref(n_l), ctypes.byref(fl)), "gemm_profile_collect")
    L.medp_gemm_profile_enable(0)
    if world > 1:
        tt = torch.tensor([dt], device=device, dtype=torch.float64)
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        dt = float(tt.item())
    if rank != 0:
        if world > 1:
            torch.distributed.destroy_process_group()
        return

    value = world * B * args.steps / dt
    achieved = fl.value / (ms.value * 1e-3) / 1e12 if ms.value > 0 else 0.0
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath):
        try:
            traffic = json.load(open(tpath)).get("vit_gemm_hbm_bytes_per_launch")
        except Exception:
            traffic = None
    res = {
        "metric": "multimodal train samples/sec", "value": round(value, 2), "unit": "samples/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
        "config": {"workload": ("SIDE MEASUREMENT (--unfreeze_cxr, SURVEY 8f1): configs[2] shapes with the CXR encoder TRAINED as well, "
                                f"batch {B} per GPU, captured graph, no encoder pipelining" if args.unfreeze_cxr else
                                "BASELINE.json configs[4] shapes (STRESS side measurement, not the metric's configuration): full multimodal "
                                f"teacher, CXR {img}x{img} ViT-B/14 + DuETT T={T}/F={V}, batch {B} per GPU" if args.stress else
                                "BASELINE.json configs[2]: full multimodal teacher (main_train_teacher_duett, perceiver_type=dual_patch, "
                                "--freeze_duett, frozen CXR): CXR 224x224 ViT-B/14 + DuETT T=96/F=48, batch 64 per GPU, bf16 MFMA / fp32 "
                                "accumulate, random-init weights, synthetic cohort seed 1234, perceiver dropout 0.2 ON"),
                   "per_gpu_batch": B, "global_batch": B * world, "parallelism": f"dp{world}",
                   "gflop_per_sample": None if (args.stress or args.unfreeze_cxr) else GFLOP_PER_SAMPLE,
                   "step_mfma_fraction_of_peak": None if (args.stress or args.unfreeze_cxr) else round(value * GFLOP_PER_SAMPLE / 1e3 / (PEAK_BF16_TFLOPS * world), 4),
                   "last_loss": round(float(last["loss"]), 5), "batch_location": "host" if args.host_batch else "hbm",
                   "execution": "eager (engine.py from Python)" if args.eager else (
                       "captured HIP graph replay (graph_step.py), two-stream step" if args.no_pipeline else
                       "captured HIP graph replay (graph_step.py): two-stream step + frozen CXR encoder of batch k+1 run beside the "
                       "step of batch k (one encoder forward, one fusion fwd/bwd and one update per replay; 4 distinct batches rotate)")},
        "roofline": {"bound": "mfma", "kernel": "gemm_bf16_nt_v6_kernel<1> / gemm_bf16_nt_v7_kernel<1> (CXR-encoder block GEMMs: proj, fc2 / qkv, fc1 — one K-loop: 256x256x64 tiles, 8 waves ping-pong, 128x64 per wave; v7 = persistent over the tile list)",
                     "achieved": round(achieved, 1), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": round(achieved / PEAK_BF16_TFLOPS, 4),
                     "traffic": traffic, "launches": int(n_l.value),
                     "note": "the matrix peak is the nominal roofline; L2 counters (profiles/r01_pmc_tcc_gemm_v6_v7.txt) show the L2 channels 79 % busy "
                             "at 8.1 TB/s of LDS staging traffic: at 128 FLOP per staged byte the binding ceiling is ~1.05 PFLOP/s",
                     "avg_launch_us": round(ms.value * 1e3 / max(n_l.value, 1), 2),
                     "algorithmic_flops_per_launch": round(fl.value / max(n_l.value, 1), 1)},
    }
    if world == 1 and not args.no_cpu_baseline and not args.stress and not args.unfreeze_cxr:
        nb = 4
        cb = make_batch(ccfg, start=10_000, batch_size=nb, mode="teacher")
        res["cpu_baseline"] = cpu_baseline(teacher, ccfg, K, cb)
    else:
        res["cpu_baseline"] = None
    print(json.dumps(res), flush=True)
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
