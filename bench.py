#!/usr/bin/env python3
"""Headline benchmark: multimodal training-step throughput on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config teacher|student|probe] [--strong]

--config teacher (default; BASELINE.json configs[2], the configuration the metric is quoted on): full `dual_patch` teacher —
    CXR 224x224 through a frozen ViT-B/14 + frozen DuETT over T=96 / V=48 vitals + trainable pathology-query fusion head, batch
    64 per GPU, DualPathologyLoss, AdamW with the reference's LR groups and warm-up/cosine schedule
    (`train_teacher_dual_pathology_batch`, reference training_duett/engine.py:135-190).
--config student (configs[3]): teacher->student distillation, `train_student_batch` (engine.py:270-301): frozen teacher forward
    under no-grad + DuETT student trained end to end (BatchNorm batch statistics) + StudentKDLoss, 35.6 MB gradient exchange.
--config probe (configs[1]): CXR-encoder linear probe (cxr_linear_training.ipynb:396-437), head-only training.

One step = the whole step function: host->device transfer of the batch, `feats_to_input` (device kernel), forward of every
encoder, loss, backward of every trainable parameter, gradient all-reduce when N > 1, optimiser step, scheduler step.
`value` is the PCIe-INCLUSIVE rate ("around the full step incl. H2D of the batch"): batches sit in pinned host
memory, as a DataLoader(pin_memory=True) hands them over, and are staged one call ahead on a copy stream.  The rate with the
batches already resident in HBM is measured right after the timed region and reported as `config.resident_batch_samples_per_s`.

N > 1: `python bench.py --gpus N` starts `python -m torch.distributed.run --nproc-per-node N bench.py ...` as a CHILD process
before any GPU call and relays rank 0's JSON line; under torch.distributed.run (RANK set) it is one rank per GPU over RCCL.
Weak scaling (default, accelerate's semantics: 64 samples per GPU) or --strong (global batch 64 sharded: 64/N per GPU).

Prints ONE JSON line (rank 0) with the driver's contract plus `roofline` (the dominant kernel = the CXR-encoder block GEMMs,
timed IN the timed region by in-kernel launch clocks, and isolated with HIP events; the HBM-bound DuETT / norm kernels against
8 TB/s) and `cpu_baseline` (the CPU oracle timed on the host cores, N=1 only).
"""
from __future__ import annotations

import argparse
import ctypes
import json
import os

import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0          # dense bf16 MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md
PEAK_HBM_GBS = 8000.0              # HBM3E, same guide
GFLOP_PER_SAMPLE = {"teacher": 48.25,     # ViT 46.32 + DuETT fwd 1.04 + 3 x fusion 0.294
                    "student": 50.78,     # teacher fwd 47.66 + 3 x DuETT 1.04
                    "probe": 46.32}       # ViT forward


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", choices=("teacher", "student", "probe"), default="teacher")
    ap.add_argument("--batch", type=int, default=64, help="batch: per GPU (weak scaling, accelerate semantics) or global (--strong)")
    ap.add_argument("--strong", action="store_true", help="strong scaling: the global batch (--batch, 64) is sharded, 64/N per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pipeline", action="store_true",
                    help="graph mode: run the frozen part inside its own batch's step instead of one batch ahead")
    ap.add_argument("--resident", action="store_true", help="A/B: time the step on HBM-resident batches (a side measurement; the line says so)")
    ap.add_argument("--eager", action="store_true", help="run the step eagerly from Python (engine.py) instead of replaying captured HIP graphs")
    ap.add_argument("--unfreeze-cxr", action="store_true",
                    help="train the CXR encoder too (run.py --unfreeze_cxr; SURVEY 8f1): a side measurement, never the contract line")
    ap.add_argument("--stress", action="store_true",
                    help="BASELINE.json configs[4] shapes per GPU (CXR 512x512, T=256, F=96, batch 32 = 256 / 8 GPUs); a side measurement")
    ap.add_argument("--no-hbm-table", action="store_true", help="skip the per-kernel HBM table of the roofline object")
    ap.add_argument("--launch-check", action="store_true",
                    help="no GPU work: every rank joins a gloo group, takes part in the barrier / MAX-over-ranks timing exchange and rank 0 "
                         "prints a JSON line - checks the self-launch, rendezvous and relay path on a CPU box (tests/test_bench_launch.py)")
    return ap.parse_args()


def self_launch(args) -> int:
    """`python bench.py --gpus N` (N > 1, no RANK): become the launcher.  Nothing here touches the GPU."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for out in proc.stdout:
        out = out.rstrip("\n")
        if out.startswith("{") and '"metric"' in out:
            line = out
        else:
            print(out, file=sys.stderr, flush=True)
    rc = proc.wait()
    if line is not None:
        print(line, flush=True)
    return rc if rc != 0 else (0 if line is not None else 1)


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


def build_teacher(T, V, DS, K, device, seed=0, freeze_cxr=True, freeze_all=False):
    import torch
    from multimodal_edema_prediction_amd.main_architecture_duett import (CXREncoder, PatchDualPathologyPerceiver, TeacherModel,
                                                                           load_duett_backbone)
    torch.manual_seed(seed)
    backbone = load_duett_backbone("synthetic", d_static_num=DS, d_time_series_num=V, n_timesteps=T, freeze=True)   # --freeze_duett
    cxr = CXREncoder("synthetic", freeze=freeze_cxr, return_patches=True)
    perceiver = PatchDualPathologyPerceiver(n_pathologies=K, d_ts=backbone.d_representation, d_latent=256, n_heads=4, dropout=0.2)
    teacher = TeacherModel(backbone, cxr, perceiver, head_hidden=128, head_dropout=0.2, cxr_return_patches=True, d_img=cxr.d_out,
                           use_aux_cxr=False, patch_dual_pathology_mode=True)
    if freeze_all:                                      # the KD teacher (trainer.py:856-865)
        for p in teacher.parameters():
            p.requires_grad = False
        teacher.eval()
    return teacher.to(device)


def build_student(T, V, DS, device, seed=1):
    import torch
    from multimodal_edema_prediction_amd.main_architecture_duett import StudentModel, load_duett_backbone
    torch.manual_seed(seed)
    backbone = load_duett_backbone("synthetic", d_static_num=DS, d_time_series_num=V, n_timesteps=T, freeze=False)   # trainer.py:868-881
    return StudentModel(backbone, pool="mean", head_hidden=128, head_dropout=0.1).to(device)


def cpu_baseline(config, model, teacher, ccfg, K, batch_cpu, target_seconds=15.0):
    """The CPU oracle (oracle/step_ref.py) on the host cores, same step, bounded sample."""
    import torch
    from oracle import duett_ref, optim_ref, step_ref, vit_ref
    cores = usable_cores()
    torch.set_num_threads(cores)
    dcfg = duett_ref.DuettCfg(d_static_num=ccfg.d_static, d_time_series_num=ccfg.n_vars, n_timesteps=ccfg.n_timesteps)
    vcfg = vit_ref.VitCfg()
    lrs = optim_ref.group_lrs(8e-5)
    lr_of = lambda name: lrs[optim_ref.param_group_of(name)] * 1e-4
    state = {"step": 0, "m": {}, "v": {}}
    n = batch_cpu["y"].shape[0]
    tsd = {k: v.detach().float().cpu().clone() for k, v in teacher.state_dict().items()}
    if config == "teacher":
        run = lambda: step_ref.teacher_step(tsd, dcfg, vcfg, batch_cpu, state, lr_of)
        what = "teacher steps"
    elif config == "student":
        ssd = {k: v.detach().float().cpu().clone() for k, v in model.state_dict().items()}

        def run():
            with torch.no_grad():
                z_t = step_ref.teacher_forward(tsd, dcfg, vcfg, batch_cpu)["main_logit"]
            step_ref.student_step(ssd, dcfg, batch_cpu, z_t, state, lambda name: 8e-5 * 1e-4)
        what = "student-KD steps (frozen teacher forward + student forward/backward/AdamW)"
    else:
        vsd = {k[len("encoder.backbone."):]: v.detach().float().cpu().clone() for k, v in model.state_dict().items()
               if k.startswith("encoder.backbone.")}
        W = model.classifier[1].weight.detach().float().cpu().clone().requires_grad_(True)
        bb = model.classifier[1].bias.detach().float().cpu().clone().requires_grad_(True)
        from oracle import losses_ref

        def run():
            with torch.no_grad():
                cls, _ = vit_ref.vit_forward(vsd, vcfg, batch_cpu["pixel_values"])
            loss = losses_ref.masked_bce_global(torch.nn.functional.linear(cls, W, bb), batch_cpu["y_multi"], batch_cpu["y_multi_mask"])
            loss.backward()
        what = "linear-probe steps (frozen ViT forward + head forward/backward)"
    t0 = time.perf_counter()
    run()                                                            # warm-up (thread pools, allocator)
    warm = time.perf_counter() - t0
    steps = max(1, min(64, int((target_seconds - warm) / max(warm, 1e-3))))      # ~15 s of CPU work
    t0 = time.perf_counter()
    for _ in range(steps):
        run()
    dt = time.perf_counter() - t0
    return {"value": round(n * steps / dt, 3), "unit": "samples/s", "cores": cores, "kind": "port",
            "sample": f"{steps} {what} of batch {n} (same shapes: 224x224 CXR, T={ccfg.n_timesteps}, V={ccfg.n_vars}) "
                      f"through the fp32 CPU oracle, torch.set_num_threads({cores}), {dt:.1f} s"}


def hbm_kernel_table(B, T, V, device, duett=None):
    """The HBM-bound kernels SURVEY.md §8(d) asks to be reported one by one: algorithmic bytes (each operand read once, each result
    written once), live HIP-event time of isolated launches at the step's shapes, fraction of 8 TB/s."""
    import torch
    from multimodal_edema_prediction_amd import functional as Fn
    from multimodal_edema_prediction_amd.abi import check, lib, ptr, stream

    def timeit(fn, iters=50, warm=5):
        for _ in range(warm):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / iters * 1e3       # us

    rows = []
    E, T1, V1 = 24, T + 1, V + 1
    psi = torch.randn(B, T1, V1, E, device=device)
    out = torch.empty(B, V1, T1, E, device=device)
    us = timeit(lambda: check(lib().medp_axis_swap(ptr(psi), ptr(out), B, T1, V1, E, stream()), "axis_swap"))
    rows.append(("axis_swap_kernel (psi <-> event view, fp32)", 2 * psi.numel() * 4, us))
    for name, rows_n, D in (("scalenorm (time view rows, fp32 -> bf16)", B * T1, V1 * E), ("scalenorm (event view rows, fp32 -> bf16)", B * V1, T1 * E)):
        x = torch.randn(rows_n, D, device=device)
        g = torch.ones(1, device=device)
        us = timeit(lambda: Fn.scalenorm(x, g))
        rows.append((name, x.numel() * 6, us))
    M = B * 257
    x = torch.randn(M, 768, device=device)
    w, b = torch.ones(768, device=device), torch.zeros(768, device=device)
    us = timeit(lambda: Fn.layernorm(x, w, b, 1e-6))
    rows.append(("layernorm_fwd_reg_kernel (ViT tokens, fp32 -> bf16)", x.numel() * 6, us))
    if duett is not None:                      # the fused DuETT front end (csrc/duett.hip), at the step's shapes
        psi_elems = B * T1 * V1 * E
        xs_static, xs_ts, xs_times = torch.randn(B, 8, device=device), torch.zeros(B, T, 2 * V + 1, device=device), torch.rand(B, T, device=device)
        xs_ts[:, :, :V] = torch.randn(B, T, V, device=device)
        xs_ts[:, :, V:2 * V] = torch.randint(0, 4, (B, T, V), device=device).float()
        w = duett._prepare()[0]
        xe, h = torch.empty(psi_elems, device=device), torch.empty(psi_elems, device=device, dtype=torch.bfloat16)
        temb, tab = torch.empty(psi_elems, device=device), torch.empty(B * E, device=device)
        emb = lambda st: check(lib().medp_duett_embed_fwd(ctypes.byref(w), ptr(xs_static), ptr(xs_ts), ptr(xs_times), B, T, ptr(xe), ptr(h),
                                                          ptr(temb), None, ptr(tab), st, stream()), "duett_embed_fwd")
        us = timeit(lambda: emb(1))
        rows.append(("tab_encoder + psi_embed_event_kernel (psi build + swap + event embedding + ScaleNorm -> fp32 + bf16)",
                     xs_ts.numel() * 4 + V1 * T1 * E * 4 + psi_elems * 6, us))
        us = timeit(lambda: emb(2))
        rows.append(("time_embed_kernel (cve 1 -> 34 -> 1176, REP row appended, fp32)", xs_times.numel() * 4 + psi_elems * 4, us))
        g1, rn = torch.ones(1, device=device), torch.rand(B * V1, device=device)
        us = timeit(lambda: check(lib().medp_duett_swap_add_norm(ptr(xe), ptr(rn), ptr(g1), ptr(temb), T1 * V1 * E, ptr(g1), 1e-12, ptr(psi), ptr(h),
                                                                 B, V1, T1, E, stream()), "swap_add_norm"))
        rows.append(("swap_add_norm_kernel (event -> time view + time embedding + ScaleNorm -> fp32 + bf16)", psi_elems * (4 + 4 + 4 + 2), us))
    return [{"kernel": n, "algorithmic_bytes": int(by), "us": round(us, 2), "GBps": round(by / us * 1e-3, 1),
             "frac_of_hbm_peak": round(by / us * 1e-3 / PEAK_HBM_GBS, 4)} for n, by, us in rows]


def main():
    args = parse_args()
    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(self_launch(args))

    import torch
    if args.launch_check:
        from multimodal_edema_prediction_amd import dp
        rank, local, world = dp.init_distributed("gloo")
        assert world == args.gpus, (world, args.gpus)
        if world > 1:
            torch.distributed.barrier()
        tt = torch.tensor([float(rank + 1)], dtype=torch.float64)
        if world > 1:
            torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        if rank == 0:
            print(json.dumps({"metric": "launch-check", "value": float(tt.item()), "n_gpus": world, "scaling": "strong" if args.strong else "weak"}), flush=True)
        if torch.distributed.is_initialized():
            torch.distributed.destroy_process_group()
        return
    from multimodal_edema_prediction_amd import abi, dp, engine
    from multimodal_edema_prediction_amd.build import build
    from multimodal_edema_prediction_amd.cohort import CohortCfg, make_batch
    from multimodal_edema_prediction_amd.losses_duett import DualPathologyLoss, StudentKDLoss
    from multimodal_edema_prediction_amd.optim import FusedAdamW, make_param_groups, make_scheduler

    rank, local, world = dp.init_distributed()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if rank == 0:
        build(verbose=False)
    if world > 1:
        torch.distributed.barrier()
    abi.require_gpu()
    if os.environ.get("MEDP_DIST_BACKEND") == "gloo":          # one-GPU rehearsal of N > 1: every rank on the devices there are
        local = local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)

    T, V, DS, K = 96, 48, 8, 7
    img = 224
    if args.strong:
        if args.batch % world:
            raise SystemExit(f"--strong: the global batch {args.batch} must divide by the number of GPUs {world}")
        B = args.batch // world
    else:
        B = args.batch
    if args.stress:
        T, V, img = 256, 96, 512
        B = 32 if args.batch == 64 else B
    cfg = args.config
    if cfg == "probe":
        # nothing runs beside the encoder here: fc1's ragged last rows as their own launch (3 full rounds of 256 workgroups instead of 4 of
        # 200) is the faster arrangement — the opposite of the two-branch steps (csrc/gemm_bf16.hip; read once, at the first GEMM)
        os.environ.setdefault("MEDP_GEMM_RAGGED", "1")
    ccfg = CohortCfg(n_timesteps=T, n_vars=V, d_static=DS, image_size=img, n_labels=K, seed=1234)
    side = args.stress or args.unfreeze_cxr or args.resident or args.eager or args.no_pipeline
    pipeline = not (args.no_pipeline or args.unfreeze_cxr)

    # ---- models, loss, optimiser -----------------------------------------------------------------------------------------------
    teacher = build_teacher(T, V, DS, K, device, freeze_cxr=not args.unfreeze_cxr, freeze_all=(cfg == "student"))
    student = probe = None
    if cfg == "teacher":
        trainable = teacher
        loss_fn = DualPathologyLoss(torch.ones(K), None, 0.5, 0.5, 1.0).to(device)
        opt = FusedAdamW(make_param_groups(teacher, 8e-5), weight_decay=5e-2)
    elif cfg == "student":
        student = trainable = build_student(T, V, DS, device)
        loss_fn = StudentKDLoss("vanilla_kl", 4.0, 0.5)
        opt = FusedAdamW(make_param_groups(student, 8e-5), weight_decay=5e-2)            # trainer.py:897-902
    else:
        from multimodal_edema_prediction_amd.linear_probe import PixelPrefetcher, RadDinoClassifier, masked_bce_with_logits_loss
        _probe_state = {}
        torch.manual_seed(0)
        probe = trainable = RadDinoClassifier("synthetic", num_classes=K, dropout=0.1).to(device)
        probe.train()
        loss_fn = masked_bce_with_logits_loss
        opt = FusedAdamW([p for p in probe.parameters() if p.requires_grad], lr=1e-4, weight_decay=1e-4)     # ipynb :519
    dp.broadcast_parameters(trainable)
    if cfg == "student":
        dp.broadcast_parameters(teacher)
    sched = make_scheduler(opt, total_steps=max(args.steps + args.warmup, 1000), lr=8e-5) if cfg != "probe" else None

    # ---- batches: a small pool of distinct synthetic batches; rank r takes items r, r+N, ... of each global batch (§8e) -------------
    n_pool = 4
    host_pool, dev_pool = [], []
    for i in range(n_pool):
        bt = make_batch(ccfg, start=i * B * world + rank, batch_size=B, mode="teacher", stride=world)
        # host batches as a DataLoader(pin_memory=True) hands them over: pinned pages, per-sample tuples stacked by the collate
        hb = dict(bt, x_ts=torch.stack(tuple(bt["x_ts"])).pin_memory(), x_static=torch.stack(tuple(bt["x_static"])).pin_memory(),
                  bin_ends=torch.stack(tuple(bt["bin_ends"])).pin_memory(), pixel_values=bt["pixel_values"].pin_memory(),
                  y=bt["y"].float().pin_memory(), y_multi=bt["y_multi"].float().pin_memory(), y_multi_mask=bt["y_multi_mask"].float().pin_memory())
        host_pool.append(hb)
        dev_pool.append({k: v.to(device) for k, v in hb.items()})
    torch.cuda.synchronize()
    L = abi.lib()

    # ---- the step ---------------------------------------------------------------------------------------------------------------
    gstep = None
    force_pg = os.environ.get("MEDP_FORCE_PG") == "1"       # rehearsal of the N > 1 arrangement (RCCL group + split graphs) on one GPU
    if force_pg and not torch.distributed.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
        torch.distributed.init_process_group("nccl", rank=0, world_size=1)
    reducer = None
    want_graph = not (args.eager or cfg == "probe")
    probe_graph = None
    graph_error = None
    if want_graph:
        from multimodal_edema_prediction_amd.graph_step import GraphedStudentStep, GraphedTeacherStep
        os.environ.setdefault("MEDP_PHASE_CHECK", "1")          # report which hardware-queue phase the captured step sits in (graph_step._setup)
        arm = lambda: abi.check(L.medp_gemm_profile_enable(2), "gemm_profile_enable")     # launch clocks ride in the captured GEMMs
        disarm = lambda: L.medp_gemm_profile_enable(0)                                       # ... and only in those (not in later eager launches)
        try:
            if cfg == "teacher":
                gstep = GraphedTeacherStep(teacher, loss_fn, opt, dev_pool[0], device, world=world, split=force_pg, pipeline_cxr=pipeline,
                                           before_capture=arm, after_capture=disarm)
            else:
                gstep = GraphedStudentStep(student, teacher, loss_fn, opt, dev_pool[0], device, world=world, split=force_pg,
                                           pipeline_teacher=pipeline, before_capture=arm, after_capture=disarm)
            gstep.force_collective = force_pg
        except Exception as e:                                 # never seen on one GPU; N > 1 has not run on hardware before the driver's run
            import traceback
            graph_error = f"{type(e).__name__}: {e}"
            print(f"[bench rank {rank}] captured-graph step failed to build, falling back to the eager engine step:\n" + traceback.format_exc(),
                  file=sys.stderr, flush=True)
            gstep = None
        L.medp_gemm_profile_enable(0)
        if world > 1:                                          # every rank takes the same path: fall back everywhere if any rank failed
            okt = torch.tensor([0 if gstep is None else 1], device=device)
            torch.distributed.all_reduce(okt, op=torch.distributed.ReduceOp.MIN)
            if int(okt.item()) == 0:
                gstep = None
                graph_error = graph_error or "another rank failed to build the captured step"
    if gstep is None:
        if world > 1:
            reducer = dp.GradAllReducer([p for p in trainable.parameters() if p.requires_grad]).attach(opt)
        if cfg == "probe" and not args.eager and world == 1:      # the probe step as one captured graph (N > 1: eager + hooked all-reduce)
            from multimodal_edema_prediction_amd.graph_step import GraphedProbeStep
            _probe_state["graph"] = GraphedProbeStep(probe, loss_fn, opt, dev_pool[0]["pixel_values"], dev_pool[0]["y_multi"],
                                                     dev_pool[0]["y_multi_mask"], device,
                                                     before_capture=lambda: abi.check(L.medp_gemm_profile_enable(2), "gemm_profile_enable"))
            L.medp_gemm_profile_enable(0)
            probe_graph = _probe_state["graph"]

        def as_lists(b):
            n = b["x_ts"].shape[0]
            return dict(b, x_ts=tuple(b["x_ts"][i] for i in range(n)), x_static=tuple(b["x_static"][i] for i in range(n)),
                        bin_ends=tuple(b["bin_ends"][i] for i in range(n)))

        def run_step(pool, i):
            b = pool[i % n_pool]
            if cfg == "teacher":
                out = engine.train_teacher_dual_pathology_batch(as_lists(b), teacher, loss_fn, opt, device)
            elif cfg == "student":
                out = engine.train_student_batch(as_lists(b), as_lists(b), student, teacher, loss_fn, opt, device)
            else:
                if b["pixel_values"].is_cuda:
                    px = b["pixel_values"]
                else:                                   # host batches: the next batch's pixels are staged on a copy stream beside this step
                    pre = _probe_state.setdefault("pre", PixelPrefetcher(device, b["pixel_values"]))
                    if _probe_state.get("staged_for") != i:
                        pre.n_taken = pre.n_staged     # (a fresh sequence: nothing usable is staged)
                        pre.stage(b["pixel_values"])
                    px = pre.take()
                    pre.stage(pool[(i + 1) % n_pool]["pixel_values"])
                    _probe_state["staged_for"] = i + 1
                if _probe_state.get("graph") is not None:
                    out = _probe_state["graph"].step(px, b["y_multi"], b["y_multi_mask"])
                else:
                    opt.zero_grad()
                    loss = loss_fn(probe(px), b["y_multi"].to(device, non_blocking=True), b["y_multi_mask"].to(device, non_blocking=True))
                    loss.backward()
                    opt.step()
                    out = {"loss": loss.detach()}
            if sched is not None:
                sched.step()
            return out
    else:
        # One host read of the loss per step, like the reference's per-step logging — of the PREVIOUS step: the loss is copied
        # to pinned memory behind an event, so the host enqueues replay k+1 while the GPU still runs replay k instead of
        # idling the GPU for a launch latency every step.  Every step still runs to completion inside the timed region.
        _pin = [torch.empty((), dtype=torch.float32, pin_memory=True) for _ in range(2)]
        _ev = [torch.cuda.Event() for _ in range(2)]
        _state = {"n": 0, "loss": float("nan")}

        def run_step(pool, i):
            host = not pool[0]["pixel_values"].is_cuda
            if cfg == "teacher":
                out = gstep.step(pool[i % n_pool], pool[(i + 1) % n_pool],          # (batch to train on, batch the next call will bring,
                                 pool[(i + 2) % n_pool] if (host and pipeline) else None)   # host batches: the one after, staged ahead)
            else:
                out = gstep.step(pool[i % n_pool], pool[(i + 1) % n_pool], pool[(i + 2) % n_pool] if (host and pipeline) else None)
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

    def timed(pool, steps, warmup, first=0):
        for i in range(warmup):
            run_step(pool, first + i)
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            run_step(pool, first + warmup + i)
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
        dt = time.perf_counter() - t0
        if world > 1:
            tt = torch.tensor([dt], device=device, dtype=torch.float64)
            torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
            dt = float(tt.item())
        return dt

    main_pool = dev_pool if args.resident else host_pool
    if gstep is None:                                   # eager steps: HIP events bracket the GEMM launches inside the timed region
        for i in range(args.warmup):
            run_step(main_pool, i)
        if probe_graph is None:
            L.medp_gemm_profile_enable(1)
        dt = timed(main_pool, args.steps, 0, first=args.warmup)
        last_loss = float(run_step(main_pool, 0)["loss"])
    else:
        dt = timed(main_pool, args.steps, args.warmup)
        last_loss = float(gstep.out["loss"].item())
    # ---- roofline leg 1: the dominant kernel IN the timed region -------------------------------------------------------------
    ms, n_l, fl = ctypes.c_double(), ctypes.c_longlong(), ctypes.c_double()
    in_step = None
    if True:
        abi.check(L.medp_gemm_profile_collect(ctypes.byref(ms), ctypes.byref(n_l), ctypes.byref(fl)), "gemm_profile_collect")
        L.medp_gemm_profile_enable(0)
        if n_l.value > 0 and ms.value > 0:
            in_step = {"achieved": fl.value / (ms.value * 1e-3) / 1e12, "launches": int(n_l.value),
                       "avg_launch_us": ms.value * 1e3 / n_l.value, "flops_per_launch": fl.value / n_l.value}
    # ---- side measurement: the same step on HBM-resident batches -----------------------------------------------------------
    dt_res = None
    if not args.resident and gstep is not None:
        dt_res = timed(dev_pool, args.steps, 2, first=args.warmup + args.steps)
    # ---- roofline leg 2: the same GEMMs alone on the GPU (HIP events around eager launches) -----------------------------------
    isolated = None
    L.medp_gemm_profile_enable(1)
    with torch.no_grad():
        for i in range(3):
            teacher.cxr.forward_bf16(dev_pool[0]["pixel_values"]) if cfg != "probe" else probe.encoder.forward_bf16(dev_pool[0]["pixel_values"])
    torch.cuda.synchronize()
    abi.check(L.medp_gemm_profile_collect(ctypes.byref(ms), ctypes.byref(n_l), ctypes.byref(fl)), "gemm_profile_collect")
    L.medp_gemm_profile_enable(0)
    if n_l.value > 0 and ms.value > 0:
        isolated = {"achieved": round(fl.value / (ms.value * 1e-3) / 1e12, 1), "launches": int(n_l.value),
                    "avg_launch_us": round(ms.value * 1e3 / n_l.value, 2)}
    if rank != 0:
        if torch.distributed.is_initialized():
            torch.distributed.destroy_process_group()
        return

    value = world * B * args.steps / dt
    gps = None if (args.stress or args.unfreeze_cxr) else GFLOP_PER_SAMPLE[cfg]
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath):
        try:
            traffic = json.load(open(tpath)).get("vit_gemm_hbm_bytes_per_launch")
        except Exception:
            traffic = None
    names = {"teacher": "BASELINE.json configs[2]: full multimodal teacher (main_train_teacher_duett, perceiver_type=dual_patch, --freeze_duett, "
                        "frozen CXR): CXR 224x224 ViT-B/14 + DuETT T=96/F=48, bf16 MFMA / fp32 accumulate, random-init weights, synthetic "
                        "cohort seed 1234, perceiver dropout 0.2 ON",
             "student": "BASELINE.json configs[3]: teacher->student distillation (main_train_student_duett): frozen teacher forward (CXR 224x224 "
                        "ViT-B/14 + DuETT + fusion head) + DuETT student T=96/F=48 trained end to end (BatchNorm batch statistics, head dropout "
                        "0.1), StudentKDLoss(T 4, alpha 0.5), random-init weights, synthetic cohort seed 1234",
             "probe": "BASELINE.json configs[1]: CXR-encoder-only linear probe (cxr_linear_training): frozen ViT-B/14 224x224 bf16 -> CLS -> "
                      "Dropout(0.1) -> Linear(768,7), masked BCE, AdamW on the 5,383 head parameters, synthetic images seed 1234"}
    workload = names[cfg] + f"; batch {B} per GPU"
    if args.stress:
        workload = (f"BASELINE.json configs[4] shapes (STRESS side measurement, not the metric's configuration): {cfg} step, CXR {img}x{img} "
                    f"ViT-B/14 + DuETT T={T}/F={V}, batch {B} per GPU")
    if args.unfreeze_cxr:
        workload = "SIDE MEASUREMENT (--unfreeze_cxr, SURVEY 8f1): " + workload + ", CXR encoder TRAINED as well, no encoder pipelining"
    if side and not (args.stress or args.unfreeze_cxr):
        workload = "SIDE MEASUREMENT (" + ", ".join(f for f, on in (("--resident", args.resident), ("--eager", args.eager), ("--no-pipeline", args.no_pipeline)) if on) + "): " + workload
    execution = ("eager (engine.py from Python)" + (f" — FALLBACK, the captured step failed to build: {graph_error}" if graph_error else "")) if gstep is None else (
        ("captured HIP graph replay (graph_step.py)" + (": frozen part of batch k+1 run beside the step of batch k (one frozen forward, one "
         "trainable fwd/bwd and one update per replay; 4 distinct batches rotate)" if pipeline else ", frozen part inside its own step"))
        + ("; N>1: fwd/bwd graph -> RCCL mean all-reduce of the flat gradient arena -> optimiser graph on the main stream, the frozen-forward "
           "graph of the next batch beside them on its own stream" if (world > 1 or force_pg) else ""))
    res = {
        "metric": "multimodal train samples/sec", "value": round(value, 2), "unit": "samples/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True,
        "scaling": "strong" if args.strong else "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
        "config": {"workload": workload, "config": cfg, "per_gpu_batch": B, "global_batch": B * world, "parallelism": f"dp{world}",
                   "batch_location": "hbm (resident)" if args.resident else "pinned host memory: H2D + feats_to_input inside the timed step",
                   "resident_batch_samples_per_s": round(world * B * args.steps / dt_res, 2) if dt_res else None,
                   # graph_step's hardware-queue phase check: (pad streams, ms a step loses to the staged pixel copy, ms of that copy alone)
                   "hw_queue_phase": getattr(gstep, "phase_log", None),
                   "gflop_per_sample": gps,
                   "step_mfma_fraction_of_peak": None if gps is None else round(value * gps / 1e3 / (PEAK_BF16_TFLOPS * world), 4),
                   "last_loss": round(last_loss, 5), "execution": execution,
                   "gradient_exchange_bytes": (gstep.arena.bytes_per_step if (gstep is not None and gstep.arena is not None) else
                                               (reducer.bytes_per_step if reducer is not None else 0))},
    }
    if in_step is not None:
        res["roofline"] = {
            "bound": "mfma",
            "kernel": "gemm_bf16_nt_v6_kernel<1> / gemm_bf16_nt_v7_kernel<1> (CXR-encoder block GEMMs: proj, fc2 / qkv, fc1 — one K-loop: "
                      "256x256x64 tiles, 8 waves ping-pong, 128x64 per wave; v7 = persistent over the tile list)",
            "achieved": round(in_step["achieved"], 1), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
            "frac": round(in_step["achieved"] / PEAK_BF16_TFLOPS, 4), "traffic": traffic, "launches": in_step["launches"],
            "avg_launch_us": round(in_step["avg_launch_us"], 2), "algorithmic_flops_per_launch": round(in_step["flops_per_launch"], 1),
            "timing": ("in-kernel launch clocks (first workgroup in / last workgroup out, 100-MHz wall clock) of the LAST replay inside the "
                       "timed region: the GEMMs as they run in the step, beside its other branches" if gstep is not None else
                       "HIP events around each launch, on its stream, inside the timed region"),
            "isolated": isolated,
            "note": "the matrix peak is the nominal roofline; L2 counters (profiles/r01_pmc_tcc_gemm_v6_v7.txt) show the L2 channels 79 % busy "
                    "at 8.1 TB/s of LDS staging traffic: at 128 FLOP per staged byte the binding ceiling is ~1.05 PFLOP/s"}
        if world == 1 and not args.no_hbm_table and not args.stress:
            res["roofline"]["hbm_kernels"] = hbm_kernel_table(B, T, V, device, teacher.duett)
    if world == 1 and not args.no_cpu_baseline and not args.stress and not args.unfreeze_cxr:
        cb = make_batch(ccfg, start=10_000, batch_size=4, mode="teacher")
        res["cpu_baseline"] = cpu_baseline(cfg, trainable, teacher, ccfg, K, cb)
    else:
        res["cpu_baseline"] = None
    print(json.dumps(res), flush=True)
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
