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

One step = the whole step function: `feats_to_input` (device kernel), forward of every encoder, loss, backward of every trainable
parameter, gradient all-reduce when N > 1, optimiser step, scheduler step.
`value` is the rate with the batches RESIDENT IN HBM when the timed region starts (the driver's contract).  The PCIe-INCLUSIVE rate
(batches in pinned host memory, as a DataLoader(pin_memory=True) hands them over, staged one call ahead on a copy
stream) is measured right after the timed region and reported as `config.pcie_inclusive_samples_per_s`; `--host-batches` swaps the two.
`ms_per_step_median`: per-step HIP events over max(steps, 50) further steps.

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
    ap.add_argument("--resident", action="store_true", help="(default since round 3; kept for old command lines) time the step on HBM-resident batches")
    ap.add_argument("--host-batches", action="store_true",
                    help="side measurement: time the PCIe-INCLUSIVE step (batches in pinned host memory, staged one call ahead) as `value`")
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


def shard_plan(args, rank: int, world: int, pool_index: int = 0):
    """(per-GPU batch B, cohort indices this rank trains on in pool batch `pool_index`): SURVEY.md §8(e) — rank r takes items
    r, r + N, ... of every global batch (accelerate's BatchSamplerShard split).  Weak scaling: --batch per GPU; --strong: the global
    --batch is sharded, --batch / N per GPU."""
    if args.strong:
        if args.batch % world:
            raise SystemExit(f"--strong: the global batch {args.batch} must divide by the number of GPUs {world}")
        B = args.batch // world
    else:
        B = args.batch
    if args.stress:
        B = 32 if args.batch == 64 else B
    start = pool_index * B * world + rank
    return B, [start + i * world for i in range(B)]


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


def cpu_baseline(config, model, teacher, ccfg, K, batch_cpu, small_batch_cpu, timed_steps=3, warm_steps=2):
    """The CPU oracle (oracle/step_ref.py) on the host cores — SURVEY.md §8(d): the same step at the same batch, `n` = all cores
    this job may use (2 warm-up + 3 timed steps, ~30 s) and `n` = 1 beside it (one step of a 4-sample batch: a one-core step of
    the full batch would take minutes)."""
    import torch
    from oracle import duett_ref, optim_ref, step_ref, vit_ref
    cores = usable_cores()
    dcfg = duett_ref.DuettCfg(d_static_num=ccfg.d_static, d_time_series_num=ccfg.n_vars, n_timesteps=ccfg.n_timesteps)
    vcfg = vit_ref.VitCfg()
    lrs = optim_ref.group_lrs(8e-5)
    lr_of = lambda name: lrs[optim_ref.param_group_of(name)] * 1e-4
    state = {"step": 0, "m": {}, "v": {}}
    tsd = {k: v.detach().float().cpu().clone() for k, v in teacher.state_dict().items()}
    if config == "teacher":
        run = lambda bt: step_ref.teacher_step(tsd, dcfg, vcfg, bt, state, lr_of)
        what = "teacher steps"
    elif config == "student":
        ssd = {k: v.detach().float().cpu().clone() for k, v in model.state_dict().items()}

        def run(bt):
            with torch.no_grad():
                z_t = step_ref.teacher_forward(tsd, dcfg, vcfg, bt)["main_logit"]
            step_ref.student_step(ssd, dcfg, bt, z_t, state, lambda name: 8e-5 * 1e-4)
        what = "student-KD steps (frozen teacher forward + student forward/backward/AdamW)"
    else:
        vsd = {k[len("encoder.backbone."):]: v.detach().float().cpu().clone() for k, v in model.state_dict().items()
               if k.startswith("encoder.backbone.")}
        W = model.classifier[1].weight.detach().float().cpu().clone().requires_grad_(True)
        bb = model.classifier[1].bias.detach().float().cpu().clone().requires_grad_(True)
        from oracle import losses_ref

        def run(bt):
            with torch.no_grad():
                cls, _ = vit_ref.vit_forward(vsd, vcfg, bt["pixel_values"])
            loss = losses_ref.masked_bce_global(torch.nn.functional.linear(cls, W, bb), bt["y_multi"], bt["y_multi_mask"])
            loss.backward()
        what = "linear-probe steps (frozen ViT forward + head forward/backward)"
    n, n1 = batch_cpu["y"].shape[0], small_batch_cpu["y"].shape[0]
    torch.set_num_threads(cores)
    for _ in range(warm_steps):
        run(batch_cpu)
    t0 = time.perf_counter()
    for _ in range(timed_steps):
        run(batch_cpu)
    dt = time.perf_counter() - t0
    torch.set_num_threads(1)
    t0 = time.perf_counter()
    run(small_batch_cpu)
    dt1 = time.perf_counter() - t0
    torch.set_num_threads(cores)
    return {"value": round(n * timed_steps / dt, 3), "unit": "samples/s", "cores": cores, "kind": "port",
            "sample": f"{warm_steps} warm-up + {timed_steps} timed {what} of batch {n} (the step's own shapes: {ccfg.image_size}x{ccfg.image_size} CXR, "
                      f"T={ccfg.n_timesteps}, V={ccfg.n_vars}) through the fp32 CPU oracle, torch.set_num_threads({cores}), {dt:.1f} s timed",
            "one_core": {"value": round(n1 / dt1, 3), "unit": "samples/s", "cores": 1,
                         "sample": f"1 step of batch {n1}, torch.set_num_threads(1), {dt1:.1f} s"}}


def hbm_kernel_table(B, T, V, device, duett=None):
    """The HBM-bound kernels SURVEY.md §8(d) asks to be reported one by one: algorithmic bytes (each operand read once, each result
    written once) and live HIP-event times of isolated launches at the step's shapes — COLD: the launches rotate over enough operand
    sets that more than 256 MiB (the Infinity Cache) pass between two uses of a buffer, which is what a kernel sees inside the step;
    WARM (one operand set repeated: it fits the Infinity Cache) is reported beside it, labelled.  The fraction of 8 TB/s is the cold one."""
    import torch
    from multimodal_edema_prediction_amd import functional as Fn
    from multimodal_edema_prediction_amd.abi import check, lib, ptr, stream

    def timeit(fn, nsets, iters=48, warm=6):
        for i in range(warm):
            fn(i % nsets)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(iters):
            fn(i % nsets)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / iters * 1e3       # us

    def both(make, run, set_bytes):
        """make(i) -> operand set; run(set); set_bytes = device bytes one set occupies"""
        n = int(max(3, min(16, (300 << 20) // max(set_bytes, 1) + 2)))
        sets = [make(i) for i in range(n)]
        cold = timeit(lambda i: run(sets[i]), n)
        warm_ = timeit(lambda i: run(sets[0]), 1)
        del sets
        return cold, warm_, n

    rows = []
    E, T1, V1 = 24, T + 1, V + 1
    psi_elems = B * T1 * V1 * E
    c, w_, n = both(lambda i: (torch.randn(B, T1, V1, E, device=device), torch.empty(B, V1, T1, E, device=device)),
                    lambda a: check(lib().medp_axis_swap(ptr(a[0]), ptr(a[1]), B, T1, V1, E, stream()), "axis_swap"), psi_elems * 8)
    rows.append(("axis_swap_kernel (psi <-> event view, fp32)", 2 * psi_elems * 4, c, w_, n))
    g1 = torch.ones(1, device=device)
    for name, rows_n, D in (("scalenorm (time view rows, fp32 -> bf16)", B * T1, V1 * E), ("scalenorm (event view rows, fp32 -> bf16)", B * V1, T1 * E)):
        c, w_, n = both(lambda i: torch.randn(rows_n, D, device=device), lambda x: Fn.scalenorm(x, g1), rows_n * D * 6)
        rows.append((name, rows_n * D * 6, c, w_, n))
    M = B * 257
    lw, lb = torch.ones(768, device=device), torch.zeros(768, device=device)
    c, w_, n = both(lambda i: torch.randn(M, 768, device=device), lambda x: Fn.layernorm(x, lw, lb, 1e-6), M * 768 * 6)
    rows.append(("layernorm_fwd_reg_kernel (ViT tokens, fp32 -> bf16)", M * 768 * 6, c, w_, n))
    if duett is not None:                      # the fused DuETT front end (csrc/duett.hip), at the step's shapes
        w = duett._prepare()[0]

        def mk(i):
            xs_ts = torch.zeros(B, T, 2 * V + 1, device=device)
            xs_ts[:, :, :V] = torch.randn(B, T, V, device=device)
            xs_ts[:, :, V:2 * V] = torch.randint(0, 4, (B, T, V), device=device).float()
            return {"st": torch.randn(B, 8, device=device), "ts": xs_ts, "tm": torch.rand(B, T, device=device),
                    "xe": torch.empty(psi_elems, device=device), "h": torch.empty(psi_elems, device=device, dtype=torch.bfloat16),
                    "temb": torch.empty(psi_elems, device=device), "tab": torch.empty(B * E, device=device),
                    "psi": torch.empty(psi_elems, device=device), "rn": torch.rand(B * V1, device=device)}
        emb = lambda a, st: check(lib().medp_duett_embed_fwd(ctypes.byref(w), ptr(a["st"]), ptr(a["ts"]), ptr(a["tm"]), B, T, ptr(a["xe"]), ptr(a["h"]),
                                                             ptr(a["temb"]), None, ptr(a["tab"]), st, stream()), "duett_embed_fwd")
        set_bytes = psi_elems * (4 + 2 + 4 + 4) + B * T * (2 * V + 1) * 4
        c, w_, n = both(mk, lambda a: emb(a, 1), set_bytes)
        rows.append(("tab_encoder + psi_embed_event_kernel (psi build + swap + event embedding + ScaleNorm -> fp32 + bf16)",
                     B * T * (2 * V + 1) * 4 + V1 * T1 * E * 4 + psi_elems * 6, c, w_, n))
        c, w_, n = both(mk, lambda a: emb(a, 2), set_bytes)
        rows.append(("time_embed_kernel (cve 1 -> 34 -> 1176, REP row appended, fp32)", B * T * 4 + psi_elems * 4, c, w_, n))
        c, w_, n = both(mk, lambda a: check(lib().medp_duett_swap_add_norm(ptr(a["xe"]), ptr(a["rn"]), ptr(g1), ptr(a["temb"]), T1 * V1 * E, ptr(g1), 1e-12,
                                                                           ptr(a["psi"]), ptr(a["h"]), B, V1, T1, E, stream()), "swap_add_norm"), set_bytes)
        rows.append(("swap_add_norm_kernel (event -> time view + time embedding + ScaleNorm -> fp32 + bf16)", psi_elems * (4 + 4 + 4 + 2), c, w_, n))
    return [{"kernel": nm, "algorithmic_bytes": int(by), "us": round(c, 2), "us_warm": round(w_, 2),
             "timing": f"cold: isolated launches rotating over {n} operand sets (> 256 MiB between two uses of a buffer); us_warm: one set repeated (Infinity-Cache resident)",
             "GBps": round(by / c * 1e-3, 1), "frac_of_hbm_peak": round(by / c * 1e-3 / PEAK_HBM_GBS, 4)} for nm, by, c, w_, n in rows]


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
        # shard arithmetic of the real run: every rank's cohort indices for the first two pool batches, gathered on rank 0
        Bc, mine = shard_plan(args, rank, world, 0)
        mine = mine + shard_plan(args, rank, world, 1)[1]
        parts = [None] * world
        if world > 1:
            torch.distributed.all_gather_object(parts, mine)
        else:
            parts = [mine]
        if rank == 0:
            flat = sorted(i for p_ in parts for i in p_)
            print(json.dumps({"metric": "launch-check", "value": float(tt.item()), "n_gpus": world, "scaling": "strong" if args.strong else "weak",
                              "per_gpu_batch": Bc, "global_batch": Bc * world,
                              "shards_cover_global_batches_exactly_once": flat == list(range(2 * Bc * world)),
                              "rank0_first_items": parts[0][:3]}), flush=True)
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
    B, _ = shard_plan(args, rank, world)
    if args.stress:
        T, V, img = 256, 96, 512
    cfg = args.config
    if cfg == "probe":
        # nothing runs beside the encoder here: fc1's ragged last rows as their own launch (3 full rounds of 256 workgroups instead of 4 of
        # 200) is the faster arrangement — the opposite of the two-branch steps (csrc/gemm_bf16.hip; read once, at the first GEMM)
        os.environ.setdefault("MEDP_GEMM_RAGGED", "1")
    ccfg = CohortCfg(n_timesteps=T, n_vars=V, d_static=DS, image_size=img, n_labels=K, seed=1234)
    side = args.stress or args.unfreeze_cxr or args.host_batches or args.eager or args.no_pipeline
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
        bt = make_batch(ccfg, start=shard_plan(args, rank, world, i)[1][0], batch_size=B, mode="teacher", stride=world)
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

    def timed(pool, steps, warmup, first=0, per_step=None):
        for i in range(warmup):
            run_step(pool, first + i)
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)] if per_step is not None else None
        t0 = time.perf_counter()
        if evs:
            evs[0].record()
        for i in range(steps):
            run_step(pool, first + warmup + i)
            if evs:
                evs[i + 1].record()
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
        dt = time.perf_counter() - t0
        if evs:
            per_step.extend(evs[i].elapsed_time(evs[i + 1]) for i in range(steps))
        if world > 1:
            tt = torch.tensor([dt], device=device, dtype=torch.float64)
            torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
            dt = float(tt.item())
        return dt

    # `value`: batches RESIDENT in HBM when the timed region starts; the PCIe-inclusive rate (pinned host batches, staged one call
    # ahead, `feats_to_input` inside the step) is measured right after it and reported in `config` (--host-batches swaps the two)
    main_pool = host_pool if args.host_batches else dev_pool
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
    # ---- side measurements: the other batch location; per-step HIP events over max(steps, 50) steps (median) -------
    dt_other = None
    if gstep is not None or cfg == "probe":
        dt_other = timed(dev_pool if args.host_batches else host_pool, args.steps, 2, first=args.warmup + args.steps)
    per_step_ms = []
    n_med = max(args.steps, 50)
    timed(main_pool, n_med, 2, first=args.warmup + 2 * args.steps + 2, per_step=per_step_ms)
    per_step_ms.sort()
    median_ms = per_step_ms[len(per_step_ms) // 2]
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
    # tracked reductions of rocprofv3 runs (tools/collect_profiles.sh -> tools/make_profile_json.py), each with its source / commit / date
    def _tracked(name):
        try:
            return json.load(open(os.path.join(ROOT, "profiles", name)))
        except Exception:
            return None
    tj, rj = _tracked("traffic.json"), _tracked("rocprof_gemm.json")
    traffic = tj.get("vit_gemm_hbm_bytes_per_launch") if tj else None
    traffic_source = None if not tj else {k: tj.get(k) for k in ("source", "commit", "date", "tag")}
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
        workload = "SIDE MEASUREMENT (" + ", ".join(f for f, on in (("--host-batches", args.host_batches), ("--eager", args.eager), ("--no-pipeline", args.no_pipeline)) if on) + "): " + workload
    execution = ("eager (engine.py from Python)" + (f" — FALLBACK, the captured step failed to build: {graph_error}" if graph_error else "")) if gstep is None else (
        ("captured HIP graph replay (graph_step.py)" + (": frozen part of batch k+1 run beside the step of batch k (one frozen forward, one "
         "trainable fwd/bwd and one update per replay; 4 distinct batches rotate)" if pipeline else ", frozen part inside its own step"))
        + ("; N>1: fwd/bwd graph -> RCCL mean all-reduce of the flat gradient arena -> optimiser graph on the main stream, the frozen-forward "
           "graph of the next batch beside them on its own stream" if (world > 1 or force_pg) else ""))
    res = {
        "metric": "multimodal train samples/sec", "value": round(value, 2), "unit": "samples/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3), "ms_per_step_median": round(median_ms, 3),
        "ms_per_step_median_of": f"{n_med} steps, HIP events on the step's stream, rank 0", "higher_is_better": True,
        "scaling": "strong" if args.strong else "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
        "config": {"workload": workload, "config": cfg, "per_gpu_batch": B, "global_batch": B * world, "parallelism": f"dp{world}",
                   "batch_location": ("pinned host memory: H2D + feats_to_input inside the timed step" if args.host_batches else
                                      "HBM (resident when the timed region starts); feats_to_input inside the timed step"),
                   ("resident_batch_samples_per_s" if args.host_batches else "pcie_inclusive_samples_per_s"):
                       round(world * B * args.steps / dt_other, 2) if dt_other else None,
                   # graph_step's hardware-queue phase check: (pad streams, ms a step loses to the staged pixel copy, ms of that copy alone)
                   "hw_queue_phase": getattr(gstep, "phase_log", None),
                   "gflop_per_sample": gps,
                   "step_mfma_fraction_of_peak": None if gps is None else round(value * gps / 1e3 / (PEAK_BF16_TFLOPS * world), 4),
                   "last_loss": round(last_loss, 5), "execution": execution,
                   # N > 1 (or MEDP_FORCE_PG=1): per-step HIP-event times of the gradient all-reduce and the optimiser update
                   "split_step_ms": gstep.split_timings() if (gstep is not None and hasattr(gstep, "split_timings")) else None,
                   "gradient_exchange_bytes": (gstep.arena.bytes_per_step if (gstep is not None and gstep.arena is not None) else
                                               (reducer.bytes_per_step if reducer is not None else 0))},
    }
    if in_step is not None:
        res["roofline"] = {
            "bound": "mfma",
            "kernel": "gemm_bf16_nt_v6_kernel<1> / gemm_bf16_nt_v7_kernel<1> (CXR-encoder block GEMMs: proj, fc2 / qkv, fc1 — one K-loop: "
                      "256x256x64 tiles, 8 waves ping-pong, 128x64 per wave; v7 = persistent over the tile list)",
            "achieved": round(in_step["achieved"], 1), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
            "frac": round(in_step["achieved"] / PEAK_BF16_TFLOPS, 4), "traffic": traffic, "traffic_source": traffic_source,
            # the same kernels' mean duration in the tracked rocprofv3 kernel trace (profiles/rocprof_gemm.json: written from ONE run that
            # also carries that run's in-kernel clocks, so the two figures and their gap can be read side by side)
            "frac_rocprof": None if not rj else round(in_step["flops_per_launch"] / (rj["avg_launch_us"] * 1e-6) / 1e12 / PEAK_BF16_TFLOPS, 4),
            "rocprof": None if not rj else {k: rj.get(k) for k in ("avg_launch_us", "launches", "same_run_in_kernel_avg_launch_us", "gap_us_per_launch",
                                                                   "source", "commit", "date")},
            "launches": in_step["launches"],
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
        cb = make_batch(ccfg, start=10_000, batch_size=B, mode="teacher")            # the step's own batch ("same B")
        cb1 = make_batch(ccfg, start=20_000, batch_size=min(4, B), mode="teacher")
        res["cpu_baseline"] = cpu_baseline(cfg, trainable, teacher, ccfg, K, cb, cb1)
    else:
        res["cpu_baseline"] = None
    print(json.dumps(res), flush=True)
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
