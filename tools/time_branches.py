#!/usr/bin/env python3
"""Which branch of the captured teacher step is the long one?  Builds the step as bench.py does (B 64, 224 x 224, T 96, V 48,
HBM-resident batch) and replays three graphs: the whole step, the training branch alone (DuETT + fusion head forward / loss /
backward / AdamW on the tokens already in `tok_cur`) and the frozen CXR encoder alone."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from multimodal_edema_prediction_amd.cohort import CohortCfg, make_batch
from multimodal_edema_prediction_amd.graph_step import GraphedTeacherStep
from multimodal_edema_prediction_amd.losses_duett import DualPathologyLoss
from multimodal_edema_prediction_amd.optim import FusedAdamW, make_param_groups

T, V, DS, K, B = 96, 48, 8, 7, 64
dev = torch.device("cuda", 0)
STUDENT = len(sys.argv) > 1 and sys.argv[1] == "student"
teacher = bench.build_teacher(T, V, DS, K, dev, freeze_all=STUDENT)
bt = make_batch(CohortCfg(n_timesteps=T, n_vars=V, d_static=DS, image_size=224, n_labels=K, seed=1234), 0, B, mode="teacher")
if STUDENT:
    from multimodal_edema_prediction_amd.graph_step import GraphedStudentStep
    from multimodal_edema_prediction_amd.losses_duett import StudentKDLoss
    student = bench.build_student(T, V, DS, dev)
    loss_fn = StudentKDLoss("vanilla_kl", 4.0, 0.5)
    opt = FusedAdamW(make_param_groups(student, 8e-5), weight_decay=5e-2)
else:
    loss_fn = DualPathologyLoss(torch.ones(K), None, 0.5, 0.5, 1.0).to(dev)
    opt = FusedAdamW(make_param_groups(teacher, 8e-5), weight_decay=5e-2)
import ctypes
from multimodal_edema_prediction_amd import abi
L = abi.lib()
L.medp_gemm_profile_enable(2)          # launch clocks ride in every CXR-encoder block GEMM captured from here on
gs = (GraphedStudentStep(student, teacher, loss_fn, opt, bt, dev) if STUDENT else
      GraphedTeacherStep(teacher, loss_fn, opt, bt, dev, pipeline_cxr=True))
n_step = None


def timeline(label, lo, hi):
    """begin / end stamps of the block GEMMs captured as slots lo..hi-1: span, busy time, the gaps between consecutive GEMMs"""
    buf = (ctypes.c_ulonglong * 8192)()
    n, khz = ctypes.c_int(), ctypes.c_int()
    L.medp_dbg_gemm_profile_raw.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]
    L.medp_dbg_gemm_profile_raw(buf, 4096, ctypes.byref(n), ctypes.byref(khz))
    ev = [(buf[2 * i] * 1e3 / khz.value, buf[2 * i + 1] * 1e3 / khz.value) for i in range(lo, min(hi, n.value)) if buf[2 * i + 1] > buf[2 * i]]
    if not ev:
        return n.value
    ev.sort()
    busy = sum(e - b for b, e in ev)
    gaps = [ev[i + 1][0] - ev[i][1] for i in range(len(ev) - 1)]
    names = ["qkv", "proj", "fc1", "fc2"]
    per = {k: sum(ev[i][1] - ev[i][0] for i in range(j, len(ev), 4)) / max(len(ev) // 4, 1) for j, k in enumerate(names)}
    gap_after = {k: sum(gaps[i] for i in range(j, len(gaps), 4)) / max(len(range(j, len(gaps), 4)), 1) for j, k in enumerate(names)}
    print(f"  {label}: {len(ev)} GEMMs, first begin -> last end {ev[-1][1] - ev[0][0]:.0f} us, GEMM busy {busy:.0f} us, gaps {sum(gaps):.0f} us; "
          f"mean us per GEMM {({k: round(v, 1) for k, v in per.items()})}, mean gap after {({k: round(v, 1) for k, v in gap_after.items()})}", flush=True)
    return n.value


def replay_ms(fn, n=40):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


whole = replay_ms(lambda: gs.step())
n_step = timeline("", 4096, 4096)
timeline("block GEMMs inside the whole step (last replay)", n_step - 48, n_step)        # the captured launches are the last 48 issued
side = torch.cuda.Stream()
g_train, g_vit = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
opt.zero_grad(set_to_none=True)
with torch.cuda.stream(side):
    with torch.cuda.graph(g_train, stream=side):
        gs._advance()
        gs._train_fwd_bwd()
        opt.step()
    with torch.cuda.graph(g_vit, stream=side):
        gs._frozen_forward(0)
torch.cuda.synchronize()
t_train = replay_ms(g_train.replay)
t_vit = replay_ms(g_vit.replay)
timeline("block GEMMs of the encoder alone (last replay)", n_step, 4096)
print(f"{'student' if STUDENT else 'teacher'}: whole step {whole:.3f} ms | training branch alone {t_train:.3f} ms | frozen branch alone {t_vit:.3f} ms "
      f"(MEDP_OVERLAP={os.environ.get('MEDP_OVERLAP', 'default')}, MEDP_V7_WGS={os.environ.get('MEDP_V7_WGS', 'auto')})", flush=True)
