import sys, os, json
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import torch, numpy as np
import test_gpu_model as tm
from multimodal_edema_prediction_amd import engine
from multimodal_edema_prediction_amd.graph_step import GraphedTeacherStep
from multimodal_edema_prediction_amd.optim import make_param_groups, FusedAdamW
from multimodal_edema_prediction_amd.losses_duett import DualPathologyLoss
DEV = "cuda"; K = tm.K
tb = tm.make_batch(tm.CCFG, tm.META["teacher_batch_start"], tm.B, mode="teacher")
loss_fn = DualPathologyLoss(torch.ones(K), None, 0.5, 0.5, 1.0).to(DEV)
te = tm.build_teacher(); oe = FusedAdamW(make_param_groups(te, 8e-5), weight_decay=5e-2)
tg = tm.build_teacher(); og = FusedAdamW(make_param_groups(tg, 8e-5), weight_decay=5e-2)
for _ in range(3): engine.train_teacher_dual_pathology_batch(tb, te, loss_fn, oe, torch.device(DEV))
gs = GraphedTeacherStep(tg, loss_fn, og, tb, torch.device(DEV), warmup=3)
def cmp(tag):
    worst = []
    for (k, a), (_, b) in zip(te.named_parameters(), tg.named_parameters()):
        if a.requires_grad: worst.append((float((a - b).abs().max()), k))
    worst.sort(reverse=True); print(tag, worst[:4])
cmp("after warmup")
for i in range(2):
    le = engine.train_teacher_dual_pathology_batch(tb, te, loss_fn, oe, torch.device(DEV))["loss"]
    lg = float(gs.step(tb)["loss"].item())
    print("loss", le, lg); cmp(f"after step {4+i}")
    # compare grads
    worst = []
    for (k, a), (_, b) in zip(te.named_parameters(), tg.named_parameters()):
        if a.requires_grad and a.grad is not None and b.grad is not None: worst.append((float((a.grad - b.grad).abs().max() / (a.grad.abs().max() + 1e-12)), k))
    worst.sort(reverse=True); print("  grad rel diff", worst[:4])
