import sys, os, json
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import torch, numpy as np
import test_gpu_model as tm
from multimodal_edema_prediction_amd import engine, autograd_ops as A
from multimodal_edema_prediction_amd.optim import make_param_groups, FusedAdamW
from multimodal_edema_prediction_amd.losses_duett import DualPathologyLoss
DEV = "cuda"; K = tm.K
tb = tm.make_batch(tm.CCFG, tm.META["teacher_batch_start"], tm.B, mode="teacher")
loss_fn = DualPathologyLoss(torch.ones(K), None, 0.5, 0.5, 1.0).to(DEV)
mode = sys.argv[1]
te = tm.build_teacher(); oe = FusedAdamW(make_param_groups(te, 8e-5), weight_decay=5e-2)
tg = tm.build_teacher(); og = FusedAdamW(make_param_groups(tg, 8e-5), weight_decay=5e-2)
engine._set_train_with_frozen_eval(tg)
b = engine._move_lists(tb, DEV)
def fb(m):
    out = m(b["x_ts"], b["x_static"], b["bin_ends"], b["pixel_values"])
    L = loss_fn(out["img_logits"], out["ts_logits"], out["fusion_logits"], b["y_multi"], b["y_multi_mask"])
    L["total"].backward(); return L["total"].detach()
for _ in range(3): engine.train_teacher_dual_pathology_batch(tb, te, loss_fn, oe, torch.device(DEV))
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3): og.zero_grad(); fb(tg); og.step()
torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
og.zero_grad(set_to_none=True)
if mode == "nocache":      # drop every cached bf16 / transposed weight before capture
    A._W_CACHE.clear()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    l = fb(tg)
for i in range(2):
    le = engine.train_teacher_dual_pathology_batch(tb, te, loss_fn, oe, torch.device(DEV))["loss"]
    g.replay(); og.step(); torch.cuda.synchronize()
    worst = sorted(((float((a.grad - p.grad).abs().max() / (a.grad.abs().max() + 1e-12)), k) for (k, a), (_, p) in zip(te.named_parameters(), tg.named_parameters()) if a.grad is not None), reverse=True)
    print(mode, "step", 4 + i, le, float(l), worst[:3])
