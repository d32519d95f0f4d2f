import sys, os, json
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import torch, numpy as np
import test_gpu_model as tm
from multimodal_edema_prediction_amd import engine
from multimodal_edema_prediction_amd.losses_duett import DualPathologyLoss
DEV = "cuda"; K = tm.K
tb = tm.make_batch(tm.CCFG, tm.META["teacher_batch_start"], tm.B, mode="teacher")
loss_fn = DualPathologyLoss(torch.ones(K), None, 0.5, 0.5, 1.0).to(DEV)
te = tm.build_teacher()
engine._set_train_with_frozen_eval(te)
def grads(variant=None):
    if variant: os.environ["MEDP_GEMM_VARIANT"] = variant
    b = engine._move_lists(tb, DEV)
    out = te(b["x_ts"], b["x_static"], b["bin_ends"], b["pixel_values"])
    L = loss_fn(out["img_logits"], out["ts_logits"], out["fusion_logits"], b["y_multi"], b["y_multi_mask"])
    te.zero_grad(); L["total"].backward()
    return {k: p.grad.clone() for k, p in te.named_parameters() if p.grad is not None}, float(L["total"])
g1, l1 = grads(); g2, l2 = grads(); g3, l3 = grads()
print("loss", l1, l2, l3)
for a, b, tag in ((g1, g2, "run1 vs run2"), (g2, g3, "run2 vs run3")):
    worst = sorted(((float((a[k] - b[k]).abs().max() / (a[k].abs().max() + 1e-12)), k) for k in a), reverse=True)
    print(tag, worst[:5])
