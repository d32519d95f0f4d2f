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
b = engine._move_lists(tb, DEV)
def fb():
    out = te(b["x_ts"], b["x_static"], b["bin_ends"], b["pixel_values"])
    L = loss_fn(out["img_logits"], out["ts_logits"], out["fusion_logits"], b["y_multi"], b["y_multi_mask"])
    L["total"].backward()
    return L["total"].detach()
te.zero_grad(); fb(); torch.cuda.synchronize()
ge = {k: p.grad.clone() for k, p in te.named_parameters() if p.grad is not None}
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(2): te.zero_grad(); fb()
torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
te.zero_grad(set_to_none=True)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    l = fb()
for rep in range(3):
    g.replay(); torch.cuda.synchronize()
    worst = sorted(((float((ge[k] - p.grad).abs().max() / (ge[k].abs().max() + 1e-12)), k) for k, p in te.named_parameters() if p.grad is not None), reverse=True)
    print("replay", rep, float(l), worst[:4])
