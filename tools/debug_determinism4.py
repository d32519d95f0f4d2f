"""lr = 0 replays: which loss component changes when a replay deviates?"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import test_gpu_model as T
from multimodal_edema_prediction_amd.graph_step import GraphedTeacherStep
from multimodal_edema_prediction_amd.optim import FusedAdamW, make_param_groups
from multimodal_edema_prediction_amd.losses_duett import DualPathologyLoss
DEV = torch.device("cuda")
batch = T.make_batch(T.CCFG, T.META["teacher_batch_start"], T.B, mode="teacher")
loss_fn = DualPathologyLoss(torch.ones(T.K), None, 0.5, 0.5, 1.0).to(DEV)
for inst in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
    tg = T.build_teacher(); og = FusedAdamW(make_param_groups(tg, 0.0), weight_decay=0.0)
    gs = GraphedTeacherStep(tg, loss_fn, og, batch, DEV, warmup=3)
    ref = None; hits = []
    for r in range(int(sys.argv[2]) if len(sys.argv) > 2 else 150):
        out = gs.step(batch); torch.cuda.synchronize()
        cur = {k: float(out[k].detach().double().abs().sum().item()) for k in ("img_total", "ts_total", "fus_total", "fusion_logits")}
        if ref is None: ref = cur
        elif cur != ref: hits.append((r, {k: cur[k] - ref[k] for k in cur if cur[k] != ref[k]}))
    print(f"instance {inst}: {len(hits)} deviating replays: {hits[:6]}", flush=True)
