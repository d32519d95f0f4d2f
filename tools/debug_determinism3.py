"""lr = 0: every replay of one captured step must reproduce the same loss and gradients bit for bit."""
import sys, os, hashlib
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
def gdigest(model):
    out = {}
    for k, p in model.named_parameters():
        if p.requires_grad and p.grad is not None:
            out[k] = hashlib.sha1(p.grad.detach().cpu().numpy().tobytes()).hexdigest()[:8]
    return out
for inst in range(int(sys.argv[1]) if len(sys.argv) > 1 else 4):
    tg = T.build_teacher(); og = FusedAdamW(make_param_groups(tg, 0.0), weight_decay=0.0)
    gs = GraphedTeacherStep(tg, loss_fn, og, batch, DEV, warmup=3)
    ref, refl, bad = None, None, {}
    for r in range(25):
        l = float(gs.step(batch)["loss"].item()); torch.cuda.synchronize()
        d = gdigest(tg)
        if ref is None: ref, refl = d, l
        else:
            for k in d:
                if d[k] != ref[k]: bad[k] = bad.get(k, 0) + 1
            if l != refl: bad["<loss>"] = bad.get("<loss>", 0) + 1
    print(f"instance {inst}: loss {refl:.9f} mismatching replays per tensor: {bad if bad else 'none'}", flush=True)
