import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch, numpy as np
from helpers import *
from multimodal_edema_prediction_amd import engine, autograd_ops as A
from multimodal_edema_prediction_amd.cohort import CohortCfg, make_batch
from multimodal_edema_prediction_amd.main_architecture_duett import DuettFeatureExtractor, StudentModel
from oracle import duett_ref
META = json.load(open(os.path.join(GOLDEN_DIR, "meta.json"))); SHAPES = load_shapes("shapes.json")
B, T, V, DS = META["B"], META["T"], META["V"], META["DS"]
CCFG = CohortCfg(n_timesteps=T, n_vars=V, d_static=DS, image_size=224, seed=META["cohort_seed"])
bb = DuettFeatureExtractor(d_static_num=DS, d_time_series_num=V, d_target=1, pretrain=False, masked_transform_timesteps=T, max_len=T)
s = StudentModel(bb, pool="mean", head_hidden=128, head_dropout=0.0)
sd = synth_state_dict(SHAPES["student"], seed=2)
s.load_state_dict(sd, strict=True); s = s.cuda().train()
tb = make_batch(CCFG, 100, B, mode="student"); b = engine._move_lists(tb, "cuda")
xin = s.duett.feats_to_input((b["x_ts"], b["x_static"], b["bin_ends"]), B)
dcfg = duett_ref.DuettCfg(d_static_num=DS, d_time_series_num=V, n_timesteps=T)
sdc = {k: v.clone() for k, v in sd.items()}
dsd = {k[6:]: v for k, v in sdc.items() if k.startswith("duett.")}
xin_ref = duett_ref.feats_to_input((tb["x_ts"], tb["x_static"], list(tb["bin_ends"])), max_len=T)
for i in range(3): print("xin", i, float((xin[i].cpu() - xin_ref[i]).abs().max()))
tok_ref = duett_ref.encode(dsd, dcfg, xin_ref, training=True)
tok = s.duett.encode(xin)
print("tok err", float((tok.detach().cpu() - tok_ref).abs().max()), float(tok_ref.abs().max()))
feat = A.MeanPoolFn.apply(tok, T); feat_ref = tok_ref[:, :-1].mean(1)
print("feat err", float((feat.detach().cpu() - feat_ref).abs().max()))
h = A.linear(feat, s.head[0].weight, s.head[0].bias); h_ref = torch.nn.functional.linear(feat_ref, sd["head.0.weight"], sd["head.0.bias"])
print("h err", float((h.detach().cpu() - h_ref).abs().max()), float(h_ref.abs().max()))
g = A.gelu_dropout(h, 0.0, 0, 0); g_ref = torch.nn.functional.gelu(h_ref)
print("gelu err", float((g.detach().cpu() - g_ref).abs().max()))
z = A.rowdot(g, s.head[3].weight, s.head[3].bias); z_ref = torch.nn.functional.linear(g_ref, sd["head.3.weight"], sd["head.3.bias"]).squeeze(-1)
print("z", z.detach().cpu(), z_ref)
