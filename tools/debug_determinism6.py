"""Forward-only captured graph, image half unrolled by hand next to the real time-series half on the side stream: which
intermediate is the first that is not bit-stable across replays?"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import test_gpu_model as T
from multimodal_edema_prediction_amd import engine, autograd_ops as A
from multimodal_edema_prediction_amd.main_architecture_duett import _BroadcastRowsFn, _side_stream
DEV = torch.device("cuda")
batch = T.make_batch(T.CCFG, T.META["teacher_batch_start"], T.B, mode="teacher")
for inst in range(3):
    te = T.build_teacher(); engine._set_train_with_frozen_eval(te); pc = te.perceiver
    x_ts = torch.stack(tuple(batch["x_ts"])).to(DEV); x_st = torch.stack(tuple(batch["x_static"])).to(DEV)
    be = torch.stack(tuple(batch["bin_ends"])).to(DEV); px = batch["pixel_values"].to(DEV)
    B = x_ts.shape[0]
    def fwd():
        out = {}
        duett_in = te.duett.feats_to_input((tuple(x_ts[i] for i in range(B)), tuple(x_st[i] for i in range(B)), tuple(be[i] for i in range(B))), B)
        q0 = _BroadcastRowsFn.apply(pc.shared_queries, B)
        cur = torch.cuda.current_stream(); side = _side_stream(DEV)
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            tt = te.duett.encode(duett_in)
            ts = pc._ts_branch(pc._select_ts(tt, "hourly_only"), q0, 0, False)
        tok = te.cxr.forward_bf16(px); out["tokens16"] = tok
        ip = A.linear(tok, te.img_proj.weight, te.img_proj.bias); out["img_proj"] = ip
        blk = pc.img_cross; d = 256
        W, b = blk.attn.in_proj_weight, blk.attn.in_proj_bias
        kn = A.layer_norm(ip, blk.norm_kv.weight, blk.norm_kv.bias, blk.norm_kv.eps); out["kn"] = kn
        KV = A.linear(kn, W[d:], b[d:]); out["KV"] = KV
        I = pc.img_cross(q0, ip, _kv_skip=1, _shared_q=pc.shared_queries, _seed=0); out["I_cross"] = I
        I2 = pc.img_self(I, I, _seed=0); out["I_self"] = I2
        cur.wait_stream(side)
        out["T_tok"] = ts["T_tok"]
        return out
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s), torch.no_grad():
        for _ in range(3): fwd()
    torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g), torch.no_grad():
        out = fwd()
    ref = None; hits = {}; first = {}
    for r in range(200):
        g.replay(); torch.cuda.synchronize()
        cur = {k: v.detach().clone() for k, v in out.items()}
        if ref is None: ref = cur
        else:
            f = None
            for k in out:
                if not torch.equal(cur[k], ref[k]):
                    hits[k] = hits.get(k, 0) + 1
                    if f is None:
                        f = k; first[k] = first.get(k, 0) + 1
                        if len(first) <= 2 and first[k] <= 2:
                            dd = (cur[k].float() - ref[k].float()).abs(); nz = (dd > 0).nonzero()
                            print(f"   replay {r}: first bad {k} shape {tuple(cur[k].shape)}: {int((dd>0).sum())} elems, max {float(dd.max()):.3e}, idx {nz[:2].tolist()}..{nz[-2:].tolist()}")
    print(f"instance {inst}: deviating per tensor {hits if hits else 'none'}; first-bad counts {first}", flush=True)
