"""Forward-only captured graph (teacher forward with the two-stream overlap): are replays bit-identical?"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import test_gpu_model as T
from multimodal_edema_prediction_amd import engine
DEV = torch.device("cuda")
batch = T.make_batch(T.CCFG, T.META["teacher_batch_start"], T.B, mode="teacher")
mode = sys.argv[1] if len(sys.argv) > 1 else "fwd"
for inst in range(3):
    te = T.build_teacher()
    engine._set_train_with_frozen_eval(te)
    x_ts = torch.stack(tuple(batch["x_ts"])).to(DEV); x_st = torch.stack(tuple(batch["x_static"])).to(DEV)
    be = torch.stack(tuple(batch["bin_ends"])).to(DEV); px = batch["pixel_values"].to(DEV)
    B = x_ts.shape[0]
    def fwd():
        return te(tuple(x_ts[i] for i in range(B)), tuple(x_st[i] for i in range(B)), tuple(be[i] for i in range(B)), px, return_attn=True)
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            if mode == "fwd":
                with torch.no_grad(): fwd()
            else:
                o = fwd(); (o["fusion_logits"].sum() + o["img_logits"].sum() + o["ts_logits"].sum()).backward(); te.zero_grad(set_to_none=True)
    torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        if mode == "fwd":
            with torch.no_grad(): out = fwd()
        else:
            out = fwd(); (out["fusion_logits"].sum() + out["img_logits"].sum() + out["ts_logits"].sum()).backward()
    keys = ["img_tokens", "ts_tokens", "img_logits", "ts_logits", "fusion_logits"]
    ref = None; hits = {}
    for r in range(200):
        g.replay(); torch.cuda.synchronize()
        cur = {k: out[k].detach().clone() for k in keys}
        if ref is None: ref = cur
        else:
            for k in keys:
                if not torch.equal(cur[k], ref[k]): hits[k] = hits.get(k, 0) + 1
    print(f"mode={mode} instance {inst}: deviating replays per output: {hits if hits else 'none'}", flush=True)
