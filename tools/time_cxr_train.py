import sys, os, time
sys.path.insert(0, "/root/repo")
import torch
from multimodal_edema_prediction_amd.main_architecture_duett import CXREncoder
dev = torch.device("cuda")
enc = CXREncoder("synthetic", freeze=False).to(dev).train()
for B in (8, 64):
    px = torch.randn(B, 3, 224, 224, device=dev)
    for it in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        tok = enc.forward_bf16(px); torch.cuda.synchronize(); t1 = time.perf_counter()
        tok.square().mean().backward(); torch.cuda.synchronize(); t2 = time.perf_counter()
        enc.zero_grad(set_to_none=True)
    print(f"B={B}: forward {1e3*(t1-t0):.1f} ms, backward {1e3*(t2-t1):.1f} ms", flush=True)
