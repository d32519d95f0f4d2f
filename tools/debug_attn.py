import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimodal_edema_prediction_amd import functional as Fn
torch.manual_seed(0)
for (B, S, H) in [(2, 256, 2), (2, 257, 2), (2, 272, 2), (2, 320, 2), (1, 64, 2), (1, 100, 2), (1, 1297, 2)]:
    D = H * 64
    qkv = (torch.randn(B * S, 3 * D, device="cuda")).bfloat16()
    o = Fn.attn_dh64(qkv, B, S, H, 0.125).float().view(B, S, H, 64)
    q, k, v = [t.float().view(B, S, H, 64).permute(0, 2, 1, 3) for t in qkv.split(D, dim=1)]
    ref = (torch.softmax(q @ k.transpose(-1, -2) * 0.125, -1) @ v).permute(0, 2, 1, 3)
    err = (o - ref).abs().amax(-1)          # [B, S, H]
    bad = err > 2e-2
    idx = bad[0, :, 0].nonzero().flatten().tolist()
    print(f"S={S}: bad queries (b0,h0) n={len(idx)} first {idx[:5]} last {idx[-5:]} maxerr {float(err.max()):.3f} nan {torch.isnan(o).sum().item()}")
