import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimodal_edema_prediction_amd import functional as Fn
M = 64 * 257
for (m, n, k) in [(M, 2304, 768), (M, 768, 3072), (8192, 8192, 8192)]:
    a = torch.randn(m, k, device="cuda").bfloat16(); w = torch.randn(n, k, device="cuda").bfloat16()
    out = torch.empty(m, n, device="cuda", dtype=torch.bfloat16)
    for _ in range(5): Fn.gemm(a, w, out=out)
torch.cuda.synchronize()
