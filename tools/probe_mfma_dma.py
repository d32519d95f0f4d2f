#!/usr/bin/env python3
"""One wave per SIMD (4 waves per workgroup, one workgroup per CU): cost of LDS-DMA pieces and ds_read_b128 issued inside the wave's own
stream of 64 MFMAs per iteration (debug hook medp_dbg_mfma_dma_probe, csrc/dbg_mfma_probe.hip).  Ideal: 64 x 16 = 1024 cycles per iteration."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimodal_edema_prediction_amd.abi import lib
L = lib()
L.medp_dbg_mfma_dma_probe.restype = ctypes.c_int
L.medp_dbg_mfma_dma_probe.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
src = torch.randn(64 << 20, device="cuda", dtype=torch.float32)            # 256 MB
out = torch.zeros(256, dtype=torch.int64, device="cuda")
sink = torch.zeros(1, device="cuda")
iters = 2000
for pieces, reads in ((0, 0), (0, 16), (4, 16), (8, 0), (8, 16), (16, 0), (16, 8), (16, 16)):
    for rep in range(2):
        rc = L.medp_dbg_mfma_dma_probe(pieces, reads, src.data_ptr(), src.numel() * 4, iters, out.data_ptr(), sink.data_ptr(), torch.cuda.current_stream().cuda_stream)
        assert rc == 0, rc
        torch.cuda.synchronize()
    ticks = out[:int(os.environ.get('MEDP_PROBE_GRID', '256'))].double().mean().item()                     # 100-MHz ticks for the loop
    us_per_iter = ticks / 100.0 / iters
    print(f"pieces {pieces:2d} reads {reads:2d}: {us_per_iter * 1e3:7.1f} ns per iteration of 64 MFMAs  (ideal 1024 cycles = {1024 / 2.4:.0f} ns at 2.4 GHz)", flush=True)
