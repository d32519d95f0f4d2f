#!/usr/bin/env python3
"""Which hardware queue does a pinned host->device copy block?  A two-branch graph of long kernels (captured after PAD raw HIP streams,
which shift the queues its internal streams get) is replayed beside a 38.5-MB pinned H2D copy issued on different streams.
Prints the replay time per (pad, copy stream)."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimodal_edema_prediction_amd.streams import new_stream

dev = torch.device("cuda", 0)
torch.cuda.init()
x = torch.randn(4096, 4096, device=dev).bfloat16()
y = torch.randn(4096, 4096, device=dev).bfloat16()
o1, o2 = torch.empty_like(x), torch.empty_like(x)
host = torch.empty(64, 3, 224, 224).pin_memory()
dst = torch.empty(64, 3, 224, 224, device=dev)
pool = [torch.cuda.Stream() for _ in range(4)]
raw = [new_stream(dev, raw=True) for _ in range(2)]


def build(pad):
    pads = [new_stream(dev, raw=True) for _ in range(pad)]
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    cap = torch.cuda.Stream()
    with torch.cuda.stream(cap):
        torch.matmul(x, y, out=o1)
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=cap):
            side.wait_stream(cap)
            with torch.cuda.stream(side):
                for _ in range(12):
                    torch.matmul(x, y, out=o2)            # the "frozen" branch: ~1.2 ms
            for _ in range(6):
                torch.matmul(x, y, out=o1)                # the "training" branch: ~0.6 ms
            cap.wait_stream(side)
    return g, pads


def run(g, copy_stream, n=20):
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        if copy_stream is not None:
            with torch.cuda.stream(copy_stream):
                dst.copy_(host, non_blocking=True)
        g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for pad in range(5):
    g, pads = build(pad)
    row = [f"{run(g, None):.3f}"] + [f"{run(g, s):.3f}" for s in pool + raw]
    print(f"pad {pad}: no copy / copy on pool0..3, raw0..1: " + "  ".join(row), flush=True)
