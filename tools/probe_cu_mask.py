#!/usr/bin/env python3
"""Does a CU-masked stream (hipExtStreamCreateWithCUMask) confine kernels (a) launched eagerly, (b) captured into a hipGraph from
that stream as a forked branch?  A throughput-bound kernel (torch.matmul 8192^3 bf16) is timed on an unmasked stream and on streams
masked to 128 / 64 CUs: confined kernels take ~2x / ~4x as long."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

hip = ctypes.CDLL("libamdhip64.so")
hip.hipExtStreamCreateWithCUMask.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint32)]
hip.hipExtStreamCreateWithCUMask.restype = ctypes.c_int
torch.cuda.init()
x = torch.randn(8192, 8192, device="cuda").bfloat16()
y = torch.randn(8192, 8192, device="cuda").bfloat16()
out = torch.empty(8192, 8192, device="cuda", dtype=torch.bfloat16)


def masked_stream(n_cus):
    words = (ctypes.c_uint32 * 8)()
    for i in range(n_cus):          # the first n_cus bits
        words[i // 32] |= 1 << (i % 32)
    s = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), 8, words)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value)


def t_eager(stream, n=10):
    with torch.cuda.stream(stream):
        for _ in range(3):
            torch.matmul(x, y, out=out)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            torch.matmul(x, y, out=out)
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def t_graph(stream, n=10):
    g = torch.cuda.CUDAGraph()
    cap = torch.cuda.Stream()
    with torch.cuda.stream(cap):
        torch.matmul(x, y, out=out)
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=cap):
            stream.wait_stream(cap)
            with torch.cuda.stream(stream):
                for _ in range(4):
                    torch.matmul(x, y, out=out)
            cap.wait_stream(stream)
    for _ in range(2):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n / 4 * 1e3


for label, s in (("unmasked", torch.cuda.Stream()), ("128 CUs", masked_stream(128)), ("64 CUs", masked_stream(64))):
    print(f"{label:9s}: eager {t_eager(s):7.3f} ms per matmul | captured as a forked branch {t_graph(s):7.3f} ms per matmul", flush=True)
