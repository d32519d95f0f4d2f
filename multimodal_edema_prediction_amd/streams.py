"""Where this package's extra HIP streams come from.

HIP multiplexes all streams of one priority over a few hardware queues (GPU_MAX_HW_QUEUES, default 4; a new stream goes to the
queue with the fewest streams on it) and a hardware queue executes its packets in order, so WHICH streams share a queue decides
what really runs concurrently — and it is an accident of creation order.  Measured on one box with otherwise identical code
(DESIGN.md section 6): teacher step 5.0-13 ms and student step 7.8-14 ms across GPU_MAX_HW_QUEUES = 2 ... 16; with more than 4
queues some arrangements fall off a cliff (13 ms), with 4 the spread is +-8 %.  Streams created through the HIP runtime
(`MEDP_RAW_STREAMS=1`, so that torch's pool of 32 streams per priority never exists and every live stream could have a queue of
its own with GPU_MAX_HW_QUEUES=16) were uniformly WORSE (13 ms): many concurrently active hardware queues is the slow case.
Default therefore: torch's pool streams and the runtime's default of 4 queues."""
from __future__ import annotations

import ctypes
import os

import torch

_HIP = None
_KEEP = []          # (handle, ExternalStream): raw streams live as long as the process


def new_stream(device=None, raw=None) -> "torch.cuda.Stream":
    """A stream on `device`: of torch's pool (default) or, with raw=True / MEDP_RAW_STREAMS=1, a non-blocking HIP stream made here."""
    global _HIP
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    if raw is None:
        raw = os.environ.get("MEDP_RAW_STREAMS", "0") == "1"
    if not raw:
        return torch.cuda.Stream(device=dev)
    try:
        if _HIP is None:
            _HIP = ctypes.CDLL("libamdhip64.so")
            _HIP.hipStreamCreateWithFlags.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint]
            _HIP.hipStreamCreateWithFlags.restype = ctypes.c_int
        h = ctypes.c_void_p()
        with torch.cuda.device(dev):
            rc = _HIP.hipStreamCreateWithFlags(ctypes.byref(h), 1)          # 1 = hipStreamNonBlocking
        if rc != 0 or not h.value:
            raise OSError(f"hipStreamCreateWithFlags -> {rc}")
        s = torch.cuda.ExternalStream(h.value, device=dev)
        _KEEP.append((h, s))
        return s
    except (OSError, AttributeError):
        return torch.cuda.Stream(device=dev)


# ---- stream forks inside a graph capture --------------------------------------------------------------------------------------
# On this ROCm a stream fork NESTED inside a forked branch of a capture (origin -> branch A -> branch B) makes hipStreamEndCapture
# crash (segmentation fault in capture_end).  Every fork this package makes from inside a forward goes through
# `fork_guard()`: while the current stream is being captured it must be the capture's ORIGIN stream — the stream the capture began
# on, noted by `note_capture_origin()` at the package's capture sites or, failing that, the first stream a guard sees for that
# capture id — otherwise it raises instead of letting the runtime crash later.
_ORIGIN = {}        # capture id -> cuda_stream handle of the stream the capture began on


def _capture_id(stream) -> "int | None":
    global _HIP
    try:
        if _HIP is None:
            _HIP = ctypes.CDLL("libamdhip64.so")
        fn = _HIP.hipStreamGetCaptureInfo
        fn.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_ulonglong)]
        fn.restype = ctypes.c_int
        status, cid = ctypes.c_int(0), ctypes.c_ulonglong(0)
        if fn(ctypes.c_void_p(stream.cuda_stream), ctypes.byref(status), ctypes.byref(cid)) != 0 or status.value != 1:
            return None                                    # 1 = hipStreamCaptureStatusActive
        return int(cid.value)
    except (OSError, AttributeError):
        return None


def note_capture_origin(device=None) -> None:
    """Call right after a capture begins (first statement inside `with torch.cuda.graph(g):`), on the capture's own stream."""
    if not torch.cuda.is_available() or not torch.cuda.is_current_stream_capturing():
        return
    cur = torch.cuda.current_stream(device)
    cid = _capture_id(cur)
    if cid is not None:
        _ORIGIN.setdefault(cid, cur.cuda_stream)


def fork_guard(what: str, device=None) -> None:
    """Call right before forking a side stream from the current one.  Outside a capture: nothing.  Inside: the current stream must
    be the capture's origin stream, else RuntimeError (a nested fork would crash hipStreamEndCapture)."""
    if not torch.cuda.is_current_stream_capturing():
        return
    cur = torch.cuda.current_stream(device)
    cid = _capture_id(cur)
    if cid is None:
        return
    origin = _ORIGIN.setdefault(cid, cur.cuda_stream)
    if origin != cur.cuda_stream:
        raise RuntimeError(f"{what}: about to fork a side stream from a stream that is itself a forked branch of a graph capture — a nested "
                           "fork makes hipStreamEndCapture crash on this ROCm.  Keep this forward on one stream (TeacherModel.forward("
                           "..., _overlap=False), MEDP_OVERLAP=0) or call it from the capture's origin stream")
