"""The persistent GEMM (csrc/gemm_bf16_v7.hip: qkv / fc1 of the CXR encoder) against the one-tile-per-workgroup kernel it replaces
(csrc/gemm_bf16_v6.hip): BIT-identical results on every eligible shape (full, ragged M, ragged N, K = 256 ... 3072), eager,
repeated (the ticket blocks re-arm themselves, the ring of 1024 wraps), on two streams at once and captured in a graph; both
also against an fp32 product.  The check runs three child processes (MEDP_GEMM_V7 = 1 / 0, and 1 with the ragged last rows as their own launch, csrc/gemm_ragged_rows.hip; the switches are read once per process)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_persistent_gemm_is_bit_identical_to_the_tile_per_workgroup_kernel():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_gemm_v7.py")], stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, text=True, timeout=900)
    assert r.returncode == 0 and "ALL OK" in r.stdout, r.stdout[-3000:]


@pytest.mark.gpu
def test_persistent_cap_changes_the_grid_not_the_result():
    """`medp_gemm_persistent_cap` (how many CUs the persistent GEMM may hold): returns the previous cap, any cap gives the same bits."""
    import torch
    sys.path.insert(0, ROOT)
    from multimodal_edema_prediction_amd import functional as Fn
    from multimodal_edema_prediction_amd.abi import lib
    torch.manual_seed(0)
    a = torch.randn(16448, 768, device="cuda").bfloat16()
    w = torch.randn(2304, 768, device="cuda").bfloat16()
    bias = torch.randn(2304, device="cuda")
    ref = Fn.gemm(a, w, bias=bias, act=1, out_dtype=torch.bfloat16)
    assert lib().medp_gemm_persistent_cap(64) == 0
    try:
        for cap in (64, 176, 250, 1000):
            prev = lib().medp_gemm_persistent_cap(cap)
            assert prev in (64, 176, 248, 256)
            assert torch.equal(Fn.gemm(a, w, bias=bias, act=1, out_dtype=torch.bfloat16), ref), cap
    finally:
        lib().medp_gemm_persistent_cap(0)
    assert torch.equal(Fn.gemm(a, w, bias=bias, act=1, out_dtype=torch.bfloat16), ref)
