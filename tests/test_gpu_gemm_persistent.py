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
