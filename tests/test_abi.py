"""CPU-side checks of the C-ABI boundary: the library builds/loads, exports every function include/medp_hip.h declares,
the ctypes table covers them all, and the product path refuses to run without a GPU (no CPU fallback)."""
import ctypes
import os
import re

import pytest
import torch

from multimodal_edema_prediction_amd import abi, build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    src = open(os.path.join(ROOT, "include", "medp_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    src = re.sub(r"typedef struct \{.*?\} \w+;", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(medp_\w+)\s*\(", src)))


def test_library_builds_and_exports_every_declared_symbol():
    build.build(verbose=False)
    L = ctypes.CDLL(abi.LIB_PATH)
    names = declared_functions()
    assert len(names) >= 20
    for n in names:
        assert hasattr(L, n), f"{n} declared in include/medp_hip.h but not exported"
    assert sorted(abi.SIGNATURES) == names, "abi.SIGNATURES and include/medp_hip.h disagree"


def test_version_and_arch():
    L = abi.lib()
    assert L.medp_version() >= 1
    assert L.medp_arch() == b"gfx950"


def test_invalid_arguments_are_reported_without_a_gpu():
    L = abi.lib()
    rc = L.medp_gemm_bf16_nt(None, None, None, 1, 1, 1, 1, 1, 1, None, None, None, 0, 0, 0, None)
    assert rc < 0 and b"null" in L.medp_last_error()
    with pytest.raises(ValueError):
        abi.check(rc, "gemm")


@pytest.mark.skipif(torch.cuda.is_available(), reason="CPU-only check")
def test_product_path_has_no_cpu_fallback():
    from multimodal_edema_prediction_amd import functional as Fn
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        Fn.to_bf16(torch.zeros(4, 8))
    from multimodal_edema_prediction_amd.cxr import CXREncoder, Dinov2Cfg
    enc = CXREncoder("synthetic", config=Dinov2Cfg(hidden_size=128, num_hidden_layers=1, num_attention_heads=2, image_size=56))
    with pytest.raises(RuntimeError):
        enc(torch.zeros(1, 3, 56, 56))


def test_cxr_encoder_raises_when_pretrained_weights_are_unavailable():
    """Reference behaviour (model file :137): a missing checkpoint is an error, never a silent random initialisation."""
    from multimodal_edema_prediction_amd.cxr import CXREncoder
    with pytest.raises(RuntimeError, match="not available locally"):
        CXREncoder("microsoft/rad-dino")
    enc = CXREncoder("synthetic")                      # the synthetic benchmark's seeded random weights stay available
    assert enc.d_out == 768
