"""ctypes binding of the C ABI declared in include/medp_hip.h (libmedp_hip.so, built in-tree by build.py).

The product path has NO fallback: if the library is missing or a tensor is not on the GPU the call raises.
Error mapping: rc < 0 (invalid argument) -> ValueError, rc > 0 (hipError_t) -> RuntimeError.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_char_p, c_float, c_int, c_longlong, c_size_t, c_uint, c_void_p

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MEDP_HIP_LIB") or os.path.join(_HERE, "libmedp_hip.so")   # override: the -DMEDP_V7_PHASE_TRACE profiling build

P, I, F, U, LL, SZ = c_void_p, c_int, c_float, c_uint, c_longlong, c_size_t


class MedpVitLayer(ctypes.Structure):
    _fields_ = [(n, P) for n in ("ln1_w", "ln1_b", "qkv_w", "qkv_b", "proj_w", "proj_b", "ls1", "ln2_w", "ln2_b",
                                 "fc1_w", "fc1_b", "fc2_w", "fc2_b", "ls2",
                                 "qkv_wg", "fc1_wg", "qkv_cs", "qkv_b2", "fc1_cs", "fc1_b2")]      # LayerNorm fold (include/medp_hip.h)


class MedpVitWeights(ctypes.Structure):
    _fields_ = ([(n, I) for n in ("hidden", "n_layers", "n_heads", "mlp_hidden", "patch", "pos_side", "patch_kpad")]
                + [("ln_eps", F)]
                + [(n, P) for n in ("patch_w", "patch_b", "cls", "pos", "final_ln_w", "final_ln_b")]
                + [("layers", ctypes.POINTER(MedpVitLayer))])


class MedpEncoderWeights(ctypes.Structure):
    """One x_transformers-style encoder (depth 1): see include/medp_hip.h"""
    _fields_ = [(n, P) for n in ("g_attn", "qkv_w", "out_w", "g_ff", "ff1_w", "ff1_b", "ff2_w", "ff2_b", "g_final")]


class MedpDuettWeights(ctypes.Structure):
    _fields_ = ([(n, I) for n in ("n_vars", "n_static", "d_embedding", "n_heads", "n_layers", "d_ff", "d_hidden_embed",
                                  "d_hidden_tab", "d_hidden_time", "n_obs_rows", "final_norm")]
                + [("norm_eps", F)]
                + [(n, P) for n in ("emb_w0", "emb_b0", "emb_bn_scale", "emb_bn_shift", "emb_w4", "emb_b4", "emb_l0", "emb_w4t", "n_obs_table",
                                    "tab_w0", "tab_b0", "tab_bn_scale", "tab_bn_shift", "tab_w4", "tab_b4", "special",
                                    "time_w0", "time_b0", "time_bn_scale", "time_bn_shift", "time_w3t", "time_b3",
                                    "rep_embedding", "event_embedding")]
                + [("event_enc", ctypes.POINTER(MedpEncoderWeights)), ("time_enc", ctypes.POINTER(MedpEncoderWeights))])


class MedpAdamTensor(ctypes.Structure):
    _fields_ = [("param", P), ("grad", P), ("exp_avg", P), ("exp_avg_sq", P), ("numel", LL), ("lr", F), ("weight_decay", F)]


class MedpOperandJob(ctypes.Structure):
    _fields_ = [("src", P), ("dst_plain", P), ("dst_t", P), ("rows", I), ("cols", I), ("ld_src", I), ("ld_plain", I), ("ld_t", I),
                ("reserved_", I)]


# name -> (restype, argtypes); must list every function declared in include/medp_hip.h (tests/test_abi.py checks)
SIGNATURES = {
    "medp_last_error": (c_char_p, []),
    "medp_version": (I, []),
    "medp_arch": (c_char_p, []),
    "medp_gemm_bf16_nt": (I, [P, P, P, I, I, I, I, I, I, P, P, P, I, I, I, P]),
    "medp_gemm_nt_workspace_bytes": (SZ, [I, I, I]),
    "medp_gemm_bf16_nt_ws": (I, [P, P, P, I, I, I, I, I, I, P, P, P, I, I, I, P, SZ, P]),
    "medp_gemm_f32_nt": (I, [P, P, P, I, I, I, I, I, I, P, P, P, I, I, P]),
    "medp_gemm_f32_tn": (I, [P, P, P, I, I, I, I, I, P]),
    "medp_gemm_tn_workspace_bytes": (SZ, [I, I, I]),
    "medp_gemm_bf16_tn": (I, [P, P, P, I, I, I, I, I, P, P]),
    "medp_gemm_persistent_cap": (I, [I]),
    "medp_gemm_profile_enable": (I, [I]),
    "medp_gemm_profile_collect": (I, [ctypes.POINTER(ctypes.c_double), ctypes.POINTER(c_longlong), ctypes.POINTER(ctypes.c_double)]),
    "medp_attn_fwd_dh64": (I, [P, P, P, P, I, I, I, I, I, I, I, F, P]),
    "medp_attn_dh16_fwd": (I, [P, I, P, I, I, I, I, I, F, P]),
    "medp_attn_dh16_train_supported": (I, [I, I, I, I, I, I]),
    "medp_attn_dh16_train_fwd": (I, [P, I, P, I, P, I, I, I, I, I, F, F, U, U, P]),
    "medp_attn_dh16_train_bwd": (I, [P, I, P, I, P, P, P, I, I, I, I, I, I, F, F, U, U, P]),
    "medp_attn_fwd_dh64_lse": (I, [P, P, P, P, P, I, I, I, I, I, I, I, F, P]),
    "medp_attn_bwd_dh64_prep": (I, [P, I, P, I, P, I, P, I, I, I, P]),
    "medp_attn_bwd_dh64": (I, [P, P, P, I, P, I, P, P, P, P, P, I, I, I, I, F, P]),
    "medp_attn_small_fwd": (I, [P, I, LL, P, P, I, LL, P, I, I, P, I, I, I, I, I, F, F, U, U, P]),
    "medp_attn_small_bwd": (I, [P, I, P, I, LL, P, P, I, LL, P, I, P, I, P, I, LL, I, I, I, I, I, F, F, U, U, P]),
    "medp_layernorm_fwd": (I, [P, I, P, P, P, I, I, P, P, I, I, F, P]),
    "medp_colsum_workspace_bytes": (SZ, [I, I]),
    "medp_layernorm_bwd": (I, [P, I, P, I, P, P, P, P, I, I, P, P, P, I, I, P]),
    "medp_colsum_f32": (I, [P, I, P, P, I, I, P]),
    "medp_scalenorm_fwd": (I, [P, I, P, P, I, I, P, I, I, F, P]),
    "medp_scalenorm_bwd": (I, [P, I, P, I, P, P, P, I, I, P, P, I, I, P]),
    "medp_scalenorm_bwd_add": (I, [P, I, P, I, P, P, P, I, P, I, P, P, I, I, P]),
    "medp_cast_f32_bf16": (I, [P, I, P, I, I, I, P]),
    "medp_transpose_to_bf16": (I, [P, I, I, P, I, I, I, P]),
    "medp_weight_operands_multi": (I, [P, P, P, I, P]),
    "medp_gelu_bwd": (I, [P, P, P, LL, P]),
    "medp_gelu_bf16_fwd": (I, [P, P, LL, P]),
    "medp_gelu_bf16_bwd": (I, [P, P, P, LL, P]),
    "medp_im2col_patch": (I, [P, P, I, I, I, I, I, I, P]),
    "medp_vit_assemble": (I, [P, P, P, P, I, I, I, P]),
    "medp_pos_embed_bicubic": (I, [P, P, I, I, I, I, P]),
    "medp_pos_embed_bicubic_bwd": (I, [P, P, I, I, I, I, P]),
    "medp_vit_workspace_bytes": (SZ, [ctypes.POINTER(MedpVitWeights), I, I, I]),
    "medp_vit_forward": (I, [ctypes.POINTER(MedpVitWeights), P, I, I, I, P, P, P, SZ, P]),
    "medp_vit_forward_part": (I, [ctypes.POINTER(MedpVitWeights), P, I, I, I, P, P, P, SZ, I, I, P]),
    "medp_duett_embed_fwd": (I, [ctypes.POINTER(MedpDuettWeights), P, P, P, I, I, P, P, P, P, P, I, P]),
    "medp_duett_swap_add_norm": (I, [P, P, P, P, LL, P, F, P, P, I, I, I, I, P]),
    "medp_feats_to_input": (I, [P, P, LL, P, P, LL, P, I, P, P, P, P, I, I, I, I, I, F, F, U, U, P]),
    "medp_ssl_mask_batch": (I, [P, P, P, P, P, P, P, P, P, I, I, I, P]),
    "medp_duett_workspace_bytes": (SZ, [ctypes.POINTER(MedpDuettWeights), I, I]),
    "medp_duett_encode": (I, [ctypes.POINTER(MedpDuettWeights), P, P, P, I, I, P, P, P, P, SZ, P]),
    "medp_gelu_dropout_fwd": (I, [P, P, LL, F, U, U, P]),
    "medp_gelu_dropout_bwd": (I, [P, P, P, LL, F, U, U, P]),
    "medp_gelu_dropout_fwd_bf16": (I, [P, P, LL, F, U, U, P]),
    "medp_gelu_dropout_bwd_bf16": (I, [P, P, P, P, LL, F, U, U, P]),
    "medp_dropout_add": (I, [P, P, P, LL, F, U, U, P]),
    "medp_rowdot_fwd": (I, [P, I, P, P, P, I, I, P]),
    "medp_rowdot_bwd": (I, [P, P, I, P, P, P, P, I, I, P]),
    "medp_fusion_logits_fwd": (I, [P, P, P, P, P, P, P, P, P, P, I, I, P]),
    "medp_fusion_logits_bwd": (I, [P, P, P, P, P, P, P, P, P, P, P, P, I, I, P]),
    "medp_meanpool_fwd": (I, [P, P, I, I, I, I, P]),
    "medp_meanpool_bwd": (I, [P, P, I, I, I, I, P]),
    "medp_dual_pathology_loss": (I, [P, P, P, P, P, P, P, F, F, F, F, P, P, P, P, I, I, P]),
    "medp_student_kd_loss": (I, [P, P, P, F, F, F, P, P, I, P]),
    "medp_aux_residual_kl": (I, [P, P, P, P, F, P, P, I, P]),
    "medp_sq_mean": (I, [P, F, P, P, I, P]),
    "medp_masked_bce_global": (I, [P, P, P, P, P, I, P]),
    "medp_glinear_fwd": (I, [P, P, P, P, I, I, I, I, P]),
    "medp_glinear_bwd_workspace_bytes": (SZ, [I, I, I, I]),
    "medp_glinear_bwd": (I, [P, P, P, P, P, P, P, I, I, I, I, P]),
    "medp_gmlp_supported": (I, [I, I, I]),
    "medp_gmlp_workspace_bytes": (SZ, [I, I, I, I, I]),
    "medp_gmlp_fwd": (I, [P] * 12 + [I, I, I, I, I, F, F, I, P, P]),
    "medp_gmlp_bwd": (I, [P] * 16 + [I, I, I, I, I, F, I, P, P]),
    "medp_gbn_workspace_bytes": (SZ, [I, I, I]),
    "medp_gbn_fwd": (I, [P, P, P, P, P, P, P, P, I, I, I, F, F, I, P, P]),
    "medp_gbn_bwd": (I, [P, P, P, P, P, P, P, P, I, I, I, F, I, P, P]),
    "medp_act_fwd": (I, [P, P, LL, I, P]),
    "medp_act_bwd": (I, [P, P, P, LL, I, P]),
    "medp_embed_inputs_fwd": (I, [P, P, I, P, I, I, I, I, P]),
    "medp_embed_inputs_bwd_blocks": (I, [I, I, I]),
    "medp_embed_inputs_bwd": (I, [P, P, P, I, I, I, I, I, P]),
    "medp_psi_assemble_fwd": (I, [P, P, P, P, P, I, I, I, I, P]),
    "medp_psi_assemble_bwd_slices": (I, [I, I, I]),
    "medp_psi_assemble_bwd": (I, [P, P, P, P, P, I, I, I, I, P]),
    "medp_axis_swap": (I, [P, P, I, I, I, I, P]),
    "medp_axis_swap_add": (I, [P, P, P, P, I, I, I, I, I, P]),
    "medp_add_bcast": (I, [P, P, P, LL, I, I, P]),
    "medp_masked_mse": (I, [P, P, P, P, P, I, P]),
    "medp_bce_mean": (I, [P, P, P, P, P, I, P]),
    "medp_adamw_chunk_elems": (I, []),
    "medp_adamw_multi": (I, [P, P, P, I, F, F, F, I, P, F, P]),
    "medp_traj_features": (I, [P, P, I, I, I, P]),
    "medp_gru_fwd": (I, [P, P, P, P, P, P, I, I, I, P]),
    "medp_gru_bwd": (I, [P, P, P, P, P, P, P, P, I, I, I, P]),
    "medp_rng_set_epoch_ptr": (I, [P]),
    "medp_counter_advance": (I, [P, P]),
}

_lib = None


def lib() -> ctypes.CDLL:
    """Load libmedp_hip.so; raise (never fall back) when it is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"HIP extension {LIB_PATH} is missing: build it with `python -m multimodal_edema_prediction_amd.build` "
                "(hipcc --offload-arch=gfx950). There is no CPU fallback for the hot path.")
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype, fn.argtypes = res, args
        _lib = L
    return _lib


def check(rc: int, what: str = "") -> None:
    if rc == 0:
        return
    msg = lib().medp_last_error().decode("utf-8", "replace")
    if rc < 0:
        raise ValueError(f"{what}: {msg}" if what else msg)
    raise RuntimeError(f"{what}: HIP error {rc}: {msg}")


def stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def ptr(t: torch.Tensor | None) -> int | None:
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("libmedp_hip operates on GPU memory only (tensor is on %s); there is no CPU fallback" % t.device)
    return t.data_ptr()


def require_gpu() -> None:
    lib()
    if not torch.cuda.is_available():
        raise RuntimeError("no HIP device is visible: the MI355X hot path cannot run (there is no CPU fallback)")
