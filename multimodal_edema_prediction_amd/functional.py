"""Tensor-level wrappers over the C ABI (abi.py): allocate outputs with torch (plumbing), pass raw device
pointers + the current HIP stream, map return codes to exceptions.  No arithmetic happens in Python."""
from __future__ import annotations

import os

import torch

from . import abi
from .abi import check, lib, ptr, stream

BF16 = torch.bfloat16
F32 = torch.float32

# ---- kernel precision mode ------------------------------------------------------------------------------------------------
# "bf16" (default): GEMM operands are bf16 (MFMA), accumulation / norms / softmax / losses fp32 — the mode every throughput figure
# is measured in.  "fp32": the op-level (trainable / autograd) paths keep fp32 operands end to end and run medp_gemm_f32_* — the
# tight-parity instrument of SURVEY.md §7 / §8(d) (logits 1e-4, loss 1e-5, gradients element-wise).  The switch is explicit
# (set_precision / the precision_mode context / MEDP_PRECISION), never inferred from tensor dtypes (§8b).  The whole-module C
# calls of the FROZEN encoders (medp_vit_forward, medp_duett_encode) are bf16 by construction; in fp32 mode the DuETT backbone
# runs its op-level form instead, the CXR encoder has no fp32 form.
_PRECISION = os.environ.get("MEDP_PRECISION", "bf16")


def precision() -> str:
    return _PRECISION


def set_precision(p: str) -> None:
    global _PRECISION
    if p not in ("bf16", "fp32"):
        raise ValueError(f"precision must be 'bf16' or 'fp32', got {p!r}")
    _PRECISION = p


class precision_mode:
    def __init__(self, p):
        self.p = p

    def __enter__(self):
        self.prev = precision()
        set_precision(self.p)

    def __exit__(self, *a):
        set_precision(self.prev)


def _2d(t: torch.Tensor) -> torch.Tensor:
    return t.reshape(-1, t.shape[-1])


def _ld(t: torch.Tensor) -> int:
    assert t.dim() == 2 and t.stride(1) == 1, "row-major 2-D view required"
    return t.stride(0)


def operand(x: torch.Tensor) -> torch.Tensor:
    """An activation / gradient as a GEMM operand of the op-level paths: bf16 (HIP cast kernel); in fp32 mode the tensor itself."""
    if _PRECISION == "fp32":
        return x.contiguous()
    return x.contiguous() if x.dtype == BF16 else to_bf16(x)          # already an operand (a producer wrote bf16 directly)


def operand_t(x: torch.Tensor) -> torch.Tensor:
    """[R, C] -> the transposed GEMM operand [C, Rpad8]: bf16, or fp32 in fp32 mode (same zero-padded shape)."""
    if _PRECISION != "fp32":
        return transpose_to_bf16(x)
    x2 = _2d(x)
    R, C = x2.shape
    y = torch.zeros((C, (R + 7) // 8 * 8), dtype=F32, device=x.device)
    y[:, :R] = x2.to(F32).t()
    return y


def to_bf16(x: torch.Tensor) -> torch.Tensor:
    """fp32 [.., C] -> bf16 (HIP cast kernel), whatever the precision mode (the frozen whole-module paths are bf16 by construction)."""
    x2 = _2d(x)
    y = torch.empty(x2.shape, dtype=BF16, device=x.device)
    check(lib().medp_cast_f32_bf16(ptr(x2), _ld(x2), ptr(y), y.stride(0), x2.shape[0], x2.shape[1], stream()), "cast")
    return y.view(x.shape)


def transpose_to_bf16(x: torch.Tensor) -> torch.Tensor:
    """[R, C] fp32|bf16 -> [C, Rpad] bf16 with Rpad = R rounded up to 8 (zero filled) so it can be a GEMM operand."""
    x2 = _2d(x)
    R, C = x2.shape
    Rp = (R + 7) // 8 * 8
    y = torch.zeros((C, Rp), dtype=BF16, device=x.device) if Rp != R else torch.empty((C, Rp), dtype=BF16, device=x.device)
    check(lib().medp_transpose_to_bf16(ptr(x2), int(x2.dtype == BF16), _ld(x2), ptr(y), Rp, R, C, stream()), "transpose")
    return y


def gemm(a: torch.Tensor, w: torch.Tensor, bias=None, scale=None, residual=None, act: int = 0,
         out_dtype=F32, out: torch.Tensor | None = None, k: int | None = None) -> torch.Tensor:
    """out[M,N] = epi(a[M,K] @ w[N,K]^T); a, w bf16 row-major (leading dims may exceed K); fp32 operands (fp32 mode): medp_gemm_f32_nt."""
    if a.dtype == F32 and w.dtype == F32:
        a2, w2 = _2d(a), _2d(w)
        M, N = a2.shape[0], w2.shape[0]
        K = k if k is not None else min(a2.shape[1], w2.shape[1])
        if out is None:
            out = torch.empty((M, N), dtype=F32, device=a.device)
        assert out.dtype == F32, "fp32 mode produces fp32 results"
        r2 = _2d(residual) if residual is not None else None
        check(lib().medp_gemm_f32_nt(ptr(a2), ptr(w2), ptr(out), M, N, K, _ld(a2), _ld(w2), _ld(out), ptr(bias), ptr(scale), ptr(r2),
                                     _ld(r2) if r2 is not None else 0, act, stream()), "gemm_f32")
        return out
    assert a.dtype == BF16 and w.dtype == BF16
    a2, w2 = _2d(a), _2d(w)
    M, N = a2.shape[0], w2.shape[0]
    K = k if k is not None else min(a2.shape[1], w2.shape[1])
    if out is None:
        out = torch.empty((M, N), dtype=out_dtype, device=a.device)
    r2 = _2d(residual) if residual is not None else None
    wsb = lib().medp_gemm_nt_workspace_bytes(M, N, K)          # > 0: a small grid that is split along K (deterministic two-pass sum)
    ws = torch.empty(wsb // 4, dtype=F32, device=a.device) if wsb else None
    check(lib().medp_gemm_bf16_nt_ws(ptr(a2), ptr(w2), ptr(out), M, N, K, _ld(a2), _ld(w2), _ld(out), ptr(bias), ptr(scale),
                                      ptr(r2), _ld(r2) if r2 is not None else 0, act, int(out.dtype == BF16), ptr(ws), wsb, stream()), "gemm")
    return out


def gemm_tn(dy: torch.Tensor, x: torch.Tensor, out: torch.Tensor | None = None) -> torch.Tensor:
    """dW[N,K] = dy[M,N]^T @ x[M,K] (bf16 row-major operands, fp32 result) — the weight gradient of a Linear.
    `out`: optional contiguous [N, K] fp32 destination (e.g. a row block of a larger gradient)."""
    dy2, x2 = _2d(dy), _2d(x)
    M, N = dy2.shape
    K = x2.shape[1]
    assert x2.shape[0] == M
    if out is None:
        out = torch.empty((N, K), dtype=F32, device=dy.device)
    assert out.shape == (N, K) and out.dtype == F32 and out.is_contiguous()
    if dy.dtype == F32 and x.dtype == F32:
        check(lib().medp_gemm_f32_tn(ptr(dy2), ptr(x2), ptr(out), M, N, K, _ld(dy2), _ld(x2), stream()), "gemm_f32_tn")
        return out
    assert dy.dtype == BF16 and x.dtype == BF16
    wsb = lib().medp_gemm_tn_workspace_bytes(M, N, K)
    ws = torch.empty(wsb // 4, dtype=F32, device=dy.device) if wsb else None
    check(lib().medp_gemm_bf16_tn(ptr(dy2), ptr(x2), ptr(out), M, N, K, _ld(dy2), _ld(x2), ptr(ws), stream()), "gemm_tn")
    return out


def layernorm(x: torch.Tensor, w, b, eps: float, out_dtype=BF16, save_stats: bool = False):
    x2 = _2d(x)
    rows, D = x2.shape
    y = torch.empty((rows, D), dtype=out_dtype, device=x.device)
    mean = torch.empty(rows, dtype=F32, device=x.device) if save_stats else None
    rstd = torch.empty(rows, dtype=F32, device=x.device) if save_stats else None
    check(lib().medp_layernorm_fwd(ptr(x2), _ld(x2), ptr(w), ptr(b), ptr(y), D, int(out_dtype == BF16), ptr(mean), ptr(rstd),
                                    rows, D, eps, stream()), "layernorm_fwd")
    y = y.view(x.shape)
    return (y, mean, rstd) if save_stats else y


def layernorm_bwd(dy, x, w, mean, rstd, need_dx=True, need_dwdb=True):
    dy2, x2 = _2d(dy), _2d(x)
    rows, D = x2.shape
    dx = torch.empty((rows, D), dtype=F32, device=x.device) if need_dx else None
    dw = torch.empty(D, dtype=F32, device=x.device) if need_dwdb else None
    db = torch.empty(D, dtype=F32, device=x.device) if need_dwdb else None
    ws = torch.empty(lib().medp_colsum_workspace_bytes(rows, D) // 4, dtype=F32, device=x.device) if need_dwdb else None
    check(lib().medp_layernorm_bwd(ptr(dy2), _ld(dy2), ptr(x2), _ld(x2), ptr(w), ptr(mean), ptr(rstd), ptr(dx), D, 0, ptr(dw),
                                    ptr(db), ptr(ws), rows, D, stream()), "layernorm_bwd")
    return (dx.view(x.shape) if need_dx else None), dw, db


def colsum(x: torch.Tensor, out: torch.Tensor | None = None) -> torch.Tensor:
    x2 = _2d(x)
    rows, D = x2.shape
    if out is None:
        out = torch.empty(D, dtype=F32, device=x.device)
    assert out.shape == (D,) and out.dtype == F32 and out.is_contiguous()
    ws = torch.empty(lib().medp_colsum_workspace_bytes(rows, D) // 4, dtype=F32, device=x.device)
    check(lib().medp_colsum_f32(ptr(x2), _ld(x2), ptr(out), ptr(ws), rows, D, stream()), "colsum")
    return out


def scalenorm(x: torch.Tensor, g: torch.Tensor, eps: float = 1e-12, out_dtype=BF16, save_rnorm=False):
    x2 = _2d(x)
    rows, D = x2.shape
    y = torch.empty((rows, D), dtype=out_dtype, device=x.device)
    rn = torch.empty(rows, dtype=F32, device=x.device) if save_rnorm else None
    check(lib().medp_scalenorm_fwd(ptr(x2), _ld(x2), ptr(g), ptr(y), D, int(out_dtype == BF16), ptr(rn), rows, D, eps, stream()),
          "scalenorm_fwd")
    y = y.view(x.shape)
    return (y, rn) if save_rnorm else y


def scalenorm_bwd(dy, x, g, rnorm, need_dg=True):
    dy2, x2 = _2d(dy), _2d(x)
    rows, D = x2.shape
    dx = torch.empty((rows, D), dtype=F32, device=x.device)
    dg = torch.empty(1, dtype=F32, device=x.device) if need_dg else None
    ws = torch.empty(rows, dtype=F32, device=x.device) if need_dg else None
    check(lib().medp_scalenorm_bwd(ptr(dy2), _ld(dy2), ptr(x2), _ld(x2), ptr(g), ptr(rnorm), ptr(dx), D, 0, ptr(dg), ptr(ws),
                                    rows, D, stream()), "scalenorm_bwd")
    return dx.view(x.shape), dg


def attn_dh64(qkv: torch.Tensor, B: int, S: int, H: int, scale: float) -> torch.Tensor:
    """qkv bf16 [B*S, 3*H*64] (q | k | v column blocks) -> o bf16 [B*S, H*64]"""
    assert qkv.dtype == BF16 and qkv.dim() == 2 and qkv.shape == (B * S, 3 * H * 64)
    D = H * 64
    o = torch.empty((B * S, D), dtype=BF16, device=qkv.device)
    base = qkv.data_ptr()
    check(lib().medp_attn_fwd_dh64(base, base + 2 * D, base + 4 * D, ptr(o), B, S, H, 3 * D, 3 * D, 3 * D, D, scale, stream()),
          "attn_fwd_dh64")
    return o


def attn_dh64_lse(qkv: torch.Tensor, B: int, S: int, H: int, scale: float):
    """Training form of `attn_dh64`: also returns lse fp32 [B, H, S] (log2-domain logsumexp of the scaled scores)."""
    assert qkv.dtype == BF16 and qkv.dim() == 2 and qkv.shape == (B * S, 3 * H * 64)
    D = H * 64
    o = torch.empty((B * S, D), dtype=BF16, device=qkv.device)
    lse = torch.empty((B, H, S), dtype=F32, device=qkv.device)
    base = qkv.data_ptr()
    check(lib().medp_attn_fwd_dh64_lse(base, base + 2 * D, base + 4 * D, ptr(o), ptr(lse), B, S, H, 3 * D, 3 * D, 3 * D, D, scale,
                                       stream()), "attn_fwd_dh64_lse")
    return o, lse


def attn_dh64_bwd(dout: torch.Tensor, qkv: torch.Tensor, o: torch.Tensor, lse: torch.Tensor, B: int, S: int, H: int, scale: float):
    """dout fp32 [B*S, H*64], qkv / o bf16 as in the forward -> dqkv fp32 [B*S, 3*H*64] (dq | dk | dv column blocks)."""
    D = H * 64
    d2 = dout.reshape(B * S, D).contiguous()
    dob = torch.empty((B * S, D), dtype=BF16, device=qkv.device)
    dsum = torch.empty((B, H, S), dtype=F32, device=qkv.device)
    check(lib().medp_attn_bwd_dh64_prep(ptr(d2), D, ptr(o), D, ptr(dob), D, ptr(dsum), B, S, H, stream()), "attn_bwd_dh64_prep")
    dqkv = torch.empty((B * S, 3 * D), dtype=F32, device=qkv.device)
    base, dbase = qkv.data_ptr(), dqkv.data_ptr()
    check(lib().medp_attn_bwd_dh64(base, base + 2 * D, base + 4 * D, 3 * D, ptr(dob), D, ptr(lse), ptr(dsum), dbase, dbase + 4 * D,
                                   dbase + 8 * D, 3 * D, B, S, H, scale, stream()), "attn_bwd_dh64")
    return dqkv


def attn_small_fwd(q, k, v, B, Lq, Lk, H, dh, scale, *, q_batch_stride=None, kv_batch_stride=None, out_dtype=F32,
                   dropout_p=0.0, seed=0, stream_id=0, attn_avg=None):
    """q fp32 rows [.., H*dh] (ld = q.stride(-2)); k, v fp32 with a common row stride."""
    ldq, ldkv = q.stride(-2), k.stride(-2)
    assert v.stride(-2) == ldkv
    qbs = Lq * ldq if q_batch_stride is None else q_batch_stride
    kbs = Lk * ldkv if kv_batch_stride is None else kv_batch_stride
    o = torch.empty((B, Lq, H * dh), dtype=out_dtype, device=k.device)
    check(lib().medp_attn_small_fwd(ptr(q), ldq, qbs, ptr(k), ptr(v), ldkv, kbs, ptr(o), H * dh, int(out_dtype == BF16),
                                     ptr(attn_avg), B, Lq, Lk, H, dh, scale, dropout_p, seed, stream_id, stream()), "attn_small_fwd")
    return o


def attn_small_bwd(dout, q, k, v, B, Lq, Lk, H, dh, scale, *, q_batch_stride=None, kv_batch_stride=None, dropout_p=0.0,
                   seed=0, stream_id=0, dkv_out=None):
    """Returns (dq [B,Lq,D], dk, dv).  dk/dv are the two column halves of one [B, Lk, 2D] buffer (`dkv_out`, which may be a
    strided view, e.g. rows 1.. of a [B, Lk+1, 2D] tensor) so the fused K|V projection gets its gradient without a concat."""
    ldq, ldkv = q.stride(-2), k.stride(-2)
    qbs = Lq * ldq if q_batch_stride is None else q_batch_stride
    kbs = Lk * ldkv if kv_batch_stride is None else kv_batch_stride
    D = H * dh
    dq = torch.empty((B, Lq, D), dtype=F32, device=k.device)
    if dkv_out is None:
        dkv_out = torch.empty((B, Lk, 2 * D), dtype=F32, device=k.device)
    assert dkv_out.stride(-1) == 1 and dkv_out.shape[-1] == 2 * D
    d2 = dout.reshape(B * Lq, D)
    base = dkv_out.data_ptr()
    check(lib().medp_attn_small_bwd(ptr(d2), _ld(d2), ptr(q), ldq, qbs, ptr(k), ptr(v), ldkv, kbs, ptr(dq), D, base,
                                     dkv_out.stride(-2), base + 4 * D, 0, dkv_out.stride(0), B, Lq, Lk, H, dh, scale, dropout_p,
                                     seed, stream_id, stream()), "attn_small_bwd")
    return dq, dkv_out[..., :D], dkv_out[..., D:]


def gelu_bwd(dy: torch.Tensor, pre: torch.Tensor) -> torch.Tensor:
    dx = torch.empty_like(dy)
    check(lib().medp_gelu_bwd(ptr(dy), ptr(pre), ptr(dx), dy.numel(), stream()), "gelu_bwd")
    return dx
