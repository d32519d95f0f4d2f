"""Host-side mirror of the reference's `models/main_architecture_duett.py` (live classes only):
`DuettFeatureExtractor`, `load_duett_backbone`, `CXREncoder`, `PatchDualPathologyPerceiver`, `_PerceiverBlock`,
`TeacherModel`, `StudentModel` — same names, constructor signatures, attribute / parameter names, output dict keys and
error behaviour, so `training_duett/{engine,trainer,evaluator}.py` run against it unchanged
(INTEGRATION.md shows the two-line module alias).  torch layers are parameter containers (identical state_dict keys and
default initialisation order); all arithmetic is HIP through `autograd_ops` / the whole-module C calls.

`TemporalPerceiver` and `PathologyPerceiver` are commented out in the reference at HEAD (model file :176-535) and are
deliberately absent here too: the reference trainer soft-imports them to None.  `DualPathologyPerceiver` (commented out
too, :659-741) IS built: the reference's student entry point cannot run without it (trainer.py:770-822 rebuilds a `dual`
teacher from `best.pt`) — restated from that source, pinned by a fixture the source itself produced
(tests/golden/make_golden_dual.py).
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn as nn

from . import autograd_ops as A
from . import functional as Fn
from .cxr import CXREncoder, Dinov2Cfg  # noqa: F401  (re-exported)
from .duett import DuettFeatureExtractor, load_duett_backbone  # noqa: F401  (re-exported)
from .trajectory import LocalTrajectoryEncoder  # noqa: F401  (re-exported; model file :1242-1391)

__all__ = ["DuettFeatureExtractor", "load_duett_backbone", "CXREncoder", "PatchDualPathologyPerceiver", "DualPathologyPerceiver",
           "_PerceiverBlock", "TeacherModel", "StudentModel"]

# dropout stream ids (one per dropout site; combined with a per-forward seed)
import os as _os

# 0 = one stream, 1 = whole TS half beside the CXR encoder, 2 = DuETT encoder only.  Unset: 1 when this forward runs the CXR encoder itself,
# 0 when the encoder's tokens are handed in (graph_step's pipelined step: the encoder is not in this forward, and a third branch beside the
# NEXT batch's encoder measured 2 % slower than none, profiles/r03_ab_experiments.txt)
_OVERLAP_ENV = _os.environ.get("MEDP_OVERLAP")
_OVERLAP_MODE = int(_OVERLAP_ENV) if _OVERLAP_ENV is not None else 1
_OVERLAP = _OVERLAP_MODE != 0
_SIDE_STREAMS: dict = {}
if _OVERLAP and hasattr(torch.autograd.graph, "set_warn_on_accumulate_grad_stream_mismatch"):
    # `shared_queries` feeds both halves of the forward, so its AccumulateGrad node legitimately sees producers on two streams
    torch.autograd.graph.set_warn_on_accumulate_grad_stream_mismatch(False)


def _side_stream(device):
    """One extra HIP stream per device for the time-series half of the teacher forward/backward."""
    if device.type != "cuda":
        return None
    key = device.index if device.index is not None else torch.cuda.current_device()
    if key not in _SIDE_STREAMS:
        from .streams import new_stream
        _SIDE_STREAMS[key] = new_stream(device)
    return _SIDE_STREAMS[key]


_SID = {"img_cross": 0, "img_self": 10, "ts_cross": 20, "ts_self": 30, "image_head": 40, "temporal_head": 41,
        "correction_head": 42, "residual_head": 43, "student_head": 50}


class _BroadcastRowsFn(torch.autograd.Function):
    """[L, D] -> [B, L, D] (the shared pathology queries, model :602-603); backward sums over the batch with the HIP column sum."""

    @staticmethod
    def forward(ctx, x, B):
        ctx.B = B
        return x.unsqueeze(0).expand(B, -1, -1).contiguous()

    @staticmethod
    def backward(ctx, dy):
        L, D = dy.shape[1], dy.shape[2]
        return Fn.colsum(dy.contiguous().view(ctx.B, L * D)).view(L, D), None


class _PerceiverBlock(nn.Module):
    """Mirror of model file :745-774."""
    _DEBUG_NORMS: bool = False

    def __init__(self, d: int, n_heads: int, dropout: float):
        super().__init__()
        self.norm_q = nn.LayerNorm(d)
        self.norm_kv = nn.LayerNorm(d)
        self.attn = nn.MultiheadAttention(d, n_heads, dropout=dropout, batch_first=True)
        self.norm_ff = nn.LayerNorm(d)
        self.ff = nn.Sequential(nn.Linear(d, d * 4), nn.GELU(), nn.Dropout(dropout), nn.Linear(d * 4, d), nn.Dropout(dropout))
        self._sid = 0

    def forward(self, latents, kv, return_attn: bool = False, *, _kv_skip: int = 0, _shared_q: Optional[torch.Tensor] = None,
                _self_attn: bool = False, _seed: Optional[int] = None):
        """latents [B, L, d]; kv [B, N(+skip), d].  `_shared_q` ([L, d]) marks latents as the batch-broadcast pathology
        queries so that norm_q / the Q projection run once instead of B times (identical values)."""
        d = latents.shape[-1]
        H = self.attn.num_heads
        p_attn = float(self.attn.dropout) if self.training else 0.0
        p_ff = float(self.ff[2].p) if self.training else 0.0
        p_ff2 = float(self.ff[4].p) if self.training else 0.0
        seed = (A.next_seed() if _seed is None else _seed) if (p_attn > 0 or p_ff > 0 or p_ff2 > 0) else 0
        W, b = self.attn.in_proj_weight, self.attn.in_proj_bias
        q_src = _shared_q if _shared_q is not None else latents
        qn = A.layer_norm(q_src, self.norm_q.weight, self.norm_q.bias, self.norm_q.eps, lowp=True)      # only the projections read these
        kn = A.layer_norm(kv, self.norm_kv.weight, self.norm_kv.bias, self.norm_kv.eps, lowp=True)
        Q, KV = A.in_proj(qn, kn, W, b, d)
        o, attn_w = A.attn_small(Q, KV, H, (d // H) ** -0.5, p_attn, seed, self._sid, _kv_skip, return_attn)
        latents = A.linear(o, self.attn.out_proj.weight, self.attn.out_proj.bias, residual=latents)
        h = A.layer_norm(latents, self.norm_ff.weight, self.norm_ff.bias, self.norm_ff.eps, lowp=True)
        h = A.linear(h, self.ff[0].weight, self.ff[0].bias)
        h = A.gelu_dropout(h, p_ff, seed, self._sid + 1, lowp=True)
        if p_ff2 > 0:
            y = A.linear(h, self.ff[3].weight, self.ff[3].bias)
            latents = A.dropout_add(y, latents, p_ff2, seed, self._sid + 2)
        else:
            latents = A.linear(h, self.ff[3].weight, self.ff[3].bias, residual=latents)
        if return_attn:
            return latents, attn_w
        return latents


class PatchDualPathologyPerceiver(nn.Module):
    """Mirror of model file :538-654 (construction order preserved: same default initialisation under a fixed seed)."""

    def __init__(self, n_pathologies: int, d_ts: int, d_latent: int = 256, n_heads: int = 4, dropout: float = 0.1,
                 head_hidden: int = 64, head_dropout: float = 0.1):
        super().__init__()
        self.n_pathologies = n_pathologies
        self.d_latent = d_latent
        self.d_ts = d_ts
        self.shared_queries = nn.Parameter(torch.randn(n_pathologies, d_latent) * 0.02)
        self.ts_proj = nn.Linear(d_ts, d_latent)
        self.img_cross = _PerceiverBlock(d_latent, n_heads, dropout)
        self.img_self = _PerceiverBlock(d_latent, n_heads, dropout)
        self.ts_cross = _PerceiverBlock(d_latent, n_heads, dropout)
        self.ts_self = _PerceiverBlock(d_latent, n_heads, dropout)
        for name in ("img_cross", "img_self", "ts_cross", "ts_self"):
            getattr(self, name)._sid = _SID[name]

        def _mk_head():
            return nn.Sequential(nn.Linear(d_latent, head_hidden), nn.GELU(), nn.Dropout(head_dropout), nn.Linear(head_hidden, 1))

        self.image_head = _mk_head()
        self.temporal_head = _mk_head()
        self.correction_head = nn.Sequential(nn.LayerNorm(d_latent), nn.Linear(d_latent, head_hidden), nn.GELU(),
                                             nn.Dropout(head_dropout), nn.Linear(head_hidden, 1, bias=False))
        nn.init.zeros_(self.correction_head[-1].weight)
        self.beta = nn.Parameter(torch.ones(n_pathologies))
        self.image_label_bias = nn.Parameter(torch.zeros(n_pathologies))
        self.temporal_label_bias = nn.Parameter(torch.zeros(n_pathologies))

    def _head(self, x, seq, seed, sid):
        """Linear(d,h) -> GELU -> Dropout -> Linear(h,1)   (model :572-575)"""
        p = float(seq[2].p) if seq[2].training else 0.0
        h = A.linear(x, seq[0].weight, seq[0].bias)
        h = A.gelu_dropout(h, p, seed, sid)
        return A.rowdot(h, seq[3].weight, seq[3].bias)

    def forward(self, ts_tokens, img_patches_proj, return_attn=False, ts_ablation="hourly_only", *, _img_skip: int = 0):
        ts_selected = self._select_ts(ts_tokens, ts_ablation)
        B = ts_tokens.size(0)
        seed = A.next_seed() if self.training else 0
        q0 = _BroadcastRowsFn.apply(self.shared_queries, B)             # img_q == ts_q (model :602-603)
        ts = self._ts_branch(ts_selected, q0, seed, return_attn)
        im = self._img_branch(img_patches_proj, q0, seed, return_attn, _img_skip)
        return self._fuse(im, ts, return_attn)

    # The image and time-series halves of the forward only meet in `_fuse`; TeacherModel runs `_ts_branch` (with the DuETT
    # encoder in front of it) on a second HIP stream next to the CXR encoder's GEMMs.  Same arithmetic either way.
    def _ts_branch(self, ts_selected, q0, seed, return_attn):
        ts_kv = A.linear(ts_selected, self.ts_proj.weight, self.ts_proj.bias)
        T_tok, ts_attn = self.ts_cross(q0, ts_kv, True, _shared_q=self.shared_queries, _seed=seed) \
            if return_attn else (self.ts_cross(q0, ts_kv, _shared_q=self.shared_queries, _seed=seed), None)
        T_tok = self.ts_self(T_tok, T_tok, _seed=seed)
        ht = self._head(T_tok, self.temporal_head, seed, _SID["temporal_head"])
        ch = self.correction_head
        c = A.layer_norm(T_tok, ch[0].weight, ch[0].bias, ch[0].eps)
        c = A.linear(c, ch[1].weight, ch[1].bias)
        c = A.gelu_dropout(c, float(ch[3].p) if ch[3].training else 0.0, seed, _SID["correction_head"])
        ts_correction = A.rowdot(c, ch[4].weight, None)
        return {"T_tok": T_tok, "ht": ht, "ts_correction": ts_correction, "ts_attn": ts_attn}

    def _img_branch(self, img_patches_proj, q0, seed, return_attn, _img_skip):
        I, img_attn = self.img_cross(q0, img_patches_proj, True, _kv_skip=_img_skip, _shared_q=self.shared_queries, _seed=seed) \
            if return_attn else (self.img_cross(q0, img_patches_proj, _kv_skip=_img_skip, _shared_q=self.shared_queries, _seed=seed), None)
        I = self.img_self(I, I, _seed=seed)
        hi = self._head(I, self.image_head, seed, _SID["image_head"])
        return {"I": I, "hi": hi, "img_attn": img_attn}

    def _fuse(self, im, ts, return_attn):
        I, T_tok, ts_correction = im["I"], ts["T_tok"], ts["ts_correction"]
        img_logits, ts_logits, scaled_correction, fusion_logits = A.FusionLogitsFn.apply(
            im["hi"], ts["ht"], ts_correction, self.image_label_bias, self.temporal_label_bias, self.beta)
        out = {"img_logits": img_logits, "ts_logits": ts_logits, "fusion_logits": fusion_logits, "img_tokens": I,
               "ts_tokens": T_tok, "fusion_tokens": T_tok, "ts_correction": ts_correction,
               "scaled_correction": scaled_correction}
        if return_attn:
            out["img_attn"] = im["img_attn"]
            out["ts_attn"] = ts["ts_attn"]
        return out

    def _select_ts(self, ts_tokens, ts_ablation):
        if ts_tokens.ndim != 3:
            raise ValueError(f"ts_tokens must be [B, T+1, d_ts], got {tuple(ts_tokens.shape)}")
        if ts_ablation == "full":
            return ts_tokens
        if ts_ablation == "hourly_only":
            return ts_tokens[:, :-1, :]
        if ts_ablation == "rep_only":
            return ts_tokens[:, -1:, :]
        raise ValueError(f"unknown ts_ablation={ts_ablation!r}; expected one of "
                         "{'full', 'hourly_only', 'rep_only'}")


class DualPathologyPerceiver(nn.Module):
    """Mirror of the reference's `DualPathologyPerceiver` (model file :659-741, commented out at HEAD; the student entry point
    needs it): image branch = K logits of a frozen pretrained CXR head, injected by TeacherModel; temporal branch = pathology
    queries x ts tokens (cross + self block) -> one MLP head per pathology for the TS-only logits and one for the residual;
    `fusion_logit[k] = img_logit[k] + residual_head_k(T[:, k])`.  The 2K per-pathology heads run as TWO grouped launches each
    (group = pathology) instead of 2K x 2 small Linears."""

    def __init__(self, n_pathologies: int, d_ts: int, d_latent: int = 256, n_heads: int = 4, dropout: float = 0.1,
                 head_hidden: int = 64, head_dropout: float = 0.1):
        super().__init__()
        self.n_pathologies = n_pathologies
        self.d_latent = d_latent
        self.temporal_queries = nn.Parameter(torch.randn(n_pathologies, d_latent) * 0.02)
        self.ts_proj = nn.Linear(d_ts, d_latent)
        self.ts_cross = _PerceiverBlock(d_latent, n_heads, dropout)
        self.ts_self = _PerceiverBlock(d_latent, n_heads, dropout)
        self.ts_cross._sid, self.ts_self._sid = _SID["ts_cross"], _SID["ts_self"]

        def _mk_head():
            return nn.Sequential(nn.Linear(d_latent, head_hidden), nn.GELU(), nn.Dropout(head_dropout), nn.Linear(head_hidden, 1))

        self.temporal_heads = nn.ModuleList([_mk_head() for _ in range(n_pathologies)])
        self.residual_heads = nn.ModuleList([_mk_head() for _ in range(n_pathologies)])

    def _grouped_heads(self, T_kbd, heads, seed, sid):
        """T_kbd [K, B, d] -> [B, K]: head k on rows of pathology k."""
        from .duett_train import GLinearFn
        K, B, _ = T_kbd.shape
        p = float(heads[0][2].p) if heads[0][2].training else 0.0
        h = GLinearFn.apply(T_kbd, torch.stack([m[0].weight for m in heads]), torch.stack([m[0].bias for m in heads]))
        h = A.gelu_dropout(h, p, seed, sid)
        z = GLinearFn.apply(h, torch.stack([m[3].weight for m in heads]), torch.stack([m[3].bias for m in heads]))     # [K, B, 1]
        return z.view(K, B).t().contiguous()                     # [B, K] (7 x B scalars: layout plumbing)

    def forward(self, ts_tokens, img_logits, return_attn: bool = False, ts_ablation: str = "hourly_only") -> dict:
        from .duett_train import AxisSwapFn
        ts_selected = PatchDualPathologyPerceiver._select_ts(self, ts_tokens, ts_ablation)
        B = ts_tokens.size(0)
        seed = A.next_seed() if self.training else 0
        ts_kv = A.linear(ts_selected, self.ts_proj.weight, self.ts_proj.bias)
        q0 = _BroadcastRowsFn.apply(self.temporal_queries, B)
        T_tok, ts_attn = self.ts_cross(q0, ts_kv, True, _shared_q=self.temporal_queries, _seed=seed) if return_attn else \
            (self.ts_cross(q0, ts_kv, _shared_q=self.temporal_queries, _seed=seed), None)
        T_tok = self.ts_self(T_tok, T_tok, _seed=seed)                                   # [B, K, d]
        K, d = T_tok.shape[1], T_tok.shape[2]
        T_kbd = AxisSwapFn.apply(T_tok.reshape(1, B, K, d)).view(K, B, d)
        ts_logits = self._grouped_heads(T_kbd, self.temporal_heads, seed, _SID["temporal_head"])
        residuals = self._grouped_heads(T_kbd, self.residual_heads, seed, _SID["residual_head"])
        fusion_logits = A.dropout_add(residuals, img_logits.detach(), 0.0, 0, 0)          # plain add; img_logits: passthrough, no gradient
        out = {"img_logits": img_logits, "ts_logits": ts_logits, "fusion_logits": fusion_logits, "ts_tokens": T_tok,
               "residuals": residuals}
        if return_attn:
            out["ts_attn"] = ts_attn
        return out


class TeacherModel(nn.Module):
    """Mirror of model file :993-1197.  The live configuration (`patch_dual_pathology_mode=True`, run_duett.sh:8) is built;
    the other modes need perceiver classes that are commented out in the reference and raise here."""

    def __init__(self, duett_backbone: DuettFeatureExtractor, cxr_encoder: CXREncoder, perceiver, head_hidden: int = 128,
                 head_dropout: float = 0.1, cxr_return_patches: bool = True, d_img: int = 768, use_aux_cxr: bool = True,
                 aux_head_hidden: int = 128, pathology_mode: bool = False, dual_pathology_mode: bool = False,
                 patch_dual_pathology_mode: bool = False, pretrained_cxr_head_ckpt: Optional[str] = None,
                 pathology_labels: Optional[tuple] = None):
        super().__init__()
        n_modes = sum([pathology_mode, dual_pathology_mode, patch_dual_pathology_mode])
        if n_modes > 1:
            raise ValueError("pathology_mode / dual_pathology_mode / patch_dual_pathology_mode: at most one may be True")
        self.duett = duett_backbone
        self.cxr = cxr_encoder
        self.perceiver = perceiver
        self.cxr_return_patches = cxr_return_patches
        self.pathology_mode = pathology_mode
        self.dual_pathology_mode = dual_pathology_mode
        self.patch_dual_pathology_mode = patch_dual_pathology_mode
        d = perceiver.d_latent
        self.img_proj = nn.Linear(d_img, d)
        if pathology_mode or dual_pathology_mode or patch_dual_pathology_mode:
            self.head = None
            self.aux_cxr_head = None
            self.use_aux_cxr = False
        else:
            self.head = nn.Sequential(nn.Linear(d, head_hidden), nn.GELU(), nn.Dropout(head_dropout), nn.Linear(head_hidden, 1))
            self.use_aux_cxr = use_aux_cxr
            if use_aux_cxr:
                self.aux_cxr_head = nn.Sequential(nn.Linear(d, aux_head_hidden), nn.GELU(), nn.Dropout(head_dropout),
                                                  nn.Linear(aux_head_hidden, 1))
        if dual_pathology_mode:                                             # model file :1047-1071
            if pretrained_cxr_head_ckpt is None or pathology_labels is None:
                raise ValueError("dual_pathology_mode requires pretrained_cxr_head_ckpt AND pathology_labels")
            # the linear-probe checkpoint (cxr_linear_training.ipynb :827-845) is plain data: loaded without unpickling code
            state = torch.load(pretrained_cxr_head_ckpt, map_location="cpu", weights_only=True)
            num_pretrained = int(state["num_classes"])
            pretrained_labels = list(state["label_cols"])
            missing = [lbl for lbl in pathology_labels if lbl not in pretrained_labels]
            if missing:
                raise ValueError(f"pathology_labels missing from the pretrained CXR head: {missing}. pretrained labels: {pretrained_labels}")
            keep_idx = [pretrained_labels.index(lbl) for lbl in pathology_labels]
            self.pretrained_cxr_head = nn.Linear(d_img, num_pretrained)
            clf_sd = state["classifier_state_dict"]
            self.pretrained_cxr_head.load_state_dict({"weight": clf_sd["1.weight"], "bias": clf_sd["1.bias"]})
            for p in self.pretrained_cxr_head.parameters():
                p.requires_grad = False
            self.register_buffer("cxr_head_keep_idx", torch.tensor(keep_idx, dtype=torch.long))

    def _dual_forward(self, duett_in, pixel_values, return_attn):
        """model file :1132-1150: CLS -> frozen pretrained head -> the K kept columns -> DualPathologyPerceiver.  Only the kept
        rows of the head are multiplied (the same numbers as computing all C columns and gathering K of them)."""
        ts_tokens = self.duett.encode(duett_in)
        cls = self.cxr(pixel_values)
        if isinstance(cls, tuple):
            cls = cls[0]
        with torch.no_grad():
            head = self.pretrained_cxr_head
            K = self.cxr_head_keep_idx.numel()
            key = (head.weight.data_ptr(), head.weight._version, head.bias._version, self.cxr_head_keep_idx._version)
            if getattr(self, "_kept_head", (None,))[0] != key:
                Kp = (K + 3) // 4 * 4                                       # the GEMM wants N in multiples of 4: zero rows appended
                W, b = head.weight.new_zeros((Kp, head.in_features)), head.bias.new_zeros(Kp)
                W[:K], b[:K] = head.weight[self.cxr_head_keep_idx], head.bias[self.cxr_head_keep_idx]
                self._kept_head = (key, W, b)
            _, W, b = self._kept_head
            img_logits = A.linear(cls.contiguous(), W, b)[:, :K].contiguous()
        out = self.perceiver(ts_tokens, img_logits, return_attn=return_attn)
        result = {"main_logit": out["fusion_logits"][:, 0], "img_logits": out["img_logits"], "ts_logits": out["ts_logits"],
                  "fusion_logits": out["fusion_logits"]}
        if return_attn:
            result["ts_tokens"], result["ts_attn"], result["residuals"] = out["ts_tokens"], out["ts_attn"], out["residuals"]
        return result

    def forward(self, x_ts_list, x_static_list, bin_ends_list, pixel_values: torch.Tensor, batch_size: Optional[int] = None,
                return_attn: bool = False, *, _cxr_tokens16: Optional[torch.Tensor] = None, _overlap: Optional[bool] = None):
        """`_cxr_tokens16` (bf16 [B, P+1, d_img], private): tokens of the FROZEN CXR encoder for `pixel_values`, computed
        ahead of time — graph_step.py runs the encoder for the next batch beside this batch's training step.
        `_overlap=False` (private): keep the whole forward on the current stream — needed when this forward is itself a forked
        branch of a graph capture (the frozen KD teacher beside the student's step): a fork nested inside a forked stream makes
        hipStreamEndCapture crash on this ROCm (segmentation fault in capture_end)."""
        if batch_size is None:
            batch_size = pixel_values.shape[0]
        x = (x_ts_list, x_static_list, bin_ends_list)
        if not (self.patch_dual_pathology_mode or self.dual_pathology_mode):
            raise NotImplementedError("only the patch_dual (live at HEAD) and dual (needed by the student entry point) perceiver modes "
                                      "are built (SURVEY.md F6, 8(f2))")
        duett_in = self.duett.feats_to_input(x, batch_size)
        if self.dual_pathology_mode:
            return self._dual_forward(duett_in, pixel_values, return_attn)
        pc = self.perceiver
        B = pixel_values.shape[0]
        seed = A.next_seed() if pc.training else 0
        q0 = _BroadcastRowsFn.apply(pc.shared_queries, B)                   # shared by both branches (model :602-603)
        # ---- time-series half on a side stream: DuETT encoder + ts_proj / ts_cross / ts_self / temporal + correction heads.
        # It is hundreds of short launches that do not depend on the image; next to the CXR encoder's one-workgroup-per-CU
        # GEMMs they fill otherwise idle issue slots.  autograd replays each node's backward on its forward stream, so the
        # two halves of the backward overlap the same way.  MEDP_OVERLAP=0 runs everything on one stream.
        cur = torch.cuda.current_stream()
        mode = _OVERLAP_MODE if (_OVERLAP_ENV is not None or _cxr_tokens16 is None) else 0
        side = _side_stream(pixel_values.device) if (mode != 0 and _overlap is not False) else None
        if side is not None:
            from .streams import fork_guard
            fork_guard("TeacherModel.forward", pixel_values.device)         # raises instead of a crash in capture_end (nested fork)
            side.wait_stream(cur)
            for t in (q0,) + tuple(duett_in):
                if isinstance(t, torch.Tensor) and t.is_cuda:
                    t.record_stream(side)
            with torch.cuda.stream(side):
                ts_tokens = self.duett.encode(duett_in)                     # [B, T+1, D]
                if mode == 1:
                    ts = pc._ts_branch(pc._select_ts(ts_tokens, "hourly_only"), q0, seed, return_attn)
            if mode != 1:                                                   # encoder only on the side stream
                cur.wait_stream(side)
                ts_tokens.record_stream(cur)
                ts = pc._ts_branch(pc._select_ts(ts_tokens, "hourly_only"), q0, seed, return_attn)
                side = None
        else:
            ts_tokens = self.duett.encode(duett_in)
            ts = pc._ts_branch(pc._select_ts(ts_tokens, "hourly_only"), q0, seed, return_attn)
        # ---- image half: CXR tokens after the final LayerNorm as bf16 [B, P+1, d_img]; the class row is skipped inside the
        # cross-attention
        if _cxr_tokens16 is not None:
            if any(p.requires_grad for p in self.cxr.parameters()):
                raise ValueError("_cxr_tokens16 is only valid with a frozen CXR encoder")
            tokens16 = _cxr_tokens16
        else:
            tokens16 = self.cxr.forward_bf16(pixel_values)
        img_proj_full = A.linear(tokens16, self.img_proj.weight, self.img_proj.bias)   # [B, P+1, d]
        im = pc._img_branch(img_proj_full, q0, seed, return_attn, 1)
        if side is not None:
            cur.wait_stream(side)
            for t in ts.values():
                if isinstance(t, torch.Tensor):
                    t.record_stream(cur)
        out = pc._fuse(im, ts, return_attn)
        result = {"main_logit": out["fusion_logits"][:, 0], "img_logits": out["img_logits"], "ts_logits": out["ts_logits"],
                  "fusion_logits": out["fusion_logits"], "ts_correction": out["ts_correction"],
                  "scaled_correction": out["scaled_correction"]}
        if return_attn:
            for k in ("img_tokens", "ts_tokens", "fusion_tokens", "img_attn", "ts_attn"):
                result[k] = out[k]
        return result


class StudentModel(nn.Module):
    """Mirror of model file :1202-1235: DuETT(TS) + MLP head, backbone trained end-to-end."""

    def __init__(self, duett_backbone: DuettFeatureExtractor, pool: str = "mean", head_hidden: int = 128, head_dropout: float = 0.1):
        super().__init__()
        self.duett = duett_backbone
        self.pool = pool
        d_rep = duett_backbone.d_representation
        self.head = nn.Sequential(nn.Linear(d_rep, head_hidden), nn.GELU(), nn.Dropout(head_dropout), nn.Linear(head_hidden, 1))

    def forward(self, x_ts_list, x_static_list, bin_ends_list, batch_size: Optional[int] = None) -> torch.Tensor:
        if batch_size is None:
            batch_size = len(x_ts_list)
        x = (x_ts_list, x_static_list, bin_ends_list)
        duett_in = self.duett.feats_to_input(x, batch_size)
        ts_tokens = self.duett.encode(duett_in)                             # [B, T+1, d_rep]
        if self.pool == "rep_token":
            feat = ts_tokens[:, -1, :]
        elif self.pool == "mean":
            feat = A.MeanPoolFn.apply(ts_tokens, ts_tokens.shape[1] - 1)    # REP token excluded
        else:
            raise ValueError(f"unknown pool: {self.pool}")
        p = float(self.head[2].p) if self.training else 0.0
        h = A.linear(feat, self.head[0].weight, self.head[0].bias)
        h = A.gelu_dropout(h, p, A.next_seed() if p > 0 else 0, _SID["student_head"])
        return A.rowdot(h, self.head[3].weight, self.head[3].bias)
