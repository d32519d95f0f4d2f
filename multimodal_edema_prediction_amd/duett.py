"""DuETT backbone: host-side mirror of the reference's `duett/duett.py` `Model` and of
`DuettFeatureExtractor` / `load_duett_backbone` (models/main_architecture_duett.py:24-123).

The module tree (and therefore `state_dict()` / `named_parameters()`) is key-for-key the reference's:
`embedding_layers.{v}.{0,3.batch_norm,4}`, `tab_encoder.*`, `special_embeddings`, `n_obs_embedding`,
`event_transformers.{l}.*`, `time_transformers.{l}.*`, `full_event_embedding`, `full_time_embedding.{0,2.batch_norm,3}`,
`full_rep_embedding`, `head`, `pretrain_*_proj`, `predict_events_*_proj`, buffers `MASKED_EMBEDDING_KEY`,
`REPRESENTATION_EMBEDDING_KEY`.  The torch layers are PARAMETER CONTAINERS only: the arithmetic runs in
libmedp_hip (`medp_duett_encode` for the inference form; the op-level autograd path of `duett_train.py` when the
backbone is being trained).  There is no CPU path.
"""
from __future__ import annotations

import ctypes

import numpy as np
import torch
import torch.nn as nn

from . import abi
from . import functional as Fn
from .abi import MedpDuettWeights, MedpEncoderWeights, check, lib, ptr, stream

SCALENORM_EPS = 1e-12      # F.normalize default (x_transformers 2.x ScaleNorm)
FINAL_NORM = True          # x_transformers 1.x/2.x: final norm present when pre_norm=True (switch)


class BatchNormLastDim(nn.Module):
    """Parameter container for the reference's `BatchNormLastDim` (duett.py:11-22)."""

    def __init__(self, d, **kwargs):
        super().__init__()
        self.batch_norm = nn.BatchNorm1d(d, **kwargs)

    def folded(self):
        bn = self.batch_norm
        scale = bn.weight.detach() / torch.sqrt(bn.running_var + bn.eps)
        return scale, bn.bias.detach() - bn.running_mean * scale


def simple_mlp(d_in, d_out, n_hidden, d_hidden, final_activation=False, input_batch_norm=False, hidden_batch_norm=False,
               dropout=0.0, activation=nn.ReLU):
    """Same layer list (hence the same state_dict indices) as the reference's `simple_mlp` (duett.py:24-39)."""
    if n_hidden == 0:
        layers = ([BatchNormLastDim(d_in)] if input_batch_norm else []) + [nn.Linear(d_in, d_out)]
    else:
        layers = ([BatchNormLastDim(d_in)] if input_batch_norm else []) + \
                 [nn.Linear(d_in, d_hidden), activation(), nn.Dropout(dropout)] + \
                 [l for _ in range(n_hidden - 1) for l in ([BatchNormLastDim(d_hidden)] if hidden_batch_norm else []) +
                  [nn.Linear(d_hidden, d_hidden), activation(), nn.Dropout(dropout)]] + \
                 ([BatchNormLastDim(d_hidden)] if hidden_batch_norm else []) + [nn.Linear(d_hidden, d_out)]
    if final_activation:
        layers.append(activation())
    return nn.Sequential(*layers)


class Encoder(nn.Module):
    """Parameter container with the module tree of `x_transformers.Encoder(dim, depth=1, heads, pre_norm=True,
    use_scalenorm=True, attn_dim_head, ff_glu=False, ff_mult)` as constructed at duett.py:95-105
    (keys: layers.0.0.0.g, layers.0.1.to_{q,k,v,out}.weight, layers.1.0.0.g, layers.1.1.ff.0.0.*, layers.1.1.ff.2.*,
    final_norm.g).  x_transformers itself is unpinned/unavailable: see oracle/xt_encoder.py."""

    def __init__(self, dim, depth=1, heads=2, pre_norm=True, use_scalenorm=True, attn_dim_head=12, ff_glu=False, ff_mult=4,
                 attn_dropout=0.0, ff_dropout=0.0, **kw):
        super().__init__()
        if not (depth == 1 and pre_norm and use_scalenorm and not ff_glu):
            raise ValueError("only the DuETT encoder configuration (depth=1, pre_norm, scalenorm, no GLU) is built")
        self.dim, self.heads, self.dim_head = dim, heads, attn_dim_head
        self.dropout = float(attn_dropout)
        inner = int(dim * ff_mult)
        hd = heads * attn_dim_head

        class _G(nn.Module):
            def __init__(s):
                super().__init__()
                s.g = nn.Parameter(torch.ones(1))

        class _Attn(nn.Module):
            def __init__(s):
                super().__init__()
                s.to_q = nn.Linear(dim, hd, bias=False)
                s.to_k = nn.Linear(dim, hd, bias=False)
                s.to_v = nn.Linear(dim, hd, bias=False)
                s.to_out = nn.Linear(hd, dim, bias=False)

        class _FF(nn.Module):
            def __init__(s):
                super().__init__()
                s.ff = nn.Sequential(nn.Sequential(nn.Linear(dim, inner), nn.GELU()), nn.Dropout(ff_dropout), nn.Linear(inner, dim))

        norms = lambda: nn.ModuleList([_G(), nn.Identity(), nn.Identity()])
        self.layers = nn.ModuleList([nn.ModuleList([norms(), _Attn(), nn.Identity()]),
                                     nn.ModuleList([norms(), _FF(), nn.Identity()])])
        self.final_norm = _G()

    @property
    def d_ff(self):
        return self.layers[1][1].ff[0][0].out_features


class Model(nn.Module):
    """Mirror of `duett.duett.Model.__init__` (duett.py:48-140): same constructor arguments, same parameter inventory.
    (The reference derives from LightningModule; the training-loop hooks of Lightning are out of scope.)"""

    def __init__(self, d_static_num, d_time_series_num, d_target, lr=3.e-4, weight_decay=1.e-1, glu=False, scalenorm=True,
                 n_hidden_mlp_embedding=1, d_hidden_mlp_embedding=64, d_embedding=24, d_feedforward=512, max_len=48,
                 n_transformer_head=2, n_duett_layers=2, d_hidden_tab_encoder=128, n_hidden_tab_encoder=1, norm_first=True,
                 fusion_method='masked_embed', n_hidden_head=1, d_hidden_head=64, aug_noise=0., aug_mask=0., pretrain=True,
                 pretrain_masked_steps=1, pretrain_n_hidden=0, pretrain_d_hidden=64, pretrain_dropout=0.5, pretrain_value=True,
                 pretrain_presence=True, pretrain_presence_weight=0.2, predict_events=True, transformer_dropout=0.,
                 pos_frac=None, freeze_encoder=False, seed=42, save_representation=None, masked_transform_timesteps=32, **kwargs):
        super().__init__()
        if n_hidden_mlp_embedding != 1 or n_hidden_tab_encoder != 1:
            raise ValueError("the HIP psi-embed kernel is built for one hidden layer per embedding MLP (duett.py defaults)")
        self.lr, self.weight_decay = lr, weight_decay
        self.d_static_num = d_static_num
        self.d_time_series_num = d_time_series_num
        self.d_target = d_target
        self.d_embedding = d_embedding
        self.max_len = max_len
        self.pretrain = pretrain
        self.pretrain_masked_steps = pretrain_masked_steps
        self.pretrain_dropout = pretrain_dropout
        self.freeze_encoder = freeze_encoder
        self.rng = np.random.default_rng(seed)
        self.aug_noise, self.aug_mask = aug_noise, aug_mask
        self.fusion_method = fusion_method
        self.pretrain_presence = pretrain_presence
        self.pretrain_presence_weight = pretrain_presence_weight
        self.predict_events = predict_events
        self.masked_transform_timesteps = masked_transform_timesteps
        self.pretrain_value = pretrain_value
        self.save_representation = save_representation
        self.n_transformer_head = n_transformer_head
        self.transformer_dropout = transformer_dropout
        self.register_buffer("MASKED_EMBEDDING_KEY", torch.tensor(0))
        self.register_buffer("REPRESENTATION_EMBEDDING_KEY", torch.tensor(1))

        self.special_embeddings = nn.Embedding(8, d_embedding)
        self.embedding_layers = nn.ModuleList([
            simple_mlp(2, d_embedding, n_hidden_mlp_embedding, d_hidden_mlp_embedding, hidden_batch_norm=True)
            for _ in range(d_time_series_num)])
        self.n_obs_embedding = nn.Embedding(16, 1)
        if d_feedforward is None:
            d_feedforward = d_embedding * 4
        et_dim = d_embedding * (masked_transform_timesteps + 1)
        tt_dim = d_embedding * (d_time_series_num + 1)
        mk = lambda dim: Encoder(dim=dim, depth=1, heads=n_transformer_head, pre_norm=norm_first, use_scalenorm=scalenorm,
                                 attn_dim_head=d_embedding // n_transformer_head, ff_glu=glu, ff_mult=d_feedforward / dim,
                                 attn_dropout=transformer_dropout, ff_dropout=transformer_dropout)
        self.event_transformers = nn.ModuleList([mk(et_dim) for _ in range(n_duett_layers)])
        self.full_event_embedding = nn.Embedding(d_time_series_num + 1, et_dim)
        self.time_transformers = nn.ModuleList([mk(tt_dim) for _ in range(n_duett_layers)])
        self.full_time_embedding = self.cve(batch_norm=True, d_embedding=tt_dim)
        self.full_rep_embedding = nn.Embedding(tt_dim, 1)

        d_representation = d_embedding * (d_time_series_num + 1)
        self.head = simple_mlp(d_representation, d_target, n_hidden_head, d_hidden_head, hidden_batch_norm=True,
                               final_activation=False, activation=nn.ReLU)
        self.pretrain_value_proj = simple_mlp(d_representation, d_time_series_num, pretrain_n_hidden, pretrain_d_hidden,
                                              hidden_batch_norm=True)
        if self.pretrain_presence:
            self.pretrain_presence_proj = simple_mlp(d_representation, d_time_series_num, pretrain_n_hidden, pretrain_d_hidden,
                                                     hidden_batch_norm=True)
        if self.predict_events:
            self.predict_events_proj = simple_mlp(et_dim, masked_transform_timesteps, pretrain_n_hidden, pretrain_d_hidden,
                                                  hidden_batch_norm=True)
            if self.pretrain_presence:
                self.predict_events_presence_proj = simple_mlp(et_dim, masked_transform_timesteps, pretrain_n_hidden,
                                                               pretrain_d_hidden, hidden_batch_norm=True)
        self.tab_encoder = simple_mlp(d_static_num, d_embedding, n_hidden_tab_encoder, d_hidden_tab_encoder, hidden_batch_norm=True)
        self._prep = None
        self._prep_key = None
        self._ws = None

    @property
    def device(self):
        return next(self.parameters()).device

    def cve(self, d_embedding=None, batch_norm=False):
        """duett.py:151-157."""
        if d_embedding is None:
            d_embedding = self.d_embedding
        d_hidden = int(np.sqrt(d_embedding))
        if batch_norm:
            return nn.Sequential(nn.Linear(1, d_hidden), nn.Tanh(), BatchNormLastDim(d_hidden), nn.Linear(d_hidden, d_embedding))
        return nn.Sequential(nn.Linear(1, d_hidden), nn.Tanh(), nn.Linear(d_hidden, d_embedding))

    # ------------------------------------------------------------------------------------------ a3: batch assembly
    _FEATS_SID = 200          # RNG stream ids 200..202 of the augmentation draws (values, dropped timesteps, static)

    @staticmethod
    def _as_rows_of_one_buffer(ts):
        """(base_ptr, stride_in_elements) when the per-sample tensors are equally shaped, contiguous, equally spaced views of
        one buffer (the collate of a resident batch / the static buffers of a captured step), else None."""
        t0 = ts[0]
        if not all(t.shape == t0.shape and t.is_contiguous() and t.dtype == torch.float32 for t in ts):
            return None
        if len(ts) == 1:
            return t0.data_ptr(), t0.numel()
        step = ts[1].data_ptr() - t0.data_ptr()
        if step < t0.numel() * 4 or step % 4 or any(t.data_ptr() != t0.data_ptr() + i * step for i, t in enumerate(ts)):
            return None
        return t0.data_ptr(), step // 4

    def _device_table(self, values, dtype):
        """Small host list -> device tensor, cached on its content (a captured step sees the same views every time)."""
        key = (dtype, tuple(values))
        cache = self.__dict__.setdefault("_tables", {})
        if key not in cache:
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError("feats_to_input: a new pointer/length table would have to be uploaded inside a graph capture; "
                                   "pass views of one stacked buffer (graph_step.py does) or warm the step up first")
            if len(cache) >= 16:
                cache.pop(next(iter(cache)))
            cache[key] = torch.tensor(list(values), dtype=dtype, device=self.device)
        return cache[key]

    def feats_to_input(self, x, batch_size, limits=None):
        """Same contract as duett.py:159-187: (tuples of per-sample tensors) -> (xs_static [B,Ds], xs_ts [B,Tpad,2V+1],
        xs_times [B,Tpad], n_timesteps) — ONE launch of `medp_feats_to_input` instead of the host loop over the batch:
        truncation to the last max_len steps, the mask column, zero padding to the longest series and the training
        augmentation all happen on the device.  Inputs may live on the host (one concatenated upload) or on
        the device (rows of one stacked buffer: no copy at all; separate tensors: a pointer table).  With augmentation off the
        result is bit-identical to the reference's; with it on the draws come from the library's counter RNG, not torch's.
        Unlike the reference the caller's tensors are never modified in place."""
        abi.require_gpu()
        xs_ts, xs_static, times = x
        xs_ts, xs_static, times = list(xs_ts), list(xs_static), list(times)
        dev = self.device
        if dev.type != "cuda":
            raise RuntimeError("feats_to_input: the model is on %s; the batch assembly kernel needs the GPU (no CPU fallback)" % dev)
        B = len(xs_ts)
        V2 = int(xs_ts[0].shape[1])
        V, Ds = V2 // 2, int(xs_static[0].shape[0])
        lens = [int(f.shape[0]) for f in xs_ts]
        if any(int(t.shape[0]) != n for t, n in zip(times, lens)) or any(int(f.shape[1]) != V2 for f in xs_ts):
            raise ValueError("feats_to_input: every series needs [T_i, 2V] features and [T_i] times")
        n_timesteps = [min(n, self.max_len) for n in lens]
        Tpad = max(n_timesteps)
        aug = self.training and not self.pretrain
        noise, maskp = (float(self.aug_noise), float(self.aug_mask)) if aug else (0.0, 0.0)
        keep = []                                   # temporaries the launch reads

        def locate(ts, width):
            """-> (ptr_table, base, stride) for a list of [T_i(, width)] series"""
            if all(t.device.type == "cpu" for t in ts):                       # host batch: one upload
                flat = torch.cat([t.reshape(-1) for t in ts]).to(dev, torch.float32, non_blocking=True)
                keep.append(flat)
                if len(set(lens)) == 1:
                    return None, flat.data_ptr(), lens[0] * width
                offs, o = [], 0
                for n in lens:
                    offs.append(flat.data_ptr() + 4 * o)
                    o += n * width
                return self._device_table(offs, torch.int64), None, 0
            ts = [t if (t.device == dev and t.dtype == torch.float32 and t.is_contiguous()) else t.to(dev, torch.float32).contiguous() for t in ts]
            keep.extend(ts)
            rows = self._as_rows_of_one_buffer(ts)
            if rows is not None:
                return None, rows[0], rows[1]
            return self._device_table([t.data_ptr() for t in ts], torch.int64), None, 0

        ts_tab, ts_base, ts_stride = locate(xs_ts, V2)
        tm_tab, tm_base, tm_stride = locate(times, 1)
        srows = self._as_rows_of_one_buffer(xs_static) if all(t.device == dev for t in xs_static) else None
        if srows is not None and (B == 1 or srows[1] == Ds):
            static_ptr = srows[0]
        else:
            st = torch.stack([t.to(torch.float32) for t in xs_static]).to(dev).contiguous()
            keep.append(st)
            static_ptr = st.data_ptr()
        uniform = len(set(lens)) == 1
        len_tab = None if uniform else self._device_table(lens, torch.int32)
        out_ts = torch.empty((B, Tpad, V2 + 1), dtype=torch.float32, device=dev)
        out_tm = torch.empty((B, Tpad), dtype=torch.float32, device=dev)
        out_st = torch.empty((B, Ds), dtype=torch.float32, device=dev)
        seed = 0
        if noise > 0 or maskp > 0:
            from . import autograd_ops as A
            seed = A.next_seed()
        check(lib().medp_feats_to_input(ptr(ts_tab), ts_base, ts_stride, ptr(tm_tab), tm_base, tm_stride, ptr(len_tab),
                                        lens[0] if uniform else 0, static_ptr, ptr(out_ts), ptr(out_tm), ptr(out_st), B, V, Ds,
                                        int(self.max_len), Tpad, noise, maskp, seed, self._FEATS_SID, stream()), "feats_to_input")
        return out_st, out_ts, out_tm, n_timesteps

    # ------------------------------------------------------------------------------------------ weight preparation
    def _prepare(self):
        """Stack the V per-variable MLPs, fold eval-mode BatchNorm, cast GEMM weights to bf16, fill the C structs.
        Rebuilt only when a parameter/buffer was modified in place (optimizer step, load_state_dict)."""
        tensors = list(self.parameters()) + list(self.buffers())
        key = tuple((t.data_ptr(), t._version) for t in tensors)
        if self._prep is not None and self._prep_key == key:
            return self._prep
        keep = []

        def f32(t):
            t = t.detach().to(torch.float32).contiguous()
            keep.append(t)
            return ptr(t)

        def bf(t2d):
            t = Fn.to_bf16(t2d.detach().to(torch.float32).contiguous())
            keep.append(t)
            return ptr(t)

        V, E = self.d_time_series_num, self.d_embedding
        el = self.embedding_layers
        w = MedpDuettWeights()
        w.n_vars, w.n_static, w.d_embedding, w.n_heads = V, self.d_static_num, E, self.n_transformer_head
        w.n_layers = len(self.event_transformers)
        w.d_ff = self.event_transformers[0].d_ff
        w.d_hidden_embed = el[0][0].out_features
        w.d_hidden_tab = self.tab_encoder[0].out_features
        w.d_hidden_time = self.full_time_embedding[0].out_features
        w.n_obs_rows = self.n_obs_embedding.num_embeddings
        w.final_norm, w.norm_eps = int(FINAL_NORM), SCALENORM_EPS
        w.emb_w0 = f32(torch.stack([m[0].weight for m in el]))
        w.emb_b0 = f32(torch.stack([m[0].bias for m in el]))
        folded = [m[3].folded() for m in el]
        w.emb_bn_scale = f32(torch.stack([s for s, _ in folded]))
        w.emb_bn_shift = f32(torch.stack([b for _, b in folded]))
        w.emb_w4 = f32(torch.stack([m[4].weight for m in el]))
        w.emb_b4 = f32(torch.stack([m[4].bias for m in el]))
        # scalar-load layout of the same MLPs for the fused embed kernel (include/medp_hip.h)
        w0s, b0s = torch.stack([m[0].weight for m in el]).detach(), torch.stack([m[0].bias for m in el]).detach()
        l0 = torch.zeros((V, w0s.shape[1], 8), dtype=torch.float32, device=w0s.device)
        l0[..., 0:2], l0[..., 2] = w0s, b0s
        l0[..., 3], l0[..., 4] = torch.stack([s for s, _ in folded]), torch.stack([b for _, b in folded])
        w.emb_l0 = f32(l0)
        w.emb_w4t = f32(torch.stack([m[4].weight for m in el]).transpose(1, 2))
        w.n_obs_table = f32(self.n_obs_embedding.weight[:, 0])
        te = self.tab_encoder
        ts_, tb_ = te[3].folded()
        w.tab_w0, w.tab_b0, w.tab_bn_scale, w.tab_bn_shift = f32(te[0].weight), f32(te[0].bias), f32(ts_), f32(tb_)
        w.tab_w4, w.tab_b4 = f32(te[4].weight), f32(te[4].bias)
        w.special = f32(self.special_embeddings.weight)
        tm = self.full_time_embedding
        ms, mb = tm[2].folded()
        w.time_w0, w.time_b0, w.time_bn_scale, w.time_bn_shift = f32(tm[0].weight[:, 0]), f32(tm[0].bias), f32(ms), f32(mb)
        w.time_w3t, w.time_b3 = f32(tm[3].weight.t()), f32(tm[3].bias)
        w.rep_embedding = f32(self.full_rep_embedding.weight[:, 0])
        w.event_embedding = f32(self.full_event_embedding.weight)

        def enc_structs(mods):
            arr = (MedpEncoderWeights * len(mods))()
            for i, m in enumerate(mods):
                a, ff = m.layers[0][1], m.layers[1][1].ff
                arr[i].g_attn = f32(m.layers[0][0][0].g)
                arr[i].qkv_w = bf(torch.cat([a.to_q.weight, a.to_k.weight, a.to_v.weight], 0))
                arr[i].out_w = bf(a.to_out.weight)
                arr[i].g_ff = f32(m.layers[1][0][0].g)
                arr[i].ff1_w, arr[i].ff1_b = bf(ff[0][0].weight), f32(ff[0][0].bias)
                arr[i].ff2_w, arr[i].ff2_b = bf(ff[2].weight), f32(ff[2].bias)
                arr[i].g_final = f32(m.final_norm.g)
            return arr

        ev, tv = enc_structs(self.event_transformers), enc_structs(self.time_transformers)
        w.event_enc = ctypes.cast(ev, ctypes.POINTER(MedpEncoderWeights))
        w.time_enc = ctypes.cast(tv, ctypes.POINTER(MedpEncoderWeights))
        self._prep, self._prep_key = (w, ev, tv, keep), key
        return self._prep

    def _encode_inference(self, x, want_bf16=False, want_psi0=False):
        """Inference form (eval-mode BatchNorm, no dropout, no autograd): one C call."""
        abi.require_gpu()
        xs_static, xs_feats, xs_times, _ = x
        w = self._prepare()[0]
        xs_static = xs_static.detach().to(torch.float32).contiguous()
        xs_feats = xs_feats.detach().to(torch.float32).contiguous()
        xs_times = xs_times.detach().to(torch.float32).contiguous()
        B, T, Fdim = xs_feats.shape
        V, E = self.d_time_series_num, self.d_embedding
        if Fdim != 2 * V + 1:
            raise ValueError(f"xs_feats must be [B, T, 2V+1] with V={V}, got {tuple(xs_feats.shape)}")
        if T != self.masked_transform_timesteps:
            raise ValueError(f"this backbone was built for n_timesteps={self.masked_transform_timesteps}, got T={T} "
                             "(the event-axis token width is 24*(T+1), duett.py:93)")
        need = lib().medp_duett_workspace_bytes(ctypes.byref(w), B, T)
        if self._ws is None or self._ws.numel() < need or self._ws.device != xs_feats.device:
            self._ws = torch.empty(need, dtype=torch.uint8, device=xs_feats.device)
        dev = xs_feats.device
        tok = torch.empty((B, T + 1, E * (V + 1)), dtype=torch.float32, device=dev)
        tok16 = torch.empty((B, T + 1, E * (V + 1)), dtype=torch.bfloat16, device=dev) if want_bf16 else None
        psi0 = torch.empty((B, T + 1, V + 1, E), dtype=torch.float32, device=dev) if want_psi0 else None
        check(lib().medp_duett_encode(ctypes.byref(w), ptr(xs_static), ptr(xs_feats), ptr(xs_times), B, T, ptr(tok), ptr(tok16),
                                      ptr(psi0), ptr(self._ws), need, stream()), "duett_encode")
        return tok, tok16, psi0


def _attach_ssl_methods():
    """`Model.pretrain_prep_batch`, `Model.forward`, `Model.training_step` (duett.py:189-372) live in duett_ssl.py."""
    from . import duett_ssl

    Model.pretrain_prep_batch = lambda self, x, batch_size: duett_ssl.pretrain_prep_batch(self, x, batch_size)
    Model.forward = lambda self, x, pretrain=False, representation=False: duett_ssl.model_forward(self, x, pretrain, representation)
    Model.training_step = lambda self, batch, batch_idx=0: duett_ssl.training_step_loss(self, batch)


def pretrain_model(d_static_num, d_time_series_num, d_target, **kwargs):
    """duett.py:41-42."""
    return Model(d_static_num, d_time_series_num, d_target, **kwargs)


class DuettFeatureExtractor(Model):
    """Mirror of model file :24-94."""

    @property
    def d_representation(self) -> int:
        return self.d_embedding * (self.d_time_series_num + 1)

    def needs_training_path(self) -> bool:
        if not torch.is_grad_enabled():
            return False
        return any(p.requires_grad for p in self.parameters())

    def encode(self, x):
        """[B, T+1, 24*(V+1)] contextual tokens (model file :31-94)."""
        if self.needs_training_path() or (self.training and any(p.requires_grad for p in self.parameters())) or Fn.precision() == "fp32":
            from .duett_train import encode_training          # op-level form (also the only fp32-mode form, functional.set_precision)
            return encode_training(self, x)
        if self.training and self.transformer_dropout > 0:
            raise NotImplementedError("dropout inside a frozen DuETT in train() mode: the reference's step functions put frozen "
                                      "sub-modules in eval() (engine.py:7-20); call .eval() on the frozen backbone")
        return self._encode_inference(x)[0]


def load_duett_backbone(ckpt_path: str, d_static_num: int, d_time_series_num: int, n_timesteps: int, freeze: bool = False,
                        aug_noise: float = 0.0, aug_mask: float = 0.0, transformer_dropout: float = 0.0) -> DuettFeatureExtractor:
    """Mirror of model file :98-123.  `ckpt_path` is a Lightning checkpoint ({"state_dict": ...}); loading is non-strict
    like the reference's `load_from_checkpoint(strict=False)`.  `ckpt_path in (None, "", "synthetic")` keeps the seeded
    random initialisation (no pretrained DuETT checkpoint ships with the reference)."""
    model = DuettFeatureExtractor(d_static_num=d_static_num, d_time_series_num=d_time_series_num, d_target=1, pretrain=False,
                                  masked_transform_timesteps=n_timesteps, max_len=n_timesteps, aug_noise=aug_noise,
                                  aug_mask=aug_mask, transformer_dropout=transformer_dropout)
    if ckpt_path and ckpt_path != "synthetic":
        state = torch.load(ckpt_path, map_location="cpu", weights_only=True)
        sd = state.get("state_dict", state)
        own = model.state_dict()
        for k in list(sd):
            if k not in own or (k.startswith("head") and sd[k].shape != own[k].shape):
                sd.pop(k)            # on_load_checkpoint semantics, duett.py:459-481
        model.load_state_dict(sd, strict=False)
    if freeze:
        for p in model.parameters():
            p.requires_grad = False
        model.eval()
    return model


_attach_ssl_methods()
