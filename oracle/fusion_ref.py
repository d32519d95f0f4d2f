"""fp32 restatement of the cross-modal fusion head (SURVEY.md §8a rows a8–a10):
`_PerceiverBlock` (model file `:745-774`), `PatchDualPathologyPerceiver` (`:538-654`),
`TeacherModel.forward` patch-dual branch (`:1089-1129`).  `sd` keys follow the reference
module tree (`perceiver.*`, `img_proj.*`)."""
from __future__ import annotations

import torch
import torch.nn.functional as F

LN_EPS = 1e-5


def mha(q_in, kv_in, sd, p, n_heads, dropout=0.0, training=False, need_weights=False):
    """`nn.MultiheadAttention(d, h, batch_first=True)(q, k, k)` with packed in_proj."""
    d = q_in.shape[-1]
    W, b = sd[p + "in_proj_weight"], sd[p + "in_proj_bias"]
    q = F.linear(q_in, W[:d], b[:d])
    k = F.linear(kv_in, W[d:2 * d], b[d:2 * d])
    v = F.linear(kv_in, W[2 * d:], b[2 * d:])
    B, Lq, _ = q.shape
    Lk = k.shape[1]
    dh = d // n_heads
    sp = lambda t, L: t.view(B, L, n_heads, dh).transpose(1, 2)
    q, k, v = sp(q, Lq), sp(k, Lk), sp(v, Lk)
    w = torch.softmax(torch.matmul(q, k.transpose(-1, -2)) * dh ** -0.5, dim=-1)
    wd = F.dropout(w, dropout, training)
    o = torch.matmul(wd, v).transpose(1, 2).reshape(B, Lq, d)
    o = F.linear(o, sd[p + "out_proj.weight"], sd[p + "out_proj.bias"])
    return o, (wd.mean(dim=1) if need_weights else None)          # average_attn_weights=True


def perceiver_block(lat, kv, sd, p, n_heads, dropout=0.0, training=False, return_attn=False):
    """model file `:759-774`."""
    ln = lambda x, n: F.layer_norm(x, (x.shape[-1],), sd[p + n + ".weight"], sd[p + n + ".bias"], LN_EPS)
    q = ln(lat, "norm_q")
    k = ln(kv, "norm_kv")
    a, w = mha(q, k, sd, p + "attn.", n_heads, dropout, training, return_attn)
    lat = lat + a
    h = F.gelu(F.linear(ln(lat, "norm_ff"), sd[p + "ff.0.weight"], sd[p + "ff.0.bias"]))
    h = F.dropout(h, dropout, training)
    h = F.dropout(F.linear(h, sd[p + "ff.3.weight"], sd[p + "ff.3.bias"]), dropout, training)
    lat = lat + h
    return (lat, w) if return_attn else lat


def perceiver_forward(sd, ts_tokens, img_patches_proj, n_heads=4, p="", dropout=0.0, head_dropout=0.0,
                      training=False, return_attn=False, ts_ablation="hourly_only"):
    """`PatchDualPathologyPerceiver.forward`, model file `:595-654`."""
    if ts_tokens.ndim != 3:
        raise ValueError(f"ts_tokens must be [B, T+1, d_ts], got {tuple(ts_tokens.shape)}")
    B = ts_tokens.size(0)
    q0 = sd[p + "shared_queries"].unsqueeze(0).expand(B, -1, -1)
    if ts_ablation == "full":
        sel = ts_tokens
    elif ts_ablation == "hourly_only":
        sel = ts_tokens[:, :-1, :]
    elif ts_ablation == "rep_only":
        sel = ts_tokens[:, -1:, :]
    else:
        raise ValueError(f"unknown ts_ablation={ts_ablation!r}; expected one of "
                         "{'full', 'hourly_only', 'rep_only'}")
    ts_kv = F.linear(sel, sd[p + "ts_proj.weight"], sd[p + "ts_proj.bias"])
    blk = lambda lat, kv, name, ra=False: perceiver_block(lat, kv, sd, p + name + ".", n_heads, dropout, training, ra)
    if return_attn:
        I, img_attn = blk(q0, img_patches_proj, "img_cross", True)
    else:
        I, img_attn = blk(q0, img_patches_proj, "img_cross"), None
    I = blk(I, I, "img_self")
    if return_attn:
        T, ts_attn = blk(q0, ts_kv, "ts_cross", True)
    else:
        T, ts_attn = blk(q0, ts_kv, "ts_cross"), None
    T = blk(T, T, "ts_self")

    def head(x, name):
        h = F.gelu(F.linear(x, sd[p + name + ".0.weight"], sd[p + name + ".0.bias"]))
        h = F.dropout(h, head_dropout, training)
        return F.linear(h, sd[p + name + ".3.weight"], sd[p + name + ".3.bias"]).squeeze(-1)

    img_logits = head(I, "image_head") + sd[p + "image_label_bias"].unsqueeze(0)
    ts_logits = head(T, "temporal_head") + sd[p + "temporal_label_bias"].unsqueeze(0)
    c = F.layer_norm(T, (T.shape[-1],), sd[p + "correction_head.0.weight"], sd[p + "correction_head.0.bias"], LN_EPS)
    c = F.gelu(F.linear(c, sd[p + "correction_head.1.weight"], sd[p + "correction_head.1.bias"]))
    c = F.dropout(c, head_dropout, training)
    ts_correction = F.linear(c, sd[p + "correction_head.4.weight"]).squeeze(-1)
    scaled = sd[p + "beta"].unsqueeze(0) * ts_correction
    fusion = img_logits.detach() + scaled
    out = {"img_logits": img_logits, "ts_logits": ts_logits, "fusion_logits": fusion, "img_tokens": I,
           "ts_tokens": T, "fusion_tokens": T, "ts_correction": ts_correction, "scaled_correction": scaled}
    if return_attn:
        out["img_attn"], out["ts_attn"] = img_attn, ts_attn
    return out


def teacher_fusion_forward(sd, ts_tokens, img_patches, n_heads=4, **kw):
    """`TeacherModel.forward` patch-dual branch after the two encoders, model file `:1098-1129`."""
    proj = F.linear(img_patches, sd["img_proj.weight"], sd["img_proj.bias"])
    out = perceiver_forward(sd, ts_tokens, proj, n_heads, p="perceiver.", **kw)
    res = {"main_logit": out["fusion_logits"][:, 0]}
    for k in ("img_logits", "ts_logits", "fusion_logits", "ts_correction", "scaled_correction"):
        res[k] = out[k]
    if kw.get("return_attn"):
        for k in ("img_tokens", "ts_tokens", "fusion_tokens", "img_attn", "ts_attn"):
            res[k] = out[k]
    return res
