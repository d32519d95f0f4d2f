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


def dual_perceiver_forward(sd, ts_tokens, img_logits, n_heads=4, p="", dropout=0.0, head_dropout=0.0, training=False,
                           return_attn=False, ts_ablation="hourly_only"):
    """`DualPathologyPerceiver.forward` — commented out at the reference's HEAD (model file `:659-741`) but required by its
    student entry point (`training_duett/trainer.py:770-822`): temporal queries x ts tokens (cross + self block), one MLP head per
    pathology for the TS-only logits and one for the residual, `fusion = img_logits + residuals` (img_logits come from the frozen
    pretrained CXR head and carry no gradient).  Pinned by tests/golden/teacher_dual_cfg1.npz, produced by executing the
    reference's own (un-commented) text."""
    B = ts_tokens.size(0)
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
    q = sd[p + "temporal_queries"].unsqueeze(0).expand(B, -1, -1)
    blk = lambda lat, kv, name, ra=False: perceiver_block(lat, kv, sd, p + name + ".", n_heads, dropout, training, ra)
    T, ts_attn = blk(q, ts_kv, "ts_cross", True) if return_attn else (blk(q, ts_kv, "ts_cross"), None)
    T = blk(T, T, "ts_self")
    K = T.shape[1]

    def heads(name):
        cols = []
        for k in range(K):
            h = F.gelu(F.linear(T[:, k], sd[f"{p}{name}.{k}.0.weight"], sd[f"{p}{name}.{k}.0.bias"]))
            h = F.dropout(h, head_dropout, training)
            cols.append(F.linear(h, sd[f"{p}{name}.{k}.3.weight"], sd[f"{p}{name}.{k}.3.bias"]).squeeze(-1))
        return torch.stack(cols, dim=1)

    ts_logits, residuals = heads("temporal_heads"), heads("residual_heads")
    out = {"img_logits": img_logits, "ts_logits": ts_logits, "fusion_logits": img_logits + residuals, "ts_tokens": T,
           "residuals": residuals}
    if return_attn:
        out["ts_attn"] = ts_attn
    return out


def teacher_dual_forward(sd, ts_tokens, cls, n_heads=4, **kw):
    """`TeacherModel.forward`, dual branch (model file `:1132-1150`): CLS -> frozen pretrained linear CXR head -> the K kept
    columns (`cxr_head_keep_idx`, `:1047-1071`) -> `DualPathologyPerceiver`."""
    with torch.no_grad():
        pre = F.linear(cls, sd["pretrained_cxr_head.weight"], sd["pretrained_cxr_head.bias"])
    img_logits = pre[:, sd["cxr_head_keep_idx"]]
    out = dual_perceiver_forward(sd, ts_tokens, img_logits, n_heads, p="perceiver.", **kw)
    res = {"main_logit": out["fusion_logits"][:, 0], "img_logits": out["img_logits"], "ts_logits": out["ts_logits"],
           "fusion_logits": out["fusion_logits"]}
    if kw.get("return_attn"):
        res["ts_tokens"], res["ts_attn"], res["residuals"] = out["ts_tokens"], out["ts_attn"], out["residuals"]
    return res
