"""Synthetic cohort generator.  Emits exactly the item dict of the reference's
`DuettAnchorDataset.__getitem__` (`training_duett/data_processing.py:378-391`) and the batch dict
of `duett_kd_collate` (`:394-411`), from seeded CPU generators only — no MIMIC data is needed.

Recipe (the reference's own smoke-test recipe, `analysis/smoke_test_trajectory_encoder.py:23-26`,
with MIMIC's count range): obs ~ Bernoulli(0.2) over [T,V]; values N(0,1)·obs; counts obs·U{1,2,3};
x_static ~ N(0,1); bin_ends = arange(1,T+1)/24; pixels ~ N(0,1); label prevalences from
`cxr_linear_training.ipynb:322-329`; mask ~ Bernoulli(0.9).
"""
from __future__ import annotations

from dataclasses import dataclass

import torch

PATHOLOGY_LABELS = ("label_edema", "label_cardiomegaly", "label_effusion", "label_pneumonia",
                    "label_atelectasis", "label_opacity", "label_consolidation")   # data_processing.py:22-30
PREVALENCE = (0.08, 0.21, 0.05, 0.02, 0.05, 0.05, 0.02)


@dataclass
class CohortCfg:
    n_timesteps: int = 96        # T
    n_vars: int = 48             # V
    d_static: int = 8
    image_size: int = 224
    n_labels: int = 7
    seed: int = 1234
    learnable: bool = False      # labels depend on the inputs (AUROC parity runs)


def make_item(cfg: CohortCfg, index: int, with_image: bool = True, n_steps: int | None = None) -> dict:
    g = torch.Generator().manual_seed(cfg.seed * 1_000_003 + index)
    T = cfg.n_timesteps if n_steps is None else n_steps
    V = cfg.n_vars
    obs = (torch.rand(T, V, generator=g) < 0.2)
    values = torch.randn(T, V, generator=g) * obs
    counts = obs.float() * torch.randint(1, 4, (T, V), generator=g).float()
    x_ts = torch.cat((values, counts), dim=1)
    x_static = torch.randn(cfg.d_static, generator=g)
    bin_ends = torch.arange(1, T + 1, dtype=torch.float32) / 24.0        # data_processing.py:343
    prev = torch.tensor(PREVALENCE[:cfg.n_labels])
    u = torch.rand(cfg.n_labels, generator=g)
    mask = (torch.rand(cfg.n_labels, generator=g) < 0.9).float()
    chan = torch.randn(3, generator=g)             # per-channel brightness offset: the image-side signal
    noise = torch.randn(cfg.n_labels, generator=g)
    if cfg.learnable:
        # y_k = 1[<w_k, summary(x_ts)> + <u_k, channel offsets> + noise > tau_k], fixed seeded w, u
        gw = torch.Generator().manual_seed(cfg.seed + 77)
        w = torch.randn(cfg.n_labels, V, generator=gw) / V ** 0.5
        uimg = torch.randn(cfg.n_labels, 3, generator=gw) / 3 ** 0.5
        summary = values.sum(0) / obs.sum(0).clamp(min=1).float().sqrt()
        score = 0.6 * (w @ summary) + 0.6 * (uimg @ chan) + 0.5 * noise
        y_multi = (score > torch.distributions.Normal(0.0, 1.0).icdf(1 - prev)).float()
    else:
        y_multi = (u < prev).float()
    item = {"x_ts": x_ts, "x_static": x_static, "bin_ends": bin_ends, "y": y_multi[0].clone(),
            "y_multi": y_multi, "y_multi_mask": mask}
    if with_image:
        pix = torch.randn(3, cfg.image_size, cfg.image_size, generator=g)
        item["pixel_values"] = pix + (0.25 * chan).view(3, 1, 1) if cfg.learnable else pix
    return item


def collate(items: list, mode: str = "teacher") -> dict:
    """Same output layout as `duett_kd_collate` (data_processing.py:394-411)."""
    out = {"x_ts": tuple(b["x_ts"] for b in items), "x_static": tuple(b["x_static"] for b in items),
           "bin_ends": tuple(b["bin_ends"] for b in items), "y": torch.stack([b["y"] for b in items])}
    if "y_multi" in items[0]:
        out["y_multi"] = torch.stack([b["y_multi"] for b in items])
        out["y_multi_mask"] = torch.stack([b["y_multi_mask"] for b in items])
    if mode == "teacher":
        out["pixel_values"] = torch.stack([b["pixel_values"] for b in items])
    return out


def make_batch(cfg: CohortCfg, start: int, batch_size: int, mode: str = "teacher", stride: int = 1) -> dict:
    return collate([make_item(cfg, start + i * stride, with_image=(mode == "teacher"))
                    for i in range(batch_size)], mode)


class SyntheticCohort(torch.utils.data.Dataset):
    def __init__(self, cfg: CohortCfg, n: int, mode: str = "teacher", offset: int = 0):
        self.cfg, self.n, self.mode, self.offset = cfg, n, mode, offset

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        return make_item(self.cfg, self.offset + i, with_image=(self.mode == "teacher"))
