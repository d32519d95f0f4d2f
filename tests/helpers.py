"""Shared test helpers: deterministic synthetic weights keyed by state_dict name, so fixtures
hold only shapes + expected outputs and the weights are regenerated identically in every
environment (this container when the fixtures are made, the GPU box when they are checked)."""
from __future__ import annotations

import hashlib
import json
import os

import numpy as np
import torch

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _seed_of(name: str, seed: int) -> int:
    return int.from_bytes(hashlib.sha256(f"{seed}:{name}".encode()).digest()[:7], "little")


def synth_tensor(name: str, shape, dtype: str = "float32", seed: int = 0) -> torch.Tensor:
    """Value distribution chosen by the key's role so every path is exercised with realistic scales."""
    shape = tuple(shape)
    g = torch.Generator().manual_seed(_seed_of(name, seed))
    if dtype == "int64":
        if name.endswith("MASKED_EMBEDDING_KEY"):
            return torch.tensor(0)
        if name.endswith("REPRESENTATION_EMBEDDING_KEY"):
            return torch.tensor(1)
        if name.endswith("cxr_head_keep_idx"):
            return torch.arange(shape[0])
        return torch.zeros(shape, dtype=torch.int64)            # num_batches_tracked
    r = torch.randn(shape, generator=g) if len(shape) else torch.randn((), generator=g)
    leaf = name.split(".")[-1]
    if leaf == "running_var":
        return 0.5 + torch.rand(shape, generator=g)
    if leaf == "running_mean":
        return 0.1 * r
    if leaf in ("g",) or "lambda1" in leaf or leaf == "beta":
        return 1.0 + 0.1 * r
    if leaf == "weight" and len(shape) == 1:                       # LayerNorm / BatchNorm scale
        return 1.0 + 0.1 * r
    if leaf == "bias" or leaf.endswith("_bias"):
        return 0.1 * r
    if leaf in ("cls_token", "position_embeddings", "shared_queries", "mask_token"):
        return 0.2 * r
    if len(shape) >= 2:
        if "embedding" in name and "embedding_layers" not in name and "full_time" not in name:
            return 0.5 * r                                          # nn.Embedding tables
        fan_in = int(np.prod(shape[1:]))
        return r / max(fan_in, 1) ** 0.5
    return 0.1 * r


def synth_state_dict(shapes: dict, seed: int = 0) -> dict:
    """shapes: {key: [shape list, dtype str]} → {key: tensor}"""
    return {k: synth_tensor(k, sh, dt, seed) for k, (sh, dt) in shapes.items()}


def shapes_of(sd: dict) -> dict:
    return {k: [list(v.shape), str(v.dtype).replace("torch.", "")] for k, v in sd.items()}


def load_shapes(name: str) -> dict:
    with open(os.path.join(GOLDEN_DIR, name)) as f:
        return json.load(f)


def load_npz(name: str) -> dict:
    with np.load(os.path.join(GOLDEN_DIR, name), allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


def t(x) -> torch.Tensor:
    return torch.from_numpy(np.ascontiguousarray(x))
