"""CXR-encoder-only linear probe (BASELINE.json configs[1]): mirror of `RadDinoClassifier` + `masked_bce_with_logits_loss`
from the reference's cxr_linear_training.ipynb (:396-437): frozen encoder under no_grad -> CLS -> Dropout(0.1) ->
Linear(768, C); one global masked BCE mean.  The encoder stays in eval() while training (ipynb :626-627)."""
from __future__ import annotations

import torch
import torch.nn as nn

from . import autograd_ops as A
from .cxr import CXREncoder


class RadDinoClassifier(nn.Module):
    def __init__(self, model_name: str = "microsoft/rad-dino", num_classes: int = 7, dropout: float = 0.1, config=None):
        super().__init__()
        self.encoder = CXREncoder(model_name, freeze=True, return_patches=False, config=config)
        self.classifier = nn.Sequential(nn.Dropout(dropout), nn.Linear(self.encoder.d_out, num_classes))

    def train(self, mode: bool = True):
        super().train(mode)
        self.encoder.eval()
        return self

    def forward(self, pixel_values: torch.Tensor) -> torch.Tensor:
        with torch.no_grad():
            cls = self.encoder(pixel_values).contiguous()                       # last_hidden_state[:, 0]
        p = float(self.classifier[0].p) if self.training else 0.0
        h = A.DropoutFn.apply(cls, p, A.next_seed() if p > 0 else 0, 60) if p > 0 else cls
        return A.linear(h, self.classifier[1].weight, self.classifier[1].bias)


def masked_bce_with_logits_loss(logits, targets, mask):
    """ipynb :426-437: sum(bce * mask) / clamp(sum(mask), 1)."""
    return A.masked_bce_global(logits, targets, mask)
