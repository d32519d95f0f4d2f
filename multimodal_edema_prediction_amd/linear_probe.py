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


class PixelPrefetcher:
    """Host -> device staging of the NEXT batch's pixels on a copy stream, two device buffers: the loader-side half of the probe's
    step (a `DataLoader(pin_memory=True)` feeding `.to(device, non_blocking=True)` leaves the 38.5-MB copy on the compute stream:
    0.8 ms of a 5.3-ms step).  `stage(host_pixels)` issues the copy; `take()` makes the compute stream wait for the oldest staged
    batch and returns it (valid until the next-but-one `stage`)."""

    def __init__(self, device, like: torch.Tensor):
        self.device = torch.device(device)
        self.bufs = [torch.empty(like.shape, dtype=like.dtype, device=self.device) for _ in range(2)]
        self.copy_stream = torch.cuda.Stream(device=self.device)
        self.done = [torch.cuda.Event(), torch.cuda.Event()]
        self.free = [torch.cuda.Event(), torch.cuda.Event()]
        for e in self.free:
            e.record(torch.cuda.current_stream(self.device))
        self.n_staged = self.n_taken = 0

    def stage(self, host_pixels: torch.Tensor) -> None:
        i = self.n_staged % 2
        if self.n_staged - self.n_taken >= 2:
            raise RuntimeError("PixelPrefetcher: both buffers hold batches that were not taken yet")
        with torch.cuda.stream(self.copy_stream):
            self.copy_stream.wait_event(self.free[i])              # the compute stream is done with this buffer's previous batch
            self.bufs[i].copy_(host_pixels, non_blocking=True)
            self.done[i].record(self.copy_stream)
        self.n_staged += 1

    def take(self) -> torch.Tensor:
        if self.n_taken >= self.n_staged:
            raise RuntimeError("PixelPrefetcher: nothing staged")
        i = self.n_taken % 2
        cur = torch.cuda.current_stream(self.device)
        cur.wait_event(self.done[i])
        j = (i + 1) % 2                                            # the OTHER buffer's batch was consumed by everything enqueued so far
        self.free[j].record(cur)
        self.n_taken += 1
        return self.bufs[i]
