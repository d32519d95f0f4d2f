"""Optimiser side of the hot path: `FusedAdamW` — torch.optim.AdamW semantics in ONE HIP launch
over all parameters (`medp_adamw_multi`), the reference's name-pattern LR groups (`make_param_groups`, trainer.py:77-116)
and its warm-up + cosine schedule (`make_scheduler`, trainer.py:119-125)."""
from __future__ import annotations

import ctypes

import torch
from torch.optim.lr_scheduler import CosineAnnealingLR, LinearLR, SequentialLR

from .abi import MedpAdamTensor, check, lib, ptr, stream


def make_param_groups(model, lr, backbone_lr_mult=0.2, query_lr_mult=0.2, correction_lr_mult=1.0):
    """trainer.py:77-116 (same name patterns, same group order and names)."""
    backbone, correction, queries, rest = [], [], [], []
    for name, p in model.named_parameters():
        if not p.requires_grad:
            continue
        if name.startswith(("duett.", "cxr.")):
            backbone.append(p)
        elif "correction_head" in name or name.endswith(".beta") or name == "beta":
            correction.append(p)
        elif name.endswith("_queries"):
            queries.append(p)
        else:
            rest.append(p)
    groups = []
    if backbone:
        groups.append({"params": backbone, "lr": lr * backbone_lr_mult, "name": "backbone"})
    if queries:
        groups.append({"params": queries, "lr": lr * query_lr_mult, "name": "pathology_queries"})
    if correction:
        groups.append({"params": correction, "lr": lr * correction_lr_mult, "name": "correction_head"})
    if rest:
        groups.append({"params": rest, "lr": lr, "name": "rest"})
    return groups


def make_scheduler(optimizer, total_steps: int, lr: float, warmup_steps: int = 300, min_lr_ratio: float = 0.01):
    """trainer.py:119-125."""
    warmup = max(int(warmup_steps), 1)
    cosine_steps = max(int(total_steps) - warmup, 1)
    return SequentialLR(optimizer, schedulers=[LinearLR(optimizer, start_factor=1e-4, end_factor=1.0, total_iters=warmup),
                                               CosineAnnealingLR(optimizer, T_max=cosine_steps, eta_min=lr * min_lr_ratio)],
                        milestones=[warmup])


class FusedAdamW(torch.optim.Optimizer):
    """Drop-in for `torch.optim.AdamW(param_groups, weight_decay=...)` (amsgrad / maximize unsupported).

    The kernel reads a device table of tensor descriptors (pointers, numel, lr, weight decay).  The host never rewrites
    memory an in-flight copy or graph may still read: the table is built in a plain host array and goes to the device
    through a RING of pinned staging buffers, each guarded by an event recorded behind its host->device copy (the host may
    run several steps ahead of the GPU — bench.py reads the loss of step k-1, a training loop may never sync).  A captured
    graph holds only the KERNEL node; the upload for replay k is enqueued on the stream right before the replay
    (`refresh_lrs`), so replay k always sees the learning rates of step k."""

    RING = 4

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self._step = 0
        self._chunk = lib().medp_adamw_chunk_elems()
        self._map_key = None
        self._blk_t = self._blk_c = self._descs_dev = None
        self._table = None            # host master copy of the descriptor table (ctypes array; the device never reads it)
        self._ring, self._ring_ev, self._ring_n = [], [], 0
        self._entries = []
        self.dev_step = None          # device uint32 step counter: bias corrections are computed ON DEVICE from it, so the
                                      # eager step and a captured-graph replay run bit-identical arithmetic

    def _upload_table(self) -> None:
        """master table -> next ring slot (after its previous copy has been consumed) -> device, stream-ordered."""
        nbytes = ctypes.sizeof(self._table)
        i = self._ring_n % self.RING
        self._ring_n += 1
        self._ring_ev[i].synchronize()                       # a no-op unless the GPU is >= RING uploads behind
        ctypes.memmove(self._ring[i].data_ptr(), ctypes.addressof(self._table), nbytes)
        self._descs_dev.copy_(self._ring[i], non_blocking=True)
        self._ring_ev[i].record(torch.cuda.current_stream(self._descs_dev.device))

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        entries = []
        betas, eps = None, None
        for g in self.param_groups:
            if betas is None:
                betas, eps = g["betas"], g["eps"]
            elif (betas, eps) != (g["betas"], g["eps"]):
                raise ValueError("FusedAdamW: betas/eps must be the same in every group")
            for p in g["params"]:
                if p.grad is None:
                    continue
                if p.dtype != torch.float32 or p.grad.dtype != torch.float32 or not p.is_contiguous() or not p.grad.is_contiguous():
                    raise ValueError("FusedAdamW needs contiguous fp32 parameters and gradients")
                st = self.state[p]
                if not st:
                    st["exp_avg"] = torch.zeros_like(p)
                    st["exp_avg_sq"] = torch.zeros_like(p)
                entries.append((p, st, float(g["lr"]), float(g["weight_decay"])))
        if not entries:
            return loss
        self._step += 1
        dev = entries[0][0].device
        capturing = torch.cuda.is_current_stream_capturing()
        key = tuple(p.numel() for p, *_ in entries)
        if key != self._map_key:
            if capturing:
                raise RuntimeError("FusedAdamW: the set of tensors changed inside a graph capture (run warm-up steps first)")
            bt, bc = [], []
            for i, n in enumerate(key):
                nb = (n + self._chunk - 1) // self._chunk
                bt += [i] * nb
                bc += list(range(nb))
            self._blk_t = torch.tensor(bt, dtype=torch.int32, device=dev)
            self._blk_c = torch.tensor(bc, dtype=torch.int32, device=dev)
            self._table = (MedpAdamTensor * len(key))()
            nbytes = ctypes.sizeof(self._table)
            self._ring = [torch.empty(nbytes, dtype=torch.uint8).pin_memory() for _ in range(self.RING)]
            self._ring_ev = [torch.cuda.Event() for _ in range(self.RING)]
            self._descs_dev = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            self._map_key = key
        self._entries = entries
        for i, (p, st, lr, wd) in enumerate(entries):
            a = self._table[i]
            a.param, a.grad, a.exp_avg, a.exp_avg_sq = p.data_ptr(), p.grad.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr()
            a.numel, a.lr, a.weight_decay = p.numel(), lr, wd
        if not capturing:            # captured: only the kernel node is recorded; refresh_lrs() uploads before every replay
            self._upload_table()
        if self.dev_step is None or self.dev_step.device != dev:
            self.dev_step = torch.full((1,), self._step - 1, dtype=torch.int32, device=dev)
        check(lib().medp_counter_advance(ptr(self.dev_step), stream()), "counter_advance")
        check(lib().medp_adamw_multi(ptr(self._descs_dev), ptr(self._blk_t), ptr(self._blk_c), self._blk_t.numel(), betas[0], betas[1],
                                     eps, self._step, ptr(self.dev_step), 1.0, stream()), "adamw_multi")
        # the kernel wrote the parameters behind torch's back: bump their version counters (host-side only, no launch) so
        # autograd's saved-tensor checks and the bf16 weight caches keyed on `_version` see the update
        torch.autograd.graph.increment_version([p for p, *_ in entries])
        return loss

    # ---- graph-replay support: the table captured with the step (its gradient pointers) + this step's learning rates ------
    def refresh_lrs(self) -> None:
        """Call right before replaying a graph that captured `step()`, on the stream the replay is enqueued on."""
        lr_of = {id(p): float(g["lr"]) for g in self.param_groups for p in g["params"]}
        for i, (p, *_rest) in enumerate(self._entries):
            self._table[i].lr = lr_of[id(p)]
        self._upload_table()

    def relaunch(self) -> None:
        """Issue the optimiser's two launches again over the table as it stands on the device (same tensors, same gradient
        buffers — the flat-arena step of graph_step.py, where `refresh_lrs()` has just uploaded this step's rates): the whole
        update without rebuilding the table on the host and without a second graph."""
        betas, eps = self.param_groups[0]["betas"], self.param_groups[0]["eps"]
        self._step += 1
        check(lib().medp_counter_advance(ptr(self.dev_step), stream()), "counter_advance")
        check(lib().medp_adamw_multi(ptr(self._descs_dev), ptr(self._blk_t), ptr(self._blk_c), self._blk_t.numel(), betas[0], betas[1],
                                     eps, self._step, ptr(self.dev_step), 1.0, stream()), "adamw_multi")
        torch.autograd.graph.increment_version([p for p, *_ in self._entries])

    def current_lrs(self) -> list:
        """Per-tensor learning rates of the table as last built (tests)."""
        return [float(self._table[i].lr) for i in range(len(self._entries))]

    def note_external_step(self) -> None:
        """A captured replay updated the parameters: advance the host step count and the parameters' version counters."""
        self._step += 1
        torch.autograd.graph.increment_version([p for p, *_ in self._entries])
