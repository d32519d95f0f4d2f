"""Whole-step HIP graph: the MI355X-first way to run the training step (SURVEY.md §7 "HIP streams and graphs instead of a
tracing compiler").  The eager step of engine.py issues ~700 launches from Python and is host-bound; here the complete
`train_teacher_dual_pathology_batch` arithmetic — both frozen encoders, the fusion head forward, DualPathologyLoss, the
backward of every trainable parameter and the fused AdamW update — is captured ONCE into a hipGraph and replayed per
step.  Everything that must differ between replays lives in device memory: the input batch (static buffers), the dropout
RNG epoch (`medp_rng_set_epoch_ptr`), the optimiser step count (`FusedAdamW.dev_step`, advanced by the captured optimiser itself) and the per-group learning rates
(the descriptor table is re-uploaded from pinned host memory by a captured memcpy node).

N > 1: forward+backward are one graph accumulating into a flat fp32 gradient arena, the arena is all-reduced by RCCL
between the two replays (one collective, 14.8 MB), and the optimiser is a second graph.  With `pipeline_cxr` the frozen
encoder of the next batch CAN be split across the two graphs (`MEDP_SPLIT_VIT_LAYERS=L`, `medp_vit_forward_part`: layers
[0, L) beside forward/backward, layers [L, 12) beside the optimiser) so that the collective and the update need not wait for
the whole encoder.  Measured on one GPU in the N > 1 arrangement it LOSES (L = 8 / 7 / 5: 10.2 / 9.9 / 9.5 k samples/s
against 11.5 k with the encoder whole in the first graph, bit-identical results): the two branches of the step time-share
the CUs rather than fill each other's gaps, so layers moved behind the optimiser simply run later.  Default 0 (whole).

`pipeline_cxr=True` (software pipelining across steps): the CXR encoder is frozen, so its tokens for batch k+1 depend on
nothing the training step of batch k produces.  The captured step then holds THREE parallel branches: the image half +
backward + optimiser of batch k (reading the tokens the previous replay left in `tok_cur`), the time-series half on the
side stream, and the encoder forward of batch k+1 on a third stream; the replay ends by moving the new tokens into
`tok_cur`.  Every replay still runs exactly one encoder forward, one fusion forward/backward and one update — the
encoder pass is just shifted one batch ahead (the first batch's pass happens in `prime`).  Results are bit-identical to the
unpipelined step (tests/test_gpu_pipeline.py).
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist

from . import engine
from .abi import check, lib, ptr, stream


class GraphedTeacherStep:
    def __init__(self, teacher, loss_fn, optimizer, example_batch: dict, device, world: int = 1, group=None, warmup: int = 3,
                 split: bool = False, before_capture=None, pipeline_cxr: bool = False):
        self.teacher, self.loss_fn, self.opt, self.device, self.world, self.group = teacher, loss_fn, optimizer, device, world, group
        b = engine._move_lists(example_batch, device)
        # static input buffers; the per-sample tuples the model interface wants are views into the stacked buffers
        self.x_ts = torch.stack(b["x_ts"]).contiguous()
        self.x_static = torch.stack(b["x_static"]).contiguous()
        self.bin_ends = torch.stack(b["bin_ends"]).contiguous()
        self.pixels = b["pixel_values"].clone()
        self.pipeline = bool(pipeline_cxr)
        if self.pipeline:
            if any(p.requires_grad for p in teacher.cxr.parameters()):
                raise ValueError("pipeline_cxr needs a frozen CXR encoder")
            self.pixels_next = self.pixels.clone()
            self.vit_stream = torch.cuda.Stream(device=device)
            with torch.no_grad():
                self.tok_cur = teacher.cxr.forward_bf16(self.pixels).clone()
            self._expect = None        # id() of the batch whose tokens sit in tok_cur
        self.y_multi = b["y_multi"].clone().float()
        self.y_mask = b["y_multi_mask"].clone().float()
        self.epoch = torch.zeros(1, dtype=torch.int32, device=device)
        check(lib().medp_rng_set_epoch_ptr(ptr(self.epoch)), "rng_set_epoch_ptr")
        self.params = [p for g in optimizer.param_groups for p in g["params"] if p.requires_grad]
        self.flat_grad = None
        self.split = split or world > 1
        if self.split:
            n = sum(p.numel() for p in self.params)
            self.flat_grad = torch.zeros(n, dtype=torch.float32, device=device)
        # split + pipelined: encoder layers [0, vit_split) run in the forward/backward graph, the rest beside the optimiser
        self.vit_split = 0
        if self.split and self.pipeline:
            n_layers = teacher.cxr.backbone.cfg.num_hidden_layers
            self.vit_split = max(0, min(n_layers, int(os.environ.get("MEDP_SPLIT_VIT_LAYERS", "0"))))
            if self.vit_split == n_layers:
                self.vit_split = 0
        engine._set_train_with_frozen_eval(teacher)
        # warm-up on a side stream (allocator pools, lazy workspaces, optimiser state), as torch.cuda.graphs requires
        s = torch.cuda.Stream(device=device)
        s.wait_stream(torch.cuda.current_stream(device))
        with torch.cuda.stream(s):
            for _ in range(warmup):
                self._zero_grads()
                self._advance()
                self._fwd_bwd()
                self._allreduce()
                self._opt_step()
        torch.cuda.current_stream(device).wait_stream(s)
        torch.cuda.synchronize(device)
        self._zero_grads()
        if before_capture is not None:
            before_capture()                         # e.g. arm the GEMM timing events so they become nodes of the graph
        self.g_fb = torch.cuda.CUDAGraph()
        if not self.split:
            with torch.cuda.graph(self.g_fb):
                self._advance()
                self.out = self._fwd_bwd()
                self.opt.step()
            self.g_opt = None
        else:
            with torch.cuda.graph(self.g_fb):
                self.flat_grad.zero_()
                self._advance()
                self.out = self._fwd_bwd()
            self.g_opt = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.g_opt):
                self._opt_step()
        torch.cuda.synchronize(device)
        self.opt._step = int(self.opt.dev_step.item())      # capture ran opt.step() on the host without executing it

    # ---- pieces ---------------------------------------------------------------------------------------------------------
    def _zero_grads(self):
        if self.flat_grad is None:
            self.opt.zero_grad(set_to_none=True)
        else:
            off = 0
            self.flat_grad.zero_()
            for p in self.params:                  # gradients accumulate straight into the flat arena (no packing step)
                p.grad = self.flat_grad[off:off + p.numel()].view_as(p)
                off += p.numel()

    def _advance(self):
        check(lib().medp_counter_advance(ptr(self.epoch), stream()), "counter_advance")

    def _fwd_bwd(self):
        B = self.x_ts.shape[0]
        tok_next = None
        if self.pipeline:
            cur = torch.cuda.current_stream(self.device)
            self.vit_stream.wait_stream(cur)
            with torch.cuda.stream(self.vit_stream), torch.no_grad():
                if self.vit_split:
                    self.teacher.cxr.forward_bf16_part(self.pixels_next, 0, self.vit_split)      # the rest: _opt_step
                else:
                    tok_next = self.teacher.cxr.forward_bf16(self.pixels_next)      # batch k+1, beside batch k's step
        out = self.teacher(tuple(self.x_ts[i] for i in range(B)), tuple(self.x_static[i] for i in range(B)),
                           tuple(self.bin_ends[i] for i in range(B)), self.pixels,
                           **({"_cxr_tokens16": self.tok_cur} if self.pipeline else {}))
        losses = self.loss_fn(out["img_logits"], out["ts_logits"], out["fusion_logits"], self.y_multi, self.y_mask)
        losses["total"].backward()
        if self.pipeline:
            cur.wait_stream(self.vit_stream)
            if not self.vit_split:
                tok_next.record_stream(cur)
                self.tok_cur.copy_(tok_next)      # after the backward: the weight-gradient GEMM of img_proj reads tok_cur
        return {"loss": losses["total"].detach(), "img_total": losses["img_total"], "ts_total": losses["ts_total"],
                "fus_total": losses["fus_total"], "fusion_logits": out["fusion_logits"].detach(), "main_logit": out["main_logit"].detach()}

    def _opt_step(self):
        """The optimiser; in the split + pipelined step also the remaining encoder layers of the next batch, beside it."""
        if not self.vit_split:
            self.opt.step()
            return
        cur = torch.cuda.current_stream(self.device)
        self.vit_stream.wait_stream(cur)
        with torch.cuda.stream(self.vit_stream), torch.no_grad():
            n_layers = self.teacher.cxr.backbone.cfg.num_hidden_layers
            tok_next = self.teacher.cxr.forward_bf16_part(self.pixels_next, self.vit_split, n_layers)
        self.opt.step()
        cur.wait_stream(self.vit_stream)
        tok_next.record_stream(cur)
        self.tok_cur.copy_(tok_next)

    def _allreduce(self):
        if self.world > 1 and dist.is_initialized():
            op = dist.ReduceOp.AVG if dist.get_backend(self.group) == "nccl" else dist.ReduceOp.SUM
            dist.all_reduce(self.flat_grad, op=op, group=self.group)
            if op == dist.ReduceOp.SUM:
                self.flat_grad.div_(self.world)

    # ---- one training step --------------------------------------------------------------------------------------------------
    def prime(self, batch: dict) -> None:
        """Pipelined mode: run the CXR encoder for `batch` now, so the next `step(batch, ...)` finds its tokens."""
        with torch.no_grad():
            self.tok_cur.copy_(self.teacher.cxr.forward_bf16(batch["pixel_values"].to(self.device, non_blocking=True)))
        self._expect = id(batch)

    def load_batch(self, batch: dict) -> None:
        """Copy a batch (host or device) into the static input buffers (async on the current stream)."""
        def stacked(v):           # the collate layout is tuples of per-sample tensors; an already stacked tensor is taken as is
            return v if torch.is_tensor(v) else torch.stack(tuple(v))
        self.x_ts.copy_(stacked(batch["x_ts"]), non_blocking=True)
        self.x_static.copy_(stacked(batch["x_static"]), non_blocking=True)
        self.bin_ends.copy_(stacked(batch["bin_ends"]), non_blocking=True)
        if not self.pipeline:                        # pipelined: this batch's pixels were consumed by the previous replay
            self.pixels.copy_(batch["pixel_values"], non_blocking=True)
        self.y_multi.copy_(batch["y_multi"], non_blocking=True)
        self.y_mask.copy_(batch["y_multi_mask"], non_blocking=True)

    # ---- host batches: staged one call ahead on a copy stream (the pinned-buffer -> device half) ----------
    def _stage_h2d(self, batch: dict, next_batch: dict) -> None:
        """Enqueue, on the copy stream, the host->device copies the NEXT call will need: the small tensors of `batch` and the
        pixels of `next_batch`, into device staging buffers.  With pinned host tensors (a DataLoader with pin_memory=True) the
        38.5-MB pixel copy runs beside the current replay; the next call only pays device-to-device copies (~20 us)."""
        if not hasattr(self, "copy_stream"):
            self.copy_stream = torch.cuda.Stream(device=self.device)
            self.stage = {k: torch.empty_like(getattr(self, a)) for k, a in
                          (("x_ts", "x_ts"), ("x_static", "x_static"), ("bin_ends", "bin_ends"), ("y_multi", "y_multi"),
                           ("y_multi_mask", "y_mask"), ("pixel_values", "pixels_next"))}
            self.h2d_done, self.stage_free = torch.cuda.Event(), torch.cuda.Event()
            self.stage_free.record(torch.cuda.current_stream(self.device))
        def stacked(v):
            return v if torch.is_tensor(v) else torch.stack(tuple(v))
        with torch.cuda.stream(self.copy_stream):
            self.copy_stream.wait_event(self.stage_free)          # the previous call has taken its data out of the staging buffers
            for k in ("x_ts", "x_static", "bin_ends", "y_multi", "y_multi_mask"):
                self.stage[k].copy_(stacked(batch[k]), non_blocking=True)
            self.stage["pixel_values"].copy_(next_batch["pixel_values"], non_blocking=True)
            self.h2d_done.record(self.copy_stream)
        self._staged = (id(batch), id(next_batch))

    def _take_staged(self) -> None:
        cur = torch.cuda.current_stream(self.device)
        cur.wait_event(self.h2d_done)
        self.x_ts.copy_(self.stage["x_ts"]); self.x_static.copy_(self.stage["x_static"]); self.bin_ends.copy_(self.stage["bin_ends"])
        self.y_multi.copy_(self.stage["y_multi"]); self.y_mask.copy_(self.stage["y_multi_mask"])
        self.pixels_next.copy_(self.stage["pixel_values"])
        self.stage_free.record(cur)

    def step(self, batch: dict | None = None, next_batch: dict | None = None, after_next: dict | None = None) -> dict:
        """Replay the captured step; returns device tensors (no host sync — read them with .item() when needed).
        Pipelined mode: `next_batch` is the batch the NEXT call will train on (its CXR tokens are produced by this replay);
        if the caller breaks that promise the tokens are recomputed on the spot.
        `after_next` (pipelined mode, HOST batches): the batch after `next_batch`; when given, the host->device copies of the
        next call (`next_batch`'s small tensors, `after_next`'s pixels) are issued on a copy stream now and overlap this replay."""
        staged = self.pipeline and batch is not None and next_batch is not None and \
            getattr(self, "_staged", None) == (id(batch), id(next_batch))
        if staged:
            if self._expect != id(batch):
                self.prime(batch)
            self._take_staged()
            self._expect = id(next_batch)
        else:
            if batch is not None:
                self.load_batch(batch)
            if self.pipeline:
                if batch is not None and self._expect != id(batch):
                    self.prime(batch)
                nb = next_batch if next_batch is not None else batch
                if nb is not None:
                    self.pixels_next.copy_(nb["pixel_values"], non_blocking=True)
                    self._expect = id(nb)
        if self.pipeline and after_next is not None and next_batch is not None and not next_batch["pixel_values"].is_cuda:
            self._stage_h2d(next_batch, after_next)
        else:
            self._staged = None
        self.opt.refresh_lrs()                       # learning rates of this step (scheduler) -> pinned descriptor table
        self.g_fb.replay()
        if self.g_opt is not None:
            self._allreduce()
            self.g_opt.replay()
        self.opt.note_external_step()
        return self.out
