"""Whole-step HIP graphs: the MI355X-first way to run the training step (SURVEY.md §7 "HIP streams and graphs instead of a
tracing compiler").  The eager step of engine.py issues ~700 launches from Python and is host-bound; here the complete
step arithmetic — frozen encoders, trainable forward, loss, backward of every trainable parameter, fused AdamW — is captured
ONCE into hipGraphs and replayed per step.  Everything that must differ between replays lives in device memory: the input
batch (static buffers), the dropout RNG epoch (`medp_rng_set_epoch_ptr`), the optimiser step count (`FusedAdamW.dev_step`,
advanced by the captured optimiser itself) and the optimiser's descriptor table with this step's learning rates (uploaded,
stream-ordered, right before each replay through a ring of pinned staging buffers: optim.FusedAdamW.refresh_lrs).

Two step classes share the machinery:
  * `GraphedTeacherStep`  — `engine.train_teacher_dual_pathology_batch` (BASELINE configs[2], engine.py:135-190)
  * `GraphedStudentStep`  — `engine.train_student_batch` (BASELINE configs[3], engine.py:270-301): frozen teacher forward under
    no-grad + student (DuETT trained end to end) forward/backward + StudentKDLoss.

Software pipelining of the FROZEN part: whatever is frozen (the teacher's CXR encoder; in the student step the whole teacher)
depends on nothing the update of batch k produces, so its forward for batch k+1 runs beside the training step of batch k and
hands its result over at the end of the replay.  Every replay still runs exactly one frozen forward, one trainable
forward/backward and one update; results are bit-identical to the unpipelined step (tests/test_gpu_pipeline.py).

One GPU: ONE graph holds all of it (three parallel branches).  N > 1 (`split`): gradients accumulate into a flat fp32
arena (dp.FlatGradArena: only parameters that really receive a gradient, the others keep `.grad = None` as under DDP's
find_unused_parameters) and the step is
      [forward/backward graph (+ the frozen forward of batch k+1 as a parallel branch)] -> RCCL mean all-reduce -> [optimiser graph]
with ONE collective over the whole arena (14.8 MB teacher / 35.6 MB student: latency-bound on xGMI, so one large message).
No collective is captured into a graph.  Measured alternatives (one GPU, size-1 RCCL group, `bench.py` with MEDP_FORCE_PG=1;
DESIGN.md §7): the frozen forward as its OWN graph on its own stream beside the other two graphs, so that collective and update
would hide under the encoder's GEMMs, ran 23 % SLOWER (8.96 k against 11.67 k samples/s): graphs launched on different streams
execute back to back on this runtime, they do not overlap; every extra graph boundary costs ~0.15-0.2 ms, more than the
collective it could hide.  Hence the fewest boundaries: two graphs, the collective between them.
"""
from __future__ import annotations

import os

import torch

from . import autograd_ops as _A, dp, engine
from .streams import new_stream, note_capture_origin
from .abi import check, lib, ptr, stream


def _stacked(v):
    """The collate layout is tuples of per-sample tensors; an already stacked tensor is taken as is."""
    return v if torch.is_tensor(v) else torch.stack(tuple(v))


class _TrainSnapshot:
    """Everything a warm-up step changes that a training run observes: trainable parameters, optimiser state (moments, device and
    host step counters), module buffers (BatchNorm running statistics, `num_batches_tracked`) and the dropout epoch.  The warm-up
    iterations torch.cuda.graphs requires are REAL steps on the example batch; `restore()` puts the model back where the caller
    handed it over, so N graph steps equal N eager steps from the same initial state (tests/test_gpu_pipeline.py)."""

    def __init__(self, params, optimizer, modules, epoch):
        self.params, self.opt, self.modules, self.epoch = list(params), optimizer, list(modules), epoch
        with torch.no_grad():
            self.p = [q.detach().clone() for q in self.params]
            self.bufs = [(m, n, b.detach().clone()) for m in self.modules for n, b in m.named_buffers()]
            self.state = {id(q): {k: (v.detach().clone() if torch.is_tensor(v) else v) for k, v in optimizer.state.get(q, {}).items()}
                          for q in self.params}
            self.host_step = int(getattr(optimizer, "_step", 0))
            self.epoch0 = epoch.detach().clone()

    def restore(self):
        with torch.no_grad():
            for q, v in zip(self.params, self.p):
                q.copy_(v)
            torch.autograd.graph.increment_version(self.params)
            for q in self.params:
                st, old = self.opt.state.get(q, {}), self.state[id(q)]
                for k, v in st.items():
                    if torch.is_tensor(v):                 # moments created by the warm-up go back to their initial zeros
                        v.copy_(old[k]) if k in old else v.zero_()
            for m, n, v in self.bufs:                      # by NAME (the training path may re-bind buffers), and only what moved: writing a
                b = m.get_buffer(n)                        # FROZEN module's buffer bumps its version and its prepared weights are rebuilt
                if not torch.equal(b, v):
                    b.copy_(v)
            self.epoch.copy_(self.epoch0)
            if getattr(self.opt, "dev_step", None) is not None:
                self.opt.dev_step.fill_(self.host_step)
            if hasattr(self.opt, "_step"):
                self.opt._step = self.host_step


# The library mixes ONE device counter into every dropout seed (medp_rng_set_epoch_ptr): the step that registered its counter last owns
# it.  The registered tensor is held here, so the pointer never dangles after its step object is gone (an eager forward afterwards used
# to read freed — possibly re-used — memory: masks that changed from call to call); a step that is collected while it still owns the
# registration takes it back (NULL = no epoch, the per-call seeds alone).
_EPOCH_OWNER = [None]


def _register_epoch(epoch: torch.Tensor):
    check(lib().medp_rng_set_epoch_ptr(ptr(epoch)), "rng_set_epoch_ptr")
    _EPOCH_OWNER[0] = epoch


def _release_epoch(epoch: torch.Tensor):
    if _EPOCH_OWNER[0] is epoch:
        try:
            lib().medp_rng_set_epoch_ptr(None)
        finally:
            _EPOCH_OWNER[0] = None


class _GraphedStep:
    """Capture / replay machinery common to both steps.  Subclasses provide `_frozen_forward()` (frozen part for the NEXT
    batch, on the current stream, returns nothing), `_train_fwd_bwd()` (returns the output dict) and `_hand_over()`."""

    def __del__(self):
        ep = self.__dict__.get("epoch")
        if ep is not None:
            _release_epoch(ep)

    def _setup(self, optimizer, device, world, group, split, pipeline, warmup, before_capture, after_capture=None):
        self.opt, self.device, self.world, self.group = optimizer, device, world, group
        self.pipeline = bool(pipeline)
        self.epoch = torch.zeros(1, dtype=torch.int32, device=device)
        _register_epoch(self.epoch)
        self.params = [p for g in optimizer.param_groups for p in g["params"] if p.requires_grad]
        self.split = bool(split) or world > 1
        self.arena = None
        self.force_collective = False      # bench.py MEDP_FORCE_PG=1: really call RCCL on a size-1 group (one-GPU rehearsal of N > 1)
        self._captured = False
        self._pool = None                  # autograd_ops.WeightOperandPool, built from the last warm-up step
        # the frozen forward may itself be cut into sub-batches on sibling streams (`_n_frozen_parts`): every one is forked from
        # the step's own stream — a fork nested inside a forked branch crashes hipStreamEndCapture on this runtime
        self.frozen_streams = [new_stream(device) for _ in range(self._n_frozen_parts())] if self.pipeline else []
        # warm-up on a side stream (allocator pools, lazy workspaces, optimiser state), as torch.cuda.graphs requires — real steps
        # on the example batch, UNDONE afterwards (the run used to start from a model trained `warmup` times on batch 0)
        snap = _TrainSnapshot(self.params, optimizer, self._stateful_modules(), self.epoch)
        s = new_stream(device)
        s.wait_stream(torch.cuda.current_stream(device))
        with torch.cuda.stream(s):
            for it in range(max(int(warmup), 1)):
                if self.split and self.arena is None:
                    # the first backward tells which parameters the step reaches; only those enter the arena
                    # (not a training step: module buffers — BatchNorm running statistics — and the dropout epoch are restored)
                    bufs = [(m, n, b.detach().clone()) for m in self._stateful_modules() for n, b in m.named_buffers()]
                    used = dp.find_used_parameters(self.params, lambda: (self._advance(), self._whole_fwd_bwd()))
                    self.arena = dp.FlatGradArena(self.params, used=used, group=group)
                    self.arena.bind(zero=True)
                    self.epoch.sub_(1)
                    for m, n, saved in bufs:               # by NAME: the training path may re-bind buffers (stacked BatchNorm statistics)
                        m.get_buffer(n).copy_(saved)
                self._zero_grads()
                self._advance()
                if it == max(int(warmup), 1) - 1 and os.environ.get("MEDP_OPERAND_POOL", "1") == "1":
                    # which trainable weights the step asks GEMM operands of: from the capture on ONE launch makes them all
                    with _A.record_operands() as log:
                        self._whole_fwd_bwd()
                    self._pool = _A.WeightOperandPool(log, device)
                else:
                    self._whole_fwd_bwd()
                self._allreduce()
                self.opt.step()
        torch.cuda.current_stream(device).wait_stream(s)
        torch.cuda.synchronize(device)
        self._zero_grads()
        snap.restore()
        torch.cuda.synchronize(device)
        if before_capture is not None:
            before_capture()
        # Hardware-queue phase.  HIP multiplexes streams over 4 in-order hardware queues, a new stream going to the queue with the
        # fewest streams on it; every raw HIP stream created before the process's FIRST graph capture moves the graph's internal branch
        # streams on by one queue.  One of the four phases is bad — the 38.5-MB pixel copy staged for the next call then costs its full
        # 0.85 ms instead of running beside the replay: teacher 6.05-6.07 ms in phase 2 (5.16-5.25 in the others), student 8.58-8.67 ms
        # in phase 0 (7.84-7.93 in the others); period 4; independent of which stream issues the copy (pool stream, raw stream, the
        # replay on a stream of its own) — what the copy collides with was not identified.  Nor can the phase be changed once the first
        # graph exists: capturing again after more pad streams, or with dummy branches forked ahead of the frozen one, stays in the phase
        # (profiles/r02_ab_experiments.txt sections 7, 9, 10).  Three pad streams is the good phase of every one-process-per-GPU run
        # measured (MEDP_PRE_CAPTURE_STREAMS overrides).  MEDP_PHASE_CHECK=1 (bench.py sets it) MEASURES the phase the step ended up in:
        # a few steps with staged host batches against resident ones on a snapshot of everything a step writes, restored afterwards;
        # `phase_log` = (pad streams, ms lost per step to the staged copy, ms of the copy alone).
        self._pad_streams = [new_stream(device, raw=True) for _ in range(int(os.environ.get("MEDP_PRE_CAPTURE_STREAMS", str(self._default_pad_streams()))))]
        self.copy_stream = new_stream(device)
        self._capture()
        if after_capture is not None:
            after_capture()                                # (bench.py: stop arming launch clocks — the phase check below launches eagerly too)
        self.phase_log = None
        if self.pipeline and os.environ.get("MEDP_PHASE_CHECK", "0") == "1":
            penalty, t_copy = self._staging_penalty()
            self.phase_log = (len(self._pad_streams), round(penalty * 1e3, 3), round(t_copy * 1e3, 3))
        self._captured = True
        self.opt._step = int(self.opt.dev_step.item())      # capture ran opt.step() on the host without executing it
        # The graph holds raw pointers into the PREPARED weights of the frozen modules (DuETT / CXR `_prep`: stacked, folded, bf16)
        # and into the cached bf16 copies of frozen Linear weights.  Those are rebuilt — and the old tensors freed — when somebody
        # writes a frozen parameter or buffer and then calls the module eagerly: keep what the capture saw alive, and refuse to
        # replay over changed weights (`_replay`) instead of training on stale ones.
        self._prep_refs = [(m, m._prep) for root in self._stateful_modules() for m in root.modules() if getattr(m, "_prep", None) is not None]
        self._cache_refs = [v for slot in _A._W_CACHE.values() for v in slot.values()]
        # ... and into the modules' workspaces and pointer / length tables: an eager call with a larger batch or image
        # REPLACES a workspace (cxr.Dinov2Backbone._ws[slot], duett.Model._ws), the 16-entry table cache of feats_to_input evicts.
        # Holding what the capture saw keeps that memory allocated, so a replay after such a call still reads and writes live buffers
        # of its own (the eager call got new ones) instead of freed memory.
        self._keepalive = []
        for root in self._stateful_modules():
            for m in root.modules():
                ws = m.__dict__.get("_ws")
                self._keepalive += list(ws.values()) if isinstance(ws, dict) else ([ws] if torch.is_tensor(ws) else [])
                self._keepalive += list(m.__dict__.get("_tables", {}).values())

    def _check_frozen_unchanged(self):
        for m, prep in self._prep_refs:
            if m._prep is not prep:
                raise RuntimeError(f"{type(m).__name__}: frozen weights were modified after the step was captured; build a new "
                                   f"{type(self).__name__} (the captured graph still reads the old prepared weights)")

    def _capture(self):
        device = self.device
        self.g_opt = None
        self.g_fb = torch.cuda.CUDAGraph()
        if not self.split:
            with torch.cuda.graph(self.g_fb):
                note_capture_origin(device)
                self._advance()
                self.out = self._whole_fwd_bwd(then=self.opt.step)
        else:
            with torch.cuda.graph(self.g_fb):
                note_capture_origin(device)
                self.arena.flat.zero_()
                self._advance()
                self.out = self._whole_fwd_bwd()
            # the update after the collective is two launches over fixed buffers (the arena views never move): issued directly
            # (`FusedAdamW.relaunch`) rather than as a second graph — one graph boundary less per step.  MEDP_SPLIT_OPT=graph
            # keeps the two-graph form for A/B runs.
            if os.environ.get("MEDP_SPLIT_OPT", "eager") == "graph":
                self.g_opt = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self.g_opt):
                    self.opt.step()
        torch.cuda.synchronize(device)
        self.opt._step = int(self.opt.dev_step.item())      # capture ran opt.step() on the host without executing it

    def _written_state(self):
        """Every tensor a replay writes and a later replay (or the caller) reads: parameters, optimiser state and step counter, the
        dropout epoch, module buffers (BatchNorm statistics), the subclass's hand-over buffers."""
        ts = list(self.params)
        for p in self.params:
            ts += [v for v in self.opt.state.get(p, {}).values() if torch.is_tensor(v)]
        ts += [self.opt.dev_step, self.epoch] + list(self._handover_tensors())
        return ts

    def _staging_penalty(self, n: int = 6):
        """(seconds a step takes longer with HOST batches — pinned, staged one call ahead exactly as `step()` does it — than with
        device-resident ones, seconds the pixel copy takes alone).  The steps train: everything they write is snapshotted first and
        restored afterwards (buffers by name: the training path may re-bind them); with more than one rank every rank takes the
        largest penalty measured, so all decide alike."""
        import time
        dev_b, host_b = self._calib_batches()
        state = self._written_state()
        saved = [t.detach().clone() for t in state]
        bufs = [(m, k, b.detach().clone()) for m in self._stateful_modules() for k, b in m.named_buffers()]
        host_step, expect = self.opt._step, self._expect

        def timed(pool, staged):
            ts = []
            for i in range(n + 3):                             # the median step: the first calls upload the graph / prime the pipeline
                torch.cuda.synchronize(self.device)
                t0 = time.perf_counter()
                self.step(pool[i % 2], pool[(i + 1) % 2], pool[i % 2] if staged else None)
                torch.cuda.synchronize(self.device)
                ts.append(time.perf_counter() - t0)
            return sorted(ts[3:])[n // 2]

        t_res, t_host = timed(dev_b, False), timed(host_b, True)
        px = host_b[0]["pixel_values"]
        dst = torch.empty_like(dev_b[0]["pixel_values"])
        torch.cuda.synchronize(self.device)
        t0 = time.perf_counter()
        with torch.cuda.stream(self.copy_stream):
            for _ in range(n):
                dst.copy_(px, non_blocking=True)
        torch.cuda.synchronize(self.device)
        t_copy = (time.perf_counter() - t0) / n
        with torch.no_grad():
            for t, v in zip(state, saved):
                t.copy_(v)
            for m, k, v in bufs:
                b = m.get_buffer(k)
                if not torch.equal(b, v):          # only what a step really moved (BatchNorm statistics of the trained part): writing a FROZEN
                    b.copy_(v)                     # module's buffer bumps its version, its prepared weights are rebuilt — and the graph reads the old ones
        self.opt._step, self._expect, self._staged = host_step, expect, None
        torch.cuda.synchronize(self.device)
        penalty = t_host - t_res
        if self.world > 1:
            pt = torch.tensor([penalty, t_copy], device=self.device, dtype=torch.float64)
            torch.distributed.all_reduce(pt, op=torch.distributed.ReduceOp.MAX, group=self.group)
            penalty, t_copy = float(pt[0]), float(pt[1])
        return penalty, t_copy

    # ---- pieces -------------------------------------------------------------------------------------------------------------
    def _zero_grads(self):
        if self.arena is None:
            self.opt.zero_grad(set_to_none=True)
        else:
            self.arena.bind(zero=True)

    def _advance(self):
        check(lib().medp_counter_advance(ptr(self.epoch), stream()), "counter_advance")
        if self._pool is not None:
            self._pool.refresh()           # bf16 operands of every trainable weight for this step: one launch

    def _allreduce(self):
        if self.arena is not None and (self.world > 1 or self.force_collective):
            self.arena.all_reduce(force=self.force_collective)

    def _whole_fwd_bwd(self, then=None):
        """One-stream-of-control form: the frozen forward of the next batch as a parallel branch inside the same capture.
        `then` (the optimiser update of the one-graph step) runs behind the backward and BEFORE the join: it depends on nothing
        the frozen branch produces, so it hides under that branch's tail instead of extending the step."""
        if not self.pipeline:
            out = self._train_fwd_bwd()
            if then is not None:
                then()
            return out
        cur = torch.cuda.current_stream(self.device)
        for part, fs in enumerate(self.frozen_streams):
            fs.wait_stream(cur)
            with torch.cuda.stream(fs):
                self._frozen_forward(part)
        out = self._train_fwd_bwd()
        if then is not None:
            then()
        for fs in self.frozen_streams:
            cur.wait_stream(fs)
        self._hand_over()
        return out

    def _n_frozen_parts(self) -> int:
        return 1

    def _default_pad_streams(self) -> int:
        return 3

    def _replay(self):
        """refresh_lrs + the graph(s) of one step on the current stream."""
        if self._captured:
            self._check_frozen_unchanged()
        self.opt.refresh_lrs()                       # this step's learning rates -> device table, ahead of the replay
        self.g_fb.replay()
        if not self.split:
            self.opt.note_external_step()
            return self.out
        # split arrangement: HIP events on the step's stream around the collective and the update, every step (a ring of the last
        # 64 steps; read by `split_timings()` — bench.py reports them in the N > 1 line so that the first multi-GPU run shows what
        # the exchange costs per step).  The process group makes this stream wait for the collective, so ev1 fires when it is done.
        ring = self._split_events()
        ring[0].record()
        self._allreduce()                            # RCCL on its own stream, ordered after g_fb by the process group
        ring[1].record()
        if self.g_opt is not None:
            self.g_opt.replay()
            self.opt.note_external_step()
        else:
            self.opt.relaunch()
        ring[2].record()
        return self.out

    def _split_events(self):
        if not hasattr(self, "_sev"):
            self._sev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(64)]
            self._sev_n = 0
        ev = self._sev[self._sev_n % 64]
        self._sev_n += 1
        return ev

    def split_timings(self):
        """Mean milliseconds per step of the gradient all-reduce and of the optimiser update over the last (up to 64) split-mode
        steps, None when the step is not split.  Synchronises the device."""
        n = min(getattr(self, "_sev_n", 0), 64)
        if n == 0:
            return None
        torch.cuda.synchronize(self.device)
        ar = sum(e[0].elapsed_time(e[1]) for e in self._sev[:n]) / n
        up = sum(e[1].elapsed_time(e[2]) for e in self._sev[:n]) / n
        return {"allreduce_ms": round(ar, 4), "optimizer_ms": round(up, 4), "steps": n,
                "collective": "one mean all-reduce of the flat fp32 gradient arena between the forward/backward graph and the optimiser launches"}


class GraphedTeacherStep(_GraphedStep):
    def __init__(self, teacher, loss_fn, optimizer, example_batch: dict, device, world: int = 1, group=None, warmup: int = 3,
                 split: bool = False, before_capture=None, pipeline_cxr: bool = False, after_capture=None):
        self.teacher, self.loss_fn = teacher, loss_fn
        b = engine._move_lists(example_batch, device)
        # static input buffers; the per-sample tuples the model interface wants are views into the stacked buffers
        self.x_ts = torch.stack(b["x_ts"]).contiguous()
        self.x_static = torch.stack(b["x_static"]).contiguous()
        self.bin_ends = torch.stack(b["bin_ends"]).contiguous()
        self.pixels = b["pixel_values"].clone()
        if pipeline_cxr:
            if any(p.requires_grad for p in teacher.cxr.parameters()):
                raise ValueError("pipeline_cxr needs a frozen CXR encoder")
            self.pixels_next = self.pixels.clone()
            with torch.no_grad():
                self.tok_cur = teacher.cxr.forward_bf16(self.pixels).clone()
            self.tok_next = torch.empty_like(self.tok_cur)
            # sub-batches of the frozen encoder on sibling streams: the hardware dispatcher then fills the CUs one sub-batch's GEMM
            # leaves idle in its last round of tiles with the other's kernels (MEDP_CXR_PARTS; measured SLOWER on MI355X — 5.83 ms with 2 parts, 7.26 ms with 4 against 5.30 ms — so the default is 1)
            self.cxr_parts = max(1, min(int(os.environ.get("MEDP_CXR_PARTS", "1")), self.pixels.shape[0]))
            self._expect = None        # the batch (object, held alive: an id() could be reused after a free) whose tokens sit in tok_cur
        self.y_multi = b["y_multi"].clone().float()
        self.y_mask = b["y_multi_mask"].clone().float()
        engine._set_train_with_frozen_eval(teacher)
        self._setup(optimizer, device, world, group, split, pipeline_cxr, warmup, before_capture, after_capture)

    # ---- the three pieces of a step ----------------------------------------------------------------------------------------
    def _n_frozen_parts(self) -> int:
        return self.cxr_parts

    def _frozen_forward(self, part: int = 0):
        B, n = self.pixels_next.shape[0], self.cxr_parts
        lo, hi = part * B // n, (part + 1) * B // n
        with torch.no_grad():                                                    # batch k+1, beside batch k's step
            self.teacher.cxr.forward_bf16(self.pixels_next[lo:hi], slot=part, out=self.tok_next[lo:hi])

    def _stateful_modules(self):
        return [self.teacher]

    def _handover_tensors(self):
        return [self.tok_cur] if self.pipeline else []

    def _calib_batches(self):
        """Two device-resident and two pinned-host copies of the batch in the static buffers (distinct dict objects: `step` tells
        batches apart by identity)."""
        d = {"x_ts": self.x_ts.clone(), "x_static": self.x_static.clone(), "bin_ends": self.bin_ends.clone(),
             "y_multi": self.y_multi.clone(), "y_multi_mask": self.y_mask.clone(), "pixel_values": self.pixels.clone()}
        h = {k: v.cpu().pin_memory() for k, v in d.items()}
        return [dict(d), dict(d)], [dict(h), dict(h)]

    def _hand_over(self):
        self.tok_cur.copy_(self.tok_next)      # after the backward: the weight-gradient GEMM of img_proj reads tok_cur

    def _train_fwd_bwd(self):
        B = self.x_ts.shape[0]
        out = self.teacher(tuple(self.x_ts[i] for i in range(B)), tuple(self.x_static[i] for i in range(B)),
                           tuple(self.bin_ends[i] for i in range(B)), self.pixels,
                           **({"_cxr_tokens16": self.tok_cur} if self.pipeline else {}))
        losses = self.loss_fn(out["img_logits"], out["ts_logits"], out["fusion_logits"], self.y_multi, self.y_mask)
        losses["total"].backward()
        return {"loss": losses["total"].detach(), "img_total": losses["img_total"], "ts_total": losses["ts_total"],
                "fus_total": losses["fus_total"], "fusion_logits": out["fusion_logits"].detach(), "main_logit": out["main_logit"].detach()}

    # ---- one training step --------------------------------------------------------------------------------------------------
    def prime(self, batch: dict) -> None:
        """Pipelined mode: run the CXR encoder for `batch` now, so the next `step(batch, ...)` finds its tokens."""
        with torch.no_grad():
            self.tok_cur.copy_(self.teacher.cxr.forward_bf16(batch["pixel_values"].to(self.device, non_blocking=True)))
        self._expect = batch

    def load_batch(self, batch: dict) -> None:
        """Copy a batch (host or device) into the static input buffers (async on the current stream)."""
        self.x_ts.copy_(_stacked(batch["x_ts"]), non_blocking=True)
        self.x_static.copy_(_stacked(batch["x_static"]), non_blocking=True)
        self.bin_ends.copy_(_stacked(batch["bin_ends"]), non_blocking=True)
        if not self.pipeline:                        # pipelined: this batch's pixels were consumed by the previous replay
            self.pixels.copy_(batch["pixel_values"], non_blocking=True)
        self.y_multi.copy_(batch["y_multi"], non_blocking=True)
        self.y_mask.copy_(batch["y_multi_mask"], non_blocking=True)

    # ---- host batches: staged one call ahead on a copy stream (the pinned-buffer -> device half) ----------
    def _stage_h2d(self, batch: dict, next_batch: dict) -> None:
        """Enqueue, on the copy stream, the host->device copies the NEXT call will need: the small tensors of `batch` and the
        pixels of `next_batch`, into device staging buffers.  With pinned host tensors (a DataLoader with pin_memory=True) the
        38.5-MB pixel copy runs beside the current replay; the next call only pays device-to-device copies (~20 us)."""
        if not hasattr(self, "stage"):
            # (the copy stream — normal priority: a stream of its own HIP priority class, high or low, made the step 7.6-8.9 ms
            # instead of 5.05 — is made in _setup, next to the replay stream)
            self.stage = {k: torch.empty_like(getattr(self, a)) for k, a in
                          (("x_ts", "x_ts"), ("x_static", "x_static"), ("bin_ends", "bin_ends"), ("y_multi", "y_multi"),
                           ("y_multi_mask", "y_mask"), ("pixel_values", "pixels_next"))}
            self.h2d_done, self.stage_free = torch.cuda.Event(), torch.cuda.Event()
            self.stage_free.record(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(self.copy_stream):
            self.copy_stream.wait_event(self.stage_free)          # the previous call has taken its data out of the staging buffers
            for k in ("x_ts", "x_static", "bin_ends", "y_multi", "y_multi_mask"):
                self.stage[k].copy_(_stacked(batch[k]), non_blocking=True)
            self.stage["pixel_values"].copy_(next_batch["pixel_values"], non_blocking=True)
            self.h2d_done.record(self.copy_stream)
        self._staged = (batch, next_batch)

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
        st = getattr(self, "_staged", None)
        staged = self.pipeline and batch is not None and next_batch is not None and st is not None and st[0] is batch and st[1] is next_batch
        if staged:
            if self._expect is not batch:
                self.prime(batch)
            self._take_staged()
            self._expect = next_batch
        else:
            if batch is not None:
                self.load_batch(batch)
            if self.pipeline:
                if batch is not None and self._expect is not batch:
                    self.prime(batch)
                nb = next_batch if next_batch is not None else batch
                if nb is not None:
                    self.pixels_next.copy_(nb["pixel_values"], non_blocking=True)
                    self._expect = nb
        if self.pipeline and after_next is not None and next_batch is not None and not next_batch["pixel_values"].is_cuda:
            self._stage_h2d(next_batch, after_next)
        else:
            self._staged = None
        return self._replay()


class GraphedStudentStep(_GraphedStep):
    """`engine.train_student_batch` as captured graphs (engine.py:270-301; the loaders are all in "teacher" mode so one batch
    feeds both models, trainer.py:890-895,923).  `pipeline_teacher=True`: the frozen teacher's forward for batch k+1 (CXR
    encoder + its own frozen DuETT + fusion head, `main_logit` only) runs beside the student's step on batch k."""

    def __init__(self, student, teacher, kd_loss_fn, optimizer, example_batch: dict, device, world: int = 1, group=None,
                 warmup: int = 3, split: bool = False, before_capture=None, pipeline_teacher: bool = True, after_capture=None,
                 swap_roles=None):
        if any(p.requires_grad for p in teacher.parameters()):
            raise ValueError("the KD teacher must be frozen (trainer.py:856-865)")
        self.student, self.teacher, self.loss_fn = student, teacher, kd_loss_fn
        import inspect
        self._teacher_kw = set(inspect.signature(teacher.forward).parameters)      # a stand-in teacher (tests) may lack the private switch
        b = engine._move_lists(example_batch, device)
        mk = lambda: {"x_ts": torch.stack(b["x_ts"]).contiguous(), "x_static": torch.stack(b["x_static"]).contiguous(),
                      "bin_ends": torch.stack(b["bin_ends"]).contiguous(), "pixel_values": b["pixel_values"].clone()}
        self.cur = mk()                                   # the batch the student trains on (and, unpipelined, the teacher reads)
        self.nxt = mk() if pipeline_teacher else None     # the batch the teacher runs ahead on
        self.y = b["y"].clone().float()
        student.train()
        teacher.eval()
        with torch.no_grad():
            self.z_cur = self._teacher_logit(self.cur).clone()
        self.z_next = None
        self._expect = None
        # roles swapped (see _whole_fwd_bwd): the frozen teacher on the step's own stream, where it may fork its time-series half beside its
        # CXR encoder.  Round 2 measured this slower (9.00 vs 8.74 ms); with round 3's lighter training branch it is 3.7 % faster (6.46 vs
        # 6.70 ms, profiles/r03_ab_experiments.txt): default on, MEDP_STUDENT_SWAP=0 / swap_roles=False for the two-branch form.
        self.swap_roles = (os.environ.get("MEDP_STUDENT_SWAP", "1") == "1") if swap_roles is None else bool(swap_roles)
        # MEDP_STUDENT_GEMM_CAP = n: the frozen teacher's persistent GEMMs hold at most n CUs per launch.  Round 2 shipped 176 (the student's
        # branch was the long one: 8.47 ms against 8.70 uncapped); with the fused encoder halves and embedding kernels of round 3 the training
        # branch needs fewer CUs and the cap costs 4 % (7.02 ms at 176, 6.73-6.76 at 200 / 224 / none, profiles/r03_ab_experiments.txt): off.
        prev = lib().medp_gemm_persistent_cap(int(os.environ.get("MEDP_STUDENT_GEMM_CAP", "0")) if pipeline_teacher else 0)
        try:
            self._setup(optimizer, device, world, group, split, pipeline_teacher, warmup, before_capture, after_capture)
        finally:
            lib().medp_gemm_persistent_cap(prev)

    def _default_pad_streams(self) -> int:
        # three branches instead of two move the bad hardware-queue phase: with the roles swapped it is 3 pad streams (the staged pixel copy
        # then costs 0.26-0.4 ms per step: 8.58 k PCIe-inclusive against 9.54-9.57 k at 0 / 1 / 2, profiles/r03_ab_experiments.txt)
        return 1 if self.swap_roles else 3

    def _teacher_logit(self, bufs, forked: bool = False):
        B = bufs["x_ts"].shape[0]
        kw = {"_overlap": False} if (forked and "_overlap" in self._teacher_kw) else {}
        return self.teacher(tuple(bufs["x_ts"][i] for i in range(B)), tuple(bufs["x_static"][i] for i in range(B)),
                            tuple(bufs["bin_ends"][i] for i in range(B)), bufs["pixel_values"], **kw)["main_logit"]

    def _frozen_forward(self, part: int = 0):
        # as a forked branch of the capture this forward must not fork again (nested forks crash hipStreamEndCapture here)
        with torch.no_grad():
            self.z_next = self._teacher_logit(self.nxt, forked=not self.swap_roles)

    def _whole_fwd_bwd(self, then=None):
        """Roles swapped against the base class: the STUDENT's forward/backward/update is the forked branch and the frozen teacher
        runs on the step's own stream — there it may fork its time-series half beside its CXR encoder (a sibling of the training
        branch, not a nested fork), so the frozen branch is max(encoder, DuETT + heads) long instead of their sum."""
        if not (self.pipeline and self.swap_roles):
            return super()._whole_fwd_bwd(then)
        cur = torch.cuda.current_stream(self.device)
        ts = self.frozen_streams[0]
        ts.wait_stream(cur)
        with torch.cuda.stream(ts):
            out = self._train_fwd_bwd()
            if then is not None:
                then()
        self._frozen_forward()
        cur.wait_stream(ts)
        self._hand_over()
        return out

    def _stateful_modules(self):
        return [self.student, self.teacher]

    def _handover_tensors(self):
        return [self.z_cur]

    def _calib_batches(self):
        d = {k: v.clone() for k, v in self.cur.items()}
        d["y"] = self.y.clone()
        h = {k: v.cpu().pin_memory() for k, v in d.items()}
        return [dict(d), dict(d)], [dict(h), dict(h)]

    def _hand_over(self):
        if not self._captured:
            self.z_next.record_stream(torch.cuda.current_stream(self.device))
        self.z_cur.copy_(self.z_next)

    def _train_fwd_bwd(self):
        if not self.pipeline:
            with torch.no_grad():
                self.z_cur.copy_(self._teacher_logit(self.cur))
        B = self.cur["x_ts"].shape[0]
        z_s = self.student(tuple(self.cur["x_ts"][i] for i in range(B)), tuple(self.cur["x_static"][i] for i in range(B)),
                           tuple(self.cur["bin_ends"][i] for i in range(B)))
        losses = self.loss_fn(z_s, self.z_cur, self.y)
        losses["total"].backward()
        return {"loss": losses["total"].detach(), "bce": losses["bce"], "kd": losses["kd"], "logits": z_s.detach()}

    @staticmethod
    def _load(bufs, batch, with_pixels=True):
        bufs["x_ts"].copy_(_stacked(batch["x_ts"]), non_blocking=True)
        bufs["x_static"].copy_(_stacked(batch["x_static"]), non_blocking=True)
        bufs["bin_ends"].copy_(_stacked(batch["bin_ends"]), non_blocking=True)
        if with_pixels:
            bufs["pixel_values"].copy_(batch["pixel_values"], non_blocking=True)

    def prime(self, batch: dict) -> None:
        """Pipelined mode: run the teacher for `batch` now (its logit was not produced by the previous replay)."""
        self._load(self.nxt, batch)
        with torch.no_grad():
            self.z_cur.copy_(self._teacher_logit(self.nxt))
        self._expect = batch

    # ---- host batches: the next call's host->device copies beside this replay (as GraphedTeacherStep) --------------------------
    def _stage_h2d(self, next_batch: dict, after_next: dict) -> None:
        """What the NEXT call needs from the host: `next_batch`'s labels (its series are already on the device: this call's teacher
        reads them) and ALL of `after_next` (the batch the teacher will run ahead on then) -> device staging buffers, on a copy stream."""
        if not hasattr(self, "stage"):
            self.stage = {k: torch.empty_like(v) for k, v in self.nxt.items()}
            self.stage["y"] = torch.empty_like(self.y)
            self.h2d_done, self.stage_free = torch.cuda.Event(), torch.cuda.Event()
            self.stage_free.record(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(self.copy_stream):
            self.copy_stream.wait_event(self.stage_free)          # the previous call has taken its data out of the staging buffers
            self._load(self.stage, after_next)
            self.stage["y"].copy_(next_batch["y"], non_blocking=True)
            self.h2d_done.record(self.copy_stream)
        self._staged = (next_batch, after_next)

    def _take_staged(self) -> None:
        cur = torch.cuda.current_stream(self.device)
        cur.wait_event(self.h2d_done)
        for k in ("x_ts", "x_static", "bin_ends"):                 # the batch the teacher ran ahead on becomes the student's batch
            self.cur[k].copy_(self.nxt[k])
        self.y.copy_(self.stage["y"])
        for k, v in self.nxt.items():
            v.copy_(self.stage[k])
        self.stage_free.record(cur)

    def step(self, batch: dict | None = None, next_batch: dict | None = None, after_next: dict | None = None) -> dict:
        """`after_next` (pipelined mode, HOST batches in pinned memory): the batch after `next_batch`; when given, the next call's
        host->device copies are issued on a copy stream now, beside this replay, as in GraphedTeacherStep.  (Measured: unlike the
        teacher step — 98 % of its resident-batch rate this way — the student step stays ~0.85 ms behind its resident-batch time with
        or without the staging: the 38.5-MB pixel copy costs its full duration wherever it is issued (before or after the launch,
        any pool stream, a raw HIP stream), except with GPU_MAX_HW_QUEUES = 3 or 8 where it overlaps (7.8 ms) — DESIGN.md section 6.)"""
        st = getattr(self, "_staged", None)
        staged = self.pipeline and batch is not None and next_batch is not None and st is not None and st[0] is batch and st[1] is next_batch
        if staged:
            if self._expect is not batch:
                self.prime(batch)
            self._take_staged()
            self._expect = next_batch
        else:
            if batch is not None:
                if self.pipeline and self._expect is not batch:
                    self.prime(batch)
                self._load(self.cur, batch, with_pixels=not self.pipeline)
                self.y.copy_(batch["y"], non_blocking=True)
            if self.pipeline:
                nb = next_batch if next_batch is not None else batch
                if nb is not None:
                    self._load(self.nxt, nb)
                    self._expect = nb
        if self.pipeline and after_next is not None and next_batch is not None and not next_batch["pixel_values"].is_cuda:
            self._stage_h2d(next_batch, after_next)
        else:
            self._staged = None
        return self._replay()


class GraphedProbeStep:
    """The linear-probe step (BASELINE configs[1]; cxr_linear_training.ipynb:396-437, 600-640) as ONE captured graph: frozen encoder
    forward -> CLS -> dropout -> Linear -> masked BCE -> backward -> AdamW.  There is nothing to pipeline (the head is a few launches);
    the capture removes the eager step's launch gaps (110 encoder launches from one C call + ~15 head launches: 4.63 -> ~4.3 ms).
    `step(pixel_values, y_multi, y_multi_mask)` copies a batch (host or device) into the static buffers and replays."""

    def __del__(self):
        ep = self.__dict__.get("epoch")
        if ep is not None:
            _release_epoch(ep)

    def __init__(self, probe, loss_fn, optimizer, example_pixels, example_y, example_mask, device, warmup: int = 3, before_capture=None):
        self.probe, self.loss_fn, self.opt, self.device = probe, loss_fn, optimizer, device
        self.pixels = example_pixels.to(device).clone()
        self.y = example_y.to(device).float().clone()
        self.mask = example_mask.to(device).float().clone()
        self.epoch = torch.zeros(1, dtype=torch.int32, device=device)
        _register_epoch(self.epoch)
        snap = _TrainSnapshot([p for g in optimizer.param_groups for p in g["params"] if p.requires_grad], optimizer, [probe], self.epoch)
        s = new_stream(device)
        s.wait_stream(torch.cuda.current_stream(device))
        with torch.cuda.stream(s):
            for _ in range(max(int(warmup), 1)):
                self._whole()
        torch.cuda.current_stream(device).wait_stream(s)
        torch.cuda.synchronize(device)
        self.opt.zero_grad(set_to_none=True)
        snap.restore()                                      # the warm-up steps trained: put the head and its optimiser state back
        torch.cuda.synchronize(device)
        if before_capture is not None:
            before_capture()
        self.g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.g):
            note_capture_origin(device)
            self.out = self._whole()
        torch.cuda.synchronize(device)
        self.opt._step = int(self.opt.dev_step.item())      # capture ran opt.step() on the host without executing it

    def _whole(self):
        check(lib().medp_counter_advance(ptr(self.epoch), stream()), "counter_advance")
        self.opt.zero_grad(set_to_none=True)
        loss = self.loss_fn(self.probe(self.pixels), self.y, self.mask)
        loss.backward()
        self.opt.step()
        return {"loss": loss.detach()}

    def step(self, pixel_values=None, y_multi=None, y_multi_mask=None) -> dict:
        if pixel_values is not None and pixel_values.data_ptr() != self.pixels.data_ptr():
            self.pixels.copy_(pixel_values, non_blocking=True)
        if y_multi is not None:
            self.y.copy_(y_multi, non_blocking=True)
            self.mask.copy_(y_multi_mask, non_blocking=True)
        self.opt.refresh_lrs()
        self.g.replay()
        self.opt.note_external_step()
        return self.out
