"""Data parallelism for the hot path: one process per GPU, model replicated, minibatch sharded by rank,
ONE exchange per optimiser step — the mean all-reduce of the trainable-parameter gradients, issued through
torch.distributed (backend "nccl" = RCCL over xGMI on ROCm; "gloo" on CPU for the tests) and overlapped with backward.

Design for xGMI rather than NVSwitch: the payload is small (14.8 MB teacher / 35.6 MB student fp32 at cfg3), so the
cost is latency, not bandwidth: the gradients are packed into a few large flat buckets (default 2) filled in
backward order, each bucket's all-reduce is launched from the autograd hook of its last gradient (so it runs on
RCCL's stream under the rest of backward), and `optimizer.step()` waits for them through a step pre-hook.  Parameters
that never receive a gradient (DuETT's SSL heads, duett.py:110-122 — the reason the reference needs
`find_unused_parameters=True`, trainer.py:217) are simply left out of the step: they keep `.grad = None` on every rank.

The reference's own trainer gets the same behaviour from `accelerate` -> torch DDP over RCCL, which wraps this
package's modules unchanged; this reducer is what bench.py and `train_synthetic` use.
"""
from __future__ import annotations

import os
from typing import Iterable, Optional

import torch
import torch.distributed as dist


def init_distributed(backend: Optional[str] = None) -> tuple[int, int, int]:
    """Read RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment (torch.distributed.run contract)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if backend is None:
            # MEDP_DIST_BACKEND=gloo: rehearsal of the N > 1 path with several ranks on ONE GPU (RCCL refuses two ranks on a device)
            backend = os.environ.get("MEDP_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local, world


def shard_indices(n_items: int, rank: int, world: int) -> range:
    """Rank r takes items r, r+N, r+2N, ... of every global batch (accelerate's BatchSamplerShard split, §8e)."""
    return range(rank, n_items, world)


@torch.no_grad()
def broadcast_parameters(module: torch.nn.Module, src: int = 0, group=None) -> None:
    """Replicate rank `src`'s parameters and buffers (what DDP does when it wraps a model), in one flat message per dtype."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    by_dtype: dict = {}
    for t in list(module.parameters()) + list(module.buffers()):
        by_dtype.setdefault((t.dtype, t.device), []).append(t)
    for (dtype, dev), ts in by_dtype.items():
        flat = torch.cat([t.detach().reshape(-1) for t in ts])
        dist.broadcast(flat, src=src, group=group)
        off = 0
        for t in ts:
            n = t.numel()
            t.detach().copy_(flat[off:off + n].view_as(t))
            off += n


class GradAllReducer:
    """Bucketed, backward-overlapped mean all-reduce of gradients.

    usage:  red = GradAllReducer(model.parameters()); red.attach(optimizer)   # then the usual zero_grad/backward/step
    """

    def __init__(self, params: Iterable[torch.nn.Parameter], group=None, n_buckets: int = 2):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.params = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("GradAllReducer: no trainable parameters")
        dev, dtype = self.params[0].device, self.params[0].dtype
        if any(p.device != dev or p.dtype != dtype for p in self.params):
            raise ValueError("GradAllReducer: parameters must share one device and dtype")
        # gradients become ready roughly in reverse registration order: fill buckets in that order
        order = list(reversed(self.params))
        total = sum(p.numel() for p in order)
        per = (total + n_buckets - 1) // max(n_buckets, 1)
        self.buckets = []          # dicts: flat, slots {param: (off, n)}, pending, work
        cur, cur_n = [], 0
        for p in order:
            cur.append(p)
            cur_n += p.numel()
            if cur_n >= per:
                self._new_bucket(cur, dev, dtype)
                cur, cur_n = [], 0
        if cur:
            self._new_bucket(cur, dev, dtype)
        self._where = {}
        for bi, b in enumerate(self.buckets):
            for p in b["slots"]:
                self._where[p] = bi
        self._hooks = [p.register_post_accumulate_grad_hook(self._on_grad) for p in self.params]
        self._opt_hook = None
        self.bytes_per_step = total * self.params[0].element_size()
        self._avg_op = None

    def _new_bucket(self, ps, dev, dtype):
        n = sum(p.numel() for p in ps)
        slots, off = {}, 0
        for p in ps:
            slots[p] = (off, p.numel())
            off += p.numel()
        self.buckets.append({"flat": torch.zeros(n, dtype=dtype, device=dev), "slots": slots, "fired": set(), "work": None,
                             "launched": False})

    # ---- autograd hook: pack this gradient; launch the bucket's all-reduce when it is complete ----
    def _on_grad(self, p: torch.nn.Parameter) -> None:
        b = self.buckets[self._where[p]]
        if b["launched"]:
            raise RuntimeError("GradAllReducer: gradient arrived after its bucket was reduced (finalize() missing between steps?)")
        off, n = b["slots"][p]
        view = b["flat"][off:off + n].view_as(p)
        view.copy_(p.grad)
        p.grad = view                        # the optimiser reads the reduced values straight out of the bucket
        b["fired"].add(p)
        if p.is_cuda:
            # the teacher's backward runs on two HIP streams (main_architecture_duett.TeacherModel.forward): remember where
            # this slice was packed so the launch can wait for packs made on the other stream
            s = torch.cuda.current_stream(p.device)
            ev = torch.cuda.Event()
            ev.record(s)
            b.setdefault("packs", {})[s.cuda_stream] = ev
        if len(b["fired"]) == len(b["slots"]):
            self._launch(b)

    def _launch(self, b) -> None:
        b["launched"] = True
        if self.world == 1:
            b.pop("packs", None)
            return
        packs = b.pop("packs", None)
        if packs:
            cur = torch.cuda.current_stream(b["flat"].device)
            for sid, ev in packs.items():
                if sid != cur.cuda_stream:
                    cur.wait_event(ev)
        if self._avg_op is None:
            backend = dist.get_backend(self.group)
            self._avg_op = dist.ReduceOp.AVG if backend == "nccl" else dist.ReduceOp.SUM
        b["work"] = dist.all_reduce(b["flat"], op=self._avg_op, group=self.group, async_op=True)

    def finalize(self) -> None:
        """Call after backward, before the optimiser step (attach() does it for you)."""
        for b in self.buckets:
            if not b["launched"]:
                missing = [p for p in b["slots"] if p not in b["fired"]]
                if len(missing) == len(b["slots"]):
                    b["fired"].clear()
                    continue                      # nothing in this bucket took part in this step (same on every rank)
                for p in missing:                 # unused this step: contributes zeros, keeps .grad = None
                    off, n = b["slots"][p]
                    b["flat"][off:off + n].zero_()
                self._launch(b)
        for b in self.buckets:
            if b["work"] is not None:
                b["work"].wait()
                if self._avg_op == dist.ReduceOp.SUM:
                    b["flat"].div_(self.world)
            b["work"], b["launched"] = None, False
            b["fired"].clear()

    def attach(self, optimizer: torch.optim.Optimizer) -> "GradAllReducer":
        self._opt_hook = optimizer.register_step_pre_hook(lambda *a, **k: self.finalize())
        return self

    def detach(self) -> None:
        for h in self._hooks:
            h.remove()
        if self._opt_hook is not None:
            self._opt_hook.remove()


class FlatGradArena:
    """Flat fp32 gradient arena for the captured-graph step (graph_step.py): every parameter that takes part in the step gets
    its `.grad` as a VIEW into one flat buffer, so backward accumulates straight into the message RCCL sends (no packing
    kernels) and the whole exchange is `n_buckets` large collectives (xGMI is point-to-point and latency-bound at these
    sizes: few, large messages).

    `used`: the parameters that really receive a gradient in a step (found by the caller from one warm-up backward; the
    set is a property of the model, hence the same on every rank).  Parameters outside it keep `.grad = None`, exactly like
    DDP with `find_unused_parameters=True` (trainer.py:217) leaves them: the optimiser skips them, so neither weight decay
    nor Adam state ever touches DuETT's SSL heads (duett.py:110-122) whatever the world size.

    Buckets are contiguous slices in REVERSE parameter order position (gradients become ready roughly in reverse registration
    order), so `all_reduce(bucket=i, async_op=True)` can be issued as soon as a backward segment has produced bucket i.
    """

    def __init__(self, params: Iterable[torch.nn.Parameter], used: Optional[Iterable[torch.nn.Parameter]] = None, group=None,
                 n_buckets: int = 1):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.all_params = [p for p in params if p.requires_grad]
        if not self.all_params:
            raise ValueError("FlatGradArena: no trainable parameters")
        used_ids = None if used is None else {id(p) for p in used}
        self.params = [p for p in self.all_params if used_ids is None or id(p) in used_ids]
        self.unused = [p for p in self.all_params if used_ids is not None and id(p) not in used_ids]
        if not self.params:
            raise ValueError("FlatGradArena: no parameter receives a gradient")
        dev, dtype = self.params[0].device, self.params[0].dtype
        if any(p.device != dev or p.dtype != dtype for p in self.params):
            raise ValueError("FlatGradArena: parameters must share one device and dtype")
        order = list(reversed(self.params))                      # backward order
        total = sum(p.numel() for p in order)
        self.flat = torch.zeros(total, dtype=dtype, device=dev)
        self.slots, off = {}, 0
        for p in order:
            self.slots[id(p)] = (off, p.numel())
            off += p.numel()
        n_buckets = max(1, min(int(n_buckets), len(order)))
        per = (total + n_buckets - 1) // n_buckets
        self.bucket_bounds, start, acc = [], 0, 0
        for p in order:
            acc += p.numel()
            if acc - start >= per and len(self.bucket_bounds) < n_buckets - 1:
                self.bucket_bounds.append((start, acc))
                start = acc
        self.bucket_bounds.append((start, total))
        self.bytes_per_step = total * self.flat.element_size()
        self._avg = None

    def bind(self, zero: bool = True) -> None:
        """Point every used parameter's `.grad` at its arena slice (unused ones: None)."""
        if zero:
            self.flat.zero_()
        for p in self.params:
            off, n = self.slots[id(p)]
            p.grad = self.flat[off:off + n].view_as(p)
        for p in self.unused:
            p.grad = None

    def bucket_of(self, p: torch.nn.Parameter) -> int:
        off, _ = self.slots[id(p)]
        for i, (a, b) in enumerate(self.bucket_bounds):
            if a <= off < b:
                return i
        raise KeyError("parameter is not in the arena")

    def all_reduce(self, bucket: Optional[int] = None, async_op: bool = False, force: bool = False):
        """Mean over ranks of the whole arena (bucket=None) or of one bucket.  Returns the work handle when async.
        `force`: issue the collective even on a size-1 group (one-GPU rehearsal of the N > 1 step)."""
        if not dist.is_initialized() or (self.world == 1 and not force):
            return None
        if self._avg is None:
            self._avg = dist.get_backend(self.group) == "nccl"
        a, b = (0, self.flat.numel()) if bucket is None else self.bucket_bounds[bucket]
        view = self.flat[a:b]
        if self._avg:
            return dist.all_reduce(view, op=dist.ReduceOp.AVG, group=self.group, async_op=async_op)
        work = dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group, async_op=async_op)   # gloo has no AVG
        if async_op:
            return _ScaledWork(work, view, self.world)
        view.div_(self.world)
        return None


class _ScaledWork:
    def __init__(self, work, view, world):
        self.work, self.view, self.world = work, view, world

    def wait(self):
        self.work.wait()
        self.view.div_(self.world)


def find_used_parameters(params: Iterable[torch.nn.Parameter], run_backward) -> list:
    """One backward with every `.grad` cleared tells which parameters the step really reaches (the static answer to DDP's
    `find_unused_parameters`).  `run_backward()` must run forward + backward once."""
    ps = [p for p in params if p.requires_grad]
    for p in ps:
        p.grad = None
    run_backward()
    return [p for p in ps if p.grad is not None]


@torch.no_grad()
def gather_for_eval(*tensors: torch.Tensor, group=None):
    """All-gather evaluation outputs so AUROC is computed on the full split (the reference evaluates rank 0's shard only,
    evaluator.py:19-37 — reported separately)."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return tensors
    world = dist.get_world_size(group)
    out = []
    for t in tensors:
        n = torch.tensor([t.shape[0]], device=t.device)
        ns = [torch.zeros_like(n) for _ in range(world)]
        dist.all_gather(ns, n, group=group)
        m = int(max(int(x) for x in ns))
        pad = torch.zeros((m,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        pad[:t.shape[0]] = t
        parts = [torch.zeros_like(pad) for _ in range(world)]
        dist.all_gather(parts, pad, group=group)
        out.append(torch.cat([p[:int(k)] for p, k in zip(parts, ns)]))
    return tuple(out)


def broadcast_flag(value: bool, src: int = 0, device="cpu", group=None) -> bool:
    """Early-stop decision broadcast 'so ranks don't hang' (trainer.py:708-711, 970-973)."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return bool(value)
    t = torch.tensor([1 if value else 0], dtype=torch.int64, device=device)
    dist.broadcast(t, src=src, group=group)
    return bool(int(t.item()))
