"""Multi-process (world_size 2, gloo, CPU) tests of the data-parallel layer: the N>1 path of bench.py.
The reducer is model-agnostic plumbing (it moves gradients, computes nothing), so a small torch model stands in."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from multimodal_edema_prediction_amd import dp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


class _Net(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.net = torch.nn.Sequential(torch.nn.Linear(16, 32), torch.nn.GELU(), torch.nn.Linear(32, 8), torch.nn.GELU(),
                                       torch.nn.Linear(8, 1))
        self.unused = torch.nn.Linear(4, 4)    # never used in forward: must keep grad None on every rank

    def forward(self, x):
        return self.net(x)


def _model():
    torch.manual_seed(0)
    return _Net()


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    r, _, w = dp.init_distributed("gloo")
    assert (r, w) == (rank, world)
    m = _model()
    if rank == 1:                               # perturb rank 1, then broadcast must restore rank 0's weights
        with torch.no_grad():
            for p in m.parameters():
                p.add_(1.0)
    dp.broadcast_parameters(m, src=0)
    opt = torch.optim.SGD(m.parameters(), lr=0.1)
    red = dp.GradAllReducer(m.parameters(), n_buckets=2).attach(opt)
    g = torch.Generator().manual_seed(1)
    x, y = torch.randn(8, 16, generator=g), torch.randn(8, 1, generator=g)
    idx = list(dp.shard_indices(8, rank, world))
    for step in range(2):
        opt.zero_grad()
        loss = torch.nn.functional.mse_loss(m(x[idx]), y[idx])
        loss.backward()
        opt.step()
    assert m.unused.weight.grad is None
    flag = dp.broadcast_flag(rank == 0, src=0)
    logits, = dp.gather_for_eval(torch.full((2 + rank, 3), float(rank)))
    q.put((rank, [p.detach().numpy().copy() for p in m.parameters()], flag, logits.numpy().copy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_two_rank_gradient_allreduce_matches_full_batch():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = {}
    for _ in range(2):
        rank, params, flag, gathered = q.get(timeout=60)
        results[rank] = (params, flag, gathered)
    for p in procs:
        p.join(timeout=30)
        assert p.exitcode == 0
    # single-process reference: mean over the two per-rank shard losses == DP's averaged gradient
    m = _model()
    opt = torch.optim.SGD(m.parameters(), lr=0.1)
    g = torch.Generator().manual_seed(1)
    x, y = torch.randn(8, 16, generator=g), torch.randn(8, 1, generator=g)
    for step in range(2):
        opt.zero_grad()
        loss = 0.5 * (torch.nn.functional.mse_loss(m(x[0::2]), y[0::2]) + torch.nn.functional.mse_loss(m(x[1::2]), y[1::2]))
        loss.backward()
        opt.step()
    for rank in (0, 1):
        for a, b in zip(results[rank][0], m.parameters()):
            assert torch.allclose(torch.from_numpy(a), b.detach(), atol=1e-6), rank
        assert results[rank][1] is True
        gat = torch.from_numpy(results[rank][2])
        assert gat.shape == (5, 3) and gat[:2].eq(0).all() and gat[2:].eq(1).all()


def _arena_worker(rank, world, port, q):
    """The graph step's N > 1 exchange (dp.FlatGradArena, used by graph_step._GraphedStep): used-parameter discovery, grads as
    arena views, bucketed async mean all-reduce, AdamW with weight decay that must NOT touch the never-used parameters."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    dp.init_distributed("gloo")
    m = _model()
    opt = torch.optim.AdamW(m.parameters(), lr=0.05, weight_decay=0.1)
    params = [p for g in opt.param_groups for p in g["params"]]
    g = torch.Generator().manual_seed(1)
    x, y = torch.randn(8, 16, generator=g), torch.randn(8, 1, generator=g)
    idx = list(dp.shard_indices(8, rank, world))
    fb = lambda: torch.nn.functional.mse_loss(m(x[idx]), y[idx]).backward()
    used = dp.find_used_parameters(params, fb)
    arena = dp.FlatGradArena(params, used=used, n_buckets=2)
    assert len(arena.unused) == 2 and len(arena.bucket_bounds) == 2 and arena.bytes_per_step == sum(p.numel() for p in used) * 4
    unused_before = [p.detach().clone() for p in m.unused.parameters()]
    for step in range(3):
        arena.bind(zero=True)                      # what the captured forward/backward graph starts with
        fb()
        assert m.net[0].weight.grad.data_ptr() == arena.flat[arena.slots[id(m.net[0].weight)][0]:].data_ptr()   # still a view
        works = [arena.all_reduce(bucket=b, async_op=True) for b in range(len(arena.bucket_bounds))]
        for w in works:
            w.wait()
        opt.step()
    assert all(p.grad is None for p in m.unused.parameters())
    assert all(torch.equal(a, b) for a, b in zip(unused_before, m.unused.parameters()))          # no weight decay crept in
    assert all(len(opt.state.get(p, {})) == 0 for p in m.unused.parameters())                    # no Adam state either
    q.put((rank, [p.detach().numpy().copy() for p in m.parameters()]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_two_rank_flat_arena_step_matches_full_batch_and_skips_unused_parameters():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_arena_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=60) for _ in range(2))
    for p in procs:
        p.join(timeout=30)
        assert p.exitcode == 0
    m = _model()
    opt = torch.optim.AdamW(m.parameters(), lr=0.05, weight_decay=0.1)
    g = torch.Generator().manual_seed(1)
    x, y = torch.randn(8, 16, generator=g), torch.randn(8, 1, generator=g)
    for step in range(3):
        opt.zero_grad()
        (0.5 * (torch.nn.functional.mse_loss(m(x[0::2]), y[0::2]) + torch.nn.functional.mse_loss(m(x[1::2]), y[1::2]))).backward()
        opt.step()
    for rank in (0, 1):
        for a, b in zip(results[rank], m.parameters()):
            assert torch.allclose(torch.from_numpy(a), b.detach(), atol=2e-6), rank


def test_single_process_is_a_no_op_reducer():
    m = _model()
    opt = torch.optim.SGD(m.parameters(), lr=0.1)
    red = dp.GradAllReducer(m.parameters()).attach(opt)
    x = torch.randn(4, 16)
    for _ in range(2):
        opt.zero_grad()
        m(x).sum().backward()
        g_before = m.net[0].weight.grad.clone()
        opt.step()
        assert torch.equal(m.net[0].weight.grad, g_before)
    assert red.bytes_per_step == sum(p.numel() for p in m.parameters()) * 4
