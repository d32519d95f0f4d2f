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
