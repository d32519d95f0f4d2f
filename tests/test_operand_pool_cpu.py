"""Host logic of autograd_ops.WeightOperandPool (the job table behind `medp_weight_operands_multi`): which weights get a plain / a transposed
bf16 operand, where q | k | v land inside their stacked operand, how many 64 x 64 tiles each job is cut into.  No kernel runs here."""
import ctypes

import torch

from multimodal_edema_prediction_amd import autograd_ops as A
from multimodal_edema_prediction_amd.abi import MedpOperandJob


def _jobs(pool):
    raw = bytes(pool._jobs.cpu().numpy().tobytes())
    n = len(raw) // ctypes.sizeof(MedpOperandJob)
    return list((MedpOperandJob * n).from_buffer_copy(raw))


def test_job_table_of_a_recorded_step():
    torch.manual_seed(0)
    P = lambda n, k: torch.nn.Parameter(torch.randn(n, k))
    wq, wk, wv, wo, w1 = P(24, 200), P(24, 200), P(24, 200), P(200, 24), P(130, 200)
    frozen = torch.nn.Parameter(torch.randn(8, 8), requires_grad=False)
    log = [((w1,), "bf16"), ((wq, wk, wv), "bf16"), ((wo,), "bf16"), ((wo,), "t_bf16"), ((wq, wk, wv), "t_bf16"), ((w1,), "bf16")]
    with A.record_operands() as rec:                       # the recorder itself: only trainable 2-D leaves are logged
        A._log_operand((frozen,), "bf16")
        A._log_operand((w1[:4],), "bf16")
        A._log_operand((w1,), "t_bf16")
    assert len(rec) == 1 and rec[0][0][0] is w1 and rec[0][1] == "t_bf16"
    pool = A.WeightOperandPool(log, torch.device("cpu"))
    groups = {tuple(id(w) for w in ws): (plain, tr) for ws, plain, tr in pool._groups}
    plain1, tr1 = groups[(id(w1),)]
    assert plain1.shape == (130, 200) and plain1.dtype == torch.bfloat16 and tr1 is None          # forward operand only
    pq, tq = groups[(id(wq), id(wk), id(wv))]
    assert pq.shape == (72, 200) and tq.shape == (200, 72) and float(tq.float().abs().sum()) == 0.0
    po, to = groups[(id(wo),)]
    assert po.shape == (200, 24) and to.shape == (24, 200)
    jobs = _jobs(pool)
    assert pool.n_jobs == len(jobs) == 5                                                         # w1, q, k, v, wo
    by_src = {j.src: j for j in jobs}
    for i, w in enumerate((wq, wk, wv)):                                                         # row blocks of ONE stacked operand
        j = by_src[w.data_ptr()]
        assert (j.rows, j.cols, j.ld_src, j.ld_plain, j.ld_t) == (24, 200, 200, 200, 72)
        assert j.dst_plain == pq.data_ptr() + i * 24 * 200 * 2 and j.dst_t == tq.data_ptr() + i * 24 * 2
    j1 = by_src[w1.data_ptr()]
    assert j1.dst_t is None and j1.dst_plain == plain1.data_ptr()
    tiles = lambda n, k: ((n + 63) // 64) * ((k + 63) // 64)
    blk_job, blk_tile = pool._blk_job.tolist(), pool._blk_tile.tolist()
    assert pool.n_blocks == len(blk_job) == tiles(130, 200) + 3 * tiles(24, 200) + tiles(200, 24)
    for ji, j in enumerate(jobs):                                                                # every job: its tiles 0 .. n-1 exactly once
        assert sorted(t for b, t in zip(blk_job, blk_tile) if b == ji) == list(range(tiles(j.rows, j.cols)))


def test_group_cache_is_keyed_on_identity_and_versions():
    a, b = torch.nn.Parameter(torch.randn(4, 8)), torch.nn.Parameter(torch.randn(4, 8))
    made = []
    make = lambda ts: made.append(1) or torch.cat(ts, 0)
    v1 = A._cached_group((a, b), "k", make)
    assert A._cached_group((a, b), "k", make) is v1 and len(made) == 1
    with torch.no_grad():
        b.add_(1.0)                                                                              # a member changed: rebuilt
    v2 = A._cached_group((a, b), "k", make)
    assert v2 is not v1 and len(made) == 2 and torch.equal(v2[4:], b.detach())
    c = torch.nn.Parameter(torch.randn(4, 8))
    assert A._cached_group((a, c), "k", make) is not v2 and len(made) == 3                       # another partner under the same first weight
