"""`python bench.py --gpus N` as the driver invokes it (no RANK in the environment) must start one process per GPU under
torch.distributed.run itself and relay rank 0's JSON line.  `--launch-check` does everything but the GPU work
(gloo group, barrier, MAX-over-ranks exchange), so the launch / rendezvous / relay path is exercised here on CPU."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.timeout(180)
@pytest.mark.parametrize("n,extra", [(2, []), (2, ["--strong"]), (3, ["--strong", "--batch", "63"]), (1, [])])
def test_bench_self_launches_one_rank_per_gpu(n, extra):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "2", "--warmup", "1", "--launch-check"] + extra,
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=170)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout                      # exactly ONE JSON line on stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == n and out["value"] == float(n)          # MAX over ranks of (rank + 1): every rank took part
    assert out["scaling"] == ("strong" if extra else "weak")
    # SURVEY.md §8(e) shard arithmetic, as the real run computes it: weak = --batch per GPU, --strong = --batch / N per GPU; rank r takes
    # items r, r + N, ... of every global batch, and the ranks together cover each global batch exactly once
    batch = int(extra[extra.index("--batch") + 1]) if "--batch" in extra else 64
    assert out["per_gpu_batch"] == (batch // n if "--strong" in extra else batch) and out["global_batch"] == out["per_gpu_batch"] * n
    assert out["shards_cover_global_batches_exactly_once"] is True and out["rank0_first_items"] == [0, n, 2 * n]
