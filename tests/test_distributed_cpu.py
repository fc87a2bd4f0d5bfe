"""CPU suite, part 3: the N > 1 path of bench.py on gloo, world_size 2 — sharding is a
partition, the step time is the max over ranks, nothing but scalars crosses ranks."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench
    from rna_algos_amd import workloads as W
    lens = W.batch_lengths(400)
    costs = W.sweep_cost(lens)
    mine = bench.shard_banded(costs, world)[rank]
    # what the ranks exchange: a barrier and the max of the local step time
    dist.barrier()
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    cnt = torch.tensor([len(mine), int(lens[mine].sum())], dtype=torch.int64)
    dist.all_reduce(cnt, op=dist.ReduceOp.SUM)
    gathered = [None] * world
    dist.all_gather_object(gathered, mine.tolist())
    q.put((rank, float(t.item()), cnt.tolist(), gathered, float(costs[mine].sum())))
    dist.destroy_process_group()


def test_two_rank_gloo_sharding():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    sys.path.insert(0, ROOT)
    from rna_algos_amd import workloads as W
    lens = W.batch_lengths(400)
    for rank, tmax, cnt, gathered, load in res:
        assert tmax == 2.0                      # max over ranks
        assert cnt == [400, int(lens.sum())]    # shards partition the batch
        assert sorted(gathered[0] + gathered[1]) == list(range(400))
    loads = sorted(r[4] for r in res)
    assert loads[1] / loads[0] < 1.01


def test_bench_launcher_two_ranks_gloo_dry_run():
    """`python bench.py --gpus 2` outside torchrun starts the launcher itself (a child process)
    and relays one JSON line with n_gpus = 2; a world that differs from --gpus is refused."""
    import json
    import subprocess
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2",
                          "--backend", "gloo", "--dry-run", "--batch-count", "300"],
                         env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["dry_run"] is True
    assert rec["config"]["sequences_all_ranks"] == 300
    # what a multi-GPU line is judged on (round-3 verdict, item 6): every rank's pass time, its
    # share of the batch, the cost model's prediction, its golden-check count, the imbalance
    assert len(rec["per_rank_s"]) == 2 and rec["per_rank_s"] == [0.01, 0.02]
    assert sum(rec["per_rank_sequences"]) == 300 and min(rec["per_rank_sequences"]) > 0
    sys.path.insert(0, ROOT)
    from rna_algos_amd import workloads as W
    assert sum(rec["per_rank_nt"]) == int(W.batch_lengths(300).sum())
    assert rec["per_rank_golden_checked"] == [0, 0]
    assert abs(rec["imbalance"] - 0.02 / 0.015) < 1e-9
    assert 1.0 <= rec["model_imbalance"] < 1.02  # equal-cost bands under the model
    # --gpus 2 inside a one-rank world: refused, non-zero
    env2 = dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    bad = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2",
                          "--backend", "gloo", "--dry-run", "--batch-count", "300"],
                         env=env2, capture_output=True, text=True, timeout=300)
    assert bad.returncode == 2 and "refusing" in bad.stderr
