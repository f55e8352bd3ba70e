"""The N>1 path on CPU: two processes, gloo backend, world_size 2 -- shard the batch, accumulate
metric sums per rank, ONE all-reduce, and the max-over-ranks timing helper used by bench.py."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from diffuvolume_amd import distributed as D
    from diffuvolume_amd import metrics as M
    r, w, _ = D.init_from_env(backend="gloo")
    assert (r, w) == (rank, world) and dist.get_backend() == "gloo"
    # 6 "batches" of metrics, sharded contiguously: rank r accumulates its slice only
    per_batch = [{n: torch.tensor(float(10 * b + i)) for i, n in enumerate(M.NAMES)} for b in range(6)]
    lo, hi = D.shard_range(len(per_batch), r, w)
    acc = M.MetricAccumulator("cpu")
    for b in range(lo, hi):
        acc.update(per_batch[b])
    out = acc.reduce()
    slow = D.barrier_and_max(1.0 + r, torch.device("cpu"))
    q.put((rank, lo, hi, out, slow))
    dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_two_rank_metric_allreduce():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = sorted(q.get(timeout=150) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, lo0, hi0, out0, slow0), (r1, lo1, hi1, out1, slow1) = results
    assert (lo0, hi0, lo1, hi1) == (0, 3, 3, 6)
    assert out0 == out1                                   # every rank holds the global means
    assert out0["EPE"] == sum(10 * b for b in range(6)) / 6 and out0["Thres3"] == 25 + 4
    assert slow0 == slow1 == 2.0                          # MAX over ranks


def test_single_process_is_a_noop(monkeypatch):
    from diffuvolume_amd import distributed as D
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    assert D.init_from_env() == (0, 1, 0)
    assert D.barrier_and_max(3.5, torch.device("cpu")) == 3.5
