"""One process per GPU; stereo pairs are independent, so the batch is split into contiguous
shards with no data-path collective.  The reference uses nn.DataParallel, which
re-broadcasts the weights and gathers outputs on every forward
(SceneFlow/test_sceneflow_ddim.py:54-61); here weights are replicated once and the only
collective is the metric all-reduce in ``metrics.MetricAccumulator.reduce``."""
from __future__ import annotations

import os
from typing import Tuple

import torch
import torch.distributed as dist


def init_from_env(backend: str | None = None, force: bool = False) -> Tuple[int, int, int]:
    """(rank, world_size, local_rank) from torchrun's environment; initialises the default
    process group when WORLD_SIZE > 1 (backend 'nccl' == RCCL on ROCm, 'gloo' on CPU), or -- ``force`` -- also for
    a single rank (a one-rank RCCL group: the same code path as N ranks, on one GPU)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if (world > 1 or force) and not dist.is_initialized():
        if world > 1 and not ("RANK" in os.environ and "MASTER_PORT" in os.environ):
            # a bare WORLD_SIZE (a scheduler's export): every process would join as rank 0 and wait for the others forever
            raise RuntimeError(f"WORLD_SIZE={world} without RANK / MASTER_PORT: not a launcher environment (use torchrun, or "
                               "`python bench.py --gpus N`, which starts its own ranks)")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend is None:
            backend = os.environ.get("DV_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if torch.cuda.is_available():
            torch.cuda.set_device(device_index(local))
        kwargs = {}
        if backend == "nccl":       # bind the communicator to this rank's GPU (no "device unknown" barrier fallback)
            kwargs["device_id"] = torch.device("cuda", device_index(local))
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kwargs)
    return rank, world, local


def device_index(local_rank: int) -> int:
    """GPU of a local rank.  One process per GPU in production; when a box has fewer GPUs than ranks
    (plumbing tests of the N>1 path on a 1-GPU box, gloo backend) ranks wrap around."""
    n = torch.cuda.device_count()
    return local_rank % n if n > 0 else 0


def shard_range(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous [lo, hi) slice of ``n_items`` for ``rank``: sizes differ by at most one and
    the first ``n_items % world`` ranks take the extra item."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError("bad rank / world size")
    base, extra = divmod(n_items, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def barrier_and_max(seconds: float, device) -> float:
    """Barrier, then the MAX over ranks of a local duration (bench.py contract)."""
    if dist.is_available() and dist.is_initialized():
        t = torch.tensor([seconds], dtype=torch.float64, device="cpu" if dist.get_backend() == "gloo" else device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())
    return seconds


def free_port(lo: int = 20000, hi: int = 30000, tries: int = 64) -> int:
    """A TCP port on 127.0.0.1 for a rendezvous, taken from BELOW the kernel's ephemeral range (32768-60999 on Linux): a port
    found by bind(0) is released before the rank-0 store binds it again, and any outgoing connection made in between -- RCCL /
    gloo sockets of another test, for instance -- may be handed the same number (`EADDRINUSE`, seen once in a full GPU suite).
    Ports in [lo, hi) are only ever taken by an explicit bind."""
    import random
    import socket
    rng = random.Random()
    for _ in range(tries):
        port = rng.randrange(lo, hi)
        with socket.socket() as sk:
            try:
                sk.bind(("127.0.0.1", port))
            except OSError:
                continue
            return port
    raise RuntimeError(f"no free TCP port found in [{lo}, {hi})")
