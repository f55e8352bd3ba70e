"""Worker of tests/test_gpu_nccl_single.py: a ONE-rank RCCL group on the box's GPU, driven through the same helpers the
N-rank bench uses (diffuvolume_amd.distributed.init_from_env -> device binding -> device all-reduce ->
MetricAccumulator.reduce -> barrier_and_max).  Prints one JSON line."""
import json
import os
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from diffuvolume_amd import distributed as D  # noqa: E402
from diffuvolume_amd import metrics as M  # noqa: E402


def main():
    rank, world, local = D.init_from_env(backend="nccl", force=True)
    dev = torch.device("cuda", D.device_index(local))
    t = torch.arange(6, dtype=torch.float64, device=dev)
    dist.all_reduce(t)                                   # a device buffer through RCCL
    acc = M.MetricAccumulator(dev)
    # one image, 10 masked pixels: per-image values 1, 2, 3, 4, 5 (sums n_mask, n_gt>0, sum|err|, n_D1, n_>1, n_>2, n_>3, 0)
    acc.update_sums(torch.tensor([[10.0, 10.0, 10.0, 20.0, 30.0, 40.0, 50.0, 0.0]], dtype=torch.float64, device=dev))
    red = acc.reduce()
    mx = D.barrier_and_max(0.125, dev)
    dist.barrier()
    out = {"rank": rank, "world": world, "backend": dist.get_backend(), "group_size": dist.get_world_size(),
           "allreduce": t.cpu().tolist(), "metrics": red, "max_seconds": mx,
           "device": torch.cuda.current_device(), "ipc_legacy": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY")}
    dist.destroy_process_group()
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
