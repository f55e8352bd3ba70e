"""De-risking the first multi-GPU run on a one-GPU box: the RCCL backend itself -- library load,
`device_id` binding of the communicator, a device all-reduce, `MetricAccumulator.reduce`, `barrier_and_max` -- runs here
as a ONE-rank "nccl" group through `distributed.init_from_env`, and `bench.py --gpus 1` with torchrun's environment set
goes through the launcher branch (process group + RCCL collectives) instead of the single-process shortcut.  Every
N>1 test elsewhere uses gloo; this is the only place the RCCL code path executes before the driver's 8-GPU run."""
import json
import os
import socket
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parents[1]


def _env():
    from diffuvolume_amd.distributed import free_port
    port = free_port()
    e = {k: v for k, v in os.environ.items() if k not in ("DV_DIST_BACKEND", "DV_BENCH_SELF_LAUNCHED")}
    e.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
             HSA_ENABLE_IPC_MODE_LEGACY="0")
    return e


def _json_line(out):
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert lines, out[-2000:]
    return json.loads(lines[-1])


def test_one_rank_rccl_group_through_the_distributed_helpers():
    r = subprocess.run([sys.executable, str(ROOT / "tests" / "nccl_single_worker.py")], env=_env(), capture_output=True,
                       text=True, timeout=500)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _json_line(r.stdout)
    assert d["backend"] == "nccl" and d["group_size"] == 1 and d["world"] == 1
    assert d["allreduce"] == [0.0, 1.0, 2.0, 3.0, 4.0, 5.0]
    assert d["metrics"]["EPE"] == 1.0 and d["metrics"]["Thres3"] == 5.0
    assert d["max_seconds"] == 0.125


def test_bench_under_a_launcher_environment_uses_rccl_at_one_rank():
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "1", "--batch", "2",
                        "--no-cpu-baseline", "--no-extras"], env=_env(), capture_output=True, text=True, timeout=800, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _json_line(r.stdout)
    assert d["n_gpus"] == 1 and d["rccl_ranks"] == 1 and d["dist_backend"] == "nccl" and d["launcher"] == "env"      # a launcher ENVIRONMENT (RANK, LOCAL_RANK, WORLD_SIZE, MASTER_PORT), no torchrun process
    assert d["value"] > 0 and d["config"]["global_batch"] == 2
