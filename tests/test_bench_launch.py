"""bench.py --gpus N must never report a GPU count it did not run on: without a launcher it starts
the N ranks itself, with one it checks WORLD_SIZE, and with too few devices it exits non-zero.  CPU-only plumbing
(gloo) through --dry-run; the real N-GPU run is the driver's."""
import json
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]


def _run(args, env=None, timeout=240):
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    e.update(env or {})
    return subprocess.run([sys.executable, str(ROOT / "bench.py")] + args, capture_output=True, text=True, env=e,
                          timeout=timeout, cwd=ROOT)


def _json_line(out):
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out
    return json.loads(lines[0])


def test_self_launch_two_ranks():
    r = _run(["--gpus", "2", "--dry-run"])
    assert r.returncode == 0, r.stderr
    line = _json_line(r.stdout)
    assert line["n_gpus"] == 2 and line["ranks"] == 2 and line["launcher"] == "self" and line["global_batch"] == 16


def test_self_launch_eight_ranks():
    """The N = 8 launch the driver's scaling run uses (BASELINE configs 3 and 5), over gloo on the CPU: eight ranks meet,
    the collective sees all eight, the global batch is 8 pairs per rank.  The subprocess timeout is the bound."""
    r = _run(["--gpus", "8", "--dry-run"], timeout=300)
    assert r.returncode == 0, r.stderr
    line = _json_line(r.stdout)
    assert line["n_gpus"] == 8 and line["ranks"] == 8 and line["launcher"] == "self" and line["global_batch"] == 64


def test_a_bare_world_size_does_not_start_a_process_group():
    """WORLD_SIZE=1 exported by a scheduler (no RANK / LOCAL_RANK / MASTER_PORT) is not a launcher."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod2", ROOT / "bench.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    keys = ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "TORCHELASTIC_RUN_ID", "DV_BENCH_SELF_LAUNCHED")
    old = {k: os.environ.get(k) for k in keys}
    try:
        for k in keys:
            os.environ.pop(k, None)
        os.environ["WORLD_SIZE"] = "1"
        assert not mod.under_launcher() and mod.launcher_label() == "single"
        os.environ.update(RANK="0", LOCAL_RANK="0", MASTER_PORT="29999")
        assert mod.under_launcher() and mod.launcher_label() == "env"
        os.environ["TORCHELASTIC_RUN_ID"] = "x"
        assert mod.launcher_label() == "torchrun"
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def test_under_torchrun_two_ranks():
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", "29653", str(ROOT / "bench.py"), "--gpus", "2",
                        "--dry-run"], capture_output=True, text=True, timeout=240, cwd=ROOT)
    assert r.returncode == 0, r.stderr
    line = _json_line(r.stdout)
    assert line["n_gpus"] == 2 and line["ranks"] == 2 and line["launcher"] == "torchrun"


def test_world_size_mismatch_is_an_error():
    r = _run(["--gpus", "4", "--dry-run"], env={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0", "MASTER_PORT": "29654"})
    assert r.returncode != 0 and "WORLD_SIZE" in (r.stderr + r.stdout)


def test_a_bare_world_size_neither_blocks_the_self_launch_nor_hangs():
    """with a scheduler-exported WORLD_SIZE (no RANK / MASTER_PORT) `--gpus N` still starts its own ranks --
    WORLD_SIZE=1 used to die with a mismatch, WORLD_SIZE=N used to join N processes as rank 0 and hang."""
    for ws in ("1", "2"):
        r = _run(["--gpus", "2", "--dry-run"], env={"WORLD_SIZE": ws})
        assert r.returncode == 0, r.stderr
        line = _json_line(r.stdout)
        assert line["n_gpus"] == 2 and line["ranks"] == 2 and line["launcher"] == "self"
    from diffuvolume_amd import distributed as D
    import pytest
    old = {k: os.environ.pop(k, None) for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    try:
        os.environ["WORLD_SIZE"] = "8"
        with pytest.raises(RuntimeError, match="not a launcher environment"):
            D.init_from_env(backend="gloo")
    finally:
        os.environ.pop("WORLD_SIZE", None)
        for k, v in old.items():
            if v is not None:
                os.environ[k] = v


def test_every_workload_launches_eight_ranks():
    """BASELINE configs 3 / 4 / 5 through the same launcher and the same contract line: `--workload` names the config, the
    per-GPU batch follows it (8 / 4 / 4 pairs)."""
    for wl, gb in (("kitti12", 32), ("kitti15", 32)):
        r = _run(["--gpus", "8", "--dry-run", "--workload", wl], timeout=300)
        assert r.returncode == 0, r.stderr
        line = _json_line(r.stdout)
        assert line["workload"] == wl and line["n_gpus"] == 8 and line["ranks"] == 8 and line["global_batch"] == gb


def test_too_few_devices_is_an_error():
    import torch
    if torch.cuda.device_count() >= 2:
        return                                  # a multi-GPU box would really run it
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0"])
    assert r.returncode != 0 and "refusing" in r.stderr
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]      # and no JSON line with a wrong n_gpus


def test_self_launch_refused_under_a_profiler():
    """rocprofv3's preloaded tool library initialises the GPU in the parent: starting workers from there is the
    exec-after-init hop the pool forbids."""
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0"], env={"ROCPROF_OUTPUT_PATH": "/tmp/x",      # what rocprofv3 exports to its child
                                                                   "DV_BENCH_OVERSUBSCRIBE": "1"})
    assert r.returncode != 0 and "profiled parent" in r.stderr


def test_device_count_without_the_runtime():
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", ROOT / "bench.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    old = {k: os.environ.get(k) for k in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES")}
    try:
        for k in old:
            os.environ.pop(k, None)
        n = mod.visible_gpu_count()
        assert n >= 0
        os.environ["HIP_VISIBLE_DEVICES"] = ""
        assert mod.visible_gpu_count() == 0
        os.environ["HIP_VISIBLE_DEVICES"] = "0"
        assert mod.visible_gpu_count() == min(n, 1)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
