"""The N>1 path with the real kernels: two processes (gloo rendezvous, both on the box's one GPU) run the HIP hot
path on shards [0,4) and [4,8) of a batch of 8 and all-reduce their metric sums; the reduced EPE must equal, bit for
bit, what one process gets from the same two shards -- and the per-shard disparities must carry the same bits as the
pairs have inside the full batch (shard invariance, which is what makes data-parallel sharding exact)."""
import json
import os
import socket
import subprocess
import sys
from pathlib import Path

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parents[1]


@pytest.mark.timeout(900)
def test_two_process_shards_match_single_process():
    sys.path.insert(0, str(ROOT / "tests"))
    import shard_worker as W
    from diffuvolume_amd import metrics as M
    from diffuvolume_amd.distributed import free_port
    port = free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, str(ROOT / "tests" / "shard_worker.py")], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        o, e = p.communicate(timeout=800)
        assert p.returncode == 0, e[-2000:]
        outs.append(json.loads([l for l in o.splitlines() if l.startswith("{")][-1]))
    outs.sort(key=lambda d: d["rank"])
    assert [(d["lo"], d["hi"]) for d in outs] == [(0, 4), (4, 8)]
    assert outs[0]["metrics"] == outs[1]["metrics"]                 # every rank holds the reduced means

    dev = torch.device("cuda:0")
    model, x, gt = W.build(dev)
    scratch = M.MetricAccumulator(dev)
    sums, rows = [], []
    for lo, hi in ((0, 4), (4, 8)):
        final, gsum, s8 = W.shard_epe(model, x, gt, lo, hi, dev, scratch)
        sums.append((final.double().sum().item().hex(), gsum))
        rows.append(s8)
    # the one global batch of 8 in ONE process: the two ranks' rows are added by the all-reduce before the division, so the
    # reduced numbers are those of the unsharded batch (reference semantics: mean over the kept images of the GLOBAL batch)
    acc = M.MetricAccumulator(dev)
    acc.update_sums(torch.cat(rows))
    single = acc.reduce()
    for n in M.NAMES:
        assert abs(single[n] - outs[0]["metrics"][n]) <= 1e-13 * max(1.0, abs(single[n])), (n, single, outs[0]["metrics"])
    assert [s[0] for s in sums] == [d["final_hex"] for d in outs]               # the disparities themselves: bit for bit
    assert [s[1] for s in sums] == [d["gwc_sum"] for d in outs]
    # and the shards carry the bits the same pairs have inside the full batch of 8
    acc8 = M.MetricAccumulator(dev)
    full, _, _ = W.shard_epe(model, x, gt, 0, 8, dev, acc8)
    assert full[:4].double().sum().item().hex() == sums[0][0] and full[4:].double().sum().item().hex() == sums[1][0]
    assert acc8.reduce() == single                        # same per-image sums, same fp64 additions


@pytest.mark.timeout(900)
def test_bench_two_ranks_end_to_end_on_one_gpu():
    """`python bench.py --gpus 2` through its own launcher on the one GPU of the test box (DV_BENCH_OVERSUBSCRIBE=1: the
    ranks share cuda:0 and rendezvous over gloo -- RCCL needs one device per rank): the whole N > 1 code path of the
    bench -- worker launch, per-rank inputs, barriers, max-over-ranks timing, the metric all-reduce, one JSON line from
    rank 0 -- runs on real kernels.  The 8-GPU RCCL run itself is the driver's."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env["DV_BENCH_OVERSUBSCRIBE"] = "1"
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--batch", "2",
                        "--height", "128", "--width", "256", "--no-cpu-baseline", "--no-extras"],
                       capture_output=True, text=True, timeout=800, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["rccl_ranks"] == 2 and d["dist_backend"] == "gloo" and d["launcher"] == "self"
    assert d["config"]["global_batch"] == 4 and d["scaling"] == "weak" and d["value"] > 0
    assert "roofline" in d and "cpu_baseline" not in d            # CPU baseline is an N = 1 leg


@pytest.mark.timeout(1200)
@pytest.mark.parametrize("workload,extra", [("kitti12", ["--height", "128", "--width", "256"]),
                                            ("kitti15", ["--height", "128", "--width", "256", "--ddim-steps", "2", "--gru-iters", "3"])])
def test_bench_two_ranks_other_workloads(workload, extra):
    """BASELINE configs 4 / 5 through the same N-rank path (`--workload`): two ranks on the one GPU of the test box over gloo,
    small frames; the contract line names the workload and the all-reduced metric is finite."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env["DV_BENCH_OVERSUBSCRIBE"] = "1"
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--workload", workload, "--steps", "1", "--warmup", "1",
                        "--batch", "1"] + extra, capture_output=True, text=True, timeout=1000, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["dist_backend"] == "gloo" and d["launcher"] == "self" and d["scaling"] == "weak"
    assert d["config"]["global_batch"] == 2 and d["value"] > 0 and d["unit"] == "pairs/s"
    assert ("KITTI12" if workload == "kitti12" else "KITTI15") in d["config"]["workload"] and workload[:5].upper() in d["metric"].upper()
    assert d["epe_px"] == d["epe_px"] and "roofline" in d and d["cpu_baseline"] is None
