"""The N>1 path on CPU: two processes, gloo backend, world_size 2 -- shard the batch, accumulate
metric sums per rank, ONE all-reduce, and the max-over-ranks timing helper used by bench.py."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    from diffuvolume_amd.distributed import free_port
    return free_port()


def _batches():
    """3 global batches of 4 images [6, 8]; image 3 of batch 1 (it lands on rank 1) has a mask ratio below 0.1, which the
    reference skips (SceneFlow/utils/metrics.py:30-31)."""
    g = torch.Generator().manual_seed(5)
    out = []
    for b in range(3):
        gt = torch.rand(4, 6, 8, generator=g) * 100 + 1
        est = gt + torch.randn(4, 6, 8, generator=g) * 3
        mask = torch.rand(4, 6, 8, generator=g) < 0.7
        if b == 1:
            mask[3] = False
            mask[3, 0, :2] = True                          # 2 of 48 pixels: ratio 0.042 < 0.1
        out.append((est, gt, mask))
    return out


def _image_sums_cpu(est, gt, mask):
    """The 8 per-image sums `dv_masked_metrics_f32` produces, in plain torch (fp64)."""
    e = (gt - est).abs().double()
    m = mask.double()
    d1 = ((e > 3) & (e / gt.abs().double() > 0.05)).double()
    cols = [m, (gt > 0).double(), e * m, d1 * m, (e > 1).double() * m, (e > 2).double() * m, (e > 3).double() * m, 0 * m]
    return torch.stack([c.flatten(1).sum(1) for c in cols], dim=1)


def _reference_run(batches):
    """SceneFlow/utils/metrics.py:22-65 + utils/experiment.py:126-151 restated: per image (skipping mask ratio < 0.1),
    mean over the kept images of the batch, mean over batches."""
    names = ("EPE", "D1", "Thres1", "Thres2", "Thres3")
    tot = {n: 0.0 for n in names}
    for est, gt, mask in batches:
        per = {n: [] for n in names}
        for i in range(gt.shape[0]):
            if mask[i].float().mean() / (gt[i] > 0).float().mean() < 0.1:
                continue
            e = (gt[i][mask[i]] - est[i][mask[i]]).abs().double()
            g = gt[i][mask[i]].abs().double()
            per["EPE"].append(e.mean())
            per["D1"].append(((e > 3) & (e / g > 0.05)).double().mean())
            for t in (1, 2, 3):
                per[f"Thres{t}"].append((e > t).double().mean())
        for n in names:
            tot[n] += float(torch.stack(per[n]).mean()) if per[n] else 0.0
    return {n: v / len(batches) for n, v in tot.items()}


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from diffuvolume_amd import distributed as D
    from diffuvolume_amd import metrics as M
    r, w, _ = D.init_from_env(backend="gloo")
    assert (r, w) == (rank, world) and dist.get_backend() == "gloo"
    # every global batch is sharded contiguously over the ranks (what nn.DataParallel's scatter does in the reference)
    acc = M.MetricAccumulator("cpu")
    spans = []
    for est, gt, mask in _batches():
        lo, hi = D.shard_range(gt.shape[0], r, w)
        spans.append((lo, hi))
        acc.update_sums(_image_sums_cpu(est[lo:hi], gt[lo:hi], mask[lo:hi]))
    assert acc.table().shape == (3, 6)
    out = acc.reduce()                                    # the one collective: a [3, 6] fp64 SUM
    slow = D.barrier_and_max(1.0 + r, torch.device("cpu"))
    q.put((rank, spans[0], float(acc.table()[1, 5]), out, slow))
    dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_two_rank_metric_allreduce():
    """World size 2 over gloo; rank 1 skips one image of batch 1.  The reduced numbers equal the single-process run of the
    reference's bookkeeping on the unsharded batches -- a mean of per-rank batch means would not (the skipped image
    changes that batch's divisor from 4 to 3)."""
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
    (r0, span0, kept0, out0, slow0), (r1, span1, kept1, out1, slow1) = results
    assert (span0, span1) == ((0, 2), (2, 4))
    assert (kept0, kept1) == (2.0, 1.0)                   # rank 1 kept one of its two images of batch 1
    assert out0 == out1                                   # every rank holds the global numbers
    want = _reference_run(_batches())
    for n, v in want.items():
        assert abs(out0[n] - v) < 1e-12, (n, out0[n], v)
    # the old reduce (mean of per-rank batch means) differs on this data: the test would notice a regression to it
    naive = 0.0
    for est, gt, mask in _batches():
        naive += 0.5 * sum(_reference_run([(est[lo:hi], gt[lo:hi], mask[lo:hi])])["EPE"] for lo, hi in ((0, 2), (2, 4))) / 3
    assert abs(naive - want["EPE"]) > 1e-6
    assert slow0 == slow1 == 2.0                          # MAX over ranks


def test_accumulator_single_process_equals_the_reference_bookkeeping():
    from diffuvolume_amd import metrics as M
    acc = M.MetricAccumulator("cpu")
    for est, gt, mask in _batches():
        acc.update_sums(_image_sums_cpu(est, gt, mask))
    out = acc.reduce()
    for n, v in _reference_run(_batches()).items():
        assert abs(out[n] - v) < 1e-12
    # a batch whose images are ALL skipped counts as a batch with value 0 (metrics.py:36-38)
    est, gt, mask = _batches()[0]
    acc.update_sums(_image_sums_cpu(est, gt, torch.zeros_like(mask)))
    assert abs(acc.reduce()["EPE"] - 3 * out["EPE"] / 4) < 1e-12


def test_single_process_is_a_noop(monkeypatch):
    from diffuvolume_amd import distributed as D
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    assert D.init_from_env() == (0, 1, 0)
    assert D.barrier_and_max(3.5, torch.device("cpu")) == 3.5
