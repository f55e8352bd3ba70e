"""EPE / D1 / Thres metrics of SceneFlow/utils/metrics.py:22-65 and their reduction across
the GPUs of a node.

The reference evaluates each metric image by image with boolean-index gathers and one
``.item()`` sync per metric.  Here one HIP pass yields seven per-image sums
(``dv_masked_metrics_f32``); the reference's semantics (per-image mean, images whose mask
ratio is < 0.1 skipped, batch value = mean over the kept images of the GLOBAL batch, run value =
mean over batches: metrics.py:30-40 + experiment.py:146-151) are applied to those sums.  Across
ranks only [sum of per-image values x 5, kept images] per batch is all-reduced (fp64, 48 bytes per
batch, RCCL over xGMI), once, BEFORE the per-batch division -- so a rank that skips an image gives the
single-process number.
"""
from __future__ import annotations

from typing import Dict, Optional

import torch

from . import _lib

NAMES = ("EPE", "D1", "Thres1", "Thres2", "Thres3")


def image_sums(est: torch.Tensor, gt: torch.Tensor, mask: torch.Tensor) -> torch.Tensor:
    """[B,H,W] x3 -> fp64 [B,8]: n_mask, n_gt>0, sum|err|, n_D1, n_err>1, n_err>2, n_err>3, 0."""
    assert est.dim() == 3 and est.size() == gt.size() == mask.size()   # metrics.py:15-19
    if not est.is_cuda:
        raise _lib.DiffuVolumeError("metrics run on the GPU (HIP kernel); move the tensors to cuda")
    est = est.float().contiguous()
    gt = gt.float().contiguous()
    m8 = mask.to(torch.uint8).contiguous()
    b = est.shape[0]
    sums = torch.empty((b, 8), dtype=torch.float64, device=est.device)
    with torch.cuda.device(est.device):
        _lib.check(_lib.load().dv_masked_metrics_f32(est.data_ptr(), gt.data_ptr(), m8.data_ptr(),
                                                     sums.data_ptr(), b, est.shape[1] * est.shape[2],
                                                     _lib.stream_ptr()), "dv_masked_metrics_f32")
    return sums


def per_image_values(sums: torch.Tensor):
    """sums [B,8] -> (values [B,5] = EPE, D1, Thres1..3 per image, keep [B] bool)."""
    hw_ratio = sums[:, 0] / sums[:, 1].clamp(min=1.0)       # mask.mean()/(gt>0).mean()
    ratio = torch.where(sums[:, 1] > 0, hw_ratio, torch.full_like(hw_ratio, float("inf")))
    ratio = torch.where((sums[:, 1] == 0) & (sums[:, 0] == 0), torch.full_like(ratio, float("nan")), ratio)
    keep = ~(ratio < 0.1)                                    # the reference skips only when ratio < 0.1
    n = sums[:, 0].clamp(min=1.0)
    vals = torch.stack([sums[:, 2] / n, sums[:, 3] / n, sums[:, 4] / n, sums[:, 5] / n, sums[:, 6] / n], dim=1)
    return vals, keep


def batch_metrics(est: torch.Tensor, gt: torch.Tensor, mask: torch.Tensor) -> Dict[str, torch.Tensor]:
    """The five scalars ``test_sample`` reports for one batch (test_sceneflow_ddim.py:113-117)."""
    vals, keep = per_image_values(image_sums(est, gt, mask))
    k = keep.to(vals.dtype)
    cnt = k.sum()
    mean = (vals * k[:, None]).sum(0) / cnt.clamp(min=1.0)
    mean = torch.where(cnt > 0, mean, torch.zeros_like(mean))   # "return 0" branch, metrics.py:36-38
    return {n: mean[i].float() for i, n in enumerate(NAMES)}


def EPE_metric(D_est, D_gt, mask):
    return batch_metrics(D_est, D_gt, mask)["EPE"]


def D1_metric(D_est, D_gt, mask):
    return batch_metrics(D_est, D_gt, mask)["D1"]


def Thres_metric(D_est, D_gt, mask, thres):
    assert isinstance(thres, (int, float))
    if float(thres) not in (1.0, 2.0, 3.0):
        raise _lib.DiffuVolumeError("the fused metrics kernel evaluates thresholds 1, 2 and 3")
    return batch_metrics(D_est, D_gt, mask)[f"Thres{int(thres)}"]


def kept_sums(sums: torch.Tensor) -> torch.Tensor:
    """sums [B,8] (``image_sums``) -> fp64 [6] = [sum of the per-image EPE, D1, Thres1, Thres2, Thres3 over the KEPT images of
    this (shard of a) batch, number of kept images] -- the vector SURVEY 8(e) all-reduces."""
    vals, keep = per_image_values(sums)
    k = keep.to(vals.dtype)
    return torch.cat([(vals * k[:, None]).sum(0), k.sum().reshape(1)])


class MetricAccumulator:
    """The reference's metric bookkeeping across batches AND ranks (SceneFlow/utils/metrics.py:22-41 +
    utils/experiment.py:126-151), kept as fp64 sums on the device:

    * one global batch = the shards all ranks hold of it (the reference's nn.DataParallel scatters every batch over the
      GPUs and evaluates the metric on the gathered output).  Per batch every rank records ``kept_sums`` of ITS images:
      5 sums of per-image values over the images it keeps + how many it kept (an image whose mask ratio is < 0.1 is
      skipped, metrics.py:30-31);
    * ``reduce()`` all-reduces the [n_batches, 6] table ONCE (SUM; 48 bytes per batch, RCCL on GPUs / gloo in the CPU
      tests) and only then forms the reference's numbers: batch value = sum / kept images of the GLOBAL batch (0 if all
      were skipped, metrics.py:36-38), run value = mean over batches (AverageMeterDict).  A rank that skips an image
      therefore changes the divisor of that batch exactly as it does in the single-process run.

    Every rank records the same number of batches (each global batch is sharded over all ranks)."""

    def __init__(self, device):
        self.device = torch.device(device)
        self.rows = []                                      # one fp64 [6] per batch, on the device

    def update_sums(self, sums: torch.Tensor) -> None:
        """``sums`` = ``image_sums(est, gt, mask)`` of this rank's shard of one batch."""
        self.rows.append(kept_sums(sums.to(self.device, torch.float64)))

    def update_images(self, est: torch.Tensor, gt: torch.Tensor, mask: torch.Tensor) -> None:
        self.update_sums(image_sums(est, gt, mask))

    def table(self) -> torch.Tensor:
        return torch.stack(self.rows) if self.rows else torch.zeros((0, 6), dtype=torch.float64, device=self.device)

    def reduce(self, group=None) -> Dict[str, float]:
        import torch.distributed as dist
        table = self.table().clone()
        if dist.is_available() and dist.is_initialized():     # also a one-rank group: same RCCL path as N ranks
            if dist.get_backend(group) == "gloo":          # CPU rendezvous (tests on a 1-GPU box): reduce on the host
                table = table.cpu()
            dist.all_reduce(table, op=dist.ReduceOp.SUM, group=group)
        host = table.cpu()
        if host.shape[0] == 0:
            return {name: 0.0 for name in NAMES}
        kept = host[:, 5:6]
        per_batch = torch.where(kept > 0, host[:, :5] / kept.clamp(min=1.0), torch.zeros_like(host[:, :5]))
        mean = per_batch.mean(0)
        return {name: float(mean[i]) for i, name in enumerate(NAMES)}
