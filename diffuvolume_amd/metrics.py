"""EPE / D1 / Thres metrics of SceneFlow/utils/metrics.py:22-65 and their reduction across
the GPUs of a node.

The reference evaluates each metric image by image with boolean-index gathers and one
``.item()`` sync per metric.  Here one HIP pass yields seven per-image sums
(``dv_masked_metrics_f32``); the reference's semantics (per-image mean, images whose mask
ratio is < 0.1 skipped, batch value = mean over kept images, run value = mean over batches:
metrics.py:30-40 + experiment.py:146-151) are applied to those sums.  Across ranks only
the 6 x fp64 running sums are all-reduced (RCCL over xGMI), once.
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


class MetricAccumulator:
    """AverageMeterDict semantics (experiment.py:126-151: mean over *batches* of the per-batch
    means) kept as fp64 sums on the device; ``reduce()`` all-reduces them over the process
    group (one 48-byte SUM, RCCL on GPUs / gloo in the CPU tests)."""

    def __init__(self, device):
        self.state = torch.zeros(6, dtype=torch.float64, device=device)   # 5 metric sums + batch count

    def update(self, batch: Dict[str, torch.Tensor]) -> None:
        vals = torch.stack([batch[n].double() for n in NAMES])
        self.state[:5] += vals.to(self.state.device)
        self.state[5] += 1

    def reduce(self, group=None) -> Dict[str, float]:
        import torch.distributed as dist
        state = self.state.clone()
        if dist.is_available() and dist.is_initialized():     # also a one-rank group: same RCCL path as N ranks
            if dist.get_backend(group) == "gloo":          # CPU rendezvous (tests on a 1-GPU box): reduce on the host
                state = state.cpu()
            dist.all_reduce(state, op=dist.ReduceOp.SUM, group=group)
        host = state.cpu()
        n = max(float(host[5]), 1.0)
        return {name: float(host[i]) / n for i, name in enumerate(NAMES)}
