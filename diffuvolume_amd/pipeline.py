"""File-to-disparity evaluation step of the SceneFlow flavour: what `test_sample` of
SceneFlow/test_sceneflow_ddim.py:88-122 does with one dataset item, from the files on disk to the five metrics.

    sample  = load_sceneflow_sample(left.png, right.png, disparity.pfm)       # sceneflow_dataset.py:36-70 (eval branch)
    scalars = test_sample(model_origin, model_ddim, sample)                   # test_sceneflow_ddim.py:88-122

File decoding and the crop run on the host (numpy / PIL); everything from the images on is the HIP path
(origin ACVNet -> used / quarter-resolution disparity -> ACVNet_DDIM.forward -> fused metrics).
"""
from __future__ import annotations

from typing import Dict

import numpy as np
import torch
import torch.nn.functional as F

from . import data_io
from . import metrics as M


def load_sceneflow_sample(left_path: str, right_path: str, disp_path: str, crop_w: int = 960, crop_h: int = 512) -> Dict:
    """One evaluation item as SceneFlowDatset.__getitem__ builds it with training=False (sceneflow_dataset.py:36-70):
    RGB images, PFM disparity, the bottom-right 960x512 crop, ImageNet normalisation.  Tensors carry a batch axis."""
    from PIL import Image
    left = np.asarray(Image.open(left_path).convert("RGB"))
    right = np.asarray(Image.open(right_path).convert("RGB"))
    disp = data_io.load_disp(disp_path)
    left, right, disp = data_io.eval_crop(left, right, disp, crop_w, crop_h)
    return {"left": torch.from_numpy(data_io.normalize_image(left)).unsqueeze(0),
            "right": torch.from_numpy(data_io.normalize_image(right)).unsqueeze(0),
            "disparity": torch.from_numpy(np.ascontiguousarray(disp)).unsqueeze(0),
            "top_pad": 0, "right_pad": 0, "left_filename": left_path}


@torch.no_grad()
def test_sample(model_origin, model, sample: Dict, maxdisp: int = 192, device: str = "cuda") -> Dict[str, float]:
    """test_sceneflow_ddim.py:88-122: origin network -> `used` (full resolution) and the quarter-resolution disparity
    that seeds x_T -> the DDIM model -> EPE / D1 / Thres1-3 of its prediction (one fused pass, one host sync)."""
    model.eval()
    model_origin.eval()
    img_l, img_r = sample["left"].to(device), sample["right"].to(device)
    disp_gt = sample["disparity"].to(device)
    mask_gt = (disp_gt < maxdisp) & (disp_gt > 0)
    disp_ = model_origin(img_l, img_r)[-1]
    disp_net = torch.clamp(disp_, 0, maxdisp - 1).unsqueeze(1)
    b, c, h, w = disp_net.shape
    disp_net = F.interpolate(disp_net, size=(h // 4, w // 4), mode="bilinear") / 4
    disp_ests = model(img_l, img_r, disp_, disp_net, None)
    out = M.batch_metrics(disp_ests[0], disp_gt, mask_gt)
    return {k: float(v) for k, v in out.items()}


def _sintel_pad(ht: int, wd: int, divis_by: int = 32):
    """core/utils/utils.py:9-14 (`InputPadder`, mode 'sintel'): (left, right, top, bottom) replicate padding."""
    pad_ht = (((ht // divis_by) + 1) * divis_by - ht) % divis_by
    pad_wd = (((wd // divis_by) + 1) * divis_by - wd) % divis_by
    return [pad_wd // 2, pad_wd - pad_wd // 2, pad_ht // 2, pad_ht - pad_ht // 2]


@torch.no_grad()
def validate_kitti_sample(model_origin, model, image1: torch.Tensor, image2: torch.Tensor, flow_gt: torch.Tensor,
                          valid_gt: torch.Tensor, iters: int = 32, device: str = "cuda") -> Dict[str, float]:
    """One item of KITTI15/evaluate_stereo.py:80-117 (`validate_kitti`): pad to a multiple of 32, origin IGEV-Stereo ->
    `flow_pr`, its clamped quarter-resolution copy `flow_4`, `IGEVStereo_ddim` -> refined disparity, unpad, per-image EPE
    and D1 (> 3 px) over the valid pixels.  image1 / image2 [3,H,W] in 0..255, flow_gt [1,H,W], valid_gt [H,W]."""
    model.eval()
    model_origin.eval()
    image1, image2 = image1[None].to(device), image2[None].to(device)
    pad = _sintel_pad(*image1.shape[-2:])
    image1, image2 = (F.pad(x, pad, mode="replicate") for x in (image1, image2))
    flow_pr = model_origin(image1, image2, iters=iters, test_mode=True)
    b, c, h, w = image1.shape
    flow_ori = torch.clamp(flow_pr, 0, w - 1)
    flow_4 = F.interpolate(flow_ori, size=(h // 4, w // 4), mode="bilinear") / 4
    _, flow_refine = model(image1, image2, flow_pr, flow_4, iters=iters, test_mode=True)
    flow_refine = flow_refine.reshape(b, 1, h, w)
    flow_refine = flow_refine[..., pad[2]:h - pad[3], pad[0]:w - pad[1]].cpu().squeeze(0)
    assert flow_refine.shape == flow_gt.shape, (flow_refine.shape, flow_gt.shape)
    epe = torch.sum((flow_refine - flow_gt.cpu()) ** 2, dim=0).sqrt().flatten()
    val = (valid_gt.cpu().flatten() >= 0.5) & (flow_gt.cpu().abs().flatten() < 192)
    return {"epe": float(epe[val].mean()), "d1": float((epe > 3.0)[val].float().mean())}
