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
