"""File side of the SceneFlow test pipeline: PFM disparity files, file lists and the evaluation
crop, with the reference's names and return conventions (SceneFlow/datasets/data_io.py:24-66,
sceneflow_dataset.py:57-70).  Host-side numpy; nothing here touches the GPU.
"""
from __future__ import annotations

import re
from typing import List, Tuple

import numpy as np

IMAGENET_MEAN = (0.485, 0.456, 0.406)       # get_transform(), data_io.py:6-13
IMAGENET_STD = (0.229, 0.224, 0.225)

_DIMS = re.compile(rb"^(\d+)\s(\d+)\s$")


def read_all_lines(filename: str) -> List[str]:
    """data_io.py:24-28: the lines of a list file without their trailing whitespace."""
    with open(filename) as f:
        return [line.rstrip() for line in f]


def pfm_imread(filename: str) -> Tuple[np.ndarray, float]:
    """data_io.py:32-66: (image [H,W] or [H,W,3] float32 with row 0 at the TOP, |scale|).  The PFM payload is stored
    bottom row first; a negative scale marks little-endian samples.  Raises ``Exception('Not a PFM file.')`` /
    ``Exception('Malformed PFM header.')`` like the reference."""
    with open(filename, "rb") as f:
        magic = f.readline().decode("utf-8").rstrip()
        if magic not in ("PF", "Pf"):
            raise Exception("Not a PFM file.")
        m = _DIMS.match(f.readline())
        if m is None:
            raise Exception("Malformed PFM header.")
        width, height = int(m.group(1)), int(m.group(2))
        scale = float(f.readline().rstrip())
        payload = f.read()
    order = "<" if scale < 0 else ">"
    samples = np.frombuffer(payload, dtype=order + "f4")
    shape = (height, width, 3) if magic == "PF" else (height, width)
    return samples.reshape(shape)[::-1], abs(scale)


def load_disp(filename: str) -> np.ndarray:
    """SceneFlowDatset.load_disp (sceneflow_dataset.py:26-29): contiguous native float32 [H,W]."""
    data, _ = pfm_imread(filename)
    return np.ascontiguousarray(data, dtype=np.float32)


def eval_crop(left: np.ndarray, right: np.ndarray, disparity: np.ndarray, crop_w: int = 960, crop_h: int = 512):
    """The test-time crop of sceneflow_dataset.py:57-65: the bottom-right crop_h x crop_w window of a 960x540 frame
    (images [H,W,3] or [3,H,W]; disparity [H,W])."""
    h, w = disparity.shape[-2:]

    def win(a):
        if a.ndim == 3 and a.shape[-1] == 3 and a.shape[0] != 3:       # HWC
            return a[h - crop_h:h, w - crop_w:w]
        return a[..., h - crop_h:h, w - crop_w:w]

    return win(left), win(right), disparity[h - crop_h:h, w - crop_w:w]


def normalize_image(img_hwc_u8: np.ndarray) -> np.ndarray:
    """get_transform() of data_io.py:6-13 (ToTensor + ImageNet Normalize): uint8 [H,W,3] -> float32 [3,H,W]."""
    x = img_hwc_u8.astype(np.float32) / 255.0
    x = (x - np.asarray(IMAGENET_MEAN, np.float32)) / np.asarray(IMAGENET_STD, np.float32)
    return np.ascontiguousarray(x.transpose(2, 0, 1))
