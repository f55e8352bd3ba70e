"""IGEV geometry-encoding-volume lookup with the DiffuVolume noise filter, behind the
reference's class API (KITTI15/core/geometry_ddim.py:6-80).

``Combined_Geo_Encoding_Volume(init_fmap1, init_fmap2, geo_volume, num_levels=2, radius=4)``
then ``corr_fn(disp, coords, noisy) -> [B, 162, h, w]`` once per GRU iteration.  The lookup,
the `geo_volume * noise` multiply and both level-1 poolings are one HIP kernel
(``dv_geo_filter_lookup_f32``); the all-pairs correlation of ``__init__`` (one GEMM per image
row, once per pair) is a plain library GEMM via ``torch.einsum``.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

from . import _lib
from .profiling import timed
from .submodule import _dev_f32


class Combined_Geo_Encoding_Volume:
    def __init__(self, init_fmap1, init_fmap2, geo_volume, num_levels=2, radius=4):
        if num_levels != 2 or radius != 4:
            raise _lib.DiffuVolumeError("the HIP lookup implements num_levels=2, radius=4 (every IGEV config)")
        self.num_levels, self.radius = num_levels, radius
        self.geo_volume = _dev_f32(geo_volume, "geo_volume")                 # [B,C,D,h,w], no permuted copy
        b, c, d, h, w = self.geo_volume.shape
        self.channel = c
        corr = self.corr(_dev_f32(init_fmap1, "init_fmap1"), _dev_f32(init_fmap2, "init_fmap2"))
        w2 = corr.shape[-1]
        self.corr0 = corr.reshape(b, h, w, w2).contiguous()
        self.corr1 = F.avg_pool2d(corr.reshape(b * h * w, 1, 1, w2), [1, 2], stride=[1, 2]).reshape(b, h, w, w2 // 2).contiguous()

    def __call__(self, disp, coords, noisy):
        disp = _dev_f32(disp, "disp")
        coords = _dev_f32(coords, "coords")
        noisy = _dev_f32(noisy, "noisy")
        b, c, d, h, w = self.geo_volume.shape
        if disp.numel() != b * h * w or coords.numel() != b * h * w or noisy.numel() != b * h * w * d:
            raise RuntimeError("disp/coords must be [B,1,h,w] and noisy [B,D,h,w] for this volume")
        nch = 2 * (c * (2 * self.radius + 1) + (2 * self.radius + 1))
        out = torch.empty((b, nch, h, w), dtype=torch.float32, device=disp.device)
        lib = _lib.load()
        with torch.cuda.device(disp.device):
            timed("geo_filter_lookup", 0.0, 4.0 * (out.numel() + noisy.numel()),
                  lambda: _lib.check(lib.dv_geo_filter_lookup_f32(
                      self.geo_volume.data_ptr(), self.corr0.data_ptr(), self.corr1.data_ptr(), disp.data_ptr(),
                      coords.data_ptr(), noisy.data_ptr(), out.data_ptr(), b, c, d, h, w, self.corr0.shape[-1],
                      self.radius, _lib.stream_ptr()), "dv_geo_filter_lookup_f32"))
        return out

    @staticmethod
    def corr(fmap1, fmap2):
        """All-pairs correlation along the epipolar line (geometry_ddim.py:72-80): [B,H,W1,1,W2]."""
        B, D, H, W1 = fmap1.shape
        W2 = fmap2.shape[-1]
        corr = torch.einsum("aijk,aijh->ajkh", fmap1, fmap2)
        return corr.reshape(B, H, W1, 1, W2).contiguous()
