"""IGEV geometry-encoding-volume lookup with the DiffuVolume noise filter, behind the
reference's class API (KITTI15/core/geometry_ddim.py:6-80).

``Combined_Geo_Encoding_Volume(init_fmap1, init_fmap2, geo_volume, num_levels=2, radius=4)``
then ``corr_fn(disp, coords, noisy) -> [B, 162, h, w]`` once per GRU iteration.  The lookup,
the `geo_volume * noise` multiply and both level-1 poolings are one HIP kernel
(``dv_geo_filter_lookup_f32``); the all-pairs correlation of ``__init__`` (one GEMM per image
row, once per pair) and its pooled level are a second one (``dv_allpairs_corr_f32``).
"""
from __future__ import annotations

import torch

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
        self.corr0, self.corr1 = self._corr_levels(_dev_f32(init_fmap1, "init_fmap1"), _dev_f32(init_fmap2, "init_fmap2"))

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
    def _corr_levels(fmap1, fmap2):
        """corr0 [B,H,W1,W2] and its avg_pool2d([1,2]) level corr1 [B,H,W1,W2//2] in one launch."""
        B, C, H, W1 = fmap1.shape
        if fmap2.shape[:3] != fmap1.shape[:3]:
            raise RuntimeError(f"feature maps must agree in batch, channels and height: {tuple(fmap1.shape)} vs {tuple(fmap2.shape)}")
        W2 = fmap2.shape[-1]
        corr0 = torch.empty((B, H, W1, W2), dtype=torch.float32, device=fmap1.device)
        corr1 = torch.empty((B, H, W1, W2 // 2), dtype=torch.float32, device=fmap1.device)
        lib = _lib.load()
        with torch.cuda.device(fmap1.device):
            timed("allpairs_corr", 2.0 * B * H * W1 * W2 * C, 4.0 * (fmap1.numel() + fmap2.numel() + corr0.numel() + corr1.numel()),
                  lambda: _lib.check(lib.dv_allpairs_corr_f32(fmap1.data_ptr(), fmap2.data_ptr(), corr0.data_ptr(),
                                                              corr1.data_ptr(), B, C, H, W1, W2, _lib.stream_ptr()),
                                     "dv_allpairs_corr_f32"), issued=2.0 * B * H * W1 * W2 * C)
        return corr0, corr1

    @staticmethod
    def corr(fmap1, fmap2):
        """All-pairs correlation along the epipolar line (geometry_ddim.py:72-80): [B,H,W1,1,W2]."""
        fmap1, fmap2 = _dev_f32(fmap1, "fmap1"), _dev_f32(fmap2, "fmap2")
        corr0, _ = Combined_Geo_Encoding_Volume._corr_levels(fmap1, fmap2)
        B, H, W1, W2 = corr0.shape
        return corr0.reshape(B, H, W1, 1, W2)
