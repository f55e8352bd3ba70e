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


class GeoLookupRequest:
    """A lookup that has not run yet: what `Combined_Geo_Encoding_Volume.request` hands to this build's update block, whose
    motion encoder consumes the lookup through ONE 1x1 convolution (BasicMotionEncoder.convc1, KITTI15/core/update.py:79,:89)
    and can therefore ask for `conv1x1` -- lookup and convolution in one kernel, the [B,162,h,w] tensor never written.
    `materialize()` is the plain lookup (what the reference's `corr_fn(...)` returns)."""

    def __init__(self, volume, disp, coords, noisy):
        self.volume, self.disp, self.coords, self.noisy = volume, disp, coords, noisy

    def materialize(self) -> torch.Tensor:
        return self.volume(self.disp, self.coords, self.noisy)

    def conv1x1(self, wpacked: torch.Tensor, bias, act: int) -> torch.Tensor:
        return self.volume.lookup_conv1x1(self.disp, self.coords, self.noisy, wpacked, bias, act)


def pack_lookup_conv1x1(weight: torch.Tensor, channel: int = 8) -> torch.Tensor:
    """nn.Conv2d(2*(9*channel+9), 64, 1).weight -> the layout `dv_geo_filter_lookup_conv1x1_f32` reads."""
    w = _dev_f32(weight.detach().reshape(weight.shape[0], -1), "weight")
    if tuple(w.shape) != (64, 2 * (9 * channel + 9)):
        raise _lib.DiffuVolumeError(f"fused lookup + 1x1 convolution: weight [64, {2 * (9 * channel + 9)}], got {tuple(w.shape)}")
    lib = _lib.load()
    out = torch.empty(lib.dv_geo_lookup_conv1x1_packed_floats(channel), dtype=torch.float32, device=w.device)
    with torch.cuda.device(w.device):
        _lib.check(lib.dv_geo_lookup_conv1x1_pack_weights_f32(w.data_ptr(), out.data_ptr(), channel, _lib.stream_ptr()),
                   "dv_geo_lookup_conv1x1_pack_weights_f32")
    return out


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

    def request(self, disp, coords, noisy) -> GeoLookupRequest:
        return GeoLookupRequest(self, disp, coords, noisy)

    def lookup_conv1x1(self, disp, coords, noisy, wpacked, bias, act):
        """act(conv1x1(lookup(disp, coords, noisy)) + bias) -> [B,64,h,w] in one launch (`pack_lookup_conv1x1` weights)."""
        disp = _dev_f32(disp, "disp")
        coords = _dev_f32(coords, "coords")
        noisy = _dev_f32(noisy, "noisy")
        b, c, d, h, w = self.geo_volume.shape
        if disp.numel() != b * h * w or coords.numel() != b * h * w or noisy.numel() != b * h * w * d:
            raise RuntimeError("disp/coords must be [B,1,h,w] and noisy [B,D,h,w] for this volume")
        nch = 2 * (c * (2 * self.radius + 1) + (2 * self.radius + 1))
        out = torch.empty((b, 64, h, w), dtype=torch.float32, device=disp.device)
        lib = _lib.load()
        with torch.cuda.device(disp.device):
            fl = 2.0 * out.numel() * nch
            timed("geo_filter_lookup_conv1x1", fl, 4.0 * (out.numel() + noisy.numel()),
                  lambda: _lib.check(lib.dv_geo_filter_lookup_conv1x1_f32(
                      self.geo_volume.data_ptr(), self.corr0.data_ptr(), self.corr1.data_ptr(), disp.data_ptr(),
                      coords.data_ptr(), noisy.data_ptr(), wpacked.data_ptr(), _lib.ptr(bias), out.data_ptr(), b, c, d, h, w,
                      self.corr0.shape[-1], self.radius, 64, act, _lib.stream_ptr()), "dv_geo_filter_lookup_conv1x1_f32"),
                  issued=fl * 20.0 / 18.0)
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
