"""IGEV-Stereo + DiffuVolume: the DDIM volume-filter loop (KITTI15/core/igev_stereo_ddim.py:226-359).

What is here: the pieces of ``IGEVStereo_ddim`` that are on the hot path -- the time embedding with its
180 -> 48 channel interpolation (core/head.py:74-83), ``model_predictions`` (noise filter -> `iters` GRU
iterations each looking the filtered geometry volume up -> two-hot re-encoding -> noise prediction) and
``ddim_sample`` (renewal mask dif<5, output rule dif<3, fresh q_sample fill, ensemble [0.6,0.1,0.3]).
The geometry lookup, the filter and the DDIM state update are HIP kernels; the ConvGRU update block and the
convex upsampling are 2-D PyTorch modules supplied by the caller (``update_block`` / ``upsample_disp``), exactly
as the reference method calls them.  The MobileNetV2 backbone (timm, pretrained) and the rest of the
``IGEVStereo_ddim`` constructor are out of scope, so this is a loop object rather than
the full nn.Module.  Batch handling: the reference head is batch-1 only; here the shift is
taken per sample.
"""
from __future__ import annotations

import ctypes
from typing import Callable, Optional, Sequence

import torch
import torch.nn.functional as F
from torch import nn

from . import _lib
from .acv_ddim import cosine_beta_schedule
from .head import SinusoidalPositionEmbeddings
from .submodule import _dev_f32


class DynamicHead180(nn.Module):
    """KITTI15/core/head.py:51-83: DynamicHead(d_model=180) whose 180-channel shift is linearly
    interpolated to the 48 disparity bins before it is added."""

    def __init__(self, d_model: int = 180, bins: int = 48):
        super().__init__()
        self.d_model, self.bins = d_model, bins
        width = d_model * 4
        self.time_mlp = nn.Sequential(SinusoidalPositionEmbeddings(d_model), nn.Linear(d_model, width), nn.GELU(),
                                      nn.Linear(width, width))
        self.block_time_mlp = nn.Sequential(nn.SiLU(), nn.Linear(width, d_model))
        for p in self.parameters():
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)

    def shift(self, t: torch.Tensor) -> torch.Tensor:
        s = self.block_time_mlp(self.time_mlp(t))                                   # [B,180]
        return F.interpolate(s.unsqueeze(1), self.bins, mode="linear").squeeze(1)    # [B,48]

    def forward(self, noisy, t):
        return noisy + self.shift(t).unsqueeze(-1).unsqueeze(-1)


class IGEVDiffusionLoop:
    def __init__(self, time_embedding: DynamicHead180, update_block: Callable, upsample_disp: Callable,
                 n_gru_layers: int = 3, slow_fast_gru: bool = False, sampling_timesteps: int = 2,
                 ensemble_cof: Sequence[float] = (0.6, 0.1, 0.3)):
        if len(ensemble_cof) != sampling_timesteps + 1:
            raise ValueError("ensemble_cof needs sampling_timesteps + 1 entries")
        self.time_embedding, self.update_block, self.upsample_disp = time_embedding, update_block, upsample_disp
        self.n_gru_layers, self.slow_fast_gru = n_gru_layers, slow_fast_gru
        self.num_timesteps, self.sampling_timesteps, self.eta = 1000, sampling_timesteps, 1.0
        self.ensemble_cof = tuple(float(c) for c in ensemble_cof)
        ac = torch.cumprod(1.0 - cosine_beta_schedule(1000), dim=0)
        self.alphas_cumprod = ac
        self.sqrt_ac, self.sqrt_1mac = torch.sqrt(ac), torch.sqrt(1.0 - ac)
        self.sqrt_recip, self.sqrt_recipm1 = torch.sqrt(1.0 / ac), torch.sqrt(1.0 / ac - 1)

    def _time_pairs(self):
        times = torch.linspace(-1, self.num_timesteps - 1, steps=self.sampling_timesteps + 1)
        times = list(reversed(times.int().tolist()))
        return list(zip(times[:-1], times[1:]))

    def _filter(self, x_t, t):
        b, c, h, w = x_t.shape
        shift = self.time_embedding.shift(t).float().contiguous()
        lib = _lib.load()
        x_t = x_t.contiguous()
        n01 = torch.empty_like(x_t)
        if x_t.dtype == torch.float32:
            _lib.check(lib.dv_noise_prepare_f32(x_t.data_ptr(), shift.data_ptr(), n01.data_ptr(), b, c, h * w,
                                                _lib.stream_ptr()), "dv_noise_prepare_f32")
            return n01, n01
        n01f = torch.empty(x_t.shape, dtype=torch.float32, device=x_t.device)
        _lib.check(lib.dv_noise_prepare_f64(x_t.data_ptr(), shift.data_ptr(), n01.data_ptr(), n01f.data_ptr(), b, c,
                                            h * w, _lib.stream_ptr()), "dv_noise_prepare_f64")
        return n01, n01f

    def _gru_iterations(self, coords0, coords1, flow_init, iters, net_list, inp_list, corr_fn, n01f, stem_2x):
        """igev_stereo_ddim.py:233-261 -- the 2-D update block is the caller's; the lookup is HIP."""
        if flow_init is not None:
            coords1 = coords1 + flow_init
        flow_up = None
        for itr in range(iters):
            flow = coords1 - coords0
            corr = corr_fn(flow, coords1, n01f)
            if self.n_gru_layers == 3 and self.slow_fast_gru:
                net_list = self.update_block(net_list, inp_list, iter32=True, iter16=False, iter08=False, update=False)
            if self.n_gru_layers >= 2 and self.slow_fast_gru:
                net_list = self.update_block(net_list, inp_list, iter32=self.n_gru_layers == 3, iter16=True,
                                             iter08=False, update=False)
            net_list, up_mask, delta_flow = self.update_block(net_list, inp_list, corr, flow,
                                                              iter16=self.n_gru_layers == 3,
                                                              iter08=self.n_gru_layers >= 2)
            coords1 = coords1 + delta_flow
            if itr == iters - 1:
                flow_up = self.upsample_disp(coords1 - coords0, up_mask, stem_2x)[:, :1]
        return flow_up, coords1, net_list

    def _coef(self, time, time_next, cof):
        k = _lib.DvDdimCoef()
        k.sqrt_recip_alpha, k.sqrt_recipm1_alpha = float(self.sqrt_recip[time]), float(self.sqrt_recipm1[time])
        k.dif_thr, k.unc_thr, k.cof, k.last = 5.0, float("inf"), cof, int(time_next < 0)
        k.clamp_max, k.ens_dif_thr = 47.0, 3.0        # clamp(pred, 0, 48-1) :265; output rule dif<3 :323-327
        if time_next >= 0:
            alpha, alpha_next = self.alphas_cumprod[time], self.alphas_cumprod[time_next]
            sigma = self.eta * ((1 - alpha / alpha_next) * (1 - alpha_next) / (1 - alpha)).sqrt()
            k.sigma, k.c, k.sqrt_alpha_next = float(sigma), float((1 - alpha_next - sigma ** 2).sqrt()), float(alpha_next.sqrt())
        return k

    def _update(self, pred, used, coords0, n01, eps, fill, mask, ens, coef, want_pred_noise=False):
        b, c, h, w = n01.shape
        dev = pred.device
        x_start = torch.empty((b, c, h, w), dtype=torch.float32, device=dev)
        x_next = None if coef.last else torch.empty((b, c, h, w), dtype=torch.float64, device=dev)
        pn = torch.empty((b, c, h, w), dtype=torch.float64, device=dev) if want_pred_noise else None
        f32 = n01.dtype == torch.float32
        e32 = eps if (eps is not None and eps.dtype == torch.float32) else None
        e64 = eps if (eps is not None and eps.dtype == torch.float64) else None
        _lib.check(_lib.load().dv_ddim_step(pred.data_ptr(), 0, used.data_ptr(), coords0.data_ptr(),
                                            n01.data_ptr() if f32 else 0, 0 if f32 else n01.data_ptr(),
                                            _lib.ptr(e32), _lib.ptr(e64), _lib.ptr(fill), mask.data_ptr(),
                                            x_start.data_ptr(), _lib.ptr(pn), _lib.ptr(x_next), _lib.ptr(ens),
                                            b, c, h, w, ctypes.byref(coef), _lib.stream_ptr()), "dv_ddim_step")
        return x_start, x_next, pn

    @torch.no_grad()
    def model_predictions(self, coords0, coords1, flow_init, iters, net_list, inp_list, corr_fn, noise, t, stem_2x):
        """igev_stereo_ddim.py:226-292 -> (pred_noise fp64, x_start fp32, pred [B,1,H,W], coords1)."""
        n01, n01f = self._filter(noise, t)
        pred, coords1, _ = self._gru_iterations(coords0, coords1, flow_init, iters, net_list, inp_list, corr_fn, n01f, stem_2x)
        pred = _dev_f32(pred, "pred")
        b, _, hh, ww = pred.shape
        c0 = _dev_f32(coords0, "coords0").reshape(b, hh // 4, ww // 4)
        mask = torch.zeros((b, hh // 4, ww // 4), dtype=torch.float32, device=pred.device)
        coef = self._coef(int(t.reshape(-1)[0]), -1, 0.0)
        p2 = pred.reshape(b, hh, ww)
        x_start, _, pn = self._update(p2, p2, c0, n01, None, None, mask, None, coef, want_pred_noise=True)
        return pn, x_start, pred, coords1

    @torch.no_grad()
    def ddim_sample(self, coords0, coords1, flow_init, iters, net_list, inp_list, corr_fn, used, asd, stem_2x,
                    noise: Optional[Callable] = None, generator: Optional[torch.Generator] = None):
        """igev_stereo_ddim.py:294-359.  Draws in reference order: 'x_T' (randn_like(asd) :303), then per
        non-final step 'eps' (:338) and 'q' (randn_like inside q_sample :343)."""
        asd = _dev_f32(asd, "asd")
        b, d, h, w = asd.shape
        dev = asd.device
        used2 = _dev_f32(used, "used").reshape(b, 4 * h, 4 * w)

        def draw(kind, shape, dtype):
            if noise is not None:
                return noise(kind, shape, dtype).to(device=dev, dtype=dtype).contiguous()
            return torch.randn(shape, device=dev, dtype=dtype, generator=generator)

        img = draw("x_T", tuple(asd.shape), torch.float32)
        mask = torch.zeros((b, h, w), dtype=torch.float32, device=dev)
        ens = used2 * self.ensemble_cof[0]
        c0 = _dev_f32(coords0, "coords0").reshape(b, h, w)
        for i, (time, time_next) in enumerate(self._time_pairs()):
            t = torch.full((b,), time, device=dev, dtype=torch.long)
            n01, n01f = self._filter(img, t)
            pred, coords1, net_list = self._gru_iterations(coords0, coords1, flow_init, iters, net_list, inp_list,
                                                           corr_fn, n01f, stem_2x)
            pred2 = _dev_f32(pred, "pred").reshape(b, 4 * h, 4 * w)
            coef = self._coef(time, time_next, self.ensemble_cof[i + 1])
            eps = fill = None
            if time_next >= 0:
                eps = draw("eps", tuple(img.shape), img.dtype)
                fill = (self.sqrt_ac[time].item() * asd.double()
                        + self.sqrt_1mac[time].item() * draw("q", tuple(asd.shape), asd.dtype).double()).contiguous()
            x_start, x_next, _ = self._update(pred2, used2, c0, n01, eps, fill, mask, ens, coef)
            img = x_start if time_next < 0 else x_next
        return ens
