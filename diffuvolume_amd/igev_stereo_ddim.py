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
import os
from typing import Callable, Optional, Sequence

import torch
import torch.nn.functional as F
from torch import nn

from . import _lib
from .acv_ddim import cosine_beta_schedule
from .head import SinusoidalPositionEmbeddings
from .submodule import (ACT_LEAKY, ACT_NONE, ACT_RELU, Conv2dPlan, Conv3dPlan, Deconv2dK4S2Plan, Deconv3dPlan, _dev_f32, build_gwc_volume,
                        feature_gate, softmax_regress)


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

    # ---- the GRU iterations of one DDIM step as a hipGraph -------------------------------------------------------
    # One step launches `iters` x ~30 small kernels (1/8- and 1/16-scale convolutions of 0.06-0.1 ms) from Python through
    # ctypes: the launch gaps are a measurable part of config 5 (20 steps x 32 iterations per pair).  Within ONE forward
    # everything a step reads besides its carried state is constant (features, context, geometry volume, stem), so the
    # first step of a forward is captured and the other steps replay it with (coords1, hidden states, filtered noise)
    # copied into the graph's static inputs.  The eager loop below is the same code the capture records; a new forward
    # (new corr_fn object) re-captures; one graph is kept.  MEASURED (round 5, 1248x384, batch 4, 20 x 32 iterations, same
    # box): eager 2 059.9 ms per forward, graph 2 082.0 ms -- the eager loop is already GPU-bound (the host runs ahead of
    # 0.06-0.1 ms kernels), a replay only adds the state copies and the capture.  OPT-IN: `use_graph` / DV_IGEV_GRAPH=1.
    use_graph = os.environ.get("DV_IGEV_GRAPH", "0") == "1"
    _graph = None
    _graph_warm = False

    def _gru_iterations(self, coords0, coords1, flow_init, iters, net_list, inp_list, corr_fn, n01f, stem_2x):
        ok = (self.use_graph and flow_init is None and coords1.is_cuda and not torch.cuda.is_current_stream_capturing()
              and all(isinstance(t, torch.Tensor) for t in net_list))
        if not ok:
            return self._gru_iterations_eager(coords0, coords1, flow_init, iters, net_list, inp_list, corr_fn, n01f, stem_2x)
        if not self._graph_warm:           # plans / packed weights are built lazily on the first pass: never inside a capture
            self._graph_warm = True
            return self._gru_iterations_eager(coords0, coords1, flow_init, iters, net_list, inp_list, corr_fn, n01f, stem_2x)
        key = (id(corr_fn), id(inp_list), iters, tuple(coords1.shape), coords0.data_ptr(), id(stem_2x))
        g = self._graph
        if g is None or g["key"] != key:
            self._graph = g = None                                   # drop the previous graph (and its memory pool) first
            st = {"key": key, "coords1": coords1.clone(), "net": [t.clone() for t in net_list], "n01f": n01f.clone()}
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                st["out"] = self._gru_iterations_eager(coords0, st["coords1"], None, iters, list(st["net"]), inp_list,
                                                       corr_fn, st["n01f"], stem_2x)
            st["graph"] = graph
            st["keep"] = (corr_fn, inp_list, stem_2x, coords0)       # what the recorded launches point at stays alive
            self._graph = g = st
        else:
            g["coords1"].copy_(coords1)
            for dst, src in zip(g["net"], net_list):
                dst.copy_(src)
            g["n01f"].copy_(n01f)
        g["graph"].replay()
        flow_up, c1, nl = g["out"]
        # the graph's outputs live in its pool and are overwritten by the next replay: hand out copies.  (The key's objects
        # are kept alive by the cached graph, so neither their ids nor their device addresses can be reused by a later forward.)
        return flow_up.clone(), c1.clone(), [t.clone() for t in nl]

    def _gru_iterations_eager(self, coords0, coords1, flow_init, iters, net_list, inp_list, corr_fn, n01f, stem_2x):
        """igev_stereo_ddim.py:233-261 -- the 2-D update block is the caller's; the lookup is HIP."""
        if flow_init is not None:
            coords1 = coords1 + flow_init
        flow_up = None
        # mask_feat_4 is read only after the last iteration (:255-259): this build's update block can skip it elsewhere.
        # (The update block runs lookup + motion encoder on a side stream beside gru16 / gru08: update.py, OVERLAP.)
        from .update import BasicMultiUpdateBlock
        skip_mask = isinstance(self.update_block, BasicMultiUpdateBlock)
        for itr in range(iters):
            flow = coords1 - coords0
            # this build's update block takes the lookup as a request and runs it fused with its first convolution
            corr = corr_fn.request(flow, coords1, n01f) if (skip_mask and hasattr(corr_fn, "request")) else corr_fn(flow, coords1, n01f)
            if self.n_gru_layers == 3 and self.slow_fast_gru:
                net_list = self.update_block(net_list, inp_list, iter32=True, iter16=False, iter08=False, update=False)
            if self.n_gru_layers >= 2 and self.slow_fast_gru:
                net_list = self.update_block(net_list, inp_list, iter32=self.n_gru_layers == 3, iter16=True,
                                             iter08=False, update=False)
            net_list, up_mask, delta_flow = self.update_block(net_list, inp_list, corr, flow,
                                                              iter16=self.n_gru_layers == 3,
                                                              iter08=self.n_gru_layers >= 2,
                                                              **({"mask": itr == iters - 1} if skip_mask else {}))
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
        for i, (time, time_next) in enumerate(self._time_pairs()):
            eps = fill = None
            if time_next >= 0:
                eps = draw("eps", tuple(img.shape), img.dtype)
                fill = (self.sqrt_ac[time].item() * asd.double()
                        + self.sqrt_1mac[time].item() * draw("q", tuple(asd.shape), asd.dtype).double()).contiguous()
            _, x_start, x_next, coords1, net_list = self.ddim_step(i, coords0, coords1, flow_init, iters, net_list, inp_list,
                                                                   corr_fn, used2, img, mask, ens, eps, fill, stem_2x)
            img = x_start if time_next < 0 else x_next
        return ens

    @torch.no_grad()
    def ddim_step(self, i, coords0, coords1, flow_init, iters, net_list, inp_list, corr_fn, used, img, mask, ens=None,
                  eps=None, fill=None, stem_2x=None):
        """Iteration ``i`` of the loop of igev_stereo_ddim.py:306-351 from explicit state: ``img`` entering the step,
        ``mask`` (updated in place), ``coords1`` and the hidden states ``net_list`` as the previous step left them,
        ``eps`` = randn_like(img), ``fill`` = q_sample(asd, t).  Returns (pred [B,4h,4w], x_start fp32,
        x_next fp64 | None, coords1, net_list)."""
        time, time_next = self._time_pairs()[i]
        b, _, h, w = img.shape
        dev = img.device
        t = torch.full((b,), time, device=dev, dtype=torch.long)
        n01, n01f = self._filter(img, t)
        pred, coords1, net_list = self._gru_iterations(coords0, coords1, flow_init, iters, net_list, inp_list,
                                                       corr_fn, n01f, stem_2x)
        pred2 = _dev_f32(pred, "pred").reshape(b, 4 * h, 4 * w)
        used2 = _dev_f32(used, "used").reshape(b, 4 * h, 4 * w)
        c0 = _dev_f32(coords0, "coords0").reshape(b, h, w)
        coef = self._coef(time, time_next, self.ensemble_cof[i + 1])
        x_start, x_next, _ = self._update(pred2, used2, c0, n01, eps, fill, mask, ens, coef)
        return pred2, x_start, x_next, coords1, net_list


# ---------------------------------------------------------------------------------------------------
# The once-per-pair cost-volume front of IGEVStereo_ddim.forward (igev_stereo_ddim.py:377-386):
# gwc volume (8 groups) -> corr_stem -> FeatureAtt -> hourglass(8) -> classifier -> softmax -> regression.
# Module and parameter names are the reference's, so its checkpoints load unchanged
# (`corr_stem.conv.weight`, `cost_agg.feature_att_16.feat_att.1.bias`, ...).
# ---------------------------------------------------------------------------------------------------
# ---------------------------------------------------------------------------------------------------
# The once-per-pair 2-D front on the in-tree kernels (round 5): every nn.Conv2d / nn.ConvTranspose2d / BatchNorm2d (eval) /
# InstanceNorm2d / activation of the feature pyramid, the stems, the context encoder and the spx heads runs through
# `hip_conv2d` / `instance_norm_act` -- no MIOpen, so a rerun and a shard of a batch give the same bits as the batch
# (MIOpen may pick another solver on a later call or for another batch size).  Plans are cached per module and rebuilt when
# a weight is loaded, moved or overwritten in place (key = data pointers + versions).  A tensor that asks for gradients
# takes the module's own torch expression (explicit autograd dispatch, like submodule.py's builders).
# ---------------------------------------------------------------------------------------------------
import weakref

_PLAN_CACHE = weakref.WeakKeyDictionary()


def _wants_autograd(x) -> bool:
    return torch.is_grad_enabled() and x.requires_grad


def _bn_tuple(bn):
    return None if bn is None else (bn.weight, bn.bias, bn.running_mean, bn.running_var)


def hip_conv2d(conv: nn.Module, x: torch.Tensor, bn: Optional[nn.BatchNorm2d] = None, act: int = ACT_NONE,
               residual: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``act(bn(conv(x)) [+ residual])`` for nn.Conv2d (k 1 / 3 with stride 1 / 2, k 3 / 5 / 7 with <= 4 input channels) and
    nn.ConvTranspose2d (k 4, stride 2, padding 1) with eval-mode BatchNorm folded, on the HIP kernels."""
    x = _dev_f32(x, "x")
    if bn is not None and bn.training:
        raise _lib.DiffuVolumeError("BatchNorm2d in training mode: the HIP front folds running statistics (model.eval())")
    tensors = [conv.weight, conv.bias] + (list(_bn_tuple(bn)) if bn is not None else [])
    key = (act, tuple((t.data_ptr(), t._version) for t in tensors if t is not None))
    hit = _PLAN_CACHE.get(conv)
    if hit is None or hit[0] != key:
        w = conv.weight
        if isinstance(conv, nn.ConvTranspose2d):
            if conv.kernel_size != (4, 4) or conv.stride != (2, 2) or conv.padding != (1, 1):
                raise _lib.DiffuVolumeError("ConvTranspose2d on the HIP front: kernel 4, stride 2, padding 1")
            plan = Deconv2dK4S2Plan(w, _bn_tuple(bn), bias=conv.bias, act=act, eps=bn.eps if bn is not None else 1e-5)
        else:
            k, st = conv.kernel_size[0], conv.stride[0]
            if (conv.kernel_size != (k, k) or conv.stride != (st, st) or conv.padding != (k // 2, k // 2)
                    or conv.dilation != (1, 1) or conv.groups != 1 or st not in (1, 2)):
                raise _lib.DiffuVolumeError(f"Conv2d on the HIP front: square kernel, padding k/2, stride 1 or 2, got {conv}")
            if w.shape[1] <= 4 and k in (3, 5, 7):
                plan = _FewInPlan(w, conv.bias, bn, st, act)
            elif k in (1, 3):
                plan = Conv2dPlan(w, _bn_tuple(bn), act=act, bias=conv.bias, stride=st, eps=bn.eps if bn is not None else 1e-5)
            else:
                raise _lib.DiffuVolumeError(f"Conv2d on the HIP front: unsupported layer {conv}")
        _PLAN_CACHE[conv] = hit = (key, plan)
    plan = hit[1]
    if residual is not None:
        if isinstance(plan, Conv2dPlan):
            return plan(x, residual=residual)
        raise _lib.DiffuVolumeError("residual: 3x3 / 1x1 Conv2d layers only")
    return plan(x)


class _FewInPlan:
    """nn.Conv2d with <= 4 input channels (the RGB stems, the 7x7 stride-2 stem of the context encoder) [+ eval BatchNorm]
    [+ activation] on `dv_conv2d_fewin_f32`."""

    def __init__(self, w, bias, bn, stride, act):
        self.w = w.detach().float().contiguous()
        self.bias = None if bias is None else bias.detach().float().contiguous()
        self.cout, self.cin, self.k = w.shape[0], w.shape[1], w.shape[2]
        self.stride, self.act = stride, act
        self.scale = self.shift = None
        if bn is not None:
            sc = (bn.weight.detach().double() / torch.sqrt(bn.running_var.detach().double() + bn.eps))
            self.scale = sc.float().contiguous()
            self.shift = (bn.bias.detach().double() - bn.running_mean.detach().double() * sc).float().contiguous()

    def __call__(self, x):
        b, c, h, w = x.shape
        if c != self.cin:
            raise RuntimeError(f"expected {self.cin} input channels, got {c}")
        out = torch.empty((b, self.cout, (h - 1) // self.stride + 1, (w - 1) // self.stride + 1), dtype=torch.float32,
                          device=x.device)
        with torch.cuda.device(x.device):
            _lib.check(_lib.load().dv_conv2d_fewin_f32(x.data_ptr(), self.w.data_ptr(), _lib.ptr(self.bias),
                                                       _lib.ptr(self.scale), _lib.ptr(self.shift), out.data_ptr(), b, c, h, w,
                                                       self.cout, self.k, self.stride, self.act, _lib.stream_ptr()),
                       "dv_conv2d_fewin_f32")
        return out


def instance_norm_act(x: torch.Tensor, act: int = ACT_NONE, eps: float = 1e-5, inplace: bool = True) -> torch.Tensor:
    """nn.InstanceNorm2d (affine=False) + activation: `dv_instance_norm_act_f32`, one block per (b, c) plane."""
    x = _dev_f32(x, "x")
    b, c, h, w = x.shape
    out = x if inplace else torch.empty_like(x)
    with torch.cuda.device(x.device):
        _lib.check(_lib.load().dv_instance_norm_act_f32(x.data_ptr(), out.data_ptr(), b * c, h * w, float(eps), act,
                                                        _lib.stream_ptr()), "dv_instance_norm_act_f32")
    return out




def hip_sequential(seq, x: torch.Tensor) -> torch.Tensor:
    """An nn.Sequential of the front (stems, spx heads, stub backbone stages, FeatureAtt's gate) with every
    [conv][BatchNorm2d | InstanceNorm2d][ReLU | ReLU6 | LeakyReLU(0.01)] run fused on the HIP kernels; members that have
    a HIP forward of their own (BasicConv, BasicConv_IN, ResidualBlock, Conv2x...) are called."""
    mods = list(seq) if isinstance(seq, (nn.Sequential, list, tuple)) else [seq]
    i = 0
    while i < len(mods):
        m = mods[i]
        if isinstance(m, (nn.Conv2d, nn.ConvTranspose2d)):
            j, bn, inorm, act, clamp6 = i + 1, None, None, ACT_NONE, False
            if j < len(mods) and isinstance(mods[j], nn.BatchNorm2d):
                bn, j = mods[j], j + 1
            elif j < len(mods) and isinstance(mods[j], nn.InstanceNorm2d):
                inorm, j = mods[j], j + 1
                if inorm.affine or inorm.track_running_stats:
                    raise _lib.DiffuVolumeError("InstanceNorm2d on the HIP front: affine=False, no running statistics")
            if j < len(mods) and isinstance(mods[j], (nn.ReLU, nn.ReLU6, nn.LeakyReLU)):
                a = mods[j]
                if isinstance(a, nn.LeakyReLU) and abs(a.negative_slope - 0.01) > 1e-12:
                    raise _lib.DiffuVolumeError("LeakyReLU on the HIP front: negative_slope 0.01")
                act = ACT_LEAKY if isinstance(a, nn.LeakyReLU) else ACT_RELU            # (isinstance: subclasses of the three too)
                clamp6, j = isinstance(a, nn.ReLU6), j + 1
            if inorm is not None:
                x = instance_norm_act(hip_conv2d(m, x), act, inorm.eps)
            else:
                x = hip_conv2d(m, x, bn, act)
            if clamp6:
                x = x.clamp_(max=6.0)
            i = j
        elif isinstance(m, nn.Sequential):
            x = hip_sequential(m, x)
            i += 1
        elif isinstance(m, (nn.Identity, nn.Dropout, nn.Dropout2d)):
            i += 1
        else:
            x = m(x)
            i += 1
    return x


class BasicConv(nn.Module):
    """core/submodule.py:9-35: conv (bias=False) [+ BatchNorm] [+ LeakyReLU(0.01)].  The 3-D flavours run as
    fused HIP plans (``plan()``); ``forward`` is the 2-D flavour used inside FeatureAtt."""

    def __init__(self, in_channels, out_channels, deconv=False, is_3d=False, bn=True, relu=True, **kwargs):
        super().__init__()
        self.relu, self.use_bn, self.is_3d, self.deconv = relu, bn, is_3d, deconv
        if is_3d:
            self.conv = (nn.ConvTranspose3d if deconv else nn.Conv3d)(in_channels, out_channels, bias=False, **kwargs)
            self.bn = nn.BatchNorm3d(out_channels)
        else:
            self.conv = (nn.ConvTranspose2d if deconv else nn.Conv2d)(in_channels, out_channels, bias=False, **kwargs)
            self.bn = nn.BatchNorm2d(out_channels)

    def plan(self):
        if not self.is_3d:
            raise _lib.DiffuVolumeError("only the 3-D BasicConv flavours have HIP plans")
        bn = (self.bn.weight, self.bn.bias, self.bn.running_mean, self.bn.running_var) if self.use_bn else None
        act = ACT_LEAKY if self.relu else ACT_NONE
        if self.deconv:
            return Deconv3dPlan(self.conv.weight, bn, act=act, eps=self.bn.eps)
        return Conv3dPlan(self.conv.weight, bn, stride=self.conv.stride[0], act=act, eps=self.bn.eps)

    def forward(self, x):
        if self.is_3d:
            raise _lib.DiffuVolumeError("3-D BasicConv runs through its HIP plan, not nn.Module.forward")
        if _wants_autograd(x):
            x = self.conv(x)
            if self.use_bn:
                x = self.bn(x)
            return F.leaky_relu(x, 0.01) if self.relu else x
        return hip_conv2d(self.conv, x, self.bn if self.use_bn else None, ACT_LEAKY if self.relu else ACT_NONE)


class FeatureAtt(nn.Module):
    """core/submodule.py:226-239: image-feature guided channel gate of a cost volume."""

    def __init__(self, cv_chan, feat_chan):
        super().__init__()
        self.feat_att = nn.Sequential(BasicConv(feat_chan, feat_chan // 2, kernel_size=1, stride=1, padding=0),
                                      nn.Conv2d(feat_chan // 2, cv_chan, 1))

    def forward(self, cv, feat, inplace=False):
        return feature_gate(cv, hip_sequential(self.feat_att, feat), inplace=inplace)


def _seq_plans(seq):
    return [m.plan() for m in seq]


def _run(plans, x):
    for p in plans:
        x = p(x)
    return x


class hourglass(nn.Module):
    """igev_stereo_ddim.py:24-91 (`hourglass(8)`, runs once per pair on the gated gwc volume)."""

    def __init__(self, in_channels):
        super().__init__()
        c = in_channels
        k3 = dict(is_3d=True, bn=True, relu=True, kernel_size=3, padding=1, dilation=1)
        self.conv1 = nn.Sequential(BasicConv(c, c * 2, stride=2, **k3), BasicConv(c * 2, c * 2, stride=1, **k3))
        self.conv2 = nn.Sequential(BasicConv(c * 2, c * 4, stride=2, **k3), BasicConv(c * 4, c * 4, stride=1, **k3))
        self.conv3 = nn.Sequential(BasicConv(c * 4, c * 6, stride=2, **k3), BasicConv(c * 6, c * 6, stride=1, **k3))
        up = dict(deconv=True, is_3d=True, kernel_size=(4, 4, 4), padding=(1, 1, 1), stride=(2, 2, 2))
        self.conv3_up = BasicConv(c * 6, c * 4, bn=True, relu=True, **up)
        self.conv2_up = BasicConv(c * 4, c * 2, bn=True, relu=True, **up)
        self.conv1_up = BasicConv(c * 2, 8, bn=False, relu=False, **up)

        def agg(cin, cout):
            return nn.Sequential(BasicConv(cin, cout, is_3d=True, kernel_size=1, padding=0, stride=1),
                                 BasicConv(cout, cout, is_3d=True, kernel_size=3, padding=1, stride=1),
                                 BasicConv(cout, cout, is_3d=True, kernel_size=3, padding=1, stride=1))

        self.agg_0 = agg(c * 8, c * 4)
        self.agg_1 = agg(c * 4, c * 2)
        self.feature_att_8 = FeatureAtt(c * 2, 64)
        self.feature_att_16 = FeatureAtt(c * 4, 192)
        self.feature_att_32 = FeatureAtt(c * 6, 160)
        self.feature_att_up_16 = FeatureAtt(c * 4, 192)
        self.feature_att_up_8 = FeatureAtt(c * 2, 64)
        self._plans = None

    def prepare(self):
        if self._plans is None:
            self._plans = {n: _seq_plans(getattr(self, n)) for n in ("conv1", "conv2", "conv3", "agg_0", "agg_1")}
            for n in ("conv3_up", "conv2_up", "conv1_up"):
                self._plans[n] = getattr(self, n).plan()
        return self._plans

    def _apply(self, fn, *a, **k):
        self._plans = None
        return super()._apply(fn, *a, **k)

    def _load_from_state_dict(self, *a, **k):      # reached also when a parent / wrapper loads the checkpoint
        self._plans = None
        return super()._load_from_state_dict(*a, **k)

    def _replicate_for_data_parallel(self):        # nn.DataParallel replicas fold / pack their own weights
        replica = super()._replicate_for_data_parallel()
        replica._plans = None
        return replica

    def forward(self, x, features):
        p = self.prepare()
        conv1 = self.feature_att_8(_run(p["conv1"], x), features[1], inplace=True)
        conv2 = self.feature_att_16(_run(p["conv2"], conv1), features[2], inplace=True)
        conv3 = self.feature_att_32(_run(p["conv3"], conv2), features[3], inplace=True)
        conv2 = _run(p["agg_0"], torch.cat((p["conv3_up"](conv3), conv2), dim=1))
        conv2 = self.feature_att_up_16(conv2, features[2], inplace=True)
        conv1 = _run(p["agg_1"], torch.cat((p["conv2_up"](conv2), conv1), dim=1))
        conv1 = self.feature_att_up_8(conv1, features[1], inplace=True)
        return p["conv1_up"](conv1)


class IGEVCostVolume(nn.Module):
    """The volume-side modules of IGEVStereo_ddim (:196-199) and the part of its forward that uses them
    (:377-386).  ``forward(match_left, match_right, features_left)`` returns the geometry encoding volume
    [B,8,D/4,h,w] (what Combined_Geo_Encoding_Volume filters at every GRU iteration) and `init_disp`
    [B,1,h,w]."""

    def __init__(self, max_disp: int = 192):
        super().__init__()
        self.max_disp = max_disp
        self.corr_stem = BasicConv(8, 8, is_3d=True, kernel_size=3, stride=1, padding=1)
        self.corr_feature_att = FeatureAtt(8, 96)
        self.cost_agg = hourglass(8)
        self.classifier = nn.Conv3d(8, 1, 3, 1, 1, bias=False)
        self._plans = None

    def _apply(self, fn, *a, **k):
        self._plans = None
        return super()._apply(fn, *a, **k)

    def _load_from_state_dict(self, *a, **k):      # reached also when a parent / wrapper loads the checkpoint
        self._plans = None
        return super()._load_from_state_dict(*a, **k)

    def _replicate_for_data_parallel(self):        # nn.DataParallel replicas fold / pack their own weights
        replica = super()._replicate_for_data_parallel()
        replica._plans = None
        return replica

    def prepare(self):
        if self._plans is None:
            self._plans = (self.corr_stem.plan(), Conv3dPlan(self.classifier.weight, None, stride=1, act=ACT_NONE))
        return self._plans

    def forward(self, match_left, match_right, features_left):
        stem, classifier = self.prepare()
        d4 = self.max_disp // 4
        gwc = stem(build_gwc_volume(match_left, match_right, d4, 8))
        gwc = self.corr_feature_att(gwc, features_left[0], inplace=True)
        geo = self.cost_agg(gwc, features_left)
        cost = classifier(geo)
        return geo, softmax_regress(cost).unsqueeze(1)          # F.softmax + disparity_regression :382-383


# ---------------------------------------------------------------------------------------------------
# IGEVStereo_ddim: the drop-in module (KITTI15/core/igev_stereo_ddim.py:118-224 constructor, :361-427 eval forward).
# Module / buffer names are the reference's, so its checkpoints load with strict=True.  On HIP: gwc volume, corr_stem,
# FeatureAtt gates, hourglass(8), classifier + softmax + regression, the filtered geometry lookup, the whole update
# block, the convex upsampling (softmax + 9-tap gather) and the DDIM state update.  PyTorch (2-D, once per pair or
# once per DDIM step): the MobileNetV2 feature pyramid, the context encoder, the stems and the spx heads' convolutions.
# ---------------------------------------------------------------------------------------------------
def context_upsample(disp_low: torch.Tensor, up_weights: torch.Tensor, scale: float = 1.0,
                     apply_softmax: bool = False) -> torch.Tensor:
    """core/submodule.py:241-253: disp_low [B,1,h,w], up_weights [B,9,4h,4w] -> [B,4h,4w].  ``apply_softmax`` /
    ``scale`` fold the ``F.softmax(spx_pred, 1)`` and ``disp*4.`` of the call site into the same pass."""
    disp_low, up_weights = _dev_f32(disp_low, "disp_low"), _dev_f32(up_weights, "up_weights")
    b, c, h, w = disp_low.shape
    if c != 1 or tuple(up_weights.shape) != (b, 9, 4 * h, 4 * w):
        raise RuntimeError(f"context_upsample: disp_low [B,1,h,w] and up_weights [B,9,4h,4w], got "
                           f"{tuple(disp_low.shape)} and {tuple(up_weights.shape)}")
    out = torch.empty((b, 4 * h, 4 * w), dtype=torch.float32, device=disp_low.device)
    with torch.cuda.device(disp_low.device):
        _lib.check(_lib.load().dv_context_upsample_f32(disp_low.data_ptr(), up_weights.data_ptr(), out.data_ptr(), b, h,
                                                       w, float(scale), int(bool(apply_softmax)), _lib.stream_ptr()),
                   "dv_context_upsample_f32")
    return out


class BasicConv_IN(nn.Module):
    """core/submodule.py:79-107 (2-D flavours only): conv (bias=False) [+ InstanceNorm2d] [+ LeakyReLU(0.01)]."""

    def __init__(self, in_channels, out_channels, deconv=False, is_3d=False, IN=True, relu=True, **kwargs):
        super().__init__()
        if is_3d:
            raise _lib.DiffuVolumeError("IGEV uses BasicConv_IN in 2-D only")
        self.relu, self.use_in = relu, IN
        self.conv = (nn.ConvTranspose2d if deconv else nn.Conv2d)(in_channels, out_channels, bias=False, **kwargs)
        self.IN = nn.InstanceNorm2d(out_channels)

    def forward(self, x):
        if _wants_autograd(x):
            x = self.conv(x)
            if self.use_in:
                x = self.IN(x)
            return F.leaky_relu(x, 0.01) if self.relu else x
        act = ACT_LEAKY if self.relu else ACT_NONE
        if self.use_in:
            return instance_norm_act(hip_conv2d(self.conv, x), act, self.IN.eps)
        return hip_conv2d(self.conv, x, None, act)


class _Conv2xBase(nn.Module):
    """core/submodule.py:36-76 / :110-150: stride-2 (de)convolution, resize to the skip tensor, concat (or add), 3x3."""

    def _finish(self, x, rem):
        x = self.conv1(x)
        if x.shape != rem.shape:
            x = F.interpolate(x, size=(rem.shape[-2], rem.shape[-1]), mode="nearest")
        x = torch.cat((x, rem), 1) if self.concat else x + rem
        return self.conv2(x)

    def forward(self, x, rem):
        return self._finish(x, rem)


class Conv2x(_Conv2xBase):
    def __init__(self, in_channels, out_channels, deconv=False, is_3d=False, concat=True, keep_concat=True, bn=True,
                 relu=True, keep_dispc=False):
        super().__init__()
        if is_3d or keep_dispc:
            raise _lib.DiffuVolumeError("IGEV uses Conv2x in 2-D only")
        self.concat = concat
        self.conv1 = BasicConv(in_channels, out_channels, deconv, False, bn=True, relu=True,
                               kernel_size=4 if deconv else 3, stride=2, padding=1)
        cin, cout = (out_channels * 2, out_channels * (2 if keep_concat else 1)) if concat else (out_channels, out_channels)
        self.conv2 = BasicConv(cin, cout, False, False, bn, relu, kernel_size=3, stride=1, padding=1)


class Conv2x_IN(_Conv2xBase):
    def __init__(self, in_channels, out_channels, deconv=False, is_3d=False, concat=True, keep_concat=True, IN=True,
                 relu=True, keep_dispc=False):
        super().__init__()
        if is_3d or keep_dispc:
            raise _lib.DiffuVolumeError("IGEV uses Conv2x_IN in 2-D only")
        self.concat = concat
        self.conv1 = BasicConv_IN(in_channels, out_channels, deconv, False, IN=True, relu=True,
                                  kernel_size=4 if deconv else 3, stride=2, padding=1)
        cin, cout = (out_channels * 2, out_channels * (2 if keep_concat else 1)) if concat else (out_channels, out_channels)
        self.conv2 = BasicConv_IN(cin, cout, False, False, IN, relu, kernel_size=3, stride=1, padding=1)


class ResidualBlock(nn.Module):
    """core/extractor.py:10-74 with norm_fn='batch' (what MultiBasicEncoder is built with, :143).  `downsample`
    holds `norm3` a second time, so both key sets exist in the state_dict, as in the reference."""

    def __init__(self, in_planes, planes, norm_fn="batch", stride=1):
        super().__init__()
        if norm_fn != "batch":
            raise _lib.DiffuVolumeError("the context encoder is built with norm_fn='batch'")
        self.conv1 = nn.Conv2d(in_planes, planes, kernel_size=3, padding=1, stride=stride)
        self.conv2 = nn.Conv2d(planes, planes, kernel_size=3, padding=1)
        self.relu = nn.ReLU(inplace=True)
        self.norm1, self.norm2 = nn.BatchNorm2d(planes), nn.BatchNorm2d(planes)
        self.downsample = None
        if not (stride == 1 and in_planes == planes):
            self.norm3 = nn.BatchNorm2d(planes)
            self.downsample = nn.Sequential(nn.Conv2d(in_planes, planes, kernel_size=1, stride=stride), self.norm3)

    def forward(self, x):
        if _wants_autograd(x):
            y = self.relu(self.norm1(self.conv1(x)))
            y = self.relu(self.norm2(self.conv2(y)))
            if self.downsample is not None:
                x = self.downsample(x)
            return self.relu(x + y)
        y = hip_conv2d(self.conv1, x, self.norm1, ACT_RELU)
        y = hip_conv2d(self.conv2, y, self.norm2, ACT_RELU)
        if self.downsample is not None:
            x = hip_conv2d(self.downsample[0], x, self.downsample[1], ACT_NONE)
        return torch.relu_(y.add_(x))


class MultiBasicEncoder(nn.Module):
    """core/extractor.py:190-295: the context encoder (`cnet`).  2-D, once per pair: PyTorch."""

    def __init__(self, output_dim=((128, 128, 128),), norm_fn="batch", dropout=0.0, downsample=3):
        super().__init__()
        self.norm_fn, self.downsample = norm_fn, downsample
        self.norm1 = nn.BatchNorm2d(64)
        self.conv1 = nn.Conv2d(3, 64, kernel_size=7, stride=1 + (downsample > 2), padding=3)
        self.relu1 = nn.ReLU(inplace=True)
        self.in_planes = 64
        self.layer1 = self._make_layer(64, stride=1)
        self.layer2 = self._make_layer(96, stride=1 + (downsample > 1))
        self.layer3 = self._make_layer(128, stride=1 + (downsample > 0))
        self.layer4 = self._make_layer(128, stride=2)
        self.layer5 = self._make_layer(128, stride=2)
        self.outputs04 = nn.ModuleList([nn.Sequential(ResidualBlock(128, 128, norm_fn, 1), nn.Conv2d(128, d[2], 3, padding=1))
                                        for d in output_dim])
        self.outputs08 = nn.ModuleList([nn.Sequential(ResidualBlock(128, 128, norm_fn, 1), nn.Conv2d(128, d[1], 3, padding=1))
                                        for d in output_dim])
        self.outputs16 = nn.ModuleList([nn.Conv2d(128, d[0], 3, padding=1) for d in output_dim])
        self.dropout = nn.Dropout2d(p=dropout) if dropout > 0 else None
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)

    def _make_layer(self, dim, stride=1):
        layers = (ResidualBlock(self.in_planes, dim, self.norm_fn, stride=stride), ResidualBlock(dim, dim, self.norm_fn, 1))
        self.in_planes = dim
        return nn.Sequential(*layers)

    def forward(self, x, dual_inp=False, num_layers=3):
        hip = not _wants_autograd(x)
        head = (lambda f, t: hip_sequential(f, t)) if hip else (lambda f, t: f(t))
        stem = hip_conv2d(self.conv1, x, self.norm1, ACT_RELU) if hip else self.relu1(self.norm1(self.conv1(x)))
        x = self.layer3(self.layer2(self.layer1(stem)))
        v = None
        if dual_inp:
            v, x = x, x[:(x.shape[0] // 2)]
        tail = (v,) if dual_inp else ()
        outs = ([head(f, x) for f in self.outputs04],)
        if num_layers >= 2:
            y = self.layer4(x)
            outs += ([head(f, y) for f in self.outputs08],)
        if num_layers >= 3:
            z = self.layer5(y)
            outs += ([head(f, z) for f in self.outputs16],)
        return outs + tail


class Feature(nn.Module):
    """core/extractor.py:327-361.  The reference takes its stem and blocks from
    ``timm.create_model('mobilenetv2_100', pretrained=True, features_only=True)``; neither timm nor the weights
    exist offline, so the backbone object is injected: anything with ``conv_stem``, ``bn1``, ``act1`` and ``blocks``
    (7 stages with 16/24/32/64/96/160/320 output channels) -- a timm MobileNetV2 or ``synth.StubMobileNetV2``."""

    def __init__(self, backbone):
        super().__init__()
        chans = [16, 24, 32, 96, 160]
        cut = [1, 2, 3, 5, 6]
        self.conv_stem, self.bn1, self.act1 = backbone.conv_stem, backbone.bn1, backbone.act1
        blocks = list(backbone.blocks)
        self.block0 = nn.Sequential(*blocks[0:cut[0]])
        self.block1 = nn.Sequential(*blocks[cut[0]:cut[1]])
        self.block2 = nn.Sequential(*blocks[cut[1]:cut[2]])
        self.block3 = nn.Sequential(*blocks[cut[2]:cut[3]])
        self.block4 = nn.Sequential(*blocks[cut[3]:cut[4]])
        self.deconv32_16 = Conv2x_IN(chans[4], chans[3], deconv=True, concat=True)
        self.deconv16_8 = Conv2x_IN(chans[3] * 2, chans[2], deconv=True, concat=True)
        self.deconv8_4 = Conv2x_IN(chans[2] * 2, chans[1], deconv=True, concat=True)
        self.conv4 = BasicConv_IN(chans[1] * 2, chans[1] * 2, kernel_size=3, stride=1, padding=1)

    def _backbone_on_hip(self) -> bool:
        def plain(m):
            if isinstance(m, nn.Sequential):
                return all(plain(c) for c in m)
            if isinstance(m, nn.Conv2d):
                k = m.kernel_size[0]
                return (m.groups == 1 and m.dilation == (1, 1) and m.kernel_size == (k, k) and k in (1, 3)
                        and m.padding == (k // 2, k // 2) and m.stride[0] in (1, 2) and m.stride[0] == m.stride[1])
            return isinstance(m, (nn.BatchNorm2d, nn.ReLU, nn.ReLU6, nn.Identity))
        return all(plain(m) for m in (self.conv_stem, self.bn1, self.act1, self.block0, self.block1, self.block2,
                                      self.block3, self.block4))

    def forward(self, x):
        if self._backbone_on_hip() and not _wants_autograd(x):
            # a backbone made of plain [Conv2d, BatchNorm2d, ReLU / ReLU6] stages (synth.StubMobileNetV2) runs on the
            # in-tree kernels; anything else (timm's MobileNetV2: depth-wise / squeeze-excite blocks) is the injected
            # module's own business
            x2 = hip_sequential(self.block0, hip_sequential([self.conv_stem, self.bn1, self.act1], x))
            x4 = hip_sequential(self.block1, x2)
            x8 = hip_sequential(self.block2, x4)
            x16 = hip_sequential(self.block3, x8)
            x32 = hip_sequential(self.block4, x16)
        else:
            x2 = self.block0(self.act1(self.bn1(self.conv_stem(x))))
            x4 = self.block1(x2)
            x8 = self.block2(x4)
            x16 = self.block3(x8)
            x32 = self.block4(x16)
        x16 = self.deconv32_16(x32, x16)
        x8 = self.deconv16_8(x16, x8)
        x4 = self.conv4(self.deconv8_4(x8, x4))
        return [x4, x8, x16, x32]


class IGEVStereo_ddim(nn.Module):
    """``IGEVStereo_ddim(args).forward(image1, image2, flow_full, flow_gt, iters=12, flow_init=None, test_mode=False)
    -> (pred, pred)`` (eval path, igev_stereo_ddim.py:361-427).  ``args``: hidden_dims, n_gru_layers, n_downsample,
    corr_levels, corr_radius, slow_fast_gru, max_disp, mixed_precision (must be False: the HIP path is fp32).
    ``feature``: the MobileNetV2 feature pyramid (``Feature(backbone)``); None = the reference's own construction from
    timm's pretrained ``mobilenetv2_100`` (core/extractor.py:327-335), which needs ``timm`` to be importable;
    ``cnet``: optional replacement for the context encoder.  ``sampling_timesteps`` / ``ensemble_cof`` are
    hard-coded to 2 / [0.6, 0.1, 0.3] in the reference (:124, :353); BASELINE config 5 asks for 20 steps."""

    def __init__(self, args, feature: Optional[nn.Module] = None, cnet: Optional[nn.Module] = None,
                 sampling_timesteps: int = 2, ensemble_cof: Optional[Sequence[float]] = None):
        super().__init__()
        if feature is None:
            # the reference's own construction (core/extractor.py:327-335): timm's pretrained MobileNetV2, when timm exists
            try:
                import timm
            except ImportError:
                timm = None
            if timm is None or not hasattr(timm, "create_model"):
                raise _lib.DiffuVolumeError(
                    "IGEVStereo_ddim(args) builds its feature pyramid from timm.create_model('mobilenetv2_100', "
                    "pretrained=True, features_only=True) like the reference (core/extractor.py:331); timm is not "
                    "importable here -- pass feature=Feature(backbone) (a timm model or synth.StubMobileNetV2())")
            feature = Feature(timm.create_model("mobilenetv2_100", pretrained=True, features_only=True))
        if getattr(args, "mixed_precision", False):
            raise _lib.DiffuVolumeError("the HIP path computes in fp32: mixed_precision must be False")
        self.args = args
        self.scale = 1.0
        self.num_timesteps = 1000
        self.sampling_timesteps = sampling_timesteps
        self.is_ddim_sampling = sampling_timesteps < self.num_timesteps
        self.ddim_sampling_eta = 1
        self.renewal = self.use_ensemble = True
        if ensemble_cof is None:
            if sampling_timesteps != 2:
                raise ValueError("give ensemble_cof (S+1 weights) when sampling_timesteps != 2")
            ensemble_cof = (0.6, 0.1, 0.3)
        self.ensemble_cof = tuple(float(c) for c in ensemble_cof)
        betas = cosine_beta_schedule(self.num_timesteps)
        alphas = 1.0 - betas
        ac = torch.cumprod(alphas, dim=0)
        ac_prev = F.pad(ac[:-1], (1, 0), value=1.0)
        post_var = betas * (1.0 - ac_prev) / (1.0 - ac)
        for name, val in (("betas", betas), ("alphas_cumprod", ac), ("alphas_cumprod_prev", ac_prev),
                          ("sqrt_alphas_cumprod", torch.sqrt(ac)),
                          ("sqrt_one_minus_alphas_cumprod", torch.sqrt(1.0 - ac)),
                          ("log_one_minus_alphas_cumprod", torch.log(1.0 - ac)),
                          ("sqrt_recip_alphas_cumprod", torch.sqrt(1.0 / ac)),
                          ("sqrt_recipm1_alphas_cumprod", torch.sqrt(1.0 / ac - 1)),
                          ("posterior_variance", post_var),
                          ("posterior_log_variance_clipped", torch.log(post_var.clamp(min=1e-20))),
                          ("posterior_mean_coef1", betas * torch.sqrt(ac_prev) / (1.0 - ac)),
                          ("posterior_mean_coef2", (1.0 - ac_prev) * torch.sqrt(alphas) / (1.0 - ac))):
            self.register_buffer(name, val)

        from .update import BasicMultiUpdateBlock
        hidden = list(args.hidden_dims)
        self.cnet = cnet if cnet is not None else MultiBasicEncoder(output_dim=[hidden, hidden], norm_fn="batch",
                                                                    downsample=args.n_downsample)
        self.update_block = BasicMultiUpdateBlock(args, hidden_dims=hidden)
        self.context_zqr_convs = nn.ModuleList([nn.Conv2d(hidden[i], hidden[i] * 3, 3, padding=1)
                                                for i in range(args.n_gru_layers)])
        self.time_embedding = DynamicHead180(180)
        self.feature = feature

        def stem(cin, cout):
            return nn.Sequential(BasicConv_IN(cin, cout, kernel_size=3, stride=2, padding=1),
                                 nn.Conv2d(cout, cout, 3, 1, 1, bias=False), nn.InstanceNorm2d(cout), nn.ReLU())

        self.stem_2, self.stem_4 = stem(3, 32), stem(32, 48)
        self.spx = nn.Sequential(nn.ConvTranspose2d(2 * 32, 9, kernel_size=4, stride=2, padding=1))
        self.spx_2 = Conv2x_IN(24, 32, True)
        self.spx_4 = nn.Sequential(BasicConv_IN(96, 24, kernel_size=3, stride=1, padding=1),
                                   nn.Conv2d(24, 24, 3, 1, 1, bias=False), nn.InstanceNorm2d(24), nn.ReLU())
        self.spx_2_gru = Conv2x(32, 32, True)
        self.spx_gru = nn.Sequential(nn.ConvTranspose2d(2 * 32, 9, kernel_size=4, stride=2, padding=1))
        self.conv = BasicConv_IN(96, 96, kernel_size=3, padding=1, stride=1)
        self.desc = nn.Conv2d(96, 96, kernel_size=1, padding=0, stride=1)
        self.corr_stem = BasicConv(8, 8, is_3d=True, kernel_size=3, stride=1, padding=1)
        self.corr_feature_att = FeatureAtt(8, 96)
        self.cost_agg = hourglass(8)
        self.classifier = nn.Conv3d(8, 1, 3, 1, 1, bias=False)
        self._plans = None
        self._spx = None

    # ---- plan cache (same rules as the other wrappers) ----------------------------------------------
    def _apply(self, fn, *a, **k):
        self._plans = None
        return super()._apply(fn, *a, **k)

    def _load_from_state_dict(self, *a, **k):
        self._plans = None
        return super()._load_from_state_dict(*a, **k)

    def _replicate_for_data_parallel(self):
        r = super()._replicate_for_data_parallel()
        r._plans = None
        return r

    def prepare(self):
        if self._plans is None:
            self._plans = (self.corr_stem.plan(), Conv3dPlan(self.classifier.weight, None, stride=1, act=ACT_NONE))
        return self._plans

    def _spx_plans(self):
        """`spx_2_gru` (transposed conv + BN + LeakyReLU, concat with the 1/2-resolution stem, 3x3 conv + BN + LeakyReLU)
        and `spx_gru` (biased transposed conv to the 9 convex-upsampling logits) as HIP plans."""
        c1, c2, head = self.spx_2_gru.conv1, self.spx_2_gru.conv2, self.spx_gru[0]
        bn = lambda m: (m.bn.weight, m.bn.bias, m.bn.running_mean, m.bn.running_var) if m.use_bn else ()
        tensors = (c1.conv.weight, *bn(c1), c2.conv.weight, *bn(c2), head.weight, head.bias)
        key = tuple((t.data_ptr(), t._version) for t in tensors)        # children loaded / moved / overwritten in place
        if self._spx is None or self._spx[0] != key:
            act = lambda m: ACT_LEAKY if m.relu else ACT_NONE
            self._spx = (key, (Deconv2dK4S2Plan(c1.conv.weight, bn(c1) or None, act=act(c1), eps=c1.bn.eps),
                               Conv2dPlan(c2.conv.weight, bn(c2) or None, act=act(c2), eps=c2.bn.eps),
                               Deconv2dK4S2Plan(head.weight, None, bias=head.bias)))
        return self._spx[1]

    def freeze_bn(self):
        for m in self.modules():
            if isinstance(m, nn.BatchNorm2d):
                m.eval()

    # ---- pieces -----------------------------------------------------------------------------------------
    def upsample_disp(self, disp, mask_feat_4, stem_2x):
        """:209-217: spx_2_gru / spx_gru (3x3 HIP kernels; the transposed convolutions as four parity convolutions + pixel
        shuffle), then softmax over the 9 taps + context_upsample(disp*4) in one HIP pass.  Returns [B,1,4h,4w]."""
        if mask_feat_4.is_cuda and self.spx_2_gru.concat:
            up, mix, head = self._spx_plans()
            x = up(mask_feat_4)
            if x.shape != stem_2x.shape:
                x = F.interpolate(x, size=(stem_2x.shape[-2], stem_2x.shape[-1]), mode="nearest")
            spx_pred = head(mix([x, stem_2x]))              # torch.cat((x, rem), 1) is never materialised
        else:
            spx_pred = self.spx_gru(self.spx_2_gru(mask_feat_4, stem_2x))
        return context_upsample(disp, spx_pred, scale=4.0, apply_softmax=True).unsqueeze(1)

    def cost_volume(self, match_left, match_right, features_left):
        """:378-386: gwc (8 groups) -> corr_stem -> FeatureAtt -> hourglass(8) -> classifier -> softmax + regression."""
        stem, classifier = self.prepare()
        d4 = self.args.max_disp // 4
        gwc = stem(build_gwc_volume(match_left, match_right, d4, 8))
        gwc = self.corr_feature_att(gwc, features_left[0], inplace=True)
        geo = self.cost_agg(gwc, features_left)
        return geo, softmax_regress(classifier(geo)).unsqueeze(1)

    def _loop(self):
        return IGEVDiffusionLoop(self.time_embedding, self.update_block, self.upsample_disp,
                                 n_gru_layers=self.args.n_gru_layers, slow_fast_gru=self.args.slow_fast_gru,
                                 sampling_timesteps=self.sampling_timesteps, ensemble_cof=self.ensemble_cof)

    @torch.no_grad()
    def model_predictions(self, coords0, coords1, flow_init, iters, net_list, inp_list, corr_fn, noise, t, stem_2x):
        return self._loop().model_predictions(coords0, coords1, flow_init, iters, net_list, inp_list, corr_fn, noise,
                                              t, stem_2x)

    @torch.no_grad()
    def ddim_sample(self, coords0, coords1, flow_init, iters, net_list, inp_list, corr_fn, used, asd, stem_2x,
                    noise=None, generator=None):
        return self._loop().ddim_sample(coords0, coords1, flow_init, iters, net_list, inp_list, corr_fn, used, asd,
                                        stem_2x, noise=noise, generator=generator)

    @torch.no_grad()
    def encode_disparity(self, flow_gt):
        """:405-419: two-hot x_0 of the quarter-resolution origin disparity, clamped to [0, 47]."""
        dq = torch.clamp(_dev_f32(flow_gt, "flow_gt"), 0, 47).contiguous()
        b, h, w = dq.shape[0], dq.shape[-2], dq.shape[-1]
        x = torch.empty((b, 48, h, w), dtype=torch.float32, device=dq.device)
        with torch.cuda.device(dq.device):
            _lib.check(_lib.load().dv_encode_two_hot_f32(dq.data_ptr(), x.data_ptr(), b, 48, h * w, _lib.stream_ptr()),
                       "dv_encode_two_hot_f32")
        return x

    def _front(self, image1, image2):
        """:364-400 up to the GRU inputs: feature pyramid + stems, matching features, cost volume + initial disparity,
        context encoder, geometry lookup object.  Shared with the origin network (igev_stereo.py:151-194)."""
        image1 = (2 * (image1 / 255.0) - 1.0).contiguous()
        image2 = (2 * (image2 / 255.0) - 1.0).contiguous()
        features_left, features_right = self.feature(image1), self.feature(image2)
        stem_2x = hip_sequential(self.stem_2, image1)
        stem_4x = hip_sequential(self.stem_4, stem_2x)
        stem_4y = hip_sequential(self.stem_4, hip_sequential(self.stem_2, image2))
        features_left[0] = torch.cat((features_left[0], stem_4x), 1)
        features_right[0] = torch.cat((features_right[0], stem_4y), 1)
        match_left = hip_conv2d(self.desc, self.conv(features_left[0])).contiguous()
        match_right = hip_conv2d(self.desc, self.conv(features_right[0])).contiguous()
        geo, init_disp = self.cost_volume(match_left, match_right, features_left)
        cnet_list = self.cnet(image1, num_layers=self.args.n_gru_layers)
        net_list = [torch.tanh(x[0]) for x in cnet_list]
        inp_list = [torch.relu(x[1]) for x in cnet_list]
        inp_list = [list(hip_conv2d(conv, i).split(split_size=conv.out_channels // 3, dim=1))
                    for i, conv in zip(inp_list, self.context_zqr_convs)]
        inp_list = [[t.contiguous() for t in trio] for trio in inp_list]
        from .geometry_ddim import Combined_Geo_Encoding_Volume
        geo_fn = Combined_Geo_Encoding_Volume(match_left, match_right, geo, radius=self.args.corr_radius,
                                              num_levels=self.args.corr_levels)
        return features_left, stem_2x, init_disp, net_list, inp_list, geo_fn

    def forward(self, image1, image2, flow_full, flow_gt, iters=12, flow_init=None, test_mode=False, noise=None):
        if self.training:
            raise NotImplementedError("the MI355X DiffuVolume path is inference-only (model.eval())")
        with torch.no_grad():
            _, stem_2x, init_disp, net_list, inp_list, geo_fn = self._front(image1, image2)
            x0 = self.encode_disparity(flow_gt)
            pred = self.ddim_sample(init_disp, init_disp, flow_init, iters, net_list, inp_list, geo_fn, flow_full, x0,
                                    stem_2x, noise=noise)
        return pred, pred
