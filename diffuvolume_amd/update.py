"""IGEV's recurrent update block (KITTI15/core/update.py) behind the reference's module API: same classes, same
parameter names (a reference ``update_block`` state_dict loads unchanged), same ``forward`` signature and return
convention, so ``IGEVDiffusionLoop`` (or the reference's own ``ddim_sample``) can call it in place.

On HIP (csrc/conv2d.hip): every 3x3 / 1x1 convolution with its bias and ReLU / sigmoid / tanh, and the ConvGRU gate
arithmetic in the epilogues -- ``convr`` emits ``r*h`` directly, ``convq`` emits ``(1-z)*h + z*tanh(.)`` -- so a
ConvGRU is three launches; the convolutions read ``[h | x...]`` as a virtual concatenation (``dv_conv2d_cat_f32``).  The 7x7 single-channel ``convd1``, the 3x3 average
pooling and the bilinear interpolation between the three scales have their own small kernels (csrc/update_glue.hip); PyTorch
writes the disparity channel into the motion features' 128th channel (the reference's ``torch.cat``).
"""
from __future__ import annotations

import torch
import torch.nn.functional as F
from torch import nn

from . import _lib
from .submodule import ACT_NONE, ACT_RELU, ACT_SIGMOID, ACT_TANH, Conv2dPairPlan, Conv2dPlan


class _Planned(nn.Module):
    """Plans (packed weights on the device) are rebuilt after .to() / load_state_dict()."""

    def __init__(self):
        super().__init__()
        self._plans = None

    def _apply(self, fn, *a, **k):
        self._plans = None
        return super()._apply(fn, *a, **k)

    def _load_from_state_dict(self, *a, **k):      # runs for every sub-module, also when a parent is loaded
        self._plans = None
        return super()._load_from_state_dict(*a, **k)

    def _replicate_for_data_parallel(self):        # nn.DataParallel replicas fold / pack their own weights
        replica = super()._replicate_for_data_parallel()
        replica._plans = None
        return replica

    def plans(self):
        if self._plans is None:
            self._plans = self._build()
        return self._plans


def _plan(conv: nn.Conv2d, act: int) -> Conv2dPlan:
    return Conv2dPlan(conv.weight, None, dilation=1, act=act, bias=conv.bias)


class DispHead(_Planned):
    """update.py:15-24."""

    def __init__(self, input_dim=128, hidden_dim=256, output_dim=1):
        super().__init__()
        self.conv1 = nn.Conv2d(input_dim, hidden_dim, 3, padding=1)
        self.conv2 = nn.Conv2d(hidden_dim, output_dim, 3, padding=1)

    def _build(self):
        return _plan(self.conv1, ACT_RELU), _plan(self.conv2, ACT_NONE)

    def forward(self, x):
        c1, c2 = self.plans()
        return c2(c1(x))


class ConvGRU(_Planned):
    """update.py:26-40: z, r gates and candidate q from 3x3 convolutions of [h, x]; cz / cr / cq are the
    per-pair context terms added before the non-linearity."""

    def __init__(self, hidden_dim, input_dim, kernel_size=3):
        super().__init__()
        if kernel_size != 3:
            raise ValueError("ConvGRU is built with 3x3 convolutions in the reference")
        self.convz = nn.Conv2d(hidden_dim + input_dim, hidden_dim, kernel_size, padding=kernel_size // 2)
        self.convr = nn.Conv2d(hidden_dim + input_dim, hidden_dim, kernel_size, padding=kernel_size // 2)
        self.convq = nn.Conv2d(hidden_dim + input_dim, hidden_dim, kernel_size, padding=kernel_size // 2)

    def _build(self):
        # convz and convr read the same [h | x]: one launch with 2 * hidden output channels
        return (Conv2dPairPlan((self.convz.weight, self.convz.bias), (self.convr.weight, self.convr.bias), ACT_SIGMOID),
                _plan(self.convq, ACT_TANH))

    def forward(self, h, cz, cr, cq, *x_list):
        pzr, pq = self.plans()
        if len(x_list) > 3:                             # the kernel takes four sources: [h | x1 | x2 | x3]
            x_list = (torch.cat(x_list[:-2], dim=1),) + tuple(x_list[-2:])
        hx = [h, *x_list]                               # torch.cat([h, x]) is never materialised
        # z = sigmoid(convz(hx) + cz),  rh = sigmoid(convr(hx) + cr) * h
        z, rh = pzr(hx, residual=(cz, cr), mul=(None, h))
        return pq([rh, *x_list], residual=cq, blend=(z, h))                 # (1-z)*h + z*tanh(convq(.) + cq)


class BasicMotionEncoder(_Planned):
    """update.py:74-94."""

    def __init__(self, args):
        super().__init__()
        self.args = args
        cor_planes = args.corr_levels * (2 * args.corr_radius + 1) * (8 + 1)
        self.convc1 = nn.Conv2d(cor_planes, 64, 1, padding=0)
        self.convc2 = nn.Conv2d(64, 64, 3, padding=1)
        self.convd1 = nn.Conv2d(1, 64, 7, padding=3)
        self.convd2 = nn.Conv2d(64, 64, 3, padding=1)
        self.conv = nn.Conv2d(64 + 64, 128 - 1, 3, padding=1)

    def _build(self):
        p = {n: _plan(getattr(self, n), ACT_RELU) for n in ("convc1", "convc2", "convd2")}
        # convc1 fused into the geometry lookup that feeds it (csrc/geo_lookup.hip), when the caller hands over the lookup
        # as a request instead of a tensor (this build's IGEVDiffusionLoop does)
        p["convc1_lookup"] = None
        if self.convc1.out_channels == 64 and self.convc1.in_channels == 162:
            from .geometry_ddim import pack_lookup_conv1x1
            p["convc1_lookup"] = (pack_lookup_conv1x1(self.convc1.weight, 8),
                                  None if self.convc1.bias is None else self.convc1.bias.detach().float().contiguous())
        # `conv` with one all-zero output channel appended: the launch writes the [B,128,h,w] tensor the reference builds with
        # torch.cat([out, disp]) (update.py:94) and channel 127 (relu(0) = 0) is then overwritten with the disparity.  gru04
        # reads ONE 128-channel source instead of 127 + 1, so every source of its virtual concatenation is a whole number of
        # the kernel's 8-channel chunks (csrc/conv2d_wino.hip, SRC = 1: the source queue moves once per chunk).
        w, b = self.conv.weight, self.conv.bias
        p["conv"] = Conv2dPlan(torch.cat([w, w.new_zeros((1,) + tuple(w.shape[1:]))]), None, dilation=1, act=ACT_RELU,
                               bias=torch.cat([b, b.new_zeros(1)]))
        return p

    def forward(self, disp, corr):
        return self.features(disp, corr)                # update.py:94 (the reference's return value)

    def features(self, disp, corr):
        """The motion features [B,128,h,w] = [conv output (127) | disp (1)], update.py:88-94."""
        p = self.plans()
        from .geometry_ddim import GeoLookupRequest
        if isinstance(corr, GeoLookupRequest):
            if p["convc1_lookup"] is not None and corr.volume.channel == 8:
                cor = corr.conv1x1(p["convc1_lookup"][0], p["convc1_lookup"][1], ACT_RELU)
            else:
                cor = p["convc1"](corr.materialize())
        else:
            cor = p["convc1"](corr)
        cor = p["convc2"](cor)
        disp_ = p["convd2"](self._convd1(disp))
        out = p["conv"]([cor, disp_])                   # virtual concatenation: torch.cat([cor, disp_]) is never materialised
        out[:, -1:].copy_(disp)
        return out

    def _convd1(self, disp):
        """relu(convd1(disp)): the 7x7 single-input-channel convolution on its own VALU kernel (MIOpen picks a naive
        solver for this shape: 0.7 ms per call at batch 4)."""
        if not _hip_ok(disp, "convd1"):
            return F.relu(self.convd1(disp))                 # autograd dispatch (training graphs only)
        disp = disp.contiguous()
        b, _, h, w = disp.shape
        out = torch.empty((b, self.convd1.out_channels, h, w), dtype=torch.float32, device=disp.device)
        wt, bias = self.convd1.weight.contiguous(), self.convd1.bias
        with torch.cuda.device(disp.device):
            _lib.check(_lib.load().dv_conv2d_1in_f32(disp.data_ptr(), wt.data_ptr(), _lib.ptr(bias), out.data_ptr(), b, h, w,
                                                     out.shape[1], int(wt.shape[-1]), ACT_RELU, _lib.stream_ptr()),
                       "dv_conv2d_1in_f32")
        return out


def _hip_ok(x, what):
    """True: the HIP kernel runs.  False: the caller asked for gradients (grad enabled AND the tensor requires grad), the
    one case that is dispatched to the differentiable torch expression -- explicitly, like submodule.py's builders.
    Anything else (a CPU tensor, another dtype) RAISES: there is no silent eager / CPU fallback on the product path."""
    if torch.is_grad_enabled() and x.requires_grad:
        return False
    if not x.is_cuda:
        raise _lib.DiffuVolumeError(f"{what}: input is on {x.device}; the DiffuVolume hot path only runs on the MI355X "
                                    "(HIP kernels, no CPU fallback)")
    if x.dtype != torch.float32:
        raise TypeError(f"{what}: input must be float32, got {x.dtype}")
    return True


def pool2x(x):
    """update.py:96-97.  HIP (`dv_avg_pool3s2_f32`); the torch expression only when gradients are asked for."""
    if not _hip_ok(x, "pool2x"):
        return F.avg_pool2d(x, 3, stride=2, padding=1)     # autograd dispatch
    x = x.contiguous()
    b, c, h, w = x.shape
    out = torch.empty((b, c, (h - 1) // 2 + 1, (w - 1) // 2 + 1), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        _lib.check(_lib.load().dv_avg_pool3s2_f32(x.data_ptr(), out.data_ptr(), b * c, h, w, _lib.stream_ptr()),
                   "dv_avg_pool3s2_f32")
    return out


def pool4x(x):
    return F.avg_pool2d(x, 5, stride=4, padding=1)


def interp(x, dest):
    """update.py:100-102.  HIP (`dv_resize_bilinear_ac_f32`); the torch expression only when gradients are asked for."""
    if not _hip_ok(x, "interp"):
        return F.interpolate(x, dest.shape[2:], mode="bilinear", align_corners=True)     # autograd dispatch
    x = x.contiguous()
    b, c, h, w = x.shape
    H, W = int(dest.shape[2]), int(dest.shape[3])
    out = torch.empty((b, c, H, W), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        _lib.check(_lib.load().dv_resize_bilinear_ac_f32(x.data_ptr(), out.data_ptr(), b * c, h, w, H, W,
                                                         _lib.stream_ptr()), "dv_resize_bilinear_ac_f32")
    return out


class BasicMultiUpdateBlock(_Planned):
    """update.py:104-142."""

    def __init__(self, args, hidden_dims=()):
        super().__init__()
        self.args = args
        self.encoder = BasicMotionEncoder(args)
        encoder_output_dim = 128
        self.gru04 = ConvGRU(hidden_dims[2], encoder_output_dim + hidden_dims[1] * (args.n_gru_layers > 1))
        self.gru08 = ConvGRU(hidden_dims[1], hidden_dims[0] * (args.n_gru_layers == 3) + hidden_dims[2])
        self.gru16 = ConvGRU(hidden_dims[0], hidden_dims[1])
        self.disp_head = DispHead(hidden_dims[2], hidden_dim=256, output_dim=1)
        self.mask_feat_4 = nn.Sequential(nn.Conv2d(hidden_dims[2], 32, 3, padding=1), nn.ReLU(inplace=True))

    def _build(self):
        return _plan(self.mask_feat_4[0], ACT_RELU)

    import os as _os
    OVERLAP = _os.environ.get("DV_IGEV_OVERLAP", "1") != "0"
    _streams = None

    def _side_stream(self, device):
        if self._streams is None:
            object.__setattr__(self, "_streams", {})
        if device not in self._streams:
            self._streams[device] = torch.cuda.Stream(device=device)
        return self._streams[device]

    def forward(self, net, inp, corr=None, disp=None, iter04=True, iter08=True, iter16=True, update=True,
                mask=True):
        """The reference's call (update.py:119-142) plus `mask=False` to skip `mask_feat_4`, which the reference computes
        in every iteration and reads only after the last one (igev_stereo_ddim.py:255-259)."""
        if self.training:
            raise NotImplementedError("the MI355X update block is inference-only (model.eval())")
        with torch.no_grad():
            mf = None
            if self.OVERLAP and iter04 and iter08 and corr is not None and disp.is_cuda and \
                    not torch.cuda.is_current_stream_capturing():
                # The motion encoder (lookup + five convolutions) does not depend on gru16 / gru08: it runs on a side stream
                # beside them, filling their ramp-up / tail gaps and the half-empty 1/16-scale launches.  Same kernels on the
                # same inputs: bit-identical.  Round 5, batch 4, 1248x384, 20 x 32 iterations, same box, alternating:
                # 1 806-1 807 ms on one stream, 1 775 ms with the encoder on the side stream (-1.8 %; the same experiment was
                # 3.5 % SLOWER before the lookup was coalesced and the small launches K-split).  Measured and NOT kept: gru16 of
                # the next iteration on the side stream beside gru04 and the disparity head (it only needs net[2] and
                # pool2x(net[1]), final after gru08): 1 786-1 789 ms -- gru04 fills the chip, the extra launches only contend.
                # DV_IGEV_OVERLAP=0 puts everything back on one stream.
                main = torch.cuda.current_stream(disp.device)
                side = self._side_stream(disp.device)
                side.wait_stream(main)
                for t in (disp, *(getattr(corr, n, None) for n in ("disp", "coords", "noisy"))):
                    if isinstance(t, torch.Tensor):
                        t.record_stream(side)                    # (main-stream tensors the side stream reads)
                with torch.cuda.stream(side):
                    mf = self.encoder.features(disp, corr)
                mf.record_stream(main)
            if iter16:
                net[2] = self.gru16(net[2], *(inp[2]), pool2x(net[1]))
            if iter08:
                if self.args.n_gru_layers > 2:
                    net[1] = self.gru08(net[1], *(inp[1]), pool2x(net[0]), interp(net[2], net[1]))
                else:
                    net[1] = self.gru08(net[1], *(inp[1]), pool2x(net[0]))
            if iter04:
                if mf is None:
                    mf = self.encoder.features(disp, corr)           # [h | mf+disp | interp]: three sources, no cat
                else:
                    torch.cuda.current_stream(disp.device).wait_stream(self._side_stream(disp.device))
                if self.args.n_gru_layers > 1:
                    net[0] = self.gru04(net[0], *(inp[0]), mf, interp(net[1], net[0]))
                else:
                    net[0] = self.gru04(net[0], *(inp[0]), mf)
            if not update:
                return net
            delta_disp = self.disp_head(net[0])
            mask_feat_4 = self.plans()(net[0]) if mask else None
        return net, mask_feat_4, delta_disp
