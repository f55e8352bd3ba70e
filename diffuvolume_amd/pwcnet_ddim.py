"""PCWNet + DiffuVolume (KITTI12 flavour) behind the reference's module API.

Drop-in for ``PWCNet_ddim`` of KITTI12/models/pwcnet_ddim.py:335-758 (eval path): same
constructor, ``forward(left, right, used, disp, mask) -> ([disp_finetune], [pred3_volume])``,
``model_predictions`` and ``ddim_sample``; same parameter / buffer names (reference
``state_dict`` loads with ``strict=True``).

HIP (libdiffuvolume_hip.so): the four group-wise-correlation and concat volumes (KITTI12
zero-fill flavour), dres0/dres1, ``hourglassup`` and, per DDIM step, the volume filter, the three
Mish hourglasses, ``classif3``, the align_corners=True trilinear/softmax/regression tail with the
uncertainty taken about the *refined* disparity, the two-hot re-encoding and the DDIM update.
Also HIP: the per-step 2-D refinement network ``refinenet3`` (dilated 3x3 convbn +
Mish, BasicBlocks, 1x1 downsamples) on the 2-D implicit-GEMM kernel (csrc/conv2d.hip).
The multi-scale 2-D feature CNN runs on the same 2-D kernel; PyTorch keeps the bilinear feature upsampling and the
concatenations.
Differences from the ACV flavour: x_T = randn, 3 steps, fill = cumulative
q_sample(asd), thresholds dif<1 & unc<1, ensemble [0.9,0,0,0.1], Mish activations.
"""
from __future__ import annotations

import ctypes
import math
from typing import Callable, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F
from torch import nn

from . import _lib
from .acv_ddim import ProbVolumeHandle, _LoopStep, _bn_of, _plan_cb3, cosine_beta_schedule
from .head import DynamicHead
from .profiling import timed
from .submodule import (ACT_MISH, ACT_NONE, Conv2dPlan, Conv3dPlan, Deconv3dPlan, ReplicaPlanCache, _dev_f32, build_concat_volume,
                        build_gwc_volume, check_split_overflow, refine_inputs, upsample_softmax_regress)

NoiseFn = Callable[[str, Tuple[int, ...], torch.dtype], torch.Tensor]


class Mish(nn.Module):
    """x * tanh(softplus(x)) (KITTI12/models/submodule.py:11-18)."""

    def forward(self, x):
        return x * torch.tanh(F.softplus(x))


def _cb2(cin, cout, k, stride, pad, dil):
    return nn.Sequential(nn.Conv2d(cin, cout, k, stride, dil if dil > 1 else pad, dil, bias=False),
                         nn.BatchNorm2d(cout))


def _cb3(cin, cout, k, stride, pad):
    return nn.Sequential(nn.Conv3d(cin, cout, k, stride, pad, bias=False), nn.BatchNorm3d(cout))


class _Block2d(nn.Module):
    """BasicBlock with Mish (KITTI12/models/submodule.py:192-215)."""

    def __init__(self, cin, planes, stride, downsample, pad, dil):
        super().__init__()
        self.conv1 = nn.Sequential(_cb2(cin, planes, 3, stride, pad, dil), Mish())
        self.conv2 = _cb2(planes, planes, 3, 1, pad, dil)
        self.downsample = downsample

    def forward(self, x):
        y = self.conv2(self.conv1(x))
        return y + (x if self.downsample is None else self.downsample(x))


class _Stacker:
    inplanes: int

    def _stack(self, planes, blocks, stride, pad, dil):
        down = None
        if stride != 1 or self.inplanes != planes:
            down = nn.Sequential(nn.Conv2d(self.inplanes, planes, 1, stride, bias=False), nn.BatchNorm2d(planes))
        layers = [_Block2d(self.inplanes, planes, stride, down, pad, dil)]
        self.inplanes = planes
        layers += [_Block2d(planes, planes, 1, None, pad, dil) for _ in range(1, blocks)]
        return nn.Sequential(*layers)


def _head2d(cin, mid, cout):
    return nn.Sequential(_cb2(cin, mid, 3, 1, 1, 1), Mish(), nn.Conv2d(mid, cout, 1, bias=False))


class FeatureExtraction(ReplicaPlanCache, nn.Module, _Stacker):
    """Multi-scale 2-D feature CNN (pwcnet_ddim.py:12-128): gw1..gw4 at 1/4..1/32, concat features,
    refinement feature.  On the GPU (eval) every convolution runs on the 2-D HIP kernel."""

    def __init__(self, concat_feature=False, concat_feature_channel=12):
        super().__init__()
        self.concat_feature = concat_feature
        self.inplanes = 32
        self.firstconv = nn.Sequential(_cb2(3, 32, 3, 2, 1, 1), Mish(), _cb2(32, 32, 3, 1, 1, 1), Mish(),
                                       _cb2(32, 32, 3, 1, 1, 1), Mish())
        self.layer1 = self._stack(32, 3, 1, 1, 1)
        self.layer2 = self._stack(64, 16, 2, 1, 1)
        self.layer3 = self._stack(128, 3, 1, 1, 1)
        self.layer4 = self._stack(128, 3, 1, 1, 2)
        self.layer5 = self._stack(192, 3, 2, 1, 1)
        self.layer7 = self._stack(256, 3, 2, 1, 1)
        self.layer9 = self._stack(512, 3, 2, 1, 1)
        self.gw2 = _head2d(192, 320, 320)
        self.gw3 = _head2d(256, 320, 320)
        self.gw4 = _head2d(512, 320, 320)
        self.layer11 = _head2d(320, 320, 320)
        self.layer_refine = nn.Sequential(_cb2(320, 128, 3, 1, 1, 1), Mish(), _cb2(128, 32, 1, 1, 0, 1), Mish())
        if concat_feature:
            self.lastconv = _head2d(320, 128, concat_feature_channel)
            self.concat2 = _head2d(192, 128, concat_feature_channel)
            self.concat3 = _head2d(256, 128, concat_feature_channel)
            self.concat4 = _head2d(512, 128, concat_feature_channel)

    # ---- HIP plans (csrc/conv2d.hip: BN / Mish / residual fused), rebuilt when the parameters change ----
    _plans = None

    def _apply(self, fn, *a, **k):
        self._plans = None
        self._replica_clear()
        return super()._apply(fn, *a, **k)

    def _load_from_state_dict(self, *a, **k):
        self._plans = None
        self._replica_clear()
        return super()._load_from_state_dict(*a, **k)

    def _replicate_for_data_parallel(self):        # nn.DataParallel replicas: plans parked on the source, per device
        replica = super()._replicate_for_data_parallel()
        replica._plans = None
        return self._mark_replica(replica)

    def train(self, mode: bool = True):
        if mode != self.training:
            self._plans = None
            self._replica_clear()
        return super().train(mode)

    def prepare(self):
        version = sum(t._version for t in self.parameters()) + sum(t._version for t in self.buffers())
        if self._plans is not None and self._plans.get("version") != version:      # weights overwritten in place since
            self._plans = None
        if self._plans is None:
            dev = self.firstconv[0][0].weight.device
            self._plans = self._replica_lookup(dev)
        if self._plans is None:
            def head(seq):          # convbn + Mish + Conv2d 1x1
                return (_plan_cb2(seq[0], ACT_MISH), Conv2dPlan(seq[2].weight, None, act=ACT_NONE))
            with torch.no_grad():
                p = {"first": [_plan_cb2(self.firstconv[i], ACT_MISH) for i in (0, 2, 4)]}
                for n in ("layer1", "layer2", "layer3", "layer4", "layer5", "layer7", "layer9"):
                    p[n] = [_Block2dPlan(b) for b in getattr(self, n)]
                for n in ("gw2", "gw3", "gw4", "layer11") + (("lastconv", "concat2", "concat3", "concat4") if self.concat_feature else ()):
                    p[n] = head(getattr(self, n))
                p["refine"] = (_plan_cb2(self.layer_refine[0], ACT_MISH), _plan_cb2(self.layer_refine[2], ACT_MISH))
            p["version"] = version
            self._plans = p
            self._replica_store(dev, p)
        return self._plans

    def forward(self, x):
        if not x.is_cuda:
            raise _lib.DiffuVolumeError(f"input is on {x.device}: the feature CNN runs on the MI355X (no CPU fallback)")
        if self.training or (torch.is_grad_enabled() and x.requires_grad):
            return self._forward_modules(x)
        p = self.prepare()
        def run(plans, t):
            for q in plans:
                t = q(t)
            return t
        with torch.no_grad():
            for q in p["first"]:
                x = q(x)
            x = run(p["layer1"], x)
            l2 = run(p["layer2"], x)
            l3 = run(p["layer3"], l2)
            l4 = run(p["layer4"], l3)
            l5 = run(p["layer5"], l4)
            l6 = run(p["layer7"], l5)
            l7 = run(p["layer9"], l6)
            fc = torch.cat((l2, l3, l4), dim=1)
            out = {"gw1": run(p["layer11"], fc), "gw2": run(p["gw2"], l5), "gw3": run(p["gw3"], l6), "gw4": run(p["gw4"], l7)}
            if self.concat_feature:
                out.update(concat_feature1=run(p["lastconv"], fc), finetune_feature=run(p["refine"], fc),
                           concat_feature2=run(p["concat2"], l5), concat_feature3=run(p["concat3"], l6),
                           concat_feature4=run(p["concat4"], l7))
        return out

    def _forward_modules(self, x):
        """The same graph on the plain nn.Modules (training / autograd on the GPU; reference for the tests)."""
        x = self.layer1(self.firstconv(x))
        l2 = self.layer2(x)
        l3 = self.layer3(l2)
        l4 = self.layer4(l3)
        l5 = self.layer5(l4)
        l6 = self.layer7(l5)
        l7 = self.layer9(l6)
        fc = torch.cat((l2, l3, l4), dim=1)
        out = {"gw1": self.layer11(fc), "gw2": self.gw2(l5), "gw3": self.gw3(l6), "gw4": self.gw4(l7)}
        if self.concat_feature:
            out.update(concat_feature1=self.lastconv(fc), finetune_feature=self.layer_refine(fc),
                       concat_feature2=self.concat2(l5), concat_feature3=self.concat3(l6),
                       concat_feature4=self.concat4(l7))
        return out


class RefineNet(nn.Module, _Stacker):
    """refinenet_version3 (pwcnet_ddim.py:251-306): dilated 2-D residual stack -> disparity residual."""

    def __init__(self, in_channels):
        super().__init__()
        self.inplanes = 128
        self.conv1 = nn.Sequential(_cb2(in_channels, 128, 3, 1, 1, 1), Mish())
        self.conv2 = nn.Sequential(_cb2(128, 128, 3, 1, 1, 1), Mish())
        self.conv3 = nn.Sequential(_cb2(128, 128, 3, 1, 2, 2), Mish())
        self.conv4 = nn.Sequential(_cb2(128, 128, 3, 1, 4, 4), Mish())
        self.conv5 = self._stack(96, 1, 1, 1, 8)
        self.conv6 = self._stack(64, 1, 1, 1, 16)
        self.conv7 = self._stack(32, 1, 1, 1, 1)
        self.conv8 = nn.Conv2d(32, 1, 3, 1, 1, bias=False)

    def forward(self, x, disp):
        for m in (self.conv1, self.conv2, self.conv3, self.conv4, self.conv5, self.conv6, self.conv7, self.conv8):
            x = m(x)
        return disp + x


class HourglassUp(nn.Module):
    """Parameters of hourglassup (pwcnet_ddim.py:131-205): fuses the 1/8, 1/16, 1/32 volumes."""

    def __init__(self, c):
        super().__init__()
        self.conv1 = nn.Conv3d(c, 2 * c, 3, 2, 1, bias=False)
        self.conv2 = nn.Sequential(_cb3(2 * c, 2 * c, 3, 1, 1), Mish())
        self.conv3 = nn.Conv3d(2 * c, 4 * c, 3, 2, 1, bias=False)
        self.conv4 = nn.Sequential(_cb3(4 * c, 4 * c, 3, 1, 1), Mish())
        self.conv5 = nn.Conv3d(4 * c, 4 * c, 3, 2, 1, bias=False)
        self.conv6 = nn.Sequential(_cb3(4 * c, 4 * c, 3, 1, 1), Mish())
        self.conv7 = nn.Sequential(nn.ConvTranspose3d(4 * c, 4 * c, 3, padding=1, output_padding=1, stride=2, bias=False),
                                   nn.BatchNorm3d(4 * c))
        self.conv8 = nn.Sequential(nn.ConvTranspose3d(4 * c, 2 * c, 3, padding=1, output_padding=1, stride=2, bias=False),
                                   nn.BatchNorm3d(2 * c))
        self.conv9 = nn.Sequential(nn.ConvTranspose3d(2 * c, c, 3, padding=1, output_padding=1, stride=2, bias=False),
                                   nn.BatchNorm3d(c))
        self.combine1 = nn.Sequential(_cb3(4 * c, 2 * c, 3, 1, 1), Mish())
        self.combine2 = nn.Sequential(_cb3(6 * c, 4 * c, 3, 1, 1), Mish())
        self.combine3 = nn.Sequential(_cb3(6 * c, 4 * c, 3, 1, 1), Mish())
        self.redir1 = _cb3(c, c, 1, 1, 0)
        self.redir2 = _cb3(2 * c, 2 * c, 1, 1, 0)
        self.redir3 = _cb3(4 * c, 4 * c, 1, 1, 0)


class Hourglass(nn.Module):
    """Parameters of the Mish hourglass (pwcnet_ddim.py:208-248): no bottleneck attention."""

    def __init__(self, c):
        super().__init__()
        self.conv1 = nn.Sequential(_cb3(c, 2 * c, 3, 2, 1), Mish())
        self.conv2 = nn.Sequential(_cb3(2 * c, 2 * c, 3, 1, 1), Mish())
        self.conv3 = nn.Sequential(_cb3(2 * c, 4 * c, 3, 2, 1), Mish())
        self.conv4 = nn.Sequential(_cb3(4 * c, 4 * c, 3, 1, 1), Mish())
        self.conv5 = nn.Sequential(nn.ConvTranspose3d(4 * c, 2 * c, 3, padding=1, output_padding=1, stride=2, bias=False),
                                   nn.BatchNorm3d(2 * c))
        self.conv6 = nn.Sequential(nn.ConvTranspose3d(2 * c, c, 3, padding=1, output_padding=1, stride=2, bias=False),
                                   nn.BatchNorm3d(c))
        self.redir1 = _cb3(c, c, 1, 1, 0)
        self.redir2 = _cb3(2 * c, 2 * c, 1, 1, 0)


# ---- prepared hot-path layers ---------------------------------------------------------------------
def _deconv_plan(seq, act, redir=None):
    """ConvTranspose3d + BN [+ the 1x1x1 `redir` conv + BN of the skip tensor folded into the same launch]."""
    if redir is None:
        return Deconv3dPlan(seq[0].weight, _bn_of(seq[1]), act=act, eps=seq[1].eps)
    return Deconv3dPlan(seq[0].weight, _bn_of(seq[1]), act=act, eps=seq[1].eps,
                        redir=(redir[0].weight, _bn_of(redir[1])), redir_eps=redir[1].eps)


class _HourglassPlan:
    def __init__(self, hg: Hourglass):
        self.conv1 = _plan_cb3(hg.conv1[0], 2, ACT_MISH)
        self.conv2 = _plan_cb3(hg.conv2[0], 1, ACT_MISH)
        self.conv3 = _plan_cb3(hg.conv3[0], 2, ACT_MISH)
        self.conv4 = _plan_cb3(hg.conv4[0], 1, ACT_MISH)
        self.conv5 = _deconv_plan(hg.conv5, ACT_MISH, hg.redir2)
        self.conv6 = _deconv_plan(hg.conv6, ACT_MISH, hg.redir1)
        self.conv6_plain = _deconv_plan(hg.conv6, ACT_MISH)        # filtered skip (first hourglass of a step)
        self.redir1 = _plan_cb3(hg.redir1, 1, ACT_NONE)

    def __call__(self, x, in_scale=None):
        """``in_scale`` is the [0,1] volume filter: the reference feeds ``volume * noise`` to conv1 AND to
        redir1 (pwcnet_ddim.py:472-474, :245), so both take the prologue."""
        c1 = self.conv1(x, in_scale=in_scale)
        c2 = self.conv2(c1)
        c4 = self.conv4(self.conv3(c2))
        c5 = self.conv5(c4, skip=c2)                                        # FMish(deconv + redir2)
        if in_scale is None:
            return self.conv6(c5, skip=x)                                   # FMish(deconv + redir1)
        return self.conv6_plain(c5, residual=self.redir1(x, in_scale=in_scale))


class _HourglassUpPlan:
    def __init__(self, m: HourglassUp):
        self.conv1 = Conv3dPlan(m.conv1.weight, None, stride=2, act=ACT_NONE)
        self.conv3 = Conv3dPlan(m.conv3.weight, None, stride=2, act=ACT_NONE)
        self.conv5 = Conv3dPlan(m.conv5.weight, None, stride=2, act=ACT_NONE)
        self.conv2 = _plan_cb3(m.conv2[0], 1, ACT_MISH)
        self.conv4 = _plan_cb3(m.conv4[0], 1, ACT_MISH)
        self.conv6 = _plan_cb3(m.conv6[0], 1, ACT_MISH)
        self.combine1 = _plan_cb3(m.combine1[0], 1, ACT_MISH)
        self.combine2 = _plan_cb3(m.combine2[0], 1, ACT_MISH)
        self.combine3 = _plan_cb3(m.combine3[0], 1, ACT_MISH)
        self.conv7 = _deconv_plan(m.conv7, ACT_MISH, m.redir3)
        self.conv8 = _deconv_plan(m.conv8, ACT_MISH, m.redir2)
        self.conv9 = _deconv_plan(m.conv9, ACT_MISH, m.redir1)

    def __call__(self, x, f4, f5, f6):
        c1 = self.combine1(torch.cat((self.conv1(x), f4), dim=1))
        c2 = self.conv2(c1)
        c3 = self.combine2(torch.cat((self.conv3(c2), f5), dim=1))
        c4 = self.conv4(c3)
        c5 = self.combine3(torch.cat((self.conv5(c4), f6), dim=1))
        c6 = self.conv6(c5)
        c7 = self.conv7(c6, skip=c4)
        c8 = self.conv8(c7, skip=c2)
        return self.conv9(c8, skip=x)


class _PairPlan:
    def __init__(self, seq, act_last):
        self.a = _plan_cb3(seq[0], 1, ACT_MISH)
        last = seq[2]
        self.b = _plan_cb3(last, 1, act_last) if isinstance(last, nn.Sequential) else \
            Conv3dPlan(last.weight, None, stride=1, act=ACT_NONE)

    def __call__(self, x, residual_self=False):
        return self.b(self.a(x), residual=x if residual_self else None)


def _plan_cb2(seq, act, dil=None):
    """convbn 2-D (Conv2d + BatchNorm2d, submodule.py:21-24) -> fused plan.  ``dil``: the dilation LEFT of the layer's own
    when its input is already de-interleaved into sub-images (`space_to_batch2`, level l: dil = dilation >> l)."""
    conv, bn = seq[0], seq[1]
    return Conv2dPlan(conv.weight, _bn_of(bn), dilation=conv.dilation[0] if dil is None else dil, act=act, eps=bn.eps,
                      stride=conv.stride[0])


def space_to_batch2(x: torch.Tensor) -> torch.Tensor:
    """[N,C,h,w] -> [4N,C,h/2,w/2]: sub-image (y & 1, x & 1) of item n becomes item 4n + 2(y & 1) + (x & 1) (HIP)."""
    n, c, h, w = x.shape
    out = torch.empty((4 * n, c, h // 2, w // 2), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        _lib.check(_lib.load().dv_space_to_batch2_f32(x.data_ptr(), out.data_ptr(), n, c, h, w, _lib.stream_ptr()),
                   "dv_space_to_batch2_f32")
    return out


def batch_to_space(x: torch.Tensor, levels: int) -> torch.Tensor:
    """The inverse of ``levels`` applications of `space_to_batch2` at once (HIP)."""
    n, c, h, w = x.shape
    b = n >> (2 * levels)
    out = torch.empty((b, c, h << levels, w << levels), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        _lib.check(_lib.load().dv_batch_to_space_f32(x.data_ptr(), out.data_ptr(), b, c, h << levels, w << levels, levels,
                                                     _lib.stream_ptr()), "dv_batch_to_space_f32")
    return out


class _Block2dPlan:
    """BasicBlock (submodule.py:192-215): convbn+Mish, convbn, `out += x` (x through the 1x1 downsample when the
    width changes); the add rides in the second convolution's epilogue, no activation after it."""

    def __init__(self, blk: _Block2d, dil=None):
        self.dilation = blk.conv2[0].dilation[0]
        self.conv1 = _plan_cb2(blk.conv1[0], ACT_MISH, dil)
        self.conv2 = _plan_cb2(blk.conv2, ACT_NONE, dil)
        self.down = None if blk.downsample is None else _plan_cb2(blk.downsample, ACT_NONE)

    def __call__(self, x, group=1):
        skip = x if self.down is None else self.down(x)
        return self.conv2(self.conv1(x, group=group), residual=skip, group=group)


class _RefinePlan:
    """refinenet_version3.forward (pwcnet_ddim.py:292-306) on the 2-D implicit-GEMM kernel."""

    # Round 5: the dilated layers run in the sub-image domain (csrc/refine_inputs.hip, `space_to_batch2`): a layer whose
    # dilation is twice that of the tensor's current de-interleave gets one more de-interleave by 2 in front and is then a
    # plain dense convolution -- loads and stores of whole cache lines instead of 4-byte accesses at a 4 d-byte stride
    # (conv2d_wino issued 0.34 of the matrix pipe at d = 8 and 0.56 at d = 1), and the d = 16 block, too far apart for the
    # dilated Winograd kernel, becomes a Winograd layer as well.  Pointwise members of a block (1x1 down-sampling,
    # BatchNorm, Mish, the skip add) do not care.  Same arithmetic per output as the dilated kernel (which is the
    # dilation-1 kernel on a sub-image).  `sub_image_domain = False` keeps every layer on the image as it lies.
    sub_image_domain = True

    def __init__(self, m: RefineNet):
        self.mods = [getattr(m, n)[0] for n in ("conv1", "conv2", "conv3", "conv4")]        # convbn of the four head layers
        self.mods += [b for n in ("conv5", "conv6", "conv7") for b in getattr(m, n)]        # BasicBlocks
        self.dils = [mod[0].dilation[0] if isinstance(mod, nn.Sequential) else mod.conv2[0].dilation[0] for mod in self.mods]
        self._cache = {}
        self.conv8 = Conv2dPlan(m.conv8.weight, None, dilation=1, act=ACT_NONE)
        self.chain_ok = all(d & (d - 1) == 0 for d in self.dils)       # powers of two

    def plan(self, i, dil):
        """Layer i with `dil` of its dilation left to the kernel (the rest is in the tensor's de-interleave level)."""
        if (i, dil) not in self._cache:
            mod = self.mods[i]
            self._cache[(i, dil)] = (_plan_cb2(mod, ACT_MISH, dil) if isinstance(mod, nn.Sequential) else _Block2dPlan(mod, dil))
        return self._cache[(i, dil)]

    PAD_LIMIT = 1.15

    @staticmethod
    def max_level(h, w):
        """How often an h x w plane may be de-interleaved before the 16 x 16 tiles of the Winograd kernel pad the sub-planes
        by more than 15 % (384 x 1248: three times -- 48 x 156; a fourth gives 24 x 78 planes computed as 32 x 80 -- measured with
        PAD_LIMIT = 1.4: 120.9 against 118.1 ms per batch of 4, the d = 16 block no faster dense than strided)."""
        lvl = 0
        while h % 2 == 0 and w % 2 == 0:
            h, w = h // 2, w // 2
            if (-(-h // 16) * 16) * (-(-w // 16) * 16) > _RefinePlan.PAD_LIMIT * h * w:
                break
            lvl += 1
        return lvl

    def __call__(self, x, disp):
        h, w = x.shape[-2], x.shape[-1]
        lmax = self.max_level(h, w)
        if not (self.sub_image_domain and self.chain_ok and lmax > 0 and max(self.dils) > 1):
            for i, d in enumerate(self.dils):
                x = self.plan(i, d)(x)
            return self.conv8(x, residual=disp)      # disp + conv8(conv7)
        want = [min(d.bit_length() - 1, lmax) for d in self.dils]      # de-interleave level every layer runs at
        level = 0                                    # x is de-interleaved `level` times: [B * 4^level, C, h >> level, w >> level]
        for i, d in enumerate(self.dils):
            if want[i] < level:                      # back on the image (the dilation-1 tail of the stack)
                x = batch_to_space(x, level)
                level = 0
            while level < want[i]:
                x = space_to_batch2(x)
                level += 1
            plan = self.plan(i, d >> level)
            # a plain convolution in front of a level step stores its result de-interleaved itself
            # (`dv_conv2d_wino_s2b_f32`): no separate pass over the tensor
            nxt = want[i + 1] if i + 1 < len(self.dils) else 0
            fuse = (nxt == level + 1 and isinstance(plan, Conv2dPlan) and plan.wino_packed is not None and plan.dilation == 1
                    and x.shape[-2] % 2 == 0 and x.shape[-1] % 2 == 0)
            if fuse:
                x = plan(x, group=4 ** level, s2b_out=True)
                level += 1
            else:
                x = plan(x, group=4 ** level) if isinstance(plan, Conv2dPlan) else plan(x, 4 ** level)
        if level:
            x = batch_to_space(x, level)
        return self.conv8(x, residual=disp)          # disp + conv8(conv7)


class _Plans:
    def __init__(self, m: "PWCNet_ddim"):
        self.dres0 = _PairPlan(m.dres0, ACT_MISH)
        self.dres1 = _PairPlan(m.dres1, ACT_NONE)
        self.combine1 = _HourglassUpPlan(m.combine1)
        self.dres2, self.dres3, self.dres4 = (_HourglassPlan(h) for h in (m.dres2, m.dres3, m.dres4))
        self.classif3 = _PairPlan(m.classif3, ACT_NONE)
        self.refinenet3 = _RefinePlan(m.refinenet3)
        conv, bn = m.dispupsample[0][0], m.dispupsample[0][1]          # 1x1 conv (1 -> 32) + BN folded to a*d + b
        g = (bn.weight / torch.sqrt(bn.running_var + bn.eps)).detach().float()
        self.du_a = (conv.weight.detach().float().reshape(-1) * g).contiguous()
        self.du_b = (bn.bias.detach().float() - bn.running_mean.detach().float() * g).contiguous()
        if hasattr(m, "alphas_cumprod"):                                  # the origin network has no schedule
            ac = m.alphas_cumprod.detach().double().cpu()
            self.alphas_cumprod = ac
            self.sqrt_ac, self.sqrt_1mac = torch.sqrt(ac), torch.sqrt(1.0 - ac)
            self.sqrt_recip, self.sqrt_recipm1 = torch.sqrt(1.0 / ac), torch.sqrt(1.0 / ac - 1)


def groupwise_corr_pm(ref: torch.Tensor, tgt: torch.Tensor, maxdisp: int) -> torch.Tensor:
    """build_corrleation_volume(ref, tgt, maxdisp, 1) squeezed (KITTI12/models/submodule.py:121-135):
    mean over channels of ref(x) * tgt(x - i) for i >= 0; for i < 0 the reference's slices pair the first
    |i| columns of ref with the last |i| columns of tgt -- kept literally.  [B, 2*maxdisp+1, H, W]."""
    b, c, h, w = ref.shape
    out = ref.new_zeros(b, 2 * maxdisp + 1, h, w)
    for i in range(-maxdisp, maxdisp + 1):
        if i > 0:
            out[:, i + maxdisp, :, i:] = (ref[..., i:] * tgt[..., :-i]).mean(dim=1)
        elif i < 0:   # as written in the reference: `[:-i]` (first |i| columns) against `[i:]` (last |i| columns)
            out[:, i + maxdisp, :, :-i] = (ref[..., :-i] * tgt[..., i:]).mean(dim=1)
        else:
            out[:, maxdisp] = (ref * tgt).mean(dim=1)
    return out


class _PWCCommon(ReplicaPlanCache):
    """What the origin network (`PWCNet`, pwcnet.py:310-507) and `PWCNet_ddim` share: the plan cache, the fused
    multi-scale volume and the 2-D refinement of a regressed disparity."""

    # ---- plan cache ---------------------------------------------------------------------------------
    def _apply(self, fn, *a, **k):
        self._plans = None
        self._replica_clear()
        return super()._apply(fn, *a, **k)

    def _load_from_state_dict(self, *a, **k):      # reached however the checkpoint arrives (wrapper or direct)
        self._plans = None
        self._replica_clear()
        return super()._load_from_state_dict(*a, **k)

    def _replicate_for_data_parallel(self):        # nn.DataParallel replicas: plans parked on the source, per device
        replica = super()._replicate_for_data_parallel()
        replica._plans = None
        return self._mark_replica(replica)

    def train(self, mode: bool = True):
        if mode != self.training:
            self._plans = None
            self._replica_clear()
        return super().train(mode)

    def _weights_version(self) -> int:
        return sum(t._version for t in self.parameters()) + sum(t._version for t in self.buffers())

    def prepare(self, check_weights: bool = False) -> _Plans:
        if check_weights and self._plans is not None and self._plans.weights_version != self._weights_version():
            self._plans = None
        if self._plans is None:
            dev = self.dres0[0][0].weight.device
            if dev.type != "cuda":
                raise _lib.DiffuVolumeError("PWCNet_ddim hot path needs the model on the MI355X; no CPU fallback")
            self._plans = self._replica_lookup(dev)          # nn.DataParallel replica: plans parked on the source module
            if self._plans is not None:
                self._plans.weights_version = self._weights_version()
                return self._plans
            with torch.no_grad(), torch.cuda.device(dev):
                self._plans = _Plans(self)
                self._plans.weights_version = self._weights_version()
                self._plans.loop_key = self._plans.loop_steps = None
            self._replica_store(dev, self._plans)
        return self._plans

    @torch.no_grad()
    def fused_volume(self, fl, fr):
        """pwcnet_ddim.py:608-641: four gwc(+concat) volumes -> dres0/dres1 -> hourglassup -> `combine`."""
        p = self.prepare()
        vols = []
        for i, div in enumerate((4, 8, 16, 32), start=1):
            v = build_gwc_volume(fl[f"gw{i}"], fr[f"gw{i}"], self.maxdisp // div, self.num_groups)
            if self.use_concat_volume:
                cv = build_concat_volume(fl[f"concat_feature{i}"], fr[f"concat_feature{i}"], self.maxdisp // div,
                                         zero_left=True)
                v = torch.cat((v, cv), 1)
            vols.append(v)
        cost0 = p.dres0(vols[0])
        cost0 = p.dres1(cost0, residual_self=True)
        return p.combine1(cost0, vols[1], vols[2], vols[3])

    def _init_backbone(self, use_concat_volume: bool, time_embedding: bool):
        """Sub-modules under the reference's attribute names, in its construction order (pwcnet.py:318-362,
        pwcnet_ddim.py:389-431), and its weight initialisation (:364-381 / :433-447)."""
        self.concat_channels = 12 if use_concat_volume else 0
        self.feature_extraction = FeatureExtraction(use_concat_volume, 12)
        cin = self.num_groups + 2 * self.concat_channels
        self.dres0 = nn.Sequential(_cb3(cin, 32, 3, 1, 1), Mish(), _cb3(32, 32, 3, 1, 1), Mish())
        self.dres1 = nn.Sequential(_cb3(32, 32, 3, 1, 1), Mish(), _cb3(32, 32, 3, 1, 1))
        self.combine1 = HourglassUp(32)
        if time_embedding:
            self.time_embedding = DynamicHead(d_model=48)
        self.dres2, self.dres3, self.dres4 = Hourglass(32), Hourglass(32), Hourglass(32)
        for i in range(5):
            setattr(self, f"classif{i}", nn.Sequential(_cb3(32, 32, 3, 1, 1), Mish(), nn.Conv3d(32, 1, 3, 1, 1, bias=False)))
        self.refinenet3 = RefineNet(146)
        self.dispupsample = nn.Sequential(_cb2(1, 32, 1, 1, 0, 1), Mish())
        for m in self.modules():
            if isinstance(m, (nn.Conv2d, nn.Conv3d)) and not isinstance(m, nn.ConvTranspose3d):
                n = m.out_channels
                for k in m.kernel_size:
                    n *= k
                m.weight.data.normal_(0, math.sqrt(2.0 / n))
            elif isinstance(m, (nn.BatchNorm2d, nn.BatchNorm3d)):
                m.weight.data.fill_(1)
                m.bias.data.zero_()
            elif isinstance(m, nn.Linear):
                m.bias.data.zero_()
        self._plans: Optional[_Plans] = None

    @staticmethod
    def refine_features(features_left, features_right, size):
        """pwcnet_ddim.py:487-492 / pwcnet.py:481-486: both `finetune_feature` maps resized to full resolution
        (bilinear, align_corners=True).  They do not depend on the DDIM step: `ddim_sample` resizes them once per
        call instead of once per step."""
        hh, ww = size
        fl = F.interpolate(features_left["finetune_feature"], [hh, ww], mode="bilinear", align_corners=True)
        fr = F.interpolate(features_right["finetune_feature"], [hh, ww], mode="bilinear", align_corners=True)
        return fl, fr

    def _refine(self, pred3, features_left, features_right, resized=None):
        """pwcnet_ddim.py:486-502: warp the right refinement feature by pred3, +-24 correlation, concat
        (one HIP kernel), refinenet3 (2-D implicit-GEMM kernel) -> disp_finetune [B,H,W].  ``resized``: the two
        full-resolution feature maps of `refine_features` when the caller already has them."""
        p3 = pred3.unsqueeze(1)
        fl, fr = resized if resized is not None else self.refine_features(features_left, features_right, pred3.shape[-2:])
        plans = self.prepare()
        comb = refine_inputs(fl, fr, p3, plans.du_a, plans.du_b, 24)      # warp, +-24 correlation, concat
        return plans.refinenet3(comb, p3.contiguous()).squeeze(1)


class PWCNet(_PWCCommon, nn.Module):
    """The origin PCWNet of KITTI12/models/pwcnet.py:310-507 (`gwcnet-g` / `gwcnet-gc` in the registry,
    models/__init__.py:5-9) -- the network whose output is the `used` disparity of KITTI12/test.py:86-92 -- on the
    same kernels as `PWCNet_ddim`: eval ``forward(left, right) -> ([disp_finetune], [pred3])`` (:483-507).  Same
    parameter names and shapes as the reference class (no time embedding, no schedule buffers)."""

    def __init__(self, maxdisp: int, use_concat_volume: bool = False):
        super().__init__()
        if maxdisp != 192:
            raise ValueError("PWCNet is defined for maxdisp == 192 here (the volume pyramid divides it by 4..32)")
        self.maxdisp, self.use_concat_volume, self.num_groups = maxdisp, use_concat_volume, 40
        self._init_backbone(use_concat_volume, time_embedding=False)

    def forward(self, left, right):
        if self.training:
            raise NotImplementedError("the MI355X DiffuVolume path is inference-only (model.eval())")
        with torch.no_grad():
            p = self.prepare(check_weights=True)
            fl = self.feature_extraction(left)
            fr = self.feature_extraction(right)
            combine = self.fused_volume(fl, fr)
            cost3 = p.classif3(p.dres4(p.dres3(p.dres2(combine))))            # pwcnet.py:421-424, :469
            pred3, _ = upsample_softmax_regress(cost3, want_uncertainty=False, align_corners=True)
            disp_finetune = self._refine(pred3, fl, fr)
        return [disp_finetune], [pred3]


def PWCNet_G(d):
    return PWCNet(d, use_concat_volume=False)


def PWCNet_GC(d):
    return PWCNet(d, use_concat_volume=True)


class PWCNet_ddim(_PWCCommon, nn.Module):
    def __init__(self, maxdisp: int, use_concat_volume: bool = True, sampling_timesteps: int = 3,
                 ensemble_cof: Optional[Sequence[float]] = None):
        super().__init__()
        if maxdisp != 192:
            raise ValueError("PWCNet_ddim is defined for maxdisp == 192 (hard-coded 48 / 192 in the reference)")
        self.maxdisp, self.use_concat_volume, self.num_groups = maxdisp, use_concat_volume, 40
        self.scale, self.num_timesteps, self.sampling_timesteps = 1.0, 1000, sampling_timesteps
        self.ddim_sampling_eta, self.renewal, self.use_ensemble = 1.0, True, True
        if ensemble_cof is None:
            if sampling_timesteps != 3:
                raise ValueError("give ensemble_cof (S+1 weights) when sampling_timesteps != 3")
            ensemble_cof = (0.9, 0.0, 0.0, 0.1)                       # pwcnet_ddim.py:599
        if len(ensemble_cof) != sampling_timesteps + 1:
            raise ValueError("ensemble_cof needs sampling_timesteps + 1 entries")
        self.ensemble_cof = tuple(float(c) for c in ensemble_cof)
        self.dif_threshold, self.unc_threshold = 1.0, 1.0            # :571-572 (the last step's <2 mask is unused)

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

        self._init_backbone(use_concat_volume, time_embedding=True)

    def _loop_plan(self):
        """Per-step constants of the loop (time-MLP shift on the device, coefficient struct): they depend only
        on the step list, so they are computed once per weight set instead of once per step and pass."""
        p = self.prepare()
        key = (self.sampling_timesteps, self.ensemble_cof, self.dif_threshold, self.unc_threshold,
               self.ddim_sampling_eta, self.num_timesteps)
        if p.loop_key != key:
            dev = self.dres0[0][0].weight.device
            steps = []
            for i, (time, time_next) in enumerate(self._time_pairs()):
                t = torch.full((1,), time, device=dev, dtype=torch.long)
                shift = self.time_embedding.shift(t).float().reshape(-1).contiguous()
                steps.append(_LoopStep(time, time_next, self._step_coef(time, time_next, self.ensemble_cof[i + 1]), shift))
            p.loop_steps, p.loop_key = steps, key
        return p.loop_steps

    # ---- pieces ---------------------------------------------------------------------------------------
    def _time_pairs(self):
        times = torch.linspace(-1, self.num_timesteps - 1, steps=self.sampling_timesteps + 1)
        times = list(reversed(times.int().tolist()))
        return list(zip(times[:-1], times[1:]))

    def _filter(self, x_t, t, shift=None):
        b, c, h, w = x_t.shape
        if shift is None:
            shift = self.time_embedding.shift(t).float().contiguous()
        lib = _lib.load()
        x_t = x_t.contiguous()
        if x_t.dtype == torch.float32:
            n01 = torch.empty_like(x_t)
            _lib.check(lib.dv_noise_prepare_f32(x_t.data_ptr(), shift.data_ptr(), n01.data_ptr(), b, c, h * w,
                                                _lib.stream_ptr()), "dv_noise_prepare_f32")
            return n01, n01
        n01 = torch.empty_like(x_t)
        n01f = torch.empty(x_t.shape, dtype=torch.float32, device=x_t.device)
        _lib.check(lib.dv_noise_prepare_f64(x_t.data_ptr(), shift.data_ptr(), n01.data_ptr(), n01f.data_ptr(),
                                            b, c, h * w, _lib.stream_ptr()), "dv_noise_prepare_f64")
        return n01, n01f

    def _aggregate(self, volume, n01f):
        """pwcnet_ddim.py:472-477: (volume * filter) -> dres2 -> dres3 -> dres4 -> classif3."""
        p = self.prepare()
        return p.classif3(p.dres4(p.dres3(p.dres2(volume, in_scale=n01f))))

    def _uncertainty_about(self, cost, disp):
        """sum_k |disp - k| * softmax(upsampled cost)_k with ``disp`` = the refined disparity (:548-552)."""
        cost = cost[:, 0] if cost.dim() == 5 else cost
        b, d, h, w = cost.shape
        unc = torch.empty_like(disp)
        lib = _lib.load()
        timed("upsample_softmax_uncertainty", 0.0, 4.0 * (cost.numel() + 2 * disp.numel()),
              lambda: _lib.check(lib.dv_upsample_softmax_uncertainty_f32(cost.data_ptr(), disp.data_ptr(),
                                                                         unc.data_ptr(), b, d, h, w, 1,
                                                                         _lib.stream_ptr()),
                                 "dv_upsample_softmax_uncertainty_f32"))
        return unc

    def _step_coef(self, time, time_next, cof):
        p = self.prepare()
        k = _lib.DvDdimCoef()
        k.sqrt_recip_alpha, k.sqrt_recipm1_alpha = float(p.sqrt_recip[time]), float(p.sqrt_recipm1[time])
        k.dif_thr, k.unc_thr, k.cof = self.dif_threshold, self.unc_threshold, cof
        k.last = int(time_next < 0)
        k.clamp_max, k.ens_dif_thr = float(self.maxdisp - 1), 0.0
        if time_next >= 0:
            alpha, alpha_next = p.alphas_cumprod[time], p.alphas_cumprod[time_next]
            sigma = self.ddim_sampling_eta * ((1 - alpha / alpha_next) * (1 - alpha_next) / (1 - alpha)).sqrt()
            k.sigma, k.c = float(sigma), float((1 - alpha_next - sigma ** 2).sqrt())
            k.sqrt_alpha_next = float(alpha_next.sqrt())
        return k

    def _ddim_update(self, disp, unc, used, n01, eps, fill, mask, ens, coef, want_pred_noise=False):
        b, c, h, w = n01.shape
        dev = disp.device
        x_start = torch.empty((b, c, h, w), dtype=torch.float32, device=dev)
        x_next = None if coef.last else torch.empty((b, c, h, w), dtype=torch.float64, device=dev)
        pred_noise = torch.empty((b, c, h, w), dtype=torch.float64, device=dev) if want_pred_noise else None
        f32 = n01.dtype == torch.float32
        eps32 = eps if (eps is not None and eps.dtype == torch.float32) else None
        eps64 = eps if (eps is not None and eps.dtype == torch.float64) else None
        _lib.check(_lib.load().dv_ddim_step(disp.data_ptr(), unc.data_ptr(), used.data_ptr(), 0,
                                            n01.data_ptr() if f32 else 0, 0 if f32 else n01.data_ptr(),
                                            _lib.ptr(eps32), _lib.ptr(eps64), _lib.ptr(fill), mask.data_ptr(),
                                            x_start.data_ptr(), _lib.ptr(pred_noise), _lib.ptr(x_next), _lib.ptr(ens),
                                            b, c, h, w, ctypes.byref(coef), _lib.stream_ptr()), "dv_ddim_step")
        return x_start, x_next, pred_noise

    def _predict(self, volume, img, t, features_left, features_right, shift=None, resized=None):
        n01, n01f = self._filter(img, t, shift)
        cost = self._aggregate(volume, n01f)
        pred3, _ = upsample_softmax_regress(cost, want_uncertainty=False, align_corners=True)
        disp = self._refine(pred3, features_left, features_right, resized).contiguous()
        unc = self._uncertainty_about(cost, disp)
        return n01, cost, disp, unc

    # ---- reference API ---------------------------------------------------------------------------------
    @torch.no_grad()
    def model_predictions(self, volume, noise, t, features_left, features_right):
        """pwcnet_ddim.py:466-528 -> (pred_noise fp64, x_start fp32, disp_finetune [B,H,W], ProbVolumeHandle)."""
        volume = _dev_f32(volume, "volume")
        b, _, d, h, w = volume.shape
        with torch.cuda.device(volume.device):
            n01, cost, disp, unc = self._predict(volume, noise, t, features_left, features_right)
            coef = self._step_coef(int(t.reshape(-1)[0]), -1, 0.0)
            mask = torch.zeros((b, h, w), dtype=torch.float32, device=volume.device)
            x_start, _, pred_noise = self._ddim_update(disp, unc, disp, n01, None, None, mask, None, coef, True)
        return pred_noise, x_start, disp, ProbVolumeHandle(cost, unc, self.maxdisp, align_corners=True)

    @torch.no_grad()
    def ddim_sample(self, volume, used, asd, features_left, features_right, noise: Optional[NoiseFn] = None,
                    generator: Optional[torch.Generator] = None, trace=None):
        """pwcnet_ddim.py:530-602.  Random draws in reference order: 'x_T' (torch.randn, :541), then per
        non-final step 'eps' (randn_like(img), :585) and 'q' (randn_like(asd) inside q_sample, :590)."""
        volume = _dev_f32(volume, "volume")
        used = _dev_f32(used, "used")
        b, _, d, h, w = volume.shape
        dev = volume.device
        if d != 48:
            raise RuntimeError(f"the fused volume must have 48 disparity bins, got {d}")
        if used.numel() != b * 16 * h * w or tuple(used.shape[-2:]) != (4 * h, 4 * w):
            # the kernels index used / disp / ens as [B,4h,4w]; the reference fails at `disp - used` (:556)
            raise RuntimeError(f"The size of tensor a {(b, 4 * h, 4 * w)} must match the size of tensor b "
                               f"{tuple(used.shape)}: `used` must be the full-resolution disparity of the volume")
        used = used.reshape(b, 4 * h, 4 * w)
        if tuple(asd.shape) != (b, 48, h, w):
            raise RuntimeError(f"x_T must be {(b, 48, h, w)}, got {tuple(asd.shape)}")
        p = self.prepare(check_weights=True)

        def draw(kind, shape, dtype):
            if noise is not None:
                return noise(kind, shape, dtype).to(device=dev, dtype=dtype).contiguous()
            return torch.randn(shape, device=dev, dtype=dtype, generator=generator)

        with torch.cuda.device(dev):
            img = draw("x_T", (b, 48, h, w), torch.float32)
            asd = asd.to(dev)
            final = [used]
            mask = torch.zeros((b, h, w), dtype=torch.float32, device=dev)
            ens = used * self.ensemble_cof[0]
            handle = None
            # step-invariant: the two refinement feature maps at full resolution (the reference resizes them inside
            # every model_predictions call, :487-492)
            resized = self.refine_features(features_left, features_right, (4 * h, 4 * w))
            for i, st in enumerate(self._loop_plan()):
                time, time_next = st.time, st.time_next
                eps = fill = None
                if time_next >= 0:
                    eps = draw("eps", tuple(img.shape), img.dtype)
                    # asd = q_sample(asd, t): float64 from the first step on (float64 schedule buffers)
                    asd = p.sqrt_ac[time].item() * asd.double() + p.sqrt_1mac[time].item() * draw("q", tuple(asd.shape), asd.dtype).double()
                    fill = asd.contiguous()
                if trace is not None:
                    trace(i, {"when": "in", "img": img, "mask": mask.clone(), "eps": eps, "fill": fill})
                # (the last step's mask is never read again: the reference only builds the unused mask_final there)
                disp, unc, x_start, x_next, cost = self.ddim_step(i, volume, used, img, mask, ens, eps, fill,
                                                                  features_left, features_right, want_cost=True,
                                                                  resized=resized)
                if trace is not None:
                    trace(i, {"when": "out", "disp": disp, "unc": unc, "x_start": x_start, "x_next": x_next,
                              "mask": mask.clone()})
                handle = ProbVolumeHandle(cost, unc, self.maxdisp, align_corners=True)
                final.append(disp)
                img = x_start if time_next < 0 else x_next
        if getattr(p.dres0.b, "split", False):
            check_split_overflow(dev)
        if self.use_ensemble:
            return ens, handle
        return final[-1], handle

    @torch.no_grad()
    def ddim_step(self, i, volume, used, img, mask, ens=None, eps=None, fill=None, features_left=None,
                  features_right=None, want_cost=False, resized=None):
        """Iteration ``i`` of the loop of pwcnet_ddim.py:545-598 from explicit state (``img`` entering the step,
        ``mask`` updated in place, ``eps`` = randn_like(img), ``fill`` = the q_sample'd origin encoding).
        Returns (disp_finetune [B,4h,4w], uncertainty, x_start fp32, x_next fp64 | None[, cost])."""
        st = self._loop_plan()[i]
        n01, cost, disp, unc = self._predict(volume, img, None, features_left, features_right,
                                             shift=st.shift_rows(volume.shape[0]), resized=resized)
        x_start, x_next, _ = self._ddim_update(disp, unc, used, n01, eps, fill, mask, ens, st.coef)
        return (disp, unc, x_start, x_next, cost) if want_cost else (disp, unc, x_start, x_next)

    @torch.no_grad()
    def encode_disparity(self, disp):
        disp = _dev_f32(disp, "disp")
        b, h, w = disp.shape[0], disp.shape[-2], disp.shape[-1]
        x = torch.empty((b, 48, h, w), dtype=torch.float32, device=disp.device)
        with torch.cuda.device(disp.device):
            _lib.check(_lib.load().dv_encode_two_hot_f32(disp.data_ptr(), x.data_ptr(), b, 48, h * w,
                                                         _lib.stream_ptr()), "dv_encode_two_hot_f32")
        return x

    def forward(self, left, right, used, disp, mask=None):
        if self.training:
            raise NotImplementedError("the MI355X DiffuVolume path is inference-only (model.eval())")
        with torch.no_grad():
            self.prepare(check_weights=True)
            fl = self.feature_extraction(left)
            fr = self.feature_extraction(right)
            combine = self.fused_volume(fl, fr)
            x_T = self.encode_disparity(disp)
            disp_finetune, handle = self.ddim_sample(combine, used, x_T, fl, fr)
        return [disp_finetune], [handle]
