"""ACVNet + DiffuVolume (SceneFlow flavour) behind the reference's module API.

Drop-in for ``ACVNet_DDIM`` of SceneFlow/models/acv_ddim.py:122-482 (eval path):
same constructor, same ``forward(left, right, used, disp, mask_gt=None) -> [pred]``,
``model_predictions`` and ``ddim_sample``; same submodule / parameter / buffer names, so a
reference ``state_dict`` (579 keys, float64 schedule buffers) loads with ``strict=True``.

What runs where
  * cost volumes, the per-step volume filter, both 3-D hourglasses, the regression tail
    and the DDIM state update: HIP kernels of libdiffuvolume_hip.so (``submodule.py``);
  * the 2-D feature CNN, ``concatconv`` and the depthwise ``patch`` convolutions: plain
    PyTorch (MIOpen) -- outside the hot path named by the north star;
  * the 56 k-parameter time MLP: PyTorch, once per step.
The dead pre-DDIM aggregation pass of the reference's eval branch (acv_ddim.py:392-401,
its result ``pred2`` is never returned) is skipped; outputs are unchanged.
Training is out of scope: ``forward`` raises in training mode.
"""
from __future__ import annotations

import ctypes
import math
from typing import Callable, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F
from torch import nn

from . import _lib
from .head import DynamicHead
from .submodule import (ACT_NONE, ACT_RELU, Conv2dPlan, Conv3dPlan, Deconv3dPlan, Rank1FilterPlan, ReplicaPlanCache,
                        _dev_f32,
                        AttentionConcatVolume, build_concat_attention_volume, build_gwc_volume, check_split_overflow,
                        default_conv_precision, patch_volume, upsample_softmax_regress, window_attention)


def any_split_plan(plans) -> bool:
    """True if the prepared layers include a split-fp16 convolution (its range guard must be read)."""
    return plans is not None and getattr(plans.dres0.b, "split", False)

NoiseFn = Callable[[str, Tuple[int, ...], torch.dtype], torch.Tensor]


# --------------------------------------------------------------------------------------
# parameter containers (names = reference state_dict keys)
# --------------------------------------------------------------------------------------
def _cb2(cin, cout, k, stride, pad, dil):
    return nn.Sequential(nn.Conv2d(cin, cout, k, stride, dil if dil > 1 else pad, dil, bias=False),
                         nn.BatchNorm2d(cout))


def _cb3(cin, cout, k, stride, pad):
    return nn.Sequential(nn.Conv3d(cin, cout, k, stride, pad, bias=False), nn.BatchNorm3d(cout))


class _ResBlock2d(nn.Module):
    """BasicBlock of the 2-D feature CNN (SceneFlow/models/submodule.py:307-330)."""

    def __init__(self, cin, planes, stride, downsample, pad, dil):
        super().__init__()
        self.conv1 = nn.Sequential(_cb2(cin, planes, 3, stride, pad, dil), nn.ReLU(inplace=True))
        self.conv2 = _cb2(planes, planes, 3, 1, pad, dil)
        self.downsample = downsample

    def forward(self, x):
        y = self.conv2(self.conv1(x))
        return y + (x if self.downsample is None else self.downsample(x))


def _plan_cb2(seq: nn.Sequential, act: int) -> Conv2dPlan:
    """convbn 2-D (Conv2d + BatchNorm2d, submodule.py:21-24 analogue) -> fused plan (stride 1 or 2)."""
    conv, bn = seq[0], seq[1]
    return Conv2dPlan(conv.weight, (bn.weight, bn.bias, bn.running_mean, bn.running_var), dilation=conv.dilation[0],
                      act=act, eps=bn.eps, stride=conv.stride[0])


class _ResBlock2dPlan:
    """BasicBlock: convbn+ReLU, convbn, `out += x` (x through the 1x1 downsample when shape changes)."""

    def __init__(self, blk: _ResBlock2d):
        self.conv1 = _plan_cb2(blk.conv1[0], ACT_RELU)
        self.conv2 = _plan_cb2(blk.conv2, ACT_NONE)
        self.down = None if blk.downsample is None else _plan_cb2(blk.downsample, ACT_NONE)

    def __call__(self, x):
        return self.conv2(self.conv1(x), residual=x if self.down is None else self.down(x))


class FeatureExtraction(ReplicaPlanCache, nn.Module):
    """2-D feature CNN (acv_ddim.py:14-53): 320-channel 1/4-resolution ``gwc_feature``.  On the GPU (eval) all 55
    convolutions run on the 2-D implicit-GEMM kernel with BN / ReLU / the residual add fused (csrc/conv2d.hip)."""

    def __init__(self):
        super().__init__()
        self.inplanes = 32
        self.firstconv = nn.Sequential(_cb2(3, 32, 3, 2, 1, 1), nn.ReLU(inplace=True),
                                       _cb2(32, 32, 3, 1, 1, 1), nn.ReLU(inplace=True),
                                       _cb2(32, 32, 3, 1, 1, 1), nn.ReLU(inplace=True))
        self.layer1 = self._stack(32, 3, 1, 1, 1)
        self.layer2 = self._stack(64, 16, 2, 1, 1)
        self.layer3 = self._stack(128, 3, 1, 1, 1)
        self.layer4 = self._stack(128, 3, 1, 1, 2)
        self._plans = None

    def _stack(self, planes, blocks, stride, pad, dil):
        down = None
        if stride != 1 or self.inplanes != planes:
            down = nn.Sequential(nn.Conv2d(self.inplanes, planes, 1, stride, bias=False), nn.BatchNorm2d(planes))
        layers = [_ResBlock2d(self.inplanes, planes, stride, down, pad, dil)]
        self.inplanes = planes
        layers += [_ResBlock2d(planes, planes, 1, None, pad, dil) for _ in range(1, blocks)]
        return nn.Sequential(*layers)

    def _apply(self, fn, *a, **k):
        self._plans = None
        self._replica_clear()
        return super()._apply(fn, *a, **k)

    def _load_from_state_dict(self, *a, **k):
        self._plans = None
        self._replica_clear()
        return super()._load_from_state_dict(*a, **k)

    def train(self, mode: bool = True):
        if mode != self.training:
            self._plans = None
            self._replica_clear()
        return super().train(mode)

    def _replicate_for_data_parallel(self):
        replica = super()._replicate_for_data_parallel()
        replica._plans = None
        return self._mark_replica(replica)

    def prepare(self):
        version = sum(t._version for t in self.parameters()) + sum(t._version for t in self.buffers())
        if self._plans is not None and self._plans[2] != version:      # weights overwritten in place since
            self._plans = None
        if self._plans is None:
            dev = self.firstconv[0][0].weight.device
            self._plans = self._replica_lookup(dev)                      # nn.DataParallel replica: plans parked on the source
        if self._plans is None:
            with torch.no_grad():
                first = [_plan_cb2(self.firstconv[i], ACT_RELU) for i in (0, 2, 4)]
                stacks = [[_ResBlock2dPlan(b) for b in getattr(self, n)] for n in ("layer1", "layer2", "layer3", "layer4")]
            self._plans = (first, stacks, version)
            self._replica_store(dev, self._plans)
        return self._plans

    def forward(self, x):
        if not x.is_cuda:
            raise _lib.DiffuVolumeError(f"input is on {x.device}: the feature CNN runs on the MI355X (no CPU fallback)")
        if self.training or (torch.is_grad_enabled() and x.requires_grad):
            x = self.layer1(self.firstconv(x))          # training / autograd: the plain PyTorch modules, on the GPU
            l2 = self.layer2(x)
            l3 = self.layer3(l2)
            l4 = self.layer4(l3)
            return {"gwc_feature": torch.cat((l2, l3, l4), dim=1)}
        first, stacks, _ = self.prepare()
        with torch.no_grad():
            for p in first:
                x = p(x)
            outs = []
            for stack in stacks:
                for blk in stack:
                    x = blk(x)
                outs.append(x)
        return {"gwc_feature": torch.cat(outs[1:], dim=1)}


class _WindowAttention(nn.Module):
    """Parameters of attention_block (submodule.py:383-396)."""

    def __init__(self, channels: int, num_heads: int):
        super().__init__()
        self.num_heads = num_heads
        self.qkv_3d = nn.Linear(channels, channels * 3, bias=True)
        self.final1x1 = nn.Conv3d(channels, channels, 1)


class Hourglass(nn.Module):
    """3-D encoder/decoder with skip convs and bottleneck window attention (acv_ddim.py:56-93)."""

    def __init__(self, c: int):
        super().__init__()
        self.conv1 = nn.Sequential(_cb3(c, 2 * c, 3, 2, 1), nn.ReLU(inplace=True))
        self.conv2 = nn.Sequential(_cb3(2 * c, 2 * c, 3, 1, 1), nn.ReLU(inplace=True))
        self.conv3 = nn.Sequential(_cb3(2 * c, 4 * c, 3, 2, 1), nn.ReLU(inplace=True))
        self.conv4 = nn.Sequential(_cb3(4 * c, 4 * c, 3, 1, 1), nn.ReLU(inplace=True))
        self.attention_block = _WindowAttention(4 * c, 16)
        self.conv5 = nn.Sequential(nn.ConvTranspose3d(4 * c, 2 * c, 3, padding=1, output_padding=1, stride=2, bias=False),
                                   nn.BatchNorm3d(2 * c))
        self.conv6 = nn.Sequential(nn.ConvTranspose3d(2 * c, c, 3, padding=1, output_padding=1, stride=2, bias=False),
                                   nn.BatchNorm3d(c))
        self.redir1 = _cb3(c, c, 1, 1, 0)
        self.redir2 = _cb3(2 * c, 2 * c, 1, 1, 0)


# --------------------------------------------------------------------------------------
# prepared (device-resident, BN-folded, repacked) hot-path layers
# --------------------------------------------------------------------------------------
def _bn_of(bn: nn.BatchNorm3d):
    return (bn.weight, bn.bias, bn.running_mean, bn.running_var)


def _plan_cb3(seq: nn.Sequential, stride: int, act: int) -> Conv3dPlan:
    return Conv3dPlan(seq[0].weight, _bn_of(seq[1]), stride=stride, act=act, eps=seq[1].eps)


class _HourglassPlan:
    def __init__(self, hg: Hourglass):
        self.conv1 = _plan_cb3(hg.conv1[0], 2, ACT_RELU)
        self.conv2 = _plan_cb3(hg.conv2[0], 1, ACT_RELU)
        self.conv3 = _plan_cb3(hg.conv3[0], 2, ACT_RELU)
        self.conv4 = _plan_cb3(hg.conv4[0], 1, ACT_RELU)
        ab = hg.attention_block
        self.heads = ab.num_heads
        self.attn = tuple(t.detach().float().contiguous() for t in
                          (ab.qkv_3d.weight, ab.qkv_3d.bias,
                           ab.final1x1.weight.reshape(ab.final1x1.weight.shape[0], -1), ab.final1x1.bias))
        # relu(BN(deconv) + BN(redir(skip))): the 1x1x1 redir convolutions ride inside the transposed convolutions
        self.conv5 = Deconv3dPlan(hg.conv5[0].weight, _bn_of(hg.conv5[1]), act=ACT_RELU, eps=hg.conv5[1].eps,
                                  redir=(hg.redir2[0].weight, _bn_of(hg.redir2[1])), redir_eps=hg.redir2[1].eps)
        self.conv6 = Deconv3dPlan(hg.conv6[0].weight, _bn_of(hg.conv6[1]), act=ACT_RELU, eps=hg.conv6[1].eps,
                                  redir=(hg.redir1[0].weight, _bn_of(hg.redir1[1])), redir_eps=hg.redir1[1].eps)

    def __call__(self, x: torch.Tensor) -> torch.Tensor:
        c1 = self.conv1(x)
        c2 = self.conv2(c1)
        c3 = self.conv3(c2)
        c4 = self.conv4(c3)
        c4 = window_attention(c4, *self.attn, heads=self.heads)
        c5 = self.conv5(c4, skip=c2)                       # relu(deconv+bn + redir2(conv2))
        return self.conv6(c5, skip=x)                      # relu(deconv+bn + redir1(x))


# A/B switch for tools/ and tests: False keeps the first layer of a DDIM step on the generic filtered convolution
RANK1_FILTER = True


class _ConvPairPlan:
    """convbn-ReLU-conv[bn][-ReLU] stacks (dres0 / dres1 / classif*, acv_ddim.py:200-222)."""

    def __init__(self, seq: nn.Sequential, relu_last: bool):
        self.a = _plan_cb3(seq[0], 1, ACT_RELU)
        last = seq[2]
        if isinstance(last, nn.Sequential):
            self.b = _plan_cb3(last, 1, ACT_RELU if relu_last else ACT_NONE)
        else:
            self.b = Conv3dPlan(last.weight, None, stride=1, act=ACT_NONE)

    def __call__(self, x, in_scale=None, residual_self=False):
        y = self.a(x, in_scale=in_scale)
        return self.b(y, residual=x if residual_self else None)

    def second(self, y):
        return self.b(y)


class _Plans:
    def __init__(self, m: "ACVNet_DDIM"):
        self.dres0 = _ConvPairPlan(m.dres0, relu_last=True)
        # dres0[0] on the factors of a filtered attention-concat volume (acv_ddim.py:260, :388-390); exact-fp32 paths only
        conv0, bn0 = m.dres0[0][0], m.dres0[0][1]
        self.dres0_rank1 = None
        if default_conv_precision() in ("f32", "f32_direct") and RANK1_FILTER:
            self.dres0_rank1 = Rank1FilterPlan(conv0.weight, (bn0.weight, bn0.bias, bn0.running_mean, bn0.running_var),
                                               act=ACT_RELU, eps=bn0.eps)
        self.dres1 = _ConvPairPlan(m.dres1, relu_last=False)
        self.dres2 = _HourglassPlan(m.dres2)
        self.dres3 = _HourglassPlan(m.dres3)
        self.classif2 = _ConvPairPlan(m.classif2, relu_last=False)
        self.dres1_att = _ConvPairPlan(m.dres1_att_, relu_last=False)
        self.dres2_att = _HourglassPlan(m.dres2_att_)
        self.classif_att = _ConvPairPlan(m.classif_att_, relu_last=False)
        # attention branch front: the two depth-wise (1,3,3) stencils fuse into one pass; concatconv on the 2-D kernel
        self.patch_w1 = m.patch.weight.detach().float().reshape(40, 9).contiguous()
        self.patch_w2 = torch.cat([p.weight.detach().float().reshape(-1, 9) for p in (m.patch_l1, m.patch_l2, m.patch_l3)]).contiguous()
        self.patch_dil = torch.tensor([1] * 8 + [2] * 16 + [3] * 16, dtype=torch.int32, device=self.patch_w1.device)
        self.concat_a = _plan_cb2(m.concatconv[0], ACT_RELU)
        self.concat_b = Conv2dPlan(m.concatconv[2].weight, None, act=ACT_NONE)
        if hasattr(m, "alphas_cumprod"):                       # the origin ACVNet has no diffusion schedule
            ac = m.alphas_cumprod.detach().double().cpu()
            self.alphas_cumprod = ac
            self.sqrt_recip = torch.sqrt(1.0 / ac)
            self.sqrt_recipm1 = torch.sqrt(1.0 / ac - 1)
        self.loop_key = self.loop_steps = None          # per-step constants of the DDIM loop (ACVNet_DDIM._loop_plan)
        self.weights_version = -1


def _volume_arg(volume):
    """The `volume` argument of the reference API: a float32 device tensor, or the factor handle standing in for it."""
    return volume if isinstance(volume, AttentionConcatVolume) else _dev_f32(volume, "volume")


def _volume_tensor(volume) -> torch.Tensor:
    return volume.tensor() if isinstance(volume, AttentionConcatVolume) else volume


class _LoopStep:
    """Everything one DDIM step needs that depends only on its timestep: the coefficient struct handed to
    ``dv_ddim_step`` by value and the time-MLP shift (device resident; one row per batch entry on demand)."""

    def __init__(self, time: int, time_next: int, coef: "_lib.DvDdimCoef", shift: torch.Tensor):
        self.time, self.time_next, self.coef, self.shift = time, time_next, coef, shift
        self._rows = {}

    def shift_rows(self, b: int) -> torch.Tensor:
        if b not in self._rows:
            self._rows[b] = self.shift.reshape(1, -1).expand(b, -1).contiguous()
        return self._rows[b]


class ProbVolumeHandle:
    """Stand-in for ``pred_volume2`` ([B,192,H,W], 377 MB per pair in the reference): keeps the
    quarter-resolution cost and the fused uncertainty; ``dense()`` materialises the softmax
    volume with PyTorch ops for callers that really want it."""

    def __init__(self, cost: torch.Tensor, uncertainty: torch.Tensor, maxdisp: int, align_corners: bool = False):
        self.cost, self.uncertainty, self.maxdisp, self.align_corners = cost, uncertainty, maxdisp, align_corners

    def dense(self) -> torch.Tensor:
        b, _, d, h, w = self.cost.shape
        up = F.interpolate(self.cost, [self.maxdisp, h * 4, w * 4], mode="trilinear",
                           align_corners=True if self.align_corners else None)
        return F.softmax(up.squeeze(1), dim=1)


def cosine_beta_schedule(timesteps: int, s: float = 0.008) -> torch.Tensor:
    """acv_ddim.py:113-119 (float64)."""
    x = torch.linspace(0, timesteps, timesteps + 1, dtype=torch.float64)
    ac = torch.cos(((x / timesteps) + s) / (1 + s) * math.pi * 0.5) ** 2
    ac = ac / ac[0]
    return torch.clip(1 - (ac[1:] / ac[:-1]), 0, 0.999)


class _HipPlanMixin(ReplicaPlanCache, nn.Module):
    """Plan cache shared by the ACV wrappers: BatchNorm folding / weight repacking happens once per weight
    set and is redone when parameters move, are reloaded (through this module or any wrapper: ``nn.DataParallel(
    model).load_state_dict`` only reaches ``_load_from_state_dict``) or the train/eval mode really changes."""
    _plans = None

    def _drop_plans(self):
        self._plans = None
        self._replica_clear()

    def _apply(self, fn, *args, **kwargs):
        self._drop_plans()
        return super()._apply(fn, *args, **kwargs)

    def _load_from_state_dict(self, *args, **kwargs):
        self._drop_plans()
        return super()._load_from_state_dict(*args, **kwargs)

    def train(self, mode: bool = True):
        if mode != self.training:          # the reference calls model.eval() on every batch: keep the plans then
            self._drop_plans()
        return super().train(mode)

    def _replicate_for_data_parallel(self):
        """nn.DataParallel copies ``__dict__`` into its per-device replicas: a replica must fold / repack its OWN
        (broadcast) weights on its own device, not inherit the source module's device-resident plans."""
        replica = super()._replicate_for_data_parallel()
        replica._plans = None
        return self._mark_replica(replica)

    def _weights_version(self) -> int:
        """Sum of the in-place version counters of every parameter and buffer: changes whenever a weight is
        overwritten (``load_state_dict`` on any sub-module, ``p.data.copy_``, an optimizer step)."""
        return sum(t._version for t in self.parameters()) + sum(t._version for t in self.buffers())

    def prepare(self, check_weights: bool = False) -> _Plans:
        """Fold BatchNorm and repack weights for the HIP kernels (once per weight set).  The public entry points
        pass ``check_weights=True``: plans built from weights that were since modified in place are rebuilt."""
        if check_weights and self._plans is not None and self._plans.weights_version != self._weights_version():
            self._drop_plans()
        if self._plans is None:
            dev = self.dres0[0][0].weight.device
            if dev.type != "cuda":
                raise _lib.DiffuVolumeError(
                    "the ACVNet hot path needs the model on the MI355X (model.cuda()); no CPU fallback")
            self._plans = self._replica_lookup(dev)          # nn.DataParallel replica: plans parked on the source module
            if self._plans is not None:
                self._plans.weights_version = self._weights_version()
                return self._plans
            with torch.no_grad(), torch.cuda.device(dev):
                self._plans = _Plans(self)
                self._plans.weights_version = self._weights_version()
            self._replica_store(dev, self._plans)
        return self._plans


class ACVNet_DDIM(_HipPlanMixin):
    def __init__(self, maxdisp: int, attn_weights_only: bool = False, freeze_attn_weights: bool = False,
                 sampling_timesteps: int = 5, ensemble_cof: Optional[Sequence[float]] = None):
        super().__init__()
        if maxdisp != 192:
            # the reference hard-codes 48 / 192 (acv_ddim.py:278,:302,:325)
            raise ValueError("ACVNet_DDIM is defined for maxdisp == 192")
        self.maxdisp = maxdisp
        self.attn_weights_only = attn_weights_only
        self.freeze_attn_weights = freeze_attn_weights
        self.num_groups = 40
        self.concat_channels = 32
        self.scale = 1.0
        self.num_timesteps = 1000
        self.sampling_timesteps = sampling_timesteps
        self.ddim_sampling_eta = 1.0
        self.renewal = True
        self.use_ensemble = True
        # ensemble over [used, disp_1 .. disp_S] (acv_ddim.py:367)
        if ensemble_cof is None:
            if sampling_timesteps != 5:
                raise ValueError("give ensemble_cof (S+1 weights) when sampling_timesteps != 5")
            ensemble_cof = (0.5, 0.0, 0.0, 0.0, 0.2, 0.3)
        if len(ensemble_cof) != sampling_timesteps + 1:
            raise ValueError("ensemble_cof needs sampling_timesteps + 1 entries")
        self.ensemble_cof = tuple(float(c) for c in ensemble_cof)
        self.dif_threshold, self.unc_threshold = 1.0, 3.0

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

        self.feature_extraction = FeatureExtraction()
        self.concatconv = nn.Sequential(_cb2(320, 128, 3, 1, 1, 1), nn.ReLU(inplace=True),
                                        nn.Conv2d(128, self.concat_channels, 1, bias=False))
        self.patch = nn.Conv3d(40, 40, (1, 3, 3), 1, (0, 1, 1), 1, groups=40, bias=False)
        self.patch_l1 = nn.Conv3d(8, 8, (1, 3, 3), 1, (0, 1, 1), 1, groups=8, bias=False)
        self.patch_l2 = nn.Conv3d(16, 16, (1, 3, 3), 1, (0, 2, 2), 2, groups=16, bias=False)
        self.patch_l3 = nn.Conv3d(16, 16, (1, 3, 3), 1, (0, 3, 3), 3, groups=16, bias=False)
        self.dres1_att_ = nn.Sequential(_cb3(40, 32, 3, 1, 1), nn.ReLU(inplace=True), _cb3(32, 32, 3, 1, 1))
        self.dres2_att_ = Hourglass(32)
        self.classif_att_ = self._classifier()
        self.time_embedding = DynamicHead(d_model=48)
        self.dres0 = nn.Sequential(_cb3(64, 32, 3, 1, 1), nn.ReLU(inplace=True),
                                   _cb3(32, 32, 3, 1, 1), nn.ReLU(inplace=True))
        self.dres1 = nn.Sequential(_cb3(32, 32, 3, 1, 1), nn.ReLU(inplace=True), _cb3(32, 32, 3, 1, 1))
        self.dres2 = Hourglass(32)
        self.dres3 = Hourglass(32)
        self.classif0 = self._classifier()
        self.classif1 = self._classifier()
        self.classif2 = self._classifier()
        self._init_weights()
        self._plans: Optional[_Plans] = None

    @staticmethod
    def _classifier():
        return nn.Sequential(_cb3(32, 32, 3, 1, 1), nn.ReLU(inplace=True),
                             nn.Conv3d(32, 1, 3, 1, 1, bias=False))

    def _init_weights(self):  # acv_ddim.py:224-238
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

    # ---- pieces of the hot path ----------------------------------------------------------
    def _time_pairs(self) -> List[Tuple[int, int]]:
        times = torch.linspace(-1, self.num_timesteps - 1, steps=self.sampling_timesteps + 1)
        times = list(reversed(times.int().tolist()))
        return list(zip(times[:-1], times[1:]))

    def _filter(self, x_t: torch.Tensor, t: Optional[torch.Tensor], shift: Optional[torch.Tensor] = None):
        """time shift + clamp + [0,1] (head.py:74-77, acv_ddim.py:256-258) -> (n01 state dtype, n01 fp32).
        ``shift`` [B,48]: the time MLP's output when it was precomputed for this step (``_loop_plan``)."""
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
        if x_t.dtype != torch.float64:
            raise TypeError("the DDIM state is float32 (first step) or float64")
        n01 = torch.empty_like(x_t)
        n01f = torch.empty(x_t.shape, dtype=torch.float32, device=x_t.device)
        _lib.check(lib.dv_noise_prepare_f64(x_t.data_ptr(), shift.data_ptr(), n01.data_ptr(), n01f.data_ptr(),
                                            b, c, h * w, _lib.stream_ptr()), "dv_noise_prepare_f64")
        return n01, n01f

    def _aggregate(self, volume: torch.Tensor, n01f: Optional[torch.Tensor]) -> torch.Tensor:
        """acv_ddim.py:260-266: (volume * filter) -> dres0 -> dres1(+res) -> dres2 -> dres3 -> classif2."""
        p = self.prepare()
        r1 = getattr(p, "dres0_rank1", None)
        if r1 is not None and r1.applies(volume):
            cost0 = p.dres0.second(r1(volume, n01f))            # first layer on the volume's factors (Rank1FilterPlan)
        else:                                                   # any other tensor; a factor handle is materialised once
            cost0 = p.dres0(_volume_tensor(volume), in_scale=n01f)
        cost0 = p.dres1(cost0, residual_self=True)
        out2 = p.dres3(p.dres2(cost0))
        return p.classif2(out2)

    def _step_coef(self, time: int, time_next: int, cof: float) -> _lib.DvDdimCoef:
        p = self.prepare()
        k = _lib.DvDdimCoef()
        k.sqrt_recip_alpha = float(p.sqrt_recip[time])
        k.sqrt_recipm1_alpha = float(p.sqrt_recipm1[time])
        k.dif_thr, k.unc_thr, k.cof = self.dif_threshold, self.unc_threshold, cof
        k.last = int(time_next < 0)
        k.clamp_max, k.ens_dif_thr = float(self.maxdisp - 1), 0.0
        if time_next >= 0:
            alpha, alpha_next = p.alphas_cumprod[time], p.alphas_cumprod[time_next]
            sigma = self.ddim_sampling_eta * ((1 - alpha / alpha_next) * (1 - alpha_next) / (1 - alpha)).sqrt()
            k.sigma = float(sigma)
            k.c = float((1 - alpha_next - sigma ** 2).sqrt())
            k.sqrt_alpha_next = float(alpha_next.sqrt())
        return k

    def _loop_plan(self) -> List[_LoopStep]:
        """Per-step constants of ``ddim_sample`` (acv_ddim.py:306-308, :348-352 and the time MLP of head.py:74-75,
        which sees nothing but ``t``): computed once per (weights, step list) and kept on the device, so the loop
        itself launches no time-MLP kernels and copies no scalars."""
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

    def _ddim_update(self, disp, unc, used, n01, eps, fill, mask, ens, coef, want_pred_noise=False):
        b, c, h, w = n01.shape
        dev = disp.device
        x_start = torch.empty((b, c, h, w), dtype=torch.float32, device=dev)
        x_next = None if coef.last else torch.empty((b, c, h, w), dtype=torch.float64, device=dev)
        pred_noise = torch.empty((b, c, h, w), dtype=torch.float64, device=dev) if want_pred_noise else None
        f32 = n01.dtype == torch.float32
        eps32 = eps if (eps is not None and eps.dtype == torch.float32) else None
        eps64 = eps if (eps is not None and eps.dtype == torch.float64) else None
        lib = _lib.load()
        _lib.check(lib.dv_ddim_step(disp.data_ptr(), unc.data_ptr(), used.data_ptr(), 0,
                                    n01.data_ptr() if f32 else 0, 0 if f32 else n01.data_ptr(),
                                    _lib.ptr(eps32), _lib.ptr(eps64), _lib.ptr(fill), mask.data_ptr(),
                                    x_start.data_ptr(), _lib.ptr(pred_noise), _lib.ptr(x_next), _lib.ptr(ens),
                                    b, c, h, w, ctypes.byref(coef), _lib.stream_ptr()), "dv_ddim_step")
        return x_start, x_next, pred_noise

    def _check_loop_args(self, volume, used, img):
        """Shapes the kernels index by: used / disp / ens are [B,4h,4w] for a [B,C,48,h,w] volume (the reference
        fails with a broadcast error at ``disp - used``, acv_ddim.py:322)."""
        b, _, d, h, w = volume.shape
        if d != self.maxdisp // 4:
            raise RuntimeError(f"the volume must have {self.maxdisp // 4} disparity bins, got {d}")
        if used.numel() != b * 16 * h * w or tuple(used.shape[-2:]) != (4 * h, 4 * w):
            raise RuntimeError(f"The size of tensor a {(b, 4 * h, 4 * w)} must match the size of tensor b "
                               f"{tuple(used.shape)}: `used` must be the full-resolution disparity of the volume")
        if tuple(img.shape) != (b, d, h, w):
            raise RuntimeError(f"x_T must be {(b, d, h, w)}, got {tuple(img.shape)}")
        return used.reshape(b, 4 * h, 4 * w)

    # ---- reference API ---------------------------------------------------------------------
    @torch.no_grad()
    def model_predictions(self, volume: torch.Tensor, noise: torch.Tensor, t: torch.Tensor):
        """acv_ddim.py:254-296 -> (pred_noise fp64, x_start fp32, pred [B,H,W], ProbVolumeHandle)."""
        volume = _volume_arg(volume)
        b, _, d, h, w = volume.shape
        self.prepare(check_weights=True)
        with torch.cuda.device(volume.device):
            n01, n01f = self._filter(noise, t)
            cost = self._aggregate(volume, n01f)
            pred, unc = upsample_softmax_regress(cost, want_uncertainty=True)
            time = int(t.reshape(-1)[0])
            coef = self._step_coef(time, -1, 0.0)
            mask = torch.zeros((b, h, w), dtype=torch.float32, device=volume.device)
            x_start, _, pred_noise = self._ddim_update(pred, unc, pred, n01, None, None, mask, None, coef,
                                                       want_pred_noise=True)
        return pred_noise, x_start, pred, ProbVolumeHandle(cost, unc, self.maxdisp)

    @torch.no_grad()
    def ddim_step(self, i: int, volume: torch.Tensor, used: torch.Tensor, img: torch.Tensor, mask: torch.Tensor,
                  ens: Optional[torch.Tensor] = None, eps: Optional[torch.Tensor] = None,
                  fill: Optional[torch.Tensor] = None, out_disp: Optional[torch.Tensor] = None):
        """Iteration ``i`` of the loop of acv_ddim.py:313-362 from explicit state: ``img`` is the DDIM state entering
        the step (fp32 for i == 0, fp64 later), ``mask`` [B,h,w] the accumulated renewal mask (updated in place),
        ``ens`` the ensemble accumulator (updated in place when given), ``eps`` / ``fill`` this step's
        ``randn_like(img)`` / ``rand_like`` draws (not needed on the last step).
        Returns (disp [B,4h,4w], uncertainty, x_start fp32, x_next fp64 | None)."""
        st = self._loop_plan()[i]
        b = volume.shape[0]
        n01, n01f = self._filter(img, None, st.shift_rows(b))
        cost = self._aggregate(volume, n01f)
        disp, unc = upsample_softmax_regress(cost, want_uncertainty=True, out_disp=out_disp)
        x_start, x_next, _ = self._ddim_update(disp, unc, used, n01, eps, fill, mask, ens, st.coef)
        return disp, unc, x_start, x_next

    @torch.no_grad()
    def ddim_sample(self, volume: torch.Tensor, used: torch.Tensor, asd: torch.Tensor,
                    noise: Optional[NoiseFn] = None, generator: Optional[torch.Generator] = None,
                    trace: Optional[Callable[[int, dict], None]] = None):
        """acv_ddim.py:298-370.  ``noise(kind, shape, dtype)`` (kind 'eps' = randn_like(img) :354,
        'fill' = rand_like :360) injects the random draws for parity tests; by default they come
        from the device generator.  ``trace(i, state)`` (tests) receives every step's tensors.
        Returns (final_prediction [B,H,W], stack [S+1,B,H,W])."""
        volume = _volume_arg(volume)
        used = self._check_loop_args(volume, _dev_f32(used, "used"), asd)
        b, _, d, h, w = volume.shape
        dev = volume.device
        self.prepare(check_weights=True)

        def draw(kind, shape, dtype):
            if noise is not None:
                return noise(kind, shape, dtype).to(device=dev, dtype=dtype).contiguous()
            fn = torch.randn if kind == "eps" else torch.rand
            return fn(shape, device=dev, dtype=dtype, generator=generator)

        with torch.cuda.device(dev):
            steps = self._loop_plan()
            img = asd.to(dev).contiguous()
            stack = torch.empty((len(steps) + 1, b, 4 * h, 4 * w), dtype=torch.float32, device=dev)
            stack[0].copy_(used)
            mask = torch.zeros((b, h, w), dtype=torch.float32, device=dev)
            ens = used * self.ensemble_cof[0]
            for i, st in enumerate(steps):
                eps = fill = None
                if st.time_next >= 0:
                    eps = draw("eps", tuple(img.shape), img.dtype)
                    fill = draw("fill", tuple(img.shape), torch.float64)
                if trace is not None:
                    trace(i, {"when": "in", "img": img, "mask": mask.clone(), "eps": eps, "fill": fill})
                disp, unc, x_start, x_next = self.ddim_step(i, volume, used, img, mask, ens, eps, fill,
                                                            out_disp=stack[i + 1])
                if trace is not None:
                    trace(i, {"when": "out", "disp": disp, "unc": unc, "x_start": x_start, "x_next": x_next,
                              "mask": mask.clone()})
                img = x_start if st.time_next < 0 else x_next
        if any_split_plan(self._plans):
            check_split_overflow(dev)
        if self.use_ensemble:
            return ens, stack
        return stack[-1]

    @torch.no_grad()
    def encode_disparity(self, disp: torch.Tensor, mask_gt: Optional[torch.Tensor] = None) -> torch.Tensor:
        """x_T of acv_ddim.py:403-419: quarter-resolution disparity [B,1,h,w] -> [B,48,h,w] in [-1,1].  `mask_gt` (None
        at every call site of the reference; anything that broadcasts against [B,48,h,w]) puts the uniform distribution
        1/48 where it is 0 (:415-417)."""
        disp = _dev_f32(disp, "disp")
        b, h, w = disp.shape[0], disp.shape[-2], disp.shape[-1]
        nb = self.maxdisp // 4
        x = torch.empty((b, nb, h, w), dtype=torch.float32, device=disp.device)
        with torch.cuda.device(disp.device):
            _lib.check(_lib.load().dv_encode_two_hot_f32(disp.data_ptr(), x.data_ptr(), b, nb, h * w,
                                                         _lib.stream_ptr()), "dv_encode_two_hot_f32")
        if mask_gt is not None:
            # the reference selects before the *2-1 rescale; the same fp32 operations on the one constant it selects
            uniform = ((torch.ones((), dtype=torch.float32, device=x.device) / nb) * 2 - 1) * self.scale
            x = torch.where(mask_gt.to(x.device) == 0, uniform, x)
        return x

    @torch.no_grad()
    def attention_logits(self, feat_left: torch.Tensor, feat_right: torch.Tensor) -> torch.Tensor:
        """acv_ddim.py:375-384 (= acv.py:171-197): gwc volume -> patch convs -> attention aggregation -> `att_weights`
        [B,1,D/4,H/4,W/4]."""
        p = self.prepare()
        gwc = build_gwc_volume(feat_left, feat_right, self.maxdisp // 4, self.num_groups)
        att = p.dres1_att(patch_volume(gwc, p.patch_w1, p.patch_w2, p.patch_dil))     # patch, patch_l1..3 (:377-381)
        return p.classif_att(p.dres2_att(att))

    @torch.no_grad()
    def attention_concat_volume(self, feat_left: torch.Tensor, feat_right: torch.Tensor, lazy: bool = False):
        """acv_ddim.py:375-390: gwc volume -> patch convs -> attention aggregation -> logits, then the
        softmax-weighted concat volume (the tensor the DDIM loop filters), [B,64,D/4,H/4,W/4] like the reference's :390.
        ``lazy=True`` (what ``forward`` passes) returns its factors instead (``AttentionConcatVolume``: the first
        aggregation layer reads nothing else, so the 3 GB tensor is not written) -- same default as
        ``build_concat_attention_volume``, so a caller who indexes / clones the result gets a tensor."""
        p = self.prepare()
        att = self.attention_logits(feat_left, feat_right)
        cl = p.concat_b(p.concat_a(feat_left))
        cr = p.concat_b(p.concat_a(feat_right))
        return build_concat_attention_volume(cl, cr, att, self.maxdisp // 4, lazy=lazy)

    def forward(self, left, right, used, disp, mask_gt=None):
        if self.training:
            raise NotImplementedError("the MI355X DiffuVolume path is inference-only (model.eval())")
        with torch.no_grad():
            self.prepare(check_weights=True)
            fl = self.feature_extraction(left)["gwc_feature"]
            fr = self.feature_extraction(right)["gwc_feature"]
            ac_volume = self.attention_concat_volume(fl, fr, lazy=True)
            x_T = self.encode_disparity(disp, mask_gt)
            pred, _ = self.ddim_sample(ac_volume, used, x_T)
        return [pred]


class ACVNet(_HipPlanMixin):
    """The origin network that supplies ``used`` (SceneFlow/models/acv.py:94-260, eval path): same feature CNN, attention
    branch, concat volume and aggregation stack as ACVNet_DDIM, run on the same HIP kernels, without the diffusion loop.
    ``forward(left, right) -> [pred2]``, or ``[pred_attention]`` (the regression of the attention logits, acv.py:246-252)
    when built with ``attn_weights_only=True`` -- the reference builds every module either way, so the ``state_dict``
    (561 keys) is the same and loads with strict=True."""

    def __init__(self, maxdisp: int, attn_weights_only: bool = False, freeze_attn_weights: bool = False):
        super().__init__()
        if maxdisp != 192:
            raise ValueError("ACVNet on the HIP path: maxdisp == 192")
        self.maxdisp, self.attn_weights_only, self.freeze_attn_weights = maxdisp, attn_weights_only, freeze_attn_weights
        self.num_groups, self.concat_channels = 40, 32
        self.feature_extraction = FeatureExtraction()
        self.concatconv = nn.Sequential(_cb2(320, 128, 3, 1, 1, 1), nn.ReLU(inplace=True),
                                        nn.Conv2d(128, self.concat_channels, 1, bias=False))
        self.patch = nn.Conv3d(40, 40, (1, 3, 3), 1, (0, 1, 1), 1, groups=40, bias=False)
        self.patch_l1 = nn.Conv3d(8, 8, (1, 3, 3), 1, (0, 1, 1), 1, groups=8, bias=False)
        self.patch_l2 = nn.Conv3d(16, 16, (1, 3, 3), 1, (0, 2, 2), 2, groups=16, bias=False)
        self.patch_l3 = nn.Conv3d(16, 16, (1, 3, 3), 1, (0, 3, 3), 3, groups=16, bias=False)
        self.dres1_att_ = nn.Sequential(_cb3(40, 32, 3, 1, 1), nn.ReLU(inplace=True), _cb3(32, 32, 3, 1, 1))
        self.dres2_att_ = Hourglass(32)
        self.classif_att_ = ACVNet_DDIM._classifier()
        self.dres0 = nn.Sequential(_cb3(64, 32, 3, 1, 1), nn.ReLU(inplace=True),
                                   _cb3(32, 32, 3, 1, 1), nn.ReLU(inplace=True))
        self.dres1 = nn.Sequential(_cb3(32, 32, 3, 1, 1), nn.ReLU(inplace=True), _cb3(32, 32, 3, 1, 1))
        self.dres2 = Hourglass(32)
        self.dres3 = Hourglass(32)
        self.classif0 = ACVNet_DDIM._classifier()
        self.classif1 = ACVNet_DDIM._classifier()
        self.classif2 = ACVNet_DDIM._classifier()
        ACVNet_DDIM._init_weights(self)
        self._plans: Optional[_Plans] = None

    attention_logits = ACVNet_DDIM.attention_logits
    attention_concat_volume = ACVNet_DDIM.attention_concat_volume
    _aggregate = ACVNet_DDIM._aggregate

    def forward(self, left, right):
        if self.training:
            raise NotImplementedError("the MI355X path is inference-only (model.eval())")
        with torch.no_grad():
            self.prepare(check_weights=True)
            fl = self.feature_extraction(left)["gwc_feature"]
            fr = self.feature_extraction(right)["gwc_feature"]
            if self.attn_weights_only:                                           # acv.py:246-252
                cost = self.attention_logits(fl, fr)
            else:
                cost = self._aggregate(self.attention_concat_volume(fl, fr, lazy=True), None)
            pred2, _ = upsample_softmax_regress(cost, want_uncertainty=False)
            if any_split_plan(self._plans):
                check_split_overflow(pred2.device)
        return [pred2]


__models__ = {"acvnet": ACVNet, "acvnet_ddim": ACVNet_DDIM}
