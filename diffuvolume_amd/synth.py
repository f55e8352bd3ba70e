"""Deterministic synthetic weights and stereo inputs (no checkpoints or datasets ship with
the reference: README.md:8, .MISSING_LARGE_BLOBS).  Every tensor is drawn from its own
CPU generator seeded by crc32(key) ^ seed, so the values depend only on (seed, key, shape)
-- not on module construction order -- and are identical here and on the GPU box.  Used by
oracle/make_golden.py (loaded into the imported reference), the tests and bench.py."""
from __future__ import annotations

import math
import zlib
from typing import Dict, Mapping

import torch


def _gen(seed: int, key: str) -> torch.Generator:
    g = torch.Generator(device="cpu")
    g.manual_seed((zlib.crc32(key.encode()) ^ (seed * 0x9E3779B1)) & 0x7FFFFFFF)
    return g


def synth_state_dict(template: Mapping[str, torch.Tensor], seed: int = 0, logit_gain: float = 1.0,
                     scale: Mapping[str, float] | None = None) -> Dict[str, torch.Tensor]:
    """Random but well-conditioned values for every entry of ``template`` (a state_dict):
    conv weights ~ N(0, sqrt(2/(k^3*Cout))) as the reference initialises them
    (acv_ddim.py:224-238), BatchNorm with NON-trivial affine and running statistics,
    xavier-uniform Linear weights.  float64 schedule buffers are kept.  ``logit_gain``
    scales the single-channel classifier heads (sharper or flatter softmax); ``scale`` multiplies
    named tensors."""
    keys = set(template.keys())
    out: Dict[str, torch.Tensor] = {}
    for key, ref in template.items():
        g = _gen(seed, key)
        shape = tuple(ref.shape)
        leaf = key.rsplit(".", 1)[-1]
        stem = key.rsplit(".", 1)[0]
        is_bn = (stem + ".running_mean") in keys
        if ref.dtype == torch.float64 or not ref.dtype.is_floating_point:
            out[key] = ref.clone()                       # schedule buffers, num_batches_tracked
        elif leaf == "running_mean":
            out[key] = torch.randn(shape, generator=g) * 0.1
        elif leaf == "running_var":
            out[key] = torch.rand(shape, generator=g) + 0.5
        elif is_bn and leaf == "weight":
            out[key] = torch.rand(shape, generator=g) * 0.4 + 0.8
        elif is_bn and leaf == "bias":
            out[key] = torch.randn(shape, generator=g) * 0.1
        elif leaf == "weight" and ref.dim() >= 3:        # Conv2d / Conv3d / ConvTranspose3d
            kprod = 1
            for s in shape[2:]:
                kprod *= s
            std = math.sqrt(2.0 / (kprod * shape[0]))
            w = torch.randn(shape, generator=g) * std
            if shape[0] == 1 and ref.dim() == 5:
                w = w * logit_gain
            out[key] = w
        elif leaf == "weight" and ref.dim() == 2:        # Linear
            a = math.sqrt(6.0 / (shape[0] + shape[1]))
            out[key] = (torch.rand(shape, generator=g) * 2 - 1) * a
        elif leaf == "bias":
            out[key] = torch.randn(shape, generator=g) * 0.02
        else:
            out[key] = torch.randn(shape, generator=g) * 0.1
        if scale and key in scale:
            out[key] = out[key] * scale[key]       # e.g. tame an untrained residual head
        out[key] = out[key].to(ref.dtype)
    return out


def synth_features(b: int, c: int, h: int, w: int, seed: int, shifts=(6, 24, 60)) -> Dict[str, torch.Tensor]:
    """Left/right feature maps with a real correlation ridge: right = left shifted by a
    per-image disparity (in feature pixels) plus noise."""
    g = _gen(seed, f"features{b}x{c}x{h}x{w}")
    left = torch.randn(b, c, h, w, generator=g)
    right = torch.empty_like(left)
    for i in range(b):
        d = shifts[i % len(shifts)] // 4 if w > 16 else 1
        right[i] = torch.roll(left[i], shifts=-d, dims=-1)
    right = right + 0.05 * torch.randn(b, c, h, w, generator=g)
    return {"left": left, "right": right}


def synth_stereo_batch(b: int, h: int, w: int, seed: int = 0, shifts=(6, 24, 60)) -> Dict[str, torch.Tensor]:
    """left = randn, right = left rolled by d0 px + noise, gt = d0 + randn clamped
    to (0,192), used = gt + 0.5 randn (stand-in for the origin network), disp = bilinear/4."""
    import torch.nn.functional as F
    g = _gen(seed, f"stereo{b}x{h}x{w}")
    left = torch.randn(b, 3, h, w, generator=g)
    right = torch.empty_like(left)
    gt = torch.empty(b, h, w)
    for i in range(b):
        d0 = shifts[i % len(shifts)]
        right[i] = torch.roll(left[i], shifts=-d0, dims=-1)
        gt[i] = d0 + torch.randn(h, w, generator=g)
    right = right + 0.05 * torch.randn(b, 3, h, w, generator=g)
    gt = gt.clamp(0.5, 191.0)
    used = (gt + 0.5 * torch.randn(b, h, w, generator=g)).clamp(0.0, 191.0)
    disp = F.interpolate(used.clamp(0, 191).unsqueeze(1), size=(h // 4, w // 4), mode="bilinear") / 4
    return {"left": left, "right": right, "gt": gt, "used": used, "disp": disp}


def synth_hot_inputs(batch: int, h: int, w: int, seed: int, shifts=(6, 24, 60)) -> Dict[str, torch.Tensor]:
    """Inputs of the hot path at quarter resolution h x w: 320-channel gwc features and 32-channel
    concat features with a real correlation ridge (right = left rolled by the pair's disparity + noise), attention
    logits, full-resolution ground truth, the origin network's stand-in ``used`` and its quarter-resolution
    encoding input ``dq``.  CPU tensors; bench.py and the full-size parity tests move them to the device."""
    import torch.nn.functional as F
    g = _gen(seed, f"bench{batch}x{h}x{w}")

    def pair(c):
        left = torch.randn(batch, c, h, w, generator=g)
        right = torch.stack([torch.roll(left[i], -(shifts[i % 3] // 4), dims=-1) for i in range(batch)])
        return left, right + 0.05 * torch.randn(batch, c, h, w, generator=g)

    fl, fr = pair(320)
    cl, cr = pair(32)
    att = torch.randn(batch, 1, 48, h, w, generator=g) * 2
    gt = torch.stack([shifts[i % 3] + torch.randn(4 * h, 4 * w, generator=g) for i in range(batch)]).clamp(0.5, 191)
    used = (gt + 0.5 * torch.randn(batch, 4 * h, 4 * w, generator=g)).clamp(0, 191)
    dq = F.interpolate(used.unsqueeze(1), size=(h, w), mode="bilinear") / 4
    return dict(fl=fl, fr=fr, cl=cl, cr=cr, att=att, gt=gt, used=used, dq=dq)


class NoiseTape:
    """Deterministic replacement for the DDIM loop's random draws (acv_ddim.py:354 'eps' =
    randn_like(img), :360 'fill' = rand_like): the k-th draw of each kind comes from its own
    seeded CPU generator in float64 and is cast to the requested dtype, so the reference
    (patched torch.randn_like / rand_like), the CPU oracle and the HIP path all see the same
    numbers whatever their device."""

    def __init__(self, seed: int):
        self.seed = seed
        self.count = {}

    def __call__(self, kind: str, shape, dtype) -> torch.Tensor:
        """kind 'fill' is uniform [0,1); every other kind ('eps', 'x_T', 'q', ...) is standard normal."""
        k = self.count.get(kind, 0)
        self.count[kind] = k + 1
        g = _gen(self.seed, f"{kind}{k}")
        fn = torch.rand if kind == "fill" else torch.randn
        return fn(tuple(shape), generator=g, dtype=torch.float64).to(dtype)


def toy_update_block(net_list, inp_list, corr=None, flow=None, iter32=True, iter16=True, iter08=True, update=True):
    """Deterministic stand-in for IGEV's ConvGRU update block (KITTI15/core/update.py:104-142, a 2-D module that
    is out of scope): same call signature and return convention, so the reference's and this build's DDIM
    loops can be driven with identical dynamics in the parity fixtures."""
    if not update:
        return net_list
    delta = 0.3 * torch.tanh(corr.mean(dim=1, keepdim=True)) - 0.02 * flow
    return net_list, torch.ones_like(flow), delta


def toy_upsample_disp(flow, up_mask, stem_2x):
    """Stand-in for IGEVStereo_ddim.upsample_disp (:206-214): x4 bilinear upsampling of 4*flow."""
    import torch.nn.functional as F
    return F.interpolate(flow * 4.0, scale_factor=4, mode="bilinear", align_corners=False)


def _stub_stage(cin, cout, stride):
    from torch import nn
    return nn.Sequential(nn.Conv2d(cin, cout, 3, stride, 1, bias=False), nn.BatchNorm2d(cout), nn.ReLU6())


class StubMobileNetV2(__import__("torch").nn.Module):
    """Stand-in for ``timm.create_model('mobilenetv2_100', features_only=True)`` (KITTI15/core/extractor.py:331): the
    attributes IGEV's ``Feature`` takes from it -- ``conv_stem`` / ``bn1`` / ``act1`` and seven ``blocks`` with
    MobileNetV2's channel counts and strides (16 @1/2, 24 @1/4, 32 @1/8, 64 + 96 @1/16, 160 @1/32, 320) -- each stage
    one 3x3 conv + BN + ReLU6.  Neither timm nor its pretrained weights exist offline; the goldens and the tests
    drive the reference and this build with this same module (weights from ``synth_state_dict``)."""

    def __init__(self):
        from torch import nn
        super().__init__()
        self.conv_stem = nn.Conv2d(3, 32, 3, 2, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(32)
        self.act1 = nn.ReLU6()
        self.blocks = nn.Sequential(_stub_stage(32, 16, 1), _stub_stage(16, 24, 2), _stub_stage(24, 32, 2),
                                    _stub_stage(32, 64, 2), _stub_stage(64, 96, 1), _stub_stage(96, 160, 2),
                                    _stub_stage(160, 320, 1))
