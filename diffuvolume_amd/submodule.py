"""L1 ops of the DiffuVolume hot path behind the reference's function signatures.

Same names, argument meaning and error behaviour as SceneFlow/models/submodule.py
(``build_gwc_volume`` :228-238, ``build_concat_volume`` :180-191,
``disparity_regression`` :173-177) and their KITTI12 / KITTI15 twins; the work is
done by the HIP kernels of libdiffuvolume_hip.so on the current stream.  The three
builders / the regression are differentiable in the reference (they are used in training):
when autograd is recording and an input requires grad they dispatch to a differentiable
PyTorch statement of the same function; everything else is inference only and
runs on the MI355X or raises.
"""
from __future__ import annotations

import ctypes
import os
from typing import Optional, Tuple

import torch

from . import _lib
from .profiling import S2PP_MULT_REDUCTION, WINO3_MULT_REDUCTION, WINO_MULT_REDUCTION, timed

__all__ = ["build_gwc_volume", "build_concat_volume", "build_concat_attention_volume", "AttentionConcatVolume",
           "volume_factors",
           "disparity_regression", "upsample_softmax_regress", "Conv3dPlan", "Conv2dPlan", "Deconv3dPlan",
           "window_attention", "feature_gate", "softmax_regress", "refine_inputs", "patch_volume", "ACT_NONE", "ACT_RELU", "ACT_MISH", "ACT_LEAKY", "ACT_SIGMOID", "ACT_TANH"]

ACT_NONE, ACT_RELU, ACT_MISH, ACT_LEAKY, ACT_SIGMOID, ACT_TANH = 0, 1, 2, 3, 4, 5     # last two: 2-D convs only


_CONV_PRECISION = None


def set_default_conv_precision(precision=None) -> None:
    """Override DV_CONV_PRECISION for plans built from now on ('f32', 'f32_direct', 'f16x3' or None = environment)."""
    global _CONV_PRECISION
    if precision not in (None, "f32", "f32_direct", "f16x3"):
        raise ValueError("precision must be 'f32', 'f32_direct', 'f16x3' or None")
    _CONV_PRECISION = precision


_OVERFLOW_FLAGS = {}


def split_overflow_flag(device) -> torch.Tensor:
    """Device int32 the split-fp16 conv kernels raise when an activation leaves the fp16 range."""
    key = torch.device(device).index if torch.device(device).index is not None else torch.cuda.current_device()
    if key not in _OVERFLOW_FLAGS:
        _OVERFLOW_FLAGS[key] = torch.zeros(1, dtype=torch.int32, device=torch.device("cuda", key))
    return _OVERFLOW_FLAGS[key]


def check_split_overflow(device) -> None:
    """Raise (after one device sync) if a split-fp16 convolution saw an out-of-range activation since the
    last check; the wrappers call it before they hand results back."""
    flag = split_overflow_flag(device)
    if int(flag.item()) != 0:
        flag.zero_()
        raise _lib.DiffuVolumeError("an activation exceeded the split-fp16 range (|x| >= 2.6e5 or NaN): "
                                    "rebuild the plans with precision='f32' (DV_CONV_PRECISION=f32)")


def default_conv_precision() -> str:
    """'f32' = fp32 MFMA everywhere, the 3x3x3 stride-1 layers in the Winograd F(2x2,3x3) form
    (csrc/conv3d_wino.hip; default); 'f32_direct' = fp32 MFMA with direct taps everywhere (csrc/conv3d.hip);
    'f16x3' = the split-fp16 kernel for the 3x3x3 stride-1 layers it covers (csrc/conv3d_f16x3.hip; inputs must
    stay below 2.6e5 in magnitude).  Set with DV_CONV_PRECISION, set_default_conv_precision() or per plan."""
    import os
    return _CONV_PRECISION or os.environ.get("DV_CONV_PRECISION", "f32")


def _dev_f32(t: torch.Tensor, name: str) -> torch.Tensor:
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name} must be a tensor")
    if not t.is_cuda:
        raise _lib.DiffuVolumeError(
            f"{name} is on {t.device}: the DiffuVolume hot path only runs on the MI355X "
            "(HIP kernels, no CPU fallback)")
    if t.dtype != torch.float32:
        raise TypeError(f"{name} must be float32, got {t.dtype}")
    if torch.is_grad_enabled() and t.requires_grad:
        raise NotImplementedError("the HIP hot path is inference-only; call it under torch.no_grad()")
    return t.contiguous()


def _wants_grad(*tensors) -> bool:
    return torch.is_grad_enabled() and any(isinstance(t, torch.Tensor) and t.requires_grad for t in tensors)


def _shift_bank(t: torch.Tensor, maxdisp: int) -> torch.Tensor:
    """[B,C,H,W] -> [B,C,D,H,W] with bank[..., d, y, x] = t[..., y, x - d] (0 where x < d): every disparity shift
    of the target features as one strided view of a left-padded copy (differentiable)."""
    w = t.shape[-1]
    win = torch.nn.functional.pad(t, (maxdisp - 1, 0)).unfold(3, w, 1)      # [B,C,H,D,W], window k starts at D-1-d
    return win.flip(3).permute(0, 1, 3, 2, 4)


def _valid_wedge(maxdisp: int, w: int, device) -> torch.Tensor:
    """[D,1,W] bool: x >= d (the reference leaves `new_zeros` where the shifted target has no pixel)."""
    return (torch.arange(w, device=device).view(1, 1, w) >= torch.arange(maxdisp, device=device).view(maxdisp, 1, 1))


def _gwc_volume_autograd(ref, tgt, maxdisp, num_groups):
    """Differentiable statement of build_gwc_volume (submodule.py:209-238) for training / autograd callers:
    products against the shift bank, mean over the channels of a group, exact zeros in the x < d wedge."""
    b, c, h, w = ref.shape
    cpg = c // num_groups
    valid = _valid_wedge(maxdisp, w, ref.device)
    step = max(1, (64 << 20) // max(1, b * cpg * maxdisp * h * w))           # groups per slab (<= ~256 MB of products)
    slabs = []
    for g0 in range(0, num_groups, step):
        g1 = min(num_groups, g0 + step)
        r = ref[:, g0 * cpg:g1 * cpg]
        prod = r.unsqueeze(2) * _shift_bank(tgt[:, g0 * cpg:g1 * cpg], maxdisp)
        slabs.append(prod.reshape(b, g1 - g0, cpg, maxdisp, h, w).mean(dim=2))
    vol = torch.cat(slabs, dim=1)
    return torch.where(valid, vol, torch.zeros((), dtype=vol.dtype, device=vol.device)).contiguous()


def _concat_volume_autograd(ref, tgt, maxdisp, zero_left):
    """Differentiable statement of build_concat_volume (submodule.py:180-191; KITTI12 :86-97 with zero_left)."""
    b, c, h, w = ref.shape
    left = ref.unsqueeze(2).expand(b, c, maxdisp, h, w)
    if zero_left:
        left = torch.where(_valid_wedge(maxdisp, w, ref.device), left, torch.zeros((), dtype=ref.dtype, device=ref.device))
    return torch.cat((left, _shift_bank(tgt, maxdisp)), dim=1).contiguous()


def build_gwc_volume(refimg_fea: torch.Tensor, targetimg_fea: torch.Tensor, maxdisp: int,
                     num_groups: int) -> torch.Tensor:
    """[B,C,H,W] x2 -> [B,num_groups,maxdisp,H,W] (submodule.py:228-238)."""
    if _wants_grad(refimg_fea, targetimg_fea):
        if refimg_fea.dim() != 4 or refimg_fea.shape != targetimg_fea.shape:
            raise RuntimeError(f"feature shapes differ or are not 4-D: {tuple(refimg_fea.shape)} vs {tuple(targetimg_fea.shape)}")
        assert refimg_fea.shape[1] % num_groups == 0          # submodule.py:211
        return _gwc_volume_autograd(refimg_fea, targetimg_fea, maxdisp, num_groups)
    ref = _dev_f32(refimg_fea, "refimg_fea")
    tgt = _dev_f32(targetimg_fea, "targetimg_fea")
    if ref.dim() != 4 or ref.shape != tgt.shape:
        raise RuntimeError(f"feature shapes differ or are not 4-D: {tuple(ref.shape)} vs {tuple(tgt.shape)}")
    b, c, h, w = ref.shape
    assert c % num_groups == 0          # submodule.py:211
    out = torch.empty((b, num_groups, maxdisp, h, w), dtype=torch.float32, device=ref.device)
    if out.numel() == 0:                # empty batch / empty image: the reference returns the empty volume
        return out
    lib = _lib.load()
    with torch.cuda.device(ref.device):
        timed("gwc_volume", 2.0 * out.numel() * (c // num_groups), 4.0 * (2 * ref.numel() + out.numel()),
              lambda: _lib.check(lib.dv_gwc_volume_f32(ref.data_ptr(), tgt.data_ptr(), out.data_ptr(), b, c, h, w,
                                                       maxdisp, num_groups, _lib.stream_ptr()),
                                 "dv_gwc_volume_f32"))
    return out


def build_concat_volume(refimg_fea: torch.Tensor, targetimg_fea: torch.Tensor, maxdisp: int,
                        zero_left: bool = False) -> torch.Tensor:
    """[B,C,H,W] x2 -> [B,2C,maxdisp,H,W] (submodule.py:180-191).  ``zero_left=True``
    is the KITTI12 flavour (KITTI12/models/submodule.py:86-97)."""
    if _wants_grad(refimg_fea, targetimg_fea):
        if refimg_fea.dim() != 4 or refimg_fea.shape != targetimg_fea.shape:
            raise RuntimeError(f"feature shapes differ or are not 4-D: {tuple(refimg_fea.shape)} vs {tuple(targetimg_fea.shape)}")
        return _concat_volume_autograd(refimg_fea, targetimg_fea, maxdisp, bool(zero_left))
    ref = _dev_f32(refimg_fea, "refimg_fea")
    tgt = _dev_f32(targetimg_fea, "targetimg_fea")
    if ref.dim() != 4 or ref.shape != tgt.shape:
        raise RuntimeError(f"feature shapes differ or are not 4-D: {tuple(ref.shape)} vs {tuple(tgt.shape)}")
    b, c, h, w = ref.shape
    out = torch.empty((b, 2 * c, maxdisp, h, w), dtype=torch.float32, device=ref.device)
    if out.numel() == 0:
        return out
    lib = _lib.load()
    with torch.cuda.device(ref.device):
        _lib.check(lib.dv_concat_volume_f32(ref.data_ptr(), tgt.data_ptr(), out.data_ptr(), b, c, h, w,
                                            maxdisp, int(bool(zero_left)), _lib.stream_ptr()),
                   "dv_concat_volume_f32")
    return out


class AttentionConcatVolume:
    """``F.softmax(att_weights, dim=2) * build_concat_volume(left, right)`` (acv_ddim.py:388-390) kept as its three
    FACTORS -- p = softmax(att) [B,D,h,w] and the two feature maps [B,C,h,w] -- instead of the [B,2C,D,h,w] tensor
    (3.0 GB at batch 8).  The first aggregation layer of every DDIM step reads the factors (``Rank1FilterPlan``), so on
    the model's own path the tensor is never written; ``tensor()`` materialises it (once) for any other consumer.
    The factors are private copies: nothing the caller does to its feature tensors afterwards can reach them."""

    def __init__(self, p_att: torch.Tensor, ref: torch.Tensor, tgt: torch.Tensor):
        self.p_att, self.ref, self.tgt = p_att, ref, tgt
        b, d, h, w = p_att.shape
        self.shape = torch.Size((b, 2 * ref.shape[1], d, h, w))
        self.device, self.dtype = p_att.device, p_att.dtype
        self._tensor: Optional[torch.Tensor] = None
        self._rank1_tables = None

    def dim(self) -> int:
        return 5

    def size(self, i: Optional[int] = None):
        return self.shape if i is None else self.shape[i]

    def numel(self) -> int:
        return self.shape.numel()

    def tensor(self) -> torch.Tensor:
        """The materialised [B,2C,D,h,w] volume (``dv_concat_attn_volume_f32`` run on p directly)."""
        if self._tensor is None:
            b, c2, d, h, w = self.shape
            out = torch.empty(tuple(self.shape), dtype=torch.float32, device=self.device)
            if out.numel() == 0:
                self._tensor = out
                return out
            lib = _lib.load()
            with torch.cuda.device(self.device):
                timed("concat_attn_volume", 2.0 * out.numel(),
                      4.0 * (2 * self.ref.numel() + self.p_att.numel() + out.numel()),
                      lambda: _lib.check(lib.dv_concat_prob_volume_f32(self.ref.data_ptr(), self.tgt.data_ptr(),
                                                                       self.p_att.data_ptr(), out.data_ptr(), b, c2 // 2,
                                                                       h, w, d, _lib.stream_ptr()),
                                         "dv_concat_prob_volume_f32"))
            self._tensor = out
        return self._tensor


def _attention_factors(refimg_fea, targetimg_fea, att_weights, maxdisp):
    ref = _dev_f32(refimg_fea, "refimg_fea")
    tgt = _dev_f32(targetimg_fea, "targetimg_fea")
    att = _dev_f32(att_weights, "att_weights")
    b, c, h, w = ref.shape
    if ref.shape != tgt.shape or tuple(att.shape) != (b, 1, maxdisp, h, w):
        raise RuntimeError("shape mismatch between features and attention weights")
    p_att = torch.empty((b, maxdisp, h, w), dtype=torch.float32, device=ref.device)
    if p_att.numel() == 0:              # empty batch / image: the reference's product is the empty volume
        return ref, tgt, att, p_att
    with torch.cuda.device(ref.device):
        _lib.check(_lib.load().dv_softmax_d_f32(att.data_ptr(), p_att.data_ptr(), b, maxdisp, h * w, _lib.stream_ptr()),
                   "dv_softmax_d_f32")
    return ref, tgt, att, p_att


def build_concat_attention_volume(refimg_fea: torch.Tensor, targetimg_fea: torch.Tensor,
                                  att_weights: torch.Tensor, maxdisp: int, lazy: bool = False):
    """``F.softmax(att_weights, dim=2) * build_concat_volume(...)`` in one pass
    (acv_ddim.py:388-390).  att_weights [B,1,maxdisp,H,W] are logits.

    ``lazy=False`` (default): the [B,2C,maxdisp,H,W] tensor, with private copies of its factors riding along so that
    ``ACVNet_DDIM`` can run its first aggregation layer on them; the ride-along is void the moment the tensor is
    edited in place (``Rank1FilterPlan.applies`` compares ``_version``).  ``lazy=True``: an ``AttentionConcatVolume``
    -- the factors only, nothing of the volume written -- which ``ACVNet_DDIM.ddim_sample / model_predictions`` accept
    in the tensor's place (what ``ACVNet_DDIM.forward`` passes itself)."""
    ref, tgt, att, p_att = _attention_factors(refimg_fea, targetimg_fea, att_weights, maxdisp)
    b, c, h, w = ref.shape
    # private copies (2 x 31 MB at batch 8): `ref` / `tgt` may BE the caller's tensors (`contiguous()` of a contiguous
    # tensor is the tensor itself), and the rank-1 tables are built from them later, on the first DDIM step
    handle = AttentionConcatVolume(p_att, ref.clone(), tgt.clone())
    if lazy:
        return handle
    out = torch.empty((b, 2 * c, maxdisp, h, w), dtype=torch.float32, device=ref.device)
    if out.numel() == 0:
        return out
    lib = _lib.load()
    with torch.cuda.device(ref.device):
        timed("concat_attn_volume", 2.0 * out.numel(), 4.0 * (2 * ref.numel() + att.numel() + out.numel()),
              lambda: _lib.check(lib.dv_concat_attn_volume_f32(ref.data_ptr(), tgt.data_ptr(), att.data_ptr(),
                                                               out.data_ptr(), b, c, h, w, maxdisp,
                                                               _lib.stream_ptr()), "dv_concat_attn_volume_f32"))
    out._dv_factors = handle
    out._dv_factors_version = out._version
    return out


def volume_factors(volume) -> Optional[AttentionConcatVolume]:
    """The factors a volume may be replaced by: the handle itself, or the ride-along of a tensor that
    ``build_concat_attention_volume`` returned and that has not been written to since (any in-place operation on the
    tensor bumps ``_version``; a clone / slice / arithmetic result is a new tensor without the attribute)."""
    if isinstance(volume, AttentionConcatVolume):
        return volume
    fac = getattr(volume, "_dv_factors", None)
    if fac is None or getattr(volume, "_dv_factors_version", None) != volume._version:
        return None
    return fac


def disparity_regression(x: torch.Tensor, maxdisp: int, keepdim: bool = False) -> torch.Tensor:
    """[B,D,H,W] probabilities -> sum_d d*p_d (submodule.py:173-177; keepdim=True is the
    KITTI15 flavour, core/submodule.py:219-223)."""
    assert len(x.shape) == 4            # submodule.py:174
    if _wants_grad(x):                  # training: the differentiable statement (sum_d d * p_d)
        if x.shape[1] != maxdisp:
            raise RuntimeError(f"The size of tensor a ({x.shape[1]}) must match the size of tensor b ({maxdisp}) "
                               "at non-singleton dimension 1")
        k = torch.arange(0, maxdisp, dtype=x.dtype, device=x.device).view(1, maxdisp, 1, 1)
        return torch.sum(x * k, 1, keepdim=keepdim)
    x = _dev_f32(x, "x")
    b, d, h, w = x.shape
    if d != maxdisp:
        raise RuntimeError(f"The size of tensor a ({d}) must match the size of tensor b ({maxdisp}) "
                           "at non-singleton dimension 1")
    out = torch.empty((b, h, w), dtype=torch.float32, device=x.device)
    if out.numel() == 0:
        return out.unsqueeze(1) if keepdim else out
    lib = _lib.load()
    with torch.cuda.device(x.device):
        _lib.check(lib.dv_disparity_regression_f32(x.data_ptr(), out.data_ptr(), b, d, h, w,
                                                   _lib.stream_ptr()), "dv_disparity_regression_f32")
    return out.unsqueeze(1) if keepdim else out


def upsample_softmax_regress(cost: torch.Tensor, want_uncertainty: bool = True,
                             align_corners: bool = False, out_disp: Optional[torch.Tensor] = None
                             ) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
    """cost [B,1,D,h,w] (or [B,D,h,w]) -> (disp [B,4h,4w], uncertainty | None): trilinear x4,
    softmax over 4D bins, soft-argmax and sum_k |disp-k| p_k (acv_ddim.py:267-270, :325-329)."""
    cost = _dev_f32(cost, "cost")
    if cost.dim() == 5:
        if cost.shape[1] != 1:
            raise RuntimeError("cost must have one channel")
        cost = cost[:, 0]
    b, d, h, w = cost.shape
    if out_disp is None:
        disp = torch.empty((b, 4 * h, 4 * w), dtype=torch.float32, device=cost.device)
    else:                                   # e.g. a slice of the per-step stack of ddim_sample
        disp = out_disp
        if tuple(disp.shape) != (b, 4 * h, 4 * w) or disp.dtype != torch.float32 or not disp.is_contiguous() \
                or disp.device != cost.device:
            raise RuntimeError("out_disp must be a contiguous float32 [B,4h,4w] tensor on the cost's device")
    unc = torch.empty_like(disp) if want_uncertainty else None
    lib = _lib.load()
    with torch.cuda.device(cost.device):
        timed("upsample_softmax_regress", 0.0, 4.0 * (cost.numel() + 2 * disp.numel()),
              lambda: _lib.check(lib.dv_upsample_softmax_regress_f32(cost.data_ptr(), disp.data_ptr(),
                                                                     _lib.ptr(unc), b, d, h, w,
                                                                     int(bool(align_corners)), _lib.stream_ptr()),
                                 "dv_upsample_softmax_regress_f32"))
    return disp, unc


def softmax_regress(cost: torch.Tensor) -> torch.Tensor:
    """`disparity_regression(F.softmax(cost, 1), D)` without upsampling (igev_stereo_ddim.py:382-383):
    cost [B,1,D,H,W] or [B,D,H,W] -> [B,H,W]."""
    cost = _dev_f32(cost, "cost")
    if cost.dim() == 5:
        if cost.shape[1] != 1:
            raise RuntimeError("cost must have one channel")
        cost = cost[:, 0]
    b, d, h, w = cost.shape
    disp = torch.empty((b, h, w), dtype=torch.float32, device=cost.device)
    lib = _lib.load()
    with torch.cuda.device(cost.device):
        timed("softmax_regress", 0.0, 4.0 * (cost.numel() + disp.numel()),
              lambda: _lib.check(lib.dv_softmax_regress_f32(cost.data_ptr(), disp.data_ptr(), b, d, h, w,
                                                            _lib.stream_ptr()), "dv_softmax_regress_f32"))
    return disp


class Conv3dPlan:
    """A Conv3d(bias=False)[+BatchNorm3d eval][+activation] layer prepared for the
    implicit-GEMM kernel: weights repacked once on the device, BN folded to a
    per-channel scale/bias applied in the epilogue (submodule.py:94-97)."""

    # 3x3x3 stride-1 layers: the F(2x2x2,3x3x3) kernel (False: the in-plane F(2x2,3x3) kernel everywhere; tests / A-B runs)
    # from WINO3_MIN_CIN input channels on -- measured at batch 8 (tools/ab_wino3.py): 64 -> 64 -6.5 %, 128 -> 128 -7.5 %,
    # 32 -> 32 +1.5 % (eight chunks do not amortise a block's prologue and depth exchange)
    WINO3 = True
    WINO3_MIN_CIN = 64

    def __init__(self, weight: torch.Tensor, bn: Optional[Tuple[torch.Tensor, ...]] = None,
                 stride: int = 1, act: int = ACT_NONE, bias: Optional[torch.Tensor] = None,
                 eps: float = 1e-5, precision: Optional[str] = None):
        w = _dev_f32(weight.detach(), "weight")
        self.cout, self.cin, k = w.shape[0], w.shape[1], w.shape[2]
        if tuple(w.shape[2:]) != (k, k, k) or k not in (1, 3):
            raise _lib.DiffuVolumeError(f"unsupported Conv3d kernel {tuple(w.shape[2:])}")
        self.k, self.stride, self.act = k, stride, act
        precision = precision or default_conv_precision()
        if precision not in ("f32", "f32_direct", "f16x3"):
            raise ValueError("precision must be 'f32' (v_mfma_f32_16x16x4_f32; Winograd F(2x2,3x3) in-plane for the "
                             "3x3x3 stride-1 layers), 'f32_direct' (the same instruction, direct taps everywhere) "
                             "or 'f16x3' (split-fp16 MFMA)")
        # the split-fp16 kernel covers the 3x3x3 stride-1 layers (32 output channels per block; wider
        # layers are split over the grid); the single-channel head stays on its vector-ALU kernel
        self.split = precision == "f16x3" and k == 3 and stride == 1 and self.cout > 1
        self.wino = precision == "f32" and k == 3 and stride == 1 and self.cout > 1
        lib = _lib.load()
        # 3x3x3 stride 2, 64 output channels per block: the polyphase minimal-filtering form (csrc/conv3d_s2pp.hip);
        # a call with a filter prologue falls back to the direct kernel (both weight images are kept)
        self.s2pp = (precision == "f32" and k == 3 and stride == 2 and os.environ.get("DV_S2PP", "1") != "0"
                     and bool(lib.dv_conv3d_s2pp_supported(self.cin, self.cout, 4, 4, 4)))
        with torch.cuda.device(w.device):
            if self.s2pp:
                self.wpacked_pp = torch.empty(lib.dv_conv3d_s2pp_packed_floats(self.cin, self.cout), dtype=torch.float32,
                                              device=w.device)
                _lib.check(lib.dv_conv3d_s2pp_pack_weights_f32(w.data_ptr(), self.wpacked_pp.data_ptr(), self.cin,
                                                               self.cout, _lib.stream_ptr()),
                           "dv_conv3d_s2pp_pack_weights_f32")
            if self.wino:
                n = lib.dv_conv3d_wino_packed_floats(self.cin, self.cout)
                self.wpacked = torch.empty(n, dtype=torch.float32, device=w.device)
                _lib.check(lib.dv_conv3d_wino_pack_weights_f32(w.data_ptr(), self.wpacked.data_ptr(), self.cin,
                                                               self.cout, _lib.stream_ptr()),
                           "dv_conv3d_wino_pack_weights_f32")
                # the F(2x2x2,3x3x3) image (csrc/conv3d_wino3.hip): every call without a filter prologue
                self.wpacked3 = torch.empty(lib.dv_conv3d_wino3_packed_floats(self.cin, self.cout), dtype=torch.float32,
                                            device=w.device)
                _lib.check(lib.dv_conv3d_wino3_pack_weights_f32(w.data_ptr(), self.wpacked3.data_ptr(), self.cin,
                                                                self.cout, _lib.stream_ptr()),
                           "dv_conv3d_wino3_pack_weights_f32")
            elif self.split:
                nbytes = lib.dv_conv3d_f16x3_packed_bytes(self.cin, self.cout)
                self.wpacked = torch.empty(nbytes // 2, dtype=torch.float16, device=w.device)
                _lib.check(lib.dv_conv3d_f16x3_pack_weights(w.data_ptr(), self.wpacked.data_ptr(), self.cin,
                                                            self.cout, _lib.stream_ptr()),
                           "dv_conv3d_f16x3_pack_weights")
            else:
                n = lib.dv_conv3d_packed_floats(self.cin, self.cout, k)
                self.wpacked = torch.empty(n, dtype=torch.float32, device=w.device)
                _lib.check(lib.dv_conv3d_pack_weights_f32(w.data_ptr(), self.wpacked.data_ptr(), self.cin,
                                                          self.cout, k, _lib.stream_ptr()),
                           "dv_conv3d_pack_weights_f32")
        self.scale, self.shift = _fold_bn(bn, bias, self.cout, w.device, eps)

    def out_shape(self, shape):
        b, _, d, h, w = shape
        s = self.stride
        f = (lambda n: (n - 1) // s + 1)
        return (b, self.cout, f(d), f(h), f(w))

    def __call__(self, x: torch.Tensor, in_scale: Optional[torch.Tensor] = None,
                 residual: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None
                 ) -> torch.Tensor:
        x = _dev_f32(x, "x")
        b, cin, d, h, w = x.shape
        if cin != self.cin:
            raise RuntimeError(f"expected {self.cin} input channels, got {cin}")
        oshape = self.out_shape(x.shape)
        if out is None:
            out = torch.empty(oshape, dtype=torch.float32, device=x.device)
        if in_scale is not None:
            in_scale = _dev_f32(in_scale, "in_scale")
            if in_scale.numel() != b * d * h * w:
                raise RuntimeError("in_scale must be [B,D,H,W]")
        if residual is not None:
            residual = _dev_f32(residual, "residual")
            if tuple(residual.shape) != tuple(oshape):
                raise RuntimeError("residual shape mismatch")
        lib = _lib.load()
        with torch.cuda.device(x.device):
            nb = 4.0 * (x.numel() + out.numel() + (0 if in_scale is None else in_scale.numel())
                        + (0 if residual is None else residual.numel()))
            if self.split:
                timed(f"conv3d_f16x3_co{self.cout}" + ("" if in_scale is None else "_filter"),
                      2.0 * out.numel() * cin * 27, nb,
                      lambda: _lib.check(lib.dv_conv3d_f16x3_f32(x.data_ptr(), self.wpacked.data_ptr(),
                                                                 _lib.ptr(self.scale), _lib.ptr(self.shift),
                                                                 _lib.ptr(in_scale), _lib.ptr(residual),
                                                                 out.data_ptr(), split_overflow_flag(x.device).data_ptr(),
                                                                 b, cin, d, h, w, self.cout,
                                                                 self.act, _lib.stream_ptr()),
                                         "dv_conv3d_f16x3_f32"))
                return out
            if self.wino and self.WINO3 and cin >= self.WINO3_MIN_CIN and in_scale is None and x.data_ptr() % 16 == 0 and \
                    lib.dv_conv3d_wino3_supported(cin, self.cout, d, h, w):
                timed(f"conv3d_k3s1_co{self.cout}", 2.0 * out.numel() * cin * 27, nb,
                      lambda: _lib.check(lib.dv_conv3d_wino3_f32(x.data_ptr(), self.wpacked3.data_ptr(),
                                                                 _lib.ptr(self.scale), _lib.ptr(self.shift),
                                                                 _lib.ptr(residual), out.data_ptr(), b, cin, d, h, w,
                                                                 self.cout, self.act, _lib.stream_ptr()),
                                         "dv_conv3d_wino3_f32"), issued=2.0 * out.numel() * cin * 27 / WINO3_MULT_REDUCTION)
                return out
            if self.wino:
                timed(f"conv3d_k3s1_co{self.cout}" + ("" if in_scale is None else "_filter"),
                      2.0 * out.numel() * cin * 27, nb,
                      lambda: _lib.check(lib.dv_conv3d_wino_f32(x.data_ptr(), self.wpacked.data_ptr(),
                                                                _lib.ptr(self.scale), _lib.ptr(self.shift),
                                                                _lib.ptr(in_scale), _lib.ptr(residual),
                                                                out.data_ptr(), b, cin, d, h, w, self.cout,
                                                                self.act, _lib.stream_ptr()),
                                         "dv_conv3d_wino_f32"), issued=2.0 * out.numel() * cin * 27 / WINO_MULT_REDUCTION)
                return out
            if self.s2pp and in_scale is None and x.data_ptr() % 16 == 0 and \
                    lib.dv_conv3d_s2pp_supported(cin, self.cout, d, h, w):
                timed(f"conv3d_k3s2_co{self.cout}", 2.0 * out.numel() * cin * 27, nb,
                      lambda: _lib.check(lib.dv_conv3d_s2pp_f32(x.data_ptr(), self.wpacked_pp.data_ptr(),
                                                                _lib.ptr(self.scale), _lib.ptr(self.shift),
                                                                _lib.ptr(residual), out.data_ptr(), b, cin, d, h, w,
                                                                self.cout, self.act, _lib.stream_ptr()),
                                         "dv_conv3d_s2pp_f32"), issued=2.0 * out.numel() * cin * 27 / S2PP_MULT_REDUCTION)
                return out
            timed(f"conv3d_k{self.k}s{self.stride}_co{self.cout}" + ("" if in_scale is None else "_filter"),
                  2.0 * out.numel() * cin * self.k ** 3, nb,
                  lambda: _lib.check(lib.dv_conv3d_f32(x.data_ptr(), self.wpacked.data_ptr(), _lib.ptr(self.scale),
                                                       _lib.ptr(self.shift), _lib.ptr(in_scale),
                                                       _lib.ptr(residual), out.data_ptr(), b, cin, d, h, w,
                                                       self.cout, self.k, self.stride, self.act,
                                                       _lib.stream_ptr()), "dv_conv3d_f32"), issued=(2.0 * out.numel() * cin * self.k ** 3 if self.cout > 1 else 0.0))
        return out


class ReplicaPlanCache:
    """Per-device plan cache for ``nn.DataParallel`` (SceneFlow/test_sceneflow_ddim.py:54-61, KITTI12/test.py): the
    wrapper re-creates its replicas on EVERY forward, so a replica that folded BatchNorm / repacked weights for itself
    would redo that work for every plan-holding module on every device and call.  Replicas keep a pointer to the module
    they were copied from and park their plans there, keyed by device and valid for one weight version of the SOURCE
    (the replicas' own weights are re-broadcast from it each time).  Mixed into the plan-holding modules."""

    def _replica_source(self):
        return self.__dict__.get("_plan_source")

    def _mark_replica(self, replica):
        replica.__dict__["_plan_source"] = self.__dict__.get("_plan_source") or self
        return replica

    def _source_version(self, src):
        """Identity of the source's weight set: (storage address, in-place version) of every parameter and buffer -- a
        weight replaced through ``param.data = ...`` changes the address, an in-place write the version."""
        return tuple((t.data_ptr(), t._version) for t in list(src.parameters()) + list(src.buffers()))

    def _replica_lookup(self, device):
        src = self._replica_source()
        if src is None:
            return None
        hit = src.__dict__.setdefault("_replica_plans", {}).get(str(device))
        return hit[1] if hit is not None and hit[0] == self._source_version(src) else None

    def _replica_store(self, device, plans):
        src = self._replica_source()
        if src is not None:
            src.__dict__.setdefault("_replica_plans", {})[str(device)] = (self._source_version(src), plans)

    def _replica_clear(self):
        self.__dict__.pop("_replica_plans", None)


class PointwiseExpandPlan:
    """A 1x1 Conv2d(bias=False) from few input channels (<= 32) to many output channels, nothing fused: the per-pair
    table convolutions of ``Rank1FilterPlan`` (csrc/pointwise_expand.hip).  weight [Cout, Cin, 1, 1] or [Cout, Cin]."""

    MAX_CIN = 32

    def __init__(self, weight: torch.Tensor):
        w = _dev_f32(weight.detach(), "weight")
        w = w.reshape(w.shape[0], -1).contiguous()
        self.cout, self.cin = w.shape
        if self.cin > self.MAX_CIN:
            raise _lib.DiffuVolumeError(f"PointwiseExpandPlan: at most {self.MAX_CIN} input channels")
        lib = _lib.load()
        with torch.cuda.device(w.device):
            self.wpacked = torch.empty(lib.dv_pointwise_expand_packed_floats(self.cin, self.cout), dtype=torch.float32,
                                       device=w.device)
            _lib.check(lib.dv_pointwise_expand_pack_weights_f32(w.data_ptr(), self.wpacked.data_ptr(), self.cin, self.cout,
                                                                _lib.stream_ptr()), "dv_pointwise_expand_pack_weights_f32")

    def __call__(self, x: torch.Tensor) -> torch.Tensor:
        x = _dev_f32(x, "x")
        b, cin, h, w = x.shape
        if cin != self.cin:
            raise RuntimeError(f"expected {self.cin} input channels, got {cin}")
        out = torch.empty((b, self.cout, h, w), dtype=torch.float32, device=x.device)
        lib = _lib.load()
        with torch.cuda.device(x.device):
            timed(f"pointwise_expand_co{self.cout}", 2.0 * out.numel() * cin, 4.0 * (x.numel() + out.numel()),
                  lambda: _lib.check(lib.dv_pointwise_expand_f32(x.data_ptr(), self.wpacked.data_ptr(), out.data_ptr(), b, cin,
                                                                 h * w, self.cout, _lib.stream_ptr()), "dv_pointwise_expand_f32"),
                  issued=2.0 * out.numel() * 32)
        return out


class Rank1FilterPlan:
    """The first layer of a DiffuVolume step, ``relu(bn(conv3d(volume * noise)))`` (acv_ddim.py:260 + dres0[0],
    :200-203), on the FACTORS of its input instead of the [B,64,48,h,w] volume: with volume = p * [L ; R(x-d)]
    (p = softmax(att), :388-390) and noise a per-voxel scalar the convolution is
    ``sum_tap (p*noise)(.) * (GL[tap] + GR[tap](x - d))``, GL / GR = 1x1 convolutions of L / R with the layer's weights,
    built once per stereo pair (csrc/rank1_filter.hip): 54 instead of 1728 multiply-adds per output, the same function
    up to fp32 re-association.  Used by ``ACVNet_DDIM`` whenever the volume it is handed carries its factors
    (``build_concat_attention_volume`` attaches them); any other volume takes the generic convolution."""

    MAX_D = 48

    def __init__(self, weight: torch.Tensor, bn, act: int = ACT_RELU, eps: float = 1e-5):
        w = _dev_f32(weight.detach(), "weight")
        self.cout, cin2, k = w.shape[0], w.shape[1], w.shape[2]
        if tuple(w.shape[2:]) != (3, 3, 3) or cin2 % 2:
            raise _lib.DiffuVolumeError("Rank1FilterPlan: a 3x3x3 convolution over [left | right] channel halves")
        self.c = cin2 // 2
        self.act = act
        # table weights: output channel = tap * Cout + co
        wl = w[:, :self.c].permute(2, 3, 4, 0, 1).reshape(27 * self.cout, self.c, 1, 1).contiguous()
        wr = w[:, self.c:].permute(2, 3, 4, 0, 1).reshape(27 * self.cout, self.c, 1, 1).contiguous()
        # few input channels -> 27 * Cout output channels, nothing fused: an HBM write stream with its own kernel
        # (csrc/pointwise_expand.hip; the generic 2-D convolution stays for more than 32 input channels)
        mk = PointwiseExpandPlan if self.c <= PointwiseExpandPlan.MAX_CIN else (lambda t: Conv2dPlan(t, None, act=ACT_NONE))
        self.table_l, self.table_r = mk(wl), mk(wr)
        self.scale, self.shift = _fold_bn(bn, None, self.cout, w.device, eps)

    def applies(self, volume) -> bool:
        fac = volume_factors(volume)
        return (fac is not None and len(volume.shape) == 5 and volume.shape[1] == 2 * self.c
                and volume.shape[2] <= self.MAX_D and fac.ref.shape[1] == self.c
                and tuple(fac.p_att.shape) == (volume.shape[0],) + tuple(volume.shape[2:]))

    def tables(self, volume):
        """(GL, GR) [B, 27*Cout, h, w] of a volume's feature maps, cached on its factor handle."""
        fac = volume_factors(volume)
        cached = fac._rank1_tables
        if cached is not None and cached[0] is self:
            return cached[1], cached[2]
        gl, gr = self.table_l(fac.ref), self.table_r(fac.tgt)
        fac._rank1_tables = (self, gl, gr)
        return gl, gr

    def __call__(self, volume, noise01: Optional[torch.Tensor] = None) -> torch.Tensor:
        """``noise01`` None: the unfiltered volume (the origin network's first aggregation layer, acv.py)."""
        p_att = volume_factors(volume).p_att
        b, d, h, w = p_att.shape
        gl, gr = self.tables(volume)
        out = torch.empty((b, self.cout, d, h, w), dtype=torch.float32, device=p_att.device)
        lib = _lib.load()
        with torch.cuda.device(p_att.device):
            if noise01 is None:
                s = p_att
            else:
                noise01 = _dev_f32(noise01, "noise01")
                if noise01.numel() != p_att.numel():
                    raise RuntimeError("the filter must be [B,D,H,W]")
                s = torch.empty_like(p_att)
                _lib.check(lib.dv_mul_f32(p_att.data_ptr(), noise01.data_ptr(), s.data_ptr(), s.numel(), _lib.stream_ptr()),
                           "dv_mul_f32")
            # algorithmic flops / bytes of the layer it replaces; issued on the vector ALU, not the matrix pipe
            timed(f"conv3d_k3s1_co{self.cout}_filter_rank1", 2.0 * out.numel() * 2 * self.c * 27,
                  4.0 * (volume.numel() + out.numel() + s.numel()),
                  lambda: _lib.check(lib.dv_conv3d_rank1_filter_f32(s.data_ptr(), gl.data_ptr(), gr.data_ptr(),
                                                                    _lib.ptr(self.scale), _lib.ptr(self.shift),
                                                                    out.data_ptr(), b, d, h, w, self.cout, self.act,
                                                                    _lib.stream_ptr()), "dv_conv3d_rank1_filter_f32"),
                  valu=out.numel() * 27 * 4.0)       # per output and tap: two fma lanes (GL term, GR term)
        return out


class Deconv2dK4S2Plan:
    """ConvTranspose2d(kernel 4, stride 2, padding 1) [+ bias] [+ BatchNorm2d eval] [+ activation] -- IGEV's `spx_2_gru.conv1`
    and `spx_gru` (KITTI15/core/igev_stereo_ddim.py:110-112, core/submodule.py:27-28) -- on the 3x3 kernels: output pixel
    (2i+a, 2j+b) reads the 2x2 inputs {i-1+a, i+a} x {j-1+b, j+b}, so the four output parities are four 3x3 stride-1
    convolutions of the input (each with five structurally zero taps) whose results interleave (pixel shuffle)."""

    def __init__(self, weight: torch.Tensor, bn=None, bias: Optional[torch.Tensor] = None, act: int = ACT_NONE,
                 eps: float = 1e-5):
        w = _dev_f32(weight.detach(), "weight")                  # [Cin, Cout, 4, 4]
        if tuple(w.shape[2:]) != (4, 4):
            raise _lib.DiffuVolumeError("Deconv2dK4S2Plan: kernel 4, stride 2, padding 1")
        cin, cout = w.shape[0], w.shape[1]
        self.cout = cout
        wc = torch.zeros((cout, 2, 2, cin, 3, 3), dtype=torch.float32, device=w.device)
        tap = {0: {0: 1, -1: 3}, 1: {1: 0, 0: 2}}              # parity -> {input offset: kernel index}
        for a in (0, 1):
            for dy, ky in tap[a].items():
                for b in (0, 1):
                    for dx, kx in tap[b].items():
                        wc[:, a, b, :, dy + 1, dx + 1] = w[:, :, ky, kx].t()
        rep = lambda t: None if t is None else t.detach().repeat_interleave(4)
        self.conv = Conv2dPlan(wc.reshape(cout * 4, cin, 3, 3), None if bn is None else tuple(rep(t) for t in bn), act=act,
                               eps=eps, bias=rep(bias))

    def __call__(self, x: torch.Tensor) -> torch.Tensor:
        return torch.nn.functional.pixel_shuffle(self.conv(x), 2)


class Conv2dPairPlan:
    """Two biased 3x3 convolutions of the same input in one Winograd launch (ConvGRU's `convz` / `convr`,
    KITTI15/core/update.py:33-35): out_g = act(conv_g(x) + bias_g + residual_g) [* mul_g].  Launches too small for the
    Winograd kernel go through the two single plans."""

    def __init__(self, conv1: Tuple[torch.Tensor, Optional[torch.Tensor]], conv2: Tuple[torch.Tensor, Optional[torch.Tensor]],
                 act: int = ACT_NONE):
        (w1, b1), (w2, b2) = conv1, conv2
        self.single = (Conv2dPlan(w1, None, act=act, bias=b1), Conv2dPlan(w2, None, act=act, bias=b2))
        p1, p2 = self.single
        self.act, self.cin, self.c1, self.c2 = act, p1.cin, p1.cout, p2.cout
        self.packed = None
        if p1.wino_packed is not None and p2.wino_packed is not None and p1.cin == p2.cin and self.c1 % 32 == 0:
            w = torch.cat([_dev_f32(w1.detach(), "weight"), _dev_f32(w2.detach(), "weight")], dim=0).contiguous()
            lib = _lib.load()
            self.packed = torch.empty(lib.dv_conv2d_wino_packed_floats(self.cin, self.c1 + self.c2), dtype=torch.float32,
                                      device=w.device)
            with torch.cuda.device(w.device):
                _lib.check(lib.dv_conv2d_wino_pack_weights_f32(w.data_ptr(), self.packed.data_ptr(), self.cin,
                                                               self.c1 + self.c2, _lib.stream_ptr()), "conv2d wino weight packing")

            def both(a, b, fill):
                if a is None and b is None:
                    return None
                mk = lambda t, c: t if t is not None else torch.full((c,), fill, dtype=torch.float32, device=w.device)
                return torch.cat([mk(a, self.c1), mk(b, self.c2)]).contiguous()
            self.scale, self.shift = both(p1.scale, p2.scale, 1.0), both(p1.shift, p2.shift, 0.0)

    def __call__(self, x, residual=(None, None), mul=(None, None)):
        parts = [_dev_f32(t, "x") for t in (x if isinstance(x, (list, tuple)) else [x])]
        b, _, h, w = parts[0].shape
        blocks = (-(-h // 16)) * (-(-w // 16)) * ((self.c1 + self.c2) // 32)      # of ONE batch item, see Conv2dPlan
        if self.packed is None or blocks < Conv2dPlan.WINO_MIN_BLOCKS or not 1 <= len(parts) <= 4:
            return (self.single[0](x, residual=residual[0], mul=mul[0]), self.single[1](x, residual=residual[1], mul=mul[1]))
        if sum(t.shape[1] for t in parts) != self.cin or any(t.shape[0] != b or t.shape[2:] != parts[0].shape[2:] for t in parts):
            raise RuntimeError("virtual concatenation: tensors with equal batch and spatial size, channels summing to Cin")
        import ctypes
        outs = [torch.empty((b, c, h, w), dtype=torch.float32, device=parts[0].device) for c in (self.c1, self.c2)]

        def same(t, o, name):
            if t is None:
                return None
            t = _dev_f32(t, name)
            if tuple(t.shape) != tuple(o.shape):
                raise RuntimeError(f"{name} shape mismatch")
            return t
        res = [same(residual[g], outs[g], "residual") for g in (0, 1)]
        mu = [same(mul[g], outs[g], "mul") for g in (0, 1)]
        ptrs = (ctypes.c_void_p * len(parts))(*[t.data_ptr() for t in parts])
        chans = (ctypes.c_int * len(parts))(*[t.shape[1] for t in parts])
        lib = _lib.load()
        n_out = outs[0].numel() + outs[1].numel()
        extra = sum(t.numel() for t in res + mu if t is not None)
        ks = lib.dv_conv2d_wino_auto_kslices(self.cin, h, w, self.c1 + self.c2, 1) if Conv2dPlan.KSPLIT else 1
        if ks > 1:            # a launch that leaves most of the chip empty: K-split, partial tiles in scratch, fixed-order sum
            scratch = torch.empty(ks * n_out, dtype=torch.float32, device=parts[0].device)
            with torch.cuda.device(parts[0].device):
                timed(f"conv2d_k3d1_co{self.c1}+{self.c2}_ksplit", 2.0 * n_out * self.cin * 9,
                      4.0 * (sum(t.numel() for t in parts) + n_out * (1 + 2 * ks) + extra),
                      lambda: _lib.check(lib.dv_conv2d_wino_cat_pair_ksplit_f32(
                          ptrs, chans, len(parts), self.packed.data_ptr(), _lib.ptr(self.scale), _lib.ptr(self.shift),
                          _lib.ptr(res[0]), _lib.ptr(mu[0]), outs[0].data_ptr(), _lib.ptr(res[1]), _lib.ptr(mu[1]),
                          outs[1].data_ptr(), scratch.data_ptr(), ks, b, h, w, self.c1, self.c2, self.act, _lib.stream_ptr()),
                          "dv_conv2d_wino_cat_pair_ksplit_f32"), issued=2.0 * n_out * self.cin * 9 / WINO_MULT_REDUCTION)
            return outs[0], outs[1]
        with torch.cuda.device(parts[0].device):
            timed(f"conv2d_k3d1_co{self.c1}+{self.c2}", 2.0 * n_out * self.cin * 9, 4.0 * (sum(t.numel() for t in parts) + n_out + extra),
                  lambda: _lib.check(lib.dv_conv2d_wino_cat_pair_f32(ptrs, chans, len(parts), self.packed.data_ptr(),
                                                                     _lib.ptr(self.scale), _lib.ptr(self.shift),
                                                                     _lib.ptr(res[0]), _lib.ptr(mu[0]), outs[0].data_ptr(),
                                                                     _lib.ptr(res[1]), _lib.ptr(mu[1]), outs[1].data_ptr(),
                                                                     b, h, w, self.c1, self.c2, self.act, _lib.stream_ptr()),
                                     "dv_conv2d_wino_cat_pair_f32"), issued=2.0 * n_out * self.cin * 9 / WINO_MULT_REDUCTION)
        return outs[0], outs[1]


class Conv2dPlan:
    """Conv2d(k 3 or 1, stride 1, padding = dilation) [+ bias] [+ BatchNorm2d eval] [+ residual] [+ activation]
    [+ ConvGRU gate arithmetic] on the 2-D implicit-GEMM kernel: `convbn` / `BasicBlock` of the KITTI12 refinement
    stack (KITTI12/models/submodule.py:21-24, :192-215) and the biased convolutions of IGEV's update block
    (KITTI15/core/update.py)."""

    def __init__(self, weight: torch.Tensor, bn: Optional[Tuple[torch.Tensor, ...]] = None, dilation: int = 1,
                 act: int = ACT_NONE, eps: float = 1e-5, bias: Optional[torch.Tensor] = None, stride: int = 1):
        w = _dev_f32(weight.detach(), "weight")
        self.cout, self.cin, k = w.shape[0], w.shape[1], w.shape[2]
        if tuple(w.shape[2:]) != (k, k) or k not in (1, 3):
            raise _lib.DiffuVolumeError(f"unsupported Conv2d kernel {tuple(w.shape[2:])}")
        if k == 3 and not 1 <= dilation <= 16:
            raise _lib.DiffuVolumeError("3x3 convolutions: dilation 1..16")
        if stride not in (1, 2) or (stride == 2 and k == 3 and dilation != 1):
            raise _lib.DiffuVolumeError("stride 1, or stride 2 with dilation 1")
        self.k, self.dilation, self.act, self.stride = k, (dilation if k == 3 else 1), act, stride
        lib = _lib.load()
        self.wpacked = torch.empty(lib.dv_conv2d_packed_floats(self.cin, self.cout, k, self.dilation),
                                   dtype=torch.float32, device=w.device)
        with torch.cuda.device(w.device):
            _lib.check(lib.dv_conv2d_pack_weights_f32(w.data_ptr(), self.wpacked.data_ptr(), self.cin, self.cout, k,
                                                      self.dilation, _lib.stream_ptr()), "conv2d weight packing")
            # 3x3, stride 1: also the Winograd F(2x2,3x3) image (csrc/conv2d_wino.hip), used when the launch has enough
            # 16x16x32 blocks to fill the chip (small images stay on the direct kernel's small tiles).  Dilated layers
            # run it on their dilation^2 sub-sampled images (up to WINO_MAX_DILATION).
            self.wino_packed = None
            if k == 3 and self.dilation <= self.WINO_MAX_DILATION and stride == 1 and default_conv_precision() == "f32":
                self.wino_packed = torch.empty(lib.dv_conv2d_wino_packed_floats(self.cin, self.cout), dtype=torch.float32,
                                               device=w.device)
                _lib.check(lib.dv_conv2d_wino_pack_weights_f32(w.data_ptr(), self.wino_packed.data_ptr(), self.cin,
                                                               self.cout, _lib.stream_ptr()), "conv2d wino weight packing")
        self.scale, self.shift = _fold_bn(bn, bias, self.cout, w.device, eps)

    # Which kernel runs (Winograd or direct, and the K-split factor of the direct one) decides the ORDER an output is summed
    # in, so it may only look at ONE batch item: a shard of a batch has to reproduce the batch's bits (round 5: config 5
    # is sharded over ranks; until then the thresholds counted the whole launch, 128 blocks = 32 per item at batch 4).
    WINO_MIN_BLOCKS = 32
    WINO_MAX_DILATION = 8       # measured at 1248x384 (tools/ab_wino2d_dil.py): d=2 -27 %, d=4 -19 %, d=8 -11..17 %, d=16 +41..57 % (the gathers spread over too many sectors)
    KSPLIT = True               # K-split the launches that are too small to fill the chip (A/B switch for tools/)

    def __call__(self, x: torch.Tensor, residual: Optional[torch.Tensor] = None, mul: Optional[torch.Tensor] = None,
                 blend: Optional[Tuple[torch.Tensor, torch.Tensor]] = None, group: int = 1,
                 s2b_out: bool = False) -> torch.Tensor:
        """act(conv(x)*scale + shift + residual) [* mul] [-> h + z*(. - h) for blend = (z, h)].
        ``x`` may be a list of up to four tensors: the convolution then runs over their channel concatenation
        without materialising it (stride 1).  ``group``: how many consecutive batch entries of ``x`` make up ONE sample (the
        sub-images of a space-to-batch tensor): the per-sample block count that picks the kernel is taken over all of them.
        ``s2b_out``: store the result de-interleaved by 2, [4B,Cout,H/2,W/2] (`dv_conv2d_wino_s2b_f32`; Winograd layers
        without residual / mul / blend, even H and W) -- what `pwcnet_ddim.space_to_batch2` would make of it."""
        parts = None
        if isinstance(x, (list, tuple)):
            parts = [_dev_f32(t, "x") for t in x]
            if not 1 <= len(parts) <= 4 or any(t.shape[0] != parts[0].shape[0] or t.shape[2:] != parts[0].shape[2:] for t in parts):
                raise RuntimeError("virtual concatenation: 1..4 tensors with equal batch and spatial size")
            if self.stride != 1:
                raise RuntimeError("virtual concatenation is stride-1 only")
            x = parts[0]
            b, _, h, w = x.shape
            cin = sum(t.shape[1] for t in parts)
        else:
            x = _dev_f32(x, "x")
            b, cin, h, w = x.shape
        if cin != self.cin:
            raise RuntimeError(f"expected {self.cin} input channels, got {cin}")
        out = torch.empty((b, self.cout, (h - 1) // self.stride + 1, (w - 1) // self.stride + 1), dtype=torch.float32,
                          device=x.device)
        if self.stride == 2:
            if mul is not None or blend is not None:
                raise RuntimeError("gate operands are stride-1 only")
            if residual is not None:
                residual = _dev_f32(residual, "residual")
                if tuple(residual.shape) != tuple(out.shape):
                    raise RuntimeError("residual shape mismatch")
            lib = _lib.load()
            with torch.cuda.device(x.device):
                timed(f"conv2d_k{self.k}s2_co{self.cout}", 2.0 * out.numel() * cin * self.k ** 2,
                      4.0 * (x.numel() + out.numel() * (1 if residual is None else 2)),
                      lambda: _lib.check(lib.dv_conv2d_s2_f32(x.data_ptr(), self.wpacked.data_ptr(), _lib.ptr(self.scale),
                                                              _lib.ptr(self.shift), _lib.ptr(residual), out.data_ptr(), b,
                                                              cin, h, w, self.cout, self.k, self.act, _lib.stream_ptr()),
                                         "dv_conv2d_s2_f32"), issued=2.0 * out.numel() * cin * self.k ** 2)
            return out

        def same(t, name):
            t = _dev_f32(t, name)
            if tuple(t.shape) != tuple(out.shape):
                raise RuntimeError(f"{name} shape mismatch")
            return t

        residual = None if residual is None else same(residual, "residual")
        mul = None if mul is None else same(mul, "mul")
        bz, bh = (None, None) if blend is None else (same(blend[0], "blend z"), same(blend[1], "blend h"))
        extra = sum(t is not None for t in (residual, mul, bz, bh))
        lib = _lib.load()
        d = self.dilation
        if s2b_out:
            if (self.wino_packed is None or d != 1 or self.stride != 1 or extra or h % 2 or w % 2):
                raise _lib.DiffuVolumeError("s2b_out: a 3x3 dilation-1 stride-1 Winograd layer without residual / mul / "
                                            "blend on even H and W")
            import ctypes
            srcs = parts if parts is not None else [x]
            ptrs = (ctypes.c_void_p * len(srcs))(*[t.data_ptr() for t in srcs])
            chans = (ctypes.c_int * len(srcs))(*[t.shape[1] for t in srcs])
            out = torch.empty((4 * b, self.cout, h // 2, w // 2), dtype=torch.float32, device=x.device)
            with torch.cuda.device(x.device):
                nb = 4.0 * (sum(t.numel() for t in srcs) + out.numel())
                timed(f"conv2d_k3d1_co{self.cout}_s2b", 2.0 * out.numel() * cin * 9, nb,
                      lambda: _lib.check(lib.dv_conv2d_wino_s2b_f32(ptrs, chans, len(srcs), self.wino_packed.data_ptr(),
                                                                    _lib.ptr(self.scale), _lib.ptr(self.shift), out.data_ptr(),
                                                                    b, h, w, self.cout, self.act, _lib.stream_ptr()),
                                         "dv_conv2d_wino_s2b_f32"), issued=2.0 * out.numel() * cin * 9 / WINO_MULT_REDUCTION)
            return out
        if self.wino_packed is not None and d <= self.WINO_MAX_DILATION and \
                group * d * d * (-(-(-(-h // d)) // 16)) * (-(-(-(-w // d)) // 16)) * (-(-self.cout // 32)) >= self.WINO_MIN_BLOCKS:
            import ctypes
            srcs = parts if parts is not None else [x]
            ptrs = (ctypes.c_void_p * len(srcs))(*[t.data_ptr() for t in srcs])
            chans = (ctypes.c_int * len(srcs))(*[t.shape[1] for t in srcs])
            ks = lib.dv_conv2d_wino_auto_kslices(cin, h, w, self.cout, d) if (self.KSPLIT and group == 1) else 1
            if ks > 1:        # (see Conv2dPairPlan)
                scratch = torch.empty(ks * out.numel(), dtype=torch.float32, device=x.device)
                with torch.cuda.device(x.device):
                    nb = 4.0 * (sum(t.numel() for t in srcs) + out.numel() * (1 + extra + 2 * ks))
                    timed(f"conv2d_k3d{d}_co{self.cout}_wksplit", 2.0 * out.numel() * cin * 9, nb,
                          lambda: _lib.check(lib.dv_conv2d_wino_cat_ksplit_f32(
                              ptrs, chans, len(srcs), self.wino_packed.data_ptr(), _lib.ptr(self.scale), _lib.ptr(self.shift),
                              _lib.ptr(residual), _lib.ptr(mul), _lib.ptr(bz), _lib.ptr(bh), out.data_ptr(), scratch.data_ptr(),
                              ks, b, h, w, self.cout, d, self.act, _lib.stream_ptr()), "dv_conv2d_wino_cat_ksplit_f32"),
                          issued=2.0 * out.numel() * cin * 9 / WINO_MULT_REDUCTION)
                return out
            with torch.cuda.device(x.device):
                nb = 4.0 * (sum(t.numel() for t in srcs) + out.numel() * (1 + extra))
                timed(f"conv2d_k3d{d}_co{self.cout}", 2.0 * out.numel() * cin * 9, nb,
                      lambda: _lib.check(lib.dv_conv2d_wino_dil_cat_f32(ptrs, chans, len(srcs), self.wino_packed.data_ptr(),
                                                                        _lib.ptr(self.scale), _lib.ptr(self.shift),
                                                                        _lib.ptr(residual), _lib.ptr(mul), _lib.ptr(bz),
                                                                        _lib.ptr(bh), out.data_ptr(), b, h, w, self.cout,
                                                                        d, self.act, _lib.stream_ptr()),
                                         "dv_conv2d_wino_dil_cat_f32"), issued=2.0 * out.numel() * cin * 9 / WINO_MULT_REDUCTION)
            return out
        # launches too small to fill the chip (a single IGEV pair at 1/8 and 1/16 resolution): K-split over the input
        # channels, partial tiles in a scratch buffer, fused epilogue in the reduction kernel
        kslices = lib.dv_conv2d_auto_kslices(1, cin, h, w, self.cout, self.k, self.dilation)     # (one batch item: see above)
        if kslices > 1 and self.KSPLIT:
            import ctypes
            srcs = parts if parts is not None else [x]
            ptrs = (ctypes.c_void_p * len(srcs))(*[t.data_ptr() for t in srcs])
            chans = (ctypes.c_int * len(srcs))(*[t.shape[1] for t in srcs])
            scratch = torch.empty(kslices * out.numel(), dtype=torch.float32, device=x.device)
            with torch.cuda.device(x.device):
                nb = 4.0 * (sum(t.numel() for t in srcs) + out.numel() * (1 + extra + 2 * kslices))
                timed(f"conv2d_k{self.k}d{self.dilation}_co{self.cout}_ksplit", 2.0 * out.numel() * cin * self.k ** 2, nb,
                      lambda: _lib.check(lib.dv_conv2d_cat_ksplit_f32(ptrs, chans, len(srcs), self.wpacked.data_ptr(),
                                                                      _lib.ptr(self.scale), _lib.ptr(self.shift),
                                                                      _lib.ptr(residual), _lib.ptr(mul), _lib.ptr(bz),
                                                                      _lib.ptr(bh), out.data_ptr(), scratch.data_ptr(),
                                                                      kslices, b, h, w, self.cout, self.k, self.dilation,
                                                                      self.act, _lib.stream_ptr()),
                                         "dv_conv2d_cat_ksplit_f32"), issued=2.0 * out.numel() * cin * self.k ** 2)
            return out
        if parts is not None:
            import ctypes
            ptrs = (ctypes.c_void_p * len(parts))(*[t.data_ptr() for t in parts])
            chans = (ctypes.c_int * len(parts))(*[t.shape[1] for t in parts])
            with torch.cuda.device(x.device):
                nb = 4.0 * (sum(t.numel() for t in parts) + out.numel() * (1 + extra))
                timed(f"conv2d_k{self.k}d{self.dilation}_co{self.cout}", 2.0 * out.numel() * cin * self.k ** 2, nb,
                      lambda: _lib.check(lib.dv_conv2d_cat_f32(ptrs, chans, len(parts), self.wpacked.data_ptr(),
                                                               _lib.ptr(self.scale), _lib.ptr(self.shift), _lib.ptr(residual),
                                                               _lib.ptr(mul), _lib.ptr(bz), _lib.ptr(bh), out.data_ptr(), b, h, w,
                                                               self.cout, self.k, self.dilation, self.act, _lib.stream_ptr()),
                                         "dv_conv2d_cat_f32"), issued=2.0 * out.numel() * cin * self.k ** 2)
            return out
        with torch.cuda.device(x.device):
            nb = 4.0 * (x.numel() + out.numel() * (1 + extra))
            timed(f"conv2d_k{self.k}d{self.dilation}_co{self.cout}", 2.0 * out.numel() * cin * self.k ** 2, nb,
                  lambda: _lib.check(lib.dv_conv2d_gated_f32(x.data_ptr(), self.wpacked.data_ptr(), _lib.ptr(self.scale),
                                                             _lib.ptr(self.shift), _lib.ptr(residual), _lib.ptr(mul),
                                                             _lib.ptr(bz), _lib.ptr(bh), out.data_ptr(), b, cin, h, w,
                                                             self.cout, self.k, self.dilation, self.act,
                                                             _lib.stream_ptr()), "dv_conv2d_gated_f32"), issued=2.0 * out.numel() * cin * self.k ** 2)
        return out


class Deconv3dPlan:
    """ConvTranspose3d(stride 2, padding 1, bias=False) + BatchNorm3d (eval) + skip add + activation that
    doubles every dimension: kernel 3 with output_padding 1 (acv_ddim.py:74-80, :91-92) or kernel 4
    (IGEV hourglass, igev_stereo_ddim.py:44-51).

    ``redir=(weight [Cout,Cskip,1,1,1], bn)``: the hourglass's 1x1x1 skip convolution + BN
    (acv_ddim.py:81-86).  Then ``plan(x, skip=t)`` computes ``act(BN(deconv(x)) + BN_r(conv1x1(t)))`` in one
    launch (both BN scales folded into the weights, the 1x1x1 product accumulated as extra K-steps); shapes the
    fused kernel does not take fall back to the two-launch form."""

    def __init__(self, weight: torch.Tensor, bn: Optional[Tuple[torch.Tensor, ...]] = None,
                 act: int = ACT_NONE, eps: float = 1e-5, redir=None, redir_eps: float = 1e-5):
        w = _dev_f32(weight.detach(), "weight")
        self.cin, self.cout, k = w.shape[0], w.shape[1], w.shape[2]
        if tuple(w.shape[2:]) != (k, k, k) or k not in (3, 4):
            raise _lib.DiffuVolumeError("transposed convolutions: k3 s2 p1 op1 and k4 s2 p1 are implemented")
        self.k, self.act = k, act
        lib = _lib.load()
        sizer, packer, self._run, self._name = (
            (lib.dv_deconv3d_packed_floats, lib.dv_deconv3d_pack_weights_f32, lib.dv_deconv3d_k3s2_f32, "dv_deconv3d_k3s2_f32")
            if k == 3 else
            (lib.dv_deconv3d_k4_packed_floats, lib.dv_deconv3d_k4_pack_weights_f32, lib.dv_deconv3d_k4s2_f32, "dv_deconv3d_k4s2_f32"))
        self.scale, self.shift = _fold_bn(bn, None, self.cout, w.device, eps)
        self.redir_w = self.redir_plan = None
        if redir is not None:
            if k != 3:
                raise _lib.DiffuVolumeError("the fused skip convolution exists for the kernel-3 flavour")
            rw, rbn = redir
            rw = _dev_f32(rw.detach(), "redir weight")
            if rw.shape[0] != self.cout or tuple(rw.shape[2:]) != (1, 1, 1):
                raise _lib.DiffuVolumeError("redir must be a 1x1x1 convolution onto the deconvolution's channels")
            self.cskip = rw.shape[1]
            rscale, rshift = _fold_bn(rbn, None, self.cout, w.device, redir_eps)
            one = torch.ones(self.cout, device=w.device)
            zero = torch.zeros(self.cout, device=w.device)
            sd, sr = (one if self.scale is None else self.scale), (one if rscale is None else rscale)
            # fold both BN scales into the weights; the fused kernel then needs a single accumulator
            w = w * sd.view(1, -1, 1, 1, 1)
            self.redir_w = (rw.reshape(self.cout, self.cskip) * sr.view(-1, 1)).contiguous()
            self.shift = ((zero if self.shift is None else self.shift) + (zero if rshift is None else rshift)).contiguous()
            self.scale = None
            self.redir_plan = Conv3dPlan(self.redir_w.view(self.cout, self.cskip, 1, 1, 1), None, stride=1, act=ACT_NONE,
                                         precision="f32")       # two-launch fallback
        self.wpacked = torch.empty(sizer(self.cin, self.cout), dtype=torch.float32, device=w.device)
        with torch.cuda.device(w.device):
            _lib.check(packer(w.contiguous().data_ptr(), self.wpacked.data_ptr(), self.cin, self.cout, _lib.stream_ptr()),
                       "deconv weight packing")

    def __call__(self, x: torch.Tensor, residual: Optional[torch.Tensor] = None,
                 skip: Optional[torch.Tensor] = None) -> torch.Tensor:
        x = _dev_f32(x, "x")
        b, cin, d, h, w = x.shape
        if cin != self.cin:
            raise RuntimeError(f"expected {self.cin} input channels, got {cin}")
        out = torch.empty((b, self.cout, 2 * d, 2 * h, 2 * w), dtype=torch.float32, device=x.device)
        if skip is not None:
            if self.redir_w is None or residual is not None:
                raise RuntimeError("skip= needs a plan built with redir=, and excludes residual=")
            skip = _dev_f32(skip, "skip")
            if tuple(skip.shape) != (b, self.cskip, 2 * d, 2 * h, 2 * w):
                raise RuntimeError("skip shape mismatch")
            if w % 2 == 0 and (self.cskip + 7) // 8 <= (cin + 7) // 8:
                lib = _lib.load()
                with torch.cuda.device(x.device):
                    nb = 4.0 * (x.numel() + out.numel() + skip.numel())
                    timed("deconv3d_k3s2_redir", 2.0 * x.numel() * self.cout * 27 + 2.0 * out.numel() * self.cskip, nb,
                          lambda: _lib.check(lib.dv_deconv3d_k3s2_redir_f32(
                              x.data_ptr(), self.wpacked.data_ptr(), _lib.ptr(self.shift), skip.data_ptr(),
                              self.redir_w.data_ptr(), out.data_ptr(), b, cin, d, h, w, self.cout, self.cskip, self.act,
                              _lib.stream_ptr()), "dv_deconv3d_k3s2_redir_f32"), issued=2.0 * x.numel() * self.cout * 27 + 2.0 * out.numel() * self.cskip)
                return out
            residual = self.redir_plan(skip)
        if residual is not None:
            residual = _dev_f32(residual, "residual")
            if tuple(residual.shape) != tuple(out.shape):
                raise RuntimeError("residual shape mismatch")
        with torch.cuda.device(x.device):
            nb = 4.0 * (x.numel() + out.numel() * (1 if residual is None else 2))
            timed(f"deconv3d_k{self.k}s2", 2.0 * x.numel() * self.cout * self.k ** 3, nb,
                  lambda: _lib.check(self._run(x.data_ptr(), self.wpacked.data_ptr(), _lib.ptr(self.scale),
                                               _lib.ptr(self.shift), _lib.ptr(residual), out.data_ptr(), b, cin, d, h,
                                               w, self.cout, self.act, _lib.stream_ptr()), self._name), issued=2.0 * x.numel() * self.cout * self.k ** 3)
        return out


def _fold_bn(bn, bias, cout, device, eps):
    """Eval-mode BatchNorm -> per-channel (scale, shift); a conv bias folds into shift."""
    if bn is None and bias is None:
        return None, None
    if bn is None:
        return None, bias.detach().to(device=device, dtype=torch.float32).contiguous()
    gamma, beta, mean, var = (t.detach().to(device=device, dtype=torch.float32) for t in bn)
    scale = gamma / torch.sqrt(var + eps)
    shift = beta - mean * scale
    if bias is not None:
        shift = shift + bias.detach().to(device=device, dtype=torch.float32) * scale
    return scale.contiguous(), shift.contiguous()


def feature_gate(cv: torch.Tensor, logit: torch.Tensor, inplace: bool = False) -> torch.Tensor:
    """`torch.sigmoid(feat_att) * cv` of FeatureAtt.forward (KITTI15/core/submodule.py:234-239); logit is the
    2-D branch's output [B,C,H,W] before the sigmoid, cv the volume [B,C,D,H,W]."""
    cv = _dev_f32(cv, "cv")
    logit = _dev_f32(logit, "logit")
    b, c, d, h, w = cv.shape
    if tuple(logit.shape) != (b, c, h, w):
        raise RuntimeError(f"gate logits {tuple(logit.shape)} do not match volume {tuple(cv.shape)}")
    out = cv if inplace else torch.empty_like(cv)
    lib = _lib.load()
    with torch.cuda.device(cv.device):
        timed("feature_gate", float(cv.numel()), 8.0 * cv.numel(),
              lambda: _lib.check(lib.dv_feature_gate_f32(cv.data_ptr(), logit.data_ptr(), out.data_ptr(), b, c, d, h, w,
                                                         _lib.stream_ptr()), "dv_feature_gate_f32"))
    return out


def refine_inputs(left: torch.Tensor, right: torch.Tensor, disp: torch.Tensor, du_a: torch.Tensor,
                  du_b: torch.Tensor, maxshift: int = 24) -> torch.Tensor:
    """cat(left - warp(right, disp), left, Mish(du_a*disp + du_b), disp, corr(left, warp(right, disp), +-24)):
    the refinement network's input (KITTI12/models/pwcnet_ddim.py:486-502) in one kernel.
    left/right [B,C,H,W], disp [B,1,H,W] or [B,H,W] -> [B, 3C+1+49, H, W]."""
    left, right = _dev_f32(left, "left"), _dev_f32(right, "right")
    disp = _dev_f32(disp, "disp")
    b, c, h, w = left.shape
    if tuple(right.shape) != (b, c, h, w) or disp.numel() != b * h * w:
        raise RuntimeError("left / right / disp shapes do not match")
    du_a, du_b = _dev_f32(du_a, "du_a"), _dev_f32(du_b, "du_b")
    out = torch.empty((b, 3 * c + 1 + 2 * maxshift + 1, h, w), dtype=torch.float32, device=left.device)
    lib = _lib.load()
    with torch.cuda.device(left.device):
        timed("refine_inputs", 2.0 * b * h * w * c * (2 * maxshift + 1), 4.0 * (2 * left.numel() + out.numel()),
              lambda: _lib.check(lib.dv_refine_inputs_f32(left.data_ptr(), right.data_ptr(), disp.data_ptr(),
                                                          du_a.data_ptr(), du_b.data_ptr(), out.data_ptr(), b, c, h, w,
                                                          maxshift, _lib.stream_ptr()), "dv_refine_inputs_f32"))
    return out


def patch_volume(gwc: torch.Tensor, w1: torch.Tensor, w2: torch.Tensor, dilation: torch.Tensor) -> torch.Tensor:
    """`patch` followed by `patch_l1/l2/l3` (acv_ddim.py:377-381): two per-channel (1,3,3) stencils in one pass.
    gwc [B,G,D,H,W]; w1, w2 [G,9] float32; dilation [G] int32 (1..3).  The dilation table is turned into runs of
    consecutive groups on the host ONCE per tensor object (cached on it: a device tensor costs one synchronising read
    the first time) so that the launcher can pick the dilation-specialised 16-byte kernels."""
    gwc = _dev_f32(gwc, "gwc")
    b, g, d, h, w = gwc.shape
    w1, w2 = _dev_f32(w1, "w1"), _dev_f32(w2, "w2")
    if tuple(w1.shape) != (g, 9) or tuple(w2.shape) != (g, 9) or dilation.numel() != g or dilation.dtype != torch.int32:
        raise RuntimeError("w1 / w2 must be [G,9] and dilation [G] int32")
    runs = getattr(dilation, "_dv_runs", None)
    key = (dilation.data_ptr(), dilation._version)       # (a `.data` swap keeps _version: the pointer is part of the key)
    if runs is None or runs[0] != key:
        vals, r = dilation.detach().cpu().tolist(), []
        for i, v in enumerate(vals):
            if r and r[-1][2] == v:
                r[-1][1] += 1
            else:
                r.append([i, 1, v])
        arr = lambda k: (ctypes.c_int * len(r))(*[x[k] for x in r])
        runs = (key, len(r), arr(0), arr(1), arr(2), all(1 <= x[2] <= 3 for x in r))
        dilation._dv_runs = runs
    out = torch.empty_like(gwc)
    lib = _lib.load()
    dil_dev = dilation.contiguous()
    with torch.cuda.device(gwc.device):
        if runs[5]:
            call = lambda: _lib.check(lib.dv_patch_volume_runs_f32(gwc.data_ptr(), w1.data_ptr(), w2.data_ptr(), dil_dev.data_ptr(),
                                                                   out.data_ptr(), b, g, d, h, w, runs[1], runs[2], runs[3],
                                                                   runs[4], _lib.stream_ptr()), "dv_patch_volume_runs_f32")
        else:
            call = lambda: _lib.check(lib.dv_patch_volume_f32(gwc.data_ptr(), w1.data_ptr(), w2.data_ptr(), dil_dev.data_ptr(),
                                                              out.data_ptr(), b, g, d, h, w, _lib.stream_ptr()),
                                      "dv_patch_volume_f32")
        timed("patch_volume", 36.0 * gwc.numel(), 8.0 * gwc.numel(), call)
    return out


def window_attention(x: torch.Tensor, qkv_w: torch.Tensor, qkv_b: torch.Tensor, proj_w: torch.Tensor,
                     proj_b: torch.Tensor, heads: int = 16) -> torch.Tensor:
    """attention_block.forward (submodule.py:398-429) on [B,C,D,H,W]."""
    x = _dev_f32(x, "x")
    b, c, d, h, w = x.shape
    qkv_w = _dev_f32(qkv_w.detach(), "qkv_w")
    qkv_b = _dev_f32(qkv_b.detach(), "qkv_b")
    proj_w = _dev_f32(proj_w.detach().reshape(c, c), "proj_w")
    proj_b = _dev_f32(proj_b.detach(), "proj_b")
    out = torch.empty_like(x)
    lib = _lib.load()
    with torch.cuda.device(x.device):
        ntok = float(b * d * h * w)
        timed("window_attn3d", ntok * (2.0 * c * 3 * c + 2.0 * c * c + 4.0 * 64 * c), 8.0 * x.numel(),
              lambda: _lib.check(lib.dv_window_attn3d_f32(x.data_ptr(), qkv_w.data_ptr(), qkv_b.data_ptr(),
                                                          proj_w.data_ptr(), proj_b.data_ptr(), out.data_ptr(),
                                                          b, c, d, h, w, heads, _lib.stream_ptr()),
                                 "dv_window_attn3d_f32"), issued=ntok * (2.0 * c * 3 * c + 2.0 * c * c + 4.0 * 64 * c))
    return out
