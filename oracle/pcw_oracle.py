"""CPU oracle for the KITTI12 flavour (PCWNet + DiffuVolume) of the hot path.

TEST INFRASTRUCTURE ONLY (same rules as oracle/acv_oracle.py).  Functional plain-PyTorch restatement of
KITTI12/models/pwcnet_ddim.py (hourglassup :131-205, Mish hourglass :208-248, refinenet_version3 :251-306,
model_predictions :466-528, ddim_sample :530-602, eval forward :604-641/:738-758) and the helpers of
KITTI12/models/submodule.py (warp :137-176, build_corrleation_volume :121-135).  Pinned by
tests/golden/pcw_*.npz, produced from the imported reference by oracle/make_golden_pcw.py.
"""
from __future__ import annotations

from typing import Callable, Dict, Sequence, Tuple

import torch
import torch.nn.functional as F

from . import acv_oracle as A

Tensor = torch.Tensor
SD = Dict[str, Tensor]


def mish(x: Tensor) -> Tensor:
    """KITTI12/models/submodule.py:11-18 / FMish :178-190."""
    return x * torch.tanh(F.softplus(x))


def _bn2(x, sd, p):
    return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"], sd[p + ".weight"], sd[p + ".bias"],
                        False, 0.0, 1e-5)


def convbn2d(x, sd, p, stride=1, pad=1, dil=1):
    """convbn (submodule.py:21-25): Conv2d(bias=False, padding = dil if dil>1 else pad) + BatchNorm2d."""
    return _bn2(F.conv2d(x, sd[p + ".0.weight"], None, stride, dil if dil > 1 else pad, dil), sd, p + ".1")


def basic_block(x, sd, p, pad, dil):
    """BasicBlock (submodule.py:192-215), stride 1."""
    y = mish(convbn2d(x, sd, p + ".conv1.0", 1, pad, dil))
    y = convbn2d(y, sd, p + ".conv2", 1, pad, dil)
    if (p + ".downsample.0.weight") in sd:
        x = _bn2(F.conv2d(x, sd[p + ".downsample.0.weight"]), sd, p + ".downsample.1")
    return y + x


def refinenet3(x, disp, sd, p="refinenet3"):
    """refinenet_version3.forward (pwcnet_ddim.py:293-306)."""
    y = mish(convbn2d(x, sd, p + ".conv1.0", 1, 1, 1))
    y = mish(convbn2d(y, sd, p + ".conv2.0", 1, 1, 1))
    y = mish(convbn2d(y, sd, p + ".conv3.0", 1, 2, 2))
    y = mish(convbn2d(y, sd, p + ".conv4.0", 1, 4, 4))
    y = basic_block(y, sd, p + ".conv5.0", 1, 8)
    y = basic_block(y, sd, p + ".conv6.0", 1, 16)
    y = basic_block(y, sd, p + ".conv7.0", 1, 1)
    return disp + F.conv2d(y, sd[p + ".conv8.weight"], None, 1, 1)


def warp(x: Tensor, disp: Tensor) -> Tensor:
    """submodule.py:137-176."""
    b, c, h, w = x.shape
    xx = torch.arange(0, w).view(1, -1).repeat(h, 1).view(1, 1, h, w).repeat(b, 1, 1, 1).float()
    yy = torch.arange(0, h).view(-1, 1).repeat(1, w).view(1, 1, h, w).repeat(b, 1, 1, 1).float()
    vgrid = torch.cat((xx - disp, yy), 1)
    vgrid[:, 0] = 2.0 * vgrid[:, 0].clone() / max(w - 1, 1) - 1.0
    vgrid[:, 1] = 2.0 * vgrid[:, 1].clone() / max(h - 1, 1) - 1.0
    vgrid = vgrid.permute(0, 2, 3, 1)
    out = F.grid_sample(x, vgrid)
    mask = F.grid_sample(torch.ones_like(x), vgrid)
    mask[mask < 0.999] = 0
    mask[mask > 0] = 1
    return out * mask


def warp_valid_mask(disp: Tensor) -> Tensor:
    """The validity mask of ``warp`` alone (submodule.py:170-174): 1 where `grid_sample(ones)` >= 0.999, else 0 -- a HARD
    threshold on a continuous function of the disparity, i.e. a decision two correct evaluations can take differently
    at a pixel whose sampling position lies within rounding of the image border.  disp [B,1,H,W] -> [B,H,W] bool."""
    b, _, h, w = disp.shape
    xx = torch.arange(0, w).view(1, -1).repeat(h, 1).view(1, 1, h, w).repeat(b, 1, 1, 1).to(disp.dtype)
    yy = torch.arange(0, h).view(-1, 1).repeat(1, w).view(1, 1, h, w).repeat(b, 1, 1, 1).to(disp.dtype)
    vgrid = torch.cat((xx - disp, yy), 1)
    vgrid[:, 0] = 2.0 * vgrid[:, 0].clone() / max(w - 1, 1) - 1.0
    vgrid[:, 1] = 2.0 * vgrid[:, 1].clone() / max(h - 1, 1) - 1.0
    mask = F.grid_sample(torch.ones_like(disp), vgrid.permute(0, 2, 3, 1))
    return (mask >= 0.999)[:, 0]


def correlation_pm(ref: Tensor, tgt: Tensor, maxdisp: int) -> Tensor:
    """build_corrleation_volume(ref, tgt, maxdisp, 1).squeeze(1) (submodule.py:121-135)."""
    b, c, h, w = ref.shape
    vol = ref.new_zeros(b, 2 * maxdisp + 1, h, w)
    for i in range(-maxdisp, maxdisp + 1):
        if i > 0:
            vol[:, i + maxdisp, :, i:] = A.groupwise_correlation(ref[..., i:], tgt[..., :-i], 1)[:, 0]
        elif i < 0:   # literal reference slicing: `:-i` with i<0 is the FIRST |i| columns, `i:` the LAST |i|
            vol[:, i + maxdisp, :, :-i] = A.groupwise_correlation(ref[..., :-i], tgt[..., i:], 1)[:, 0]
        else:
            vol[:, maxdisp] = A.groupwise_correlation(ref, tgt, 1)[:, 0]
    return vol


def _deconv_bn(x, sd, p):
    return A._bn(F.conv_transpose3d(x, sd[p + ".0.weight"], None, 2, 1, 1), sd, p + ".1")


def hourglass_mish(x, sd, p):
    """pwcnet_ddim.py:208-248."""
    c1 = mish(A.convbn_3d(x, sd, p + ".conv1.0", 2, 1))
    c2 = mish(A.convbn_3d(c1, sd, p + ".conv2.0", 1, 1))
    c3 = mish(A.convbn_3d(c2, sd, p + ".conv3.0", 2, 1))
    c4 = mish(A.convbn_3d(c3, sd, p + ".conv4.0", 1, 1))
    c5 = mish(_deconv_bn(c4, sd, p + ".conv5") + A.convbn_3d(c2, sd, p + ".redir2", 1, 0))
    return mish(_deconv_bn(c5, sd, p + ".conv6") + A.convbn_3d(x, sd, p + ".redir1", 1, 0))


def hourglassup(x, f4, f5, f6, sd, p):
    """pwcnet_ddim.py:177-205."""
    c1 = F.conv3d(x, sd[p + ".conv1.weight"], None, 2, 1)
    c1 = mish(A.convbn_3d(torch.cat((c1, f4), 1), sd, p + ".combine1.0", 1, 1))
    c2 = mish(A.convbn_3d(c1, sd, p + ".conv2.0", 1, 1))
    c3 = F.conv3d(c2, sd[p + ".conv3.weight"], None, 2, 1)
    c3 = mish(A.convbn_3d(torch.cat((c3, f5), 1), sd, p + ".combine2.0", 1, 1))
    c4 = mish(A.convbn_3d(c3, sd, p + ".conv4.0", 1, 1))
    c5 = F.conv3d(c4, sd[p + ".conv5.weight"], None, 2, 1)
    c5 = mish(A.convbn_3d(torch.cat((c5, f6), 1), sd, p + ".combine3.0", 1, 1))
    c6 = mish(A.convbn_3d(c5, sd, p + ".conv6.0", 1, 1))
    c7 = mish(_deconv_bn(c6, sd, p + ".conv7") + A.convbn_3d(c4, sd, p + ".redir3", 1, 0))
    c8 = mish(_deconv_bn(c7, sd, p + ".conv8") + A.convbn_3d(c2, sd, p + ".redir2", 1, 0))
    return mish(_deconv_bn(c8, sd, p + ".conv9") + A.convbn_3d(x, sd, p + ".redir1", 1, 0))


def conv_mish_conv(x, sd, p, mish_last, bn_last=True):
    y = mish(A.convbn_3d(x, sd, p + ".0", 1, 1))
    y = A.convbn_3d(y, sd, p + ".2", 1, 1) if bn_last else F.conv3d(y, sd[p + ".2.weight"], None, 1, 1)
    return mish(y) if mish_last else y


def fused_volume(fl: Dict[str, Tensor], fr: Dict[str, Tensor], sd: SD, maxdisp: int = 192) -> Tensor:
    """pwcnet_ddim.py:608-641 given the feature dictionaries."""
    vols = []
    for i, div in enumerate((4, 8, 16, 32), start=1):
        g = A.build_gwc_volume(fl[f"gw{i}"], fr[f"gw{i}"], maxdisp // div, 40)
        c = A.build_concat_volume(fl[f"concat_feature{i}"], fr[f"concat_feature{i}"], maxdisp // div, zero_left=True)
        vols.append(torch.cat((g, c), 1))
    cost0 = conv_mish_conv(vols[0], sd, "dres0", True)
    cost0 = conv_mish_conv(cost0, sd, "dres1", False) + cost0
    return hourglassup(cost0, vols[1], vols[2], vols[3], sd, "combine1")


class PCWDiffusionOracle:
    def __init__(self, sd: SD, maxdisp: int = 192, sampling_timesteps: int = 3,
                 cof: Sequence[float] = (0.9, 0.0, 0.0, 0.1)):
        self.sd, self.maxdisp, self.scale = sd, maxdisp, 1.0
        self.num_timesteps, self.sampling_timesteps, self.eta, self.cof = 1000, sampling_timesteps, 1.0, tuple(cof)
        ac = A.cosine_alphas_cumprod(1000)
        self.alphas_cumprod = ac
        self.sqrt_alphas_cumprod, self.sqrt_one_minus = torch.sqrt(ac), torch.sqrt(1.0 - ac)
        self.sqrt_recip, self.sqrt_recipm1 = torch.sqrt(1.0 / ac), torch.sqrt(1.0 / ac - 1)

    def aggregate(self, volume: Tensor) -> Tensor:
        out = hourglass_mish(hourglass_mish(hourglass_mish(volume, self.sd, "dres2"), self.sd, "dres3"), self.sd, "dres4")
        return conv_mish_conv(out, self.sd, "classif3", False, bn_last=False)

    def refine(self, pred3: Tensor, fl, fr) -> Tensor:
        """pwcnet_ddim.py:486-502."""
        hh, ww = pred3.shape[-2:]
        p3 = pred3.unsqueeze(1)
        left = F.interpolate(fl["finetune_feature"], [hh, ww], mode="bilinear", align_corners=True)
        right = F.interpolate(fr["finetune_feature"], [hh, ww], mode="bilinear", align_corners=True)
        rw = warp(right, p3)
        cv = correlation_pm(left, rw, 24)
        p3f = mish(convbn2d(p3, self.sd, "dispupsample.0", 1, 0, 1))
        comb = torch.cat((left - rw, left, p3f, p3, cv), dim=1)
        return refinenet3(comb, p3, self.sd).squeeze(1)

    def model_predictions(self, volume, x_t, t, fl, fr):
        shift = A.time_shift(t, self.sd)[:, :, None, None]
        n01 = ((torch.clamp(x_t + shift, -self.scale, self.scale) / self.scale) + 1) / 2
        cost = self.aggregate(volume * n01.unsqueeze(1).float())
        pred3, prob = A.upsample_softmax_regress(cost, self.maxdisp, align_corners=True)
        self.last_pred3 = pred3                 # (kept for the parity bookkeeping: the input of the 2-D refinement)
        disp = self.refine(pred3, fl, fr)
        dn = torch.clamp(disp, 0, self.maxdisp - 1).unsqueeze(1)
        hh, ww = dn.shape[-2:]
        dn = F.interpolate(dn, size=(hh // 4, ww // 4), mode="bilinear") / 4
        x_start = torch.clamp(self.scale * (A.encode_two_hot(dn, 48) * 2 - 1.0), -self.scale, self.scale)
        bs = (x_t.shape[0], 1, 1, 1)
        pred_noise = (self.sqrt_recip.gather(-1, t).reshape(bs) * n01 - x_start) / self.sqrt_recipm1.gather(-1, t).reshape(bs)
        return pred_noise, x_start, disp, prob

    def ddim_sample(self, volume, used, asd, fl, fr, draw: Callable[[str, Tuple[int, ...], torch.dtype], Tensor],
                    trace=None):
        """pwcnet_ddim.py:530-602; draws: 'x_T' (:541), then per non-final step 'eps' (:585), 'q' (:590).
        ``trace`` (a list) receives one dict per step: the state entering it, its outputs and draws
        ('fill' = the q_sample'd origin encoding that replaces never-confirmed pixels, :590-594)."""
        b, _, _, h, w = volume.shape
        img = draw("x_T", (b, 48, h, w), torch.float32)
        final = [used.unsqueeze(0)]
        mask = torch.zeros(b, h, w)
        times = torch.linspace(-1, 999, steps=self.sampling_timesteps + 1)
        times = list(reversed(times.int().tolist()))
        for time, time_next in zip(times[:-1], times[1:]):
            t = torch.full((b,), time, dtype=torch.long)
            pred_noise, x_start, disp, prob = self.model_predictions(volume, img, t, fl, fr)
            final.append(disp.unsqueeze(0))
            unc = A.disparity_uncertainty(disp, prob)
            rec = None
            if trace is not None:
                rec = {"time": time, "time_next": time_next, "img": img, "mask_in": mask, "disp": disp, "unc": unc,
                       "pred3": self.last_pred3,
                       "x_start": x_start, "eps": None, "fill": None, "img_next": None, "mask_out": None}
                trace.append(rec)
            if time_next >= 0:
                keep = ((torch.abs(disp - used) < 1) & (unc < 1)).float()
                keep = F.interpolate(keep.unsqueeze(1), size=(h, w), mode="bilinear").squeeze(1)
                mask = torch.clamp(mask + keep, 0, 1)
                if rec is not None:
                    rec["mask_out"] = mask
            else:
                img = x_start
                continue
            alpha, alpha_next = self.alphas_cumprod[time], self.alphas_cumprod[time_next]
            sigma = self.eta * ((1 - alpha / alpha_next) * (1 - alpha_next) / (1 - alpha)).sqrt()
            c = (1 - alpha_next - sigma ** 2).sqrt()
            eps = draw("eps", tuple(img.shape), img.dtype)
            img = x_start * alpha_next.sqrt() + c * pred_noise + sigma * eps
            tt = torch.full((1,), time, dtype=torch.long)
            asd = (self.sqrt_alphas_cumprod.gather(-1, tt).reshape(1, 1, 1, 1) * asd
                   + self.sqrt_one_minus.gather(-1, tt).reshape(1, 1, 1, 1) * draw("q", tuple(asd.shape), asd.dtype))
            img = torch.where(mask.unsqueeze(1) == 0, asd, img)
            if rec is not None:
                rec["eps"], rec["fill"], rec["img_next"] = eps, asd, img
        stack = torch.cat(final, dim=0)
        return torch.sum(stack * torch.tensor(self.cof).view(-1, 1, 1, 1), dim=0), stack
