"""Parity bookkeeping for the 5-step DDIM volume-filter loop: HIP path vs the CPU oracle.

TEST INFRASTRUCTURE ONLY (imported by tests/, __graft_entry__.smoke() and the cpu_baseline /
parity leg of bench.py).  Nothing under ``diffuvolume_amd/`` imports it.

Why three comparisons.  The loop of SceneFlow/models/acv_ddim.py:298-370 feeds two hard decisions back
into its state: the renewal mask ``|disp-used| < 1 & unc < 3`` (:322-338; a quarter-resolution pixel whose
accumulated mask is exactly 0 is *replaced by uniform noise*, :362) and, harmlessly, ``floor()`` of the
two-hot encoder (:280; the two-hot weights are continuous across a bin boundary).  A pixel that sits within
fp32 rounding of a threshold can therefore take a different branch in two correct implementations, and
the next step then sees a different 48-vector at that pixel.  So:

1. **teacher-forced** (``teacher_forced``): every step i of the HIP path is run from the ORACLE's state
   entering step i (img, mask, draws).  Each step is then an ordinary function comparison: the north-star
   bars (|d disp| <= 1e-3 px on 99.9 % of the pixels, |EPE_hip - EPE_oracle| < 1e-4) are asserted on it, and
   the decisions that come out differently are counted -- these are the *primary flips*;
2. **decision-forced** (``decision_forced``): the HIP path runs free on its own state, but after every
   step the discrete renewal decision (mask == 0 or not) of the oracle is imposed on the pixels where the
   two differ.  If this run stays within the bars, HIP arithmetic does not accumulate error across steps and
   every remaining difference of the free run is caused by flips;
3. **free** (``free_run``): both run on their own; reported with the flip counts per step.
"""
from __future__ import annotations

from typing import Dict, List

import torch
import torch.nn.functional as F

from diffuvolume_amd.synth import NoiseTape

BAR_PX = 1e-3          # north star: disparity maps within 1e-3 px ...
BAR_FRAC = 1e-3        # ... on 99.9 % of the pixels
BAR_EPE = 1e-4         # EPE within 1e-4
UNC_CONFIDENT = 3.0    # the reference's own confidence criterion: uncertainty < 3 px (acv_ddim.py:330)


def oracle_trajectory(orc, vol, used, x_T, seed, *features):
    """Free run of the oracle with a NoiseTape; returns (final, stack, per-step records).  ``features``: the two
    feature dictionaries of the KITTI12 flavour (pcw_oracle.PCWDiffusionOracle.ddim_sample)."""
    trace: List[dict] = []
    final, stack = orc.ddim_sample(vol, used, x_T, *features, NoiseTape(seed), trace=trace)
    return final, stack, trace


def _epe(disp, gt):
    m = (gt > 0) & (gt < 192)
    return float((disp.double() - gt.double()).abs()[m].mean())


def _keep(disp, unc, used, model):
    return ((disp - used).abs() < model.dif_threshold) & (unc < model.unc_threshold)


def _stats(d, unc=None):
    """Distance statistics of one step.
    ``frac_gt_1e-3`` is the RAW contract figure over all pixels.
    ``share_unc_lt_3`` is the share of pixels the reference itself calls confident (uncertainty = sum_k p_k |k - disp|
    < 3 px, its own renewal criterion, acv_ddim.py:325-330) and ``frac_gt_1e-3_where_unc_lt_3`` the raw 1e-3 px bar on
    exactly those pixels -- unscaled, so it can fail.
    ``frac_gt_bar`` is a builder-defined diagnostic, NOT the contract: the bar grows as unc/3 on the other pixels (a
    soft-argmax moves by at most unc * max|d cost| when the cost moves, so the disparity of a flat distribution --
    untrained weights: unc ~ 50 px -- is 17x more sensitive to the last bit of the fp32 cost than a peaked one)."""
    s = {"mean_abs_px": float(d.mean()), "frac_gt_1e-3": float((d > BAR_PX).float().mean()), "max_px": float(d.max())}
    if unc is not None:
        conf = unc < UNC_CONFIDENT
        s["share_unc_lt_3"] = float(conf.float().mean())
        s["frac_gt_1e-3_where_unc_lt_3"] = float((d[conf] > BAR_PX).float().mean()) if bool(conf.any()) else 0.0
        s["max_px_where_unc_lt_3"] = float(d[conf].max()) if bool(conf.any()) else 0.0
        bar = BAR_PX * torch.clamp(unc / UNC_CONFIDENT, min=1.0)
        s["frac_gt_bar"] = float((d > bar).float().mean())
        s["unc_mean_px"] = float(unc.mean())
        s["max_err_over_unc"] = float((d / unc.clamp(min=1e-3)).max())
    return s


@torch.no_grad()
def teacher_forced(model, trace, vol_d, used_d, used, gt, **step_kw) -> List[Dict]:
    """Per step: HIP from the oracle's state.  Returns one dict per step.  ``step_kw``: extra keyword arguments of
    the model's ``ddim_step`` (the KITTI12 flavour's feature dictionaries)."""
    dev = vol_d.device
    out = []
    for i, r in enumerate(trace):
        mask = r["mask_in"].to(dev).clone()
        eps = None if r["eps"] is None else r["eps"].to(dev)
        fill = None if r["fill"] is None else r["fill"].to(dev)
        disp, unc, xs, xn = model.ddim_step(i, vol_d, used_d, r["img"].to(dev), mask, None, eps, fill, **step_kw)
        disp, unc, xs, mask = disp.cpu(), unc.cpu(), xs.cpu(), mask.cpu()
        s = _stats((disp - r["disp"]).abs(), r["unc"])
        s["step"] = i + 1
        s["epe_hip"], s["epe_oracle"] = _epe(disp, gt), _epe(r["disp"], gt)
        s["epe_delta"] = abs(s["epe_hip"] - s["epe_oracle"])
        s["unc_mean_abs"] = float((unc - r["unc"]).abs().mean())
        # decisions
        s["flips_keep_fullres"] = int((_keep(disp, unc, used, model) != _keep(r["disp"], r["unc"], used, model)).sum())
        mask_o = mask if r["mask_out"] is None else r["mask_out"]      # KITTI12: no renewal on the last step
        zero_h, zero_o = mask == 0, mask_o == 0
        s["flips_mask_zero"] = int((zero_h != zero_o).sum())
        s["mask_max_abs"] = float((mask - mask_o).abs().max())
        same_bins = ((xs - r["x_start"]).abs() < 1e-2).all(dim=1)
        s["flips_floor_bin"] = int((~same_bins).sum())
        if xn is not None:
            agree = (zero_h == zero_o).unsqueeze(1).expand_as(r["img_next"])
            dx = (xn.cpu() - r["img_next"]).abs()[agree]
            s["x_next_max_abs_where_decisions_agree"] = float(dx.max())
            s["x_next_mean_abs_where_decisions_agree"] = float(dx.mean())
        out.append(s)
    return out


@torch.no_grad()
def pcw_teacher_forced_split(model, trace, vol_d, used_d, features_left, features_right) -> List[Dict]:
    """KITTI12 step taken apart at the one hard decision inside it.  The per-step disparity of this flavour is
    ``refine(pred3)`` (pwcnet_ddim.py:486-502), and ``warp`` inside the refinement zeroes the warped feature wherever
    ``grid_sample(ones) < 0.999`` (submodule.py:170-174): a pixel whose sampling position x - pred3 lies within rounding
    of the image border takes the other branch in two correct evaluations, and the 9 dilated 2-D layers spread that one
    flipped feature vector over a +-61-pixel neighbourhood (measured at 1248x384: one flipped pixel -> 2 % of the image
    beyond 1e-3 px, max 0.1 px).  So, per step of ``trace`` and from the ORACLE's state:
      pred3            HIP 3-D stack + regression vs the oracle's pred3                      (function comparison)
      refine           HIP refinement FROM THE ORACLE'S pred3 vs the oracle's disparity      (function comparison)
      warp_mask_flips  pixels where the validity decision differs between the two pred3's
      disp             the composite step (raw figures; within the bars whenever warp_mask_flips == 0)"""
    from . import pcw_oracle as P
    from diffuvolume_amd.submodule import upsample_softmax_regress
    dev = vol_d.device
    out = []
    for i, r in enumerate(trace):
        mask = r["mask_in"].to(dev).clone()
        eps = None if r["eps"] is None else r["eps"].to(dev)
        fill = None if r["fill"] is None else r["fill"].to(dev)
        disp, unc, xs, xn, cost = model.ddim_step(i, vol_d, used_d, r["img"].to(dev), mask, None, eps, fill,
                                                  features_left, features_right, want_cost=True)
        pred3_h, _ = upsample_softmax_regress(cost, want_uncertainty=False, align_corners=True)
        refined = model._refine(r["pred3"].to(dev), features_left, features_right).cpu()
        pred3_h = pred3_h.cpu()
        flips = P.warp_valid_mask(pred3_h.unsqueeze(1)) != P.warp_valid_mask(r["pred3"].unsqueeze(1))
        s = {"step": i + 1, "pred3": _stats((pred3_h - r["pred3"]).abs()), "refine": _stats((refined - r["disp"]).abs()),
             "disp": _stats((disp.cpu() - r["disp"]).abs()), "warp_mask_flips": int(flips.sum())}
        out.append(s)
    return out


@torch.no_grad()
def teacher_forced_vs_fp64(model, orc32, orc64, trace, vol, vol_d, used_d, oracle_args=(), **step_kw) -> List[Dict]:
    """Triangulation against a float64 evaluation of the reference's function.  For every step of ``trace`` (the fp32
    oracle's run), from the SAME entering state: disparity of the HIP path, of the fp32 oracle and of the oracle with
    float64 weights and activations.  Returns per step the raw statistics of |HIP - fp64| and |fp32 oracle - fp64|
    (``frac_gt_1e-3`` = the contract's pixel figure, unscaled) -- two fp32 evaluations of a network can agree with
    each other no better than the sum of their distances to this one.  ``oracle_args``: extra positional arguments of
    the float64 oracle's ``model_predictions`` (the KITTI12 flavour's feature dictionaries, in float64)."""
    dev = vol_d.device
    out = []
    vol64 = vol.double()
    for i, r in enumerate(trace):
        mask = r["mask_in"].to(dev).clone()
        eps = None if r["eps"] is None else r["eps"].to(dev)
        fill = None if r["fill"] is None else r["fill"].to(dev)
        disp_h = model.ddim_step(i, vol_d, used_d, r["img"].to(dev), mask, None, eps, fill, **step_kw)[0].cpu().double()
        t = torch.full((vol.shape[0],), r["time"], dtype=torch.long)
        _, _, disp64, prob64 = orc64.model_predictions(vol64, r["img"], t, *oracle_args)
        k = torch.arange(0, prob64.shape[1], dtype=torch.float64).view(1, -1, 1, 1)
        unc64 = torch.sum(torch.abs(disp64.unsqueeze(1) - k) * prob64, dim=1)
        del prob64
        sh = _stats((disp_h - disp64).abs(), unc64)
        so = _stats((r["disp"].double() - disp64).abs(), unc64)
        sm = _stats((disp_h - r["disp"].double()).abs(), unc64)
        out.append({"step": i + 1, "hip_vs_fp64": sh, "oracle32_vs_fp64": so, "hip_vs_oracle32": sm})
    return out


@torch.no_grad()
def decision_forced(model, trace, vol_d, used_d, x_T, gt, **step_kw) -> List[Dict]:
    """HIP free run on its own state with the oracle's renewal decisions imposed after every step."""
    dev = vol_d.device
    steps = model._loop_plan()
    b, _, d, h, w = vol_d.shape
    img = x_T.to(dev)
    mask = torch.zeros((b, h, w), dtype=torch.float32, device=dev)
    out = []
    for i, (st, r) in enumerate(zip(steps, trace)):
        eps = None if r["eps"] is None else r["eps"].to(dev)
        fill = None if r["fill"] is None else r["fill"].to(dev)
        img_in = img
        disp, unc, xs, xn = model.ddim_step(i, vol_d, used_d, img_in, mask, None, eps, fill, **step_kw)
        s = _stats((disp.cpu() - r["disp"]).abs(), r["unc"])
        s["step"] = i + 1
        s["epe_delta"] = abs(_epe(disp.cpu(), gt) - _epe(r["disp"], gt))
        if r["mask_out"] is None:
            s["decisions_imposed"] = 0
            out.append(s)
            img = xs if xn is None else xn
            continue
        zero_o = (r["mask_out"] == 0).to(dev)
        flip = (mask == 0) != zero_o
        s["decisions_imposed"] = int(flip.sum())
        mask = r["mask_out"].to(dev).clone()            # the accumulated mask only matters through `== 0`
        if xn is not None and bool(flip.any()):
            # acv_ddim.py:348-362 restated for the flipped pixels only, from HIP's own x_start and state
            k = st.coef
            shift = st.shift_rows(b).double().view(b, d, 1, 1)
            n01 = ((img_in.double() + shift).clamp(-1.0, 1.0) + 1.0) / 2.0
            pn = (k.sqrt_recip_alpha * n01 - xs.double()) / k.sqrt_recipm1_alpha
            if eps.dtype == torch.float32:
                se = (torch.tensor(k.sigma, dtype=torch.float64).float().to(dev) * eps).double()
            else:
                se = k.sigma * eps
            upd = (xs * float(torch.tensor(k.sqrt_alpha_next, dtype=torch.float64).float())).double() + k.c * pn + se
            want = torch.where(zero_o.unsqueeze(1), fill, upd)
            xn = torch.where(flip.unsqueeze(1), want, xn)
        img = xs if xn is None else xn
        out.append(s)
    return out


@torch.no_grad()
def free_run(model, trace, stack_o, final_o, vol_d, used_d, x_T, gt, seed, *features) -> Dict:
    """Both sides on their own state; per-step distance and the number of renewal decisions that differ."""
    dev = vol_d.device
    masks, disps = [], []

    def spy(i, st):
        if st["when"] == "out":
            masks.append(st["mask"].cpu())
            disps.append(st["disp"].cpu())

    ret = model.ddim_sample(vol_d, used_d, x_T.to(dev), *features, noise=NoiseTape(seed), trace=spy)
    final_h = ret[0].cpu()
    steps = []
    for i, r in enumerate(trace):
        s = _stats((disps[i] - stack_o[i + 1]).abs(), r["unc"])
        s["step"] = i + 1
        s["flips_mask_zero"] = 0 if r["mask_out"] is None else int(((masks[i] == 0) != (r["mask_out"] == 0)).sum())
        s["epe_delta"] = abs(_epe(disps[i], gt) - _epe(stack_o[i + 1], gt))
        steps.append(s)
    fin = _stats((final_h - final_o).abs())
    fin["epe_hip"], fin["epe_oracle"] = _epe(final_h, gt), _epe(final_o, gt)
    fin["epe_delta"] = abs(fin["epe_hip"] - fin["epe_oracle"])
    return {"steps": steps, "final": fin}


def explained_by_flips(stack_h, stack_o, flip_maps, radius: int):
    """Fraction of the offending full-resolution pixels (|d| > 1e-3 px) of each step that lie within `radius`
    quarter-resolution pixels of a renewal decision that differed at an EARLIER step."""
    out = []
    seen = torch.zeros_like(flip_maps[0], dtype=torch.float32)
    for i in range(len(flip_maps)):
        off = (stack_h[i + 1] - stack_o[i + 1]).abs() > BAR_PX
        if i == 0 or not bool(seen.any()):
            near = torch.zeros_like(off)
        else:
            k = 2 * radius + 1
            near_q = F.max_pool2d(seen.unsqueeze(1), k, 1, radius).squeeze(1) > 0
            near = near_q.repeat_interleave(4, dim=-2).repeat_interleave(4, dim=-1)
        n_off = int(off.sum())
        out.append({"step": i + 1, "offending": n_off, "explained": int((off & near).sum())})
        seen = torch.maximum(seen, flip_maps[i].float())
    return out
