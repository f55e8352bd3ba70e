"""Where do HIP and the oracle part on the conditioned KITTI12 network at 1248x384?  Stage by stage on pair 0, step 1:
cost (3-D stack) -> pred3 (regression) -> refinement inputs -> refined disparity, with the spatial distribution of the
pixels beyond 1e-3 px.  Writes gpurun_out/pcw_cond_stages.json.  (GPU box; ~1 min of host CPU.)"""
import json
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import conditioned_pcw_state_dict  # noqa: E402
from diffuvolume_amd.pwcnet_ddim import PWCNet_ddim  # noqa: E402
from diffuvolume_amd.submodule import refine_inputs, upsample_softmax_regress  # noqa: E402
from diffuvolume_amd.synth import _gen, synth_stereo_batch  # noqa: E402
from oracle import acv_oracle as A  # noqa: E402
from oracle import pcw_oracle as P  # noqa: E402

DEV = "cuda:0"
H, W = 384, 1248


def stats(a, b, bar=1e-3):
    d = (a.double().cpu() - b.double().cpu()).abs()
    return {"mean": float(d.mean()), "max": float(d.max()), "frac_gt_bar": float((d > bar).float().mean()),
            "ref_absmax": float(b.abs().max())}


def main():
    torch.set_num_threads(max(1, (os.cpu_count() or 2) // 2))
    sd, _ = conditioned_pcw_state_dict("pcw_conditioned_config4")
    m = PWCNet_ddim(192, True)
    m.load_state_dict(sd, strict=True)
    m = m.to(DEV).eval()
    batch = {k: v[:1].to(DEV) for k, v in synth_stereo_batch(4, H, W, seed=0).items()}
    out = {}
    with torch.no_grad():
        fl, fr = m.feature_extraction(batch["left"]), m.feature_extraction(batch["right"])
        fl0, fr0 = {k: v.cpu() for k, v in fl.items()}, {k: v.cpu() for k, v in fr.items()}
        vol_d = m.fused_volume(fl, fr)
        vol = P.fused_volume(fl0, fr0, sd)
        out["fused_volume"] = stats(vol_d, vol, 1e-5)
        x_t = torch.randn(1, 48, 96, 312, generator=_gen(9, "xt"))
        t = torch.full((1,), 999, dtype=torch.long)
        orc = P.PCWDiffusionOracle(sd)
        shift = A.time_shift(t, sd)[:, :, None, None]
        n01 = ((torch.clamp(x_t + shift, -1, 1)) + 1) / 2
        cost = orc.aggregate(vol * n01.unsqueeze(1))
        pred3, prob = A.upsample_softmax_regress(cost, 192, align_corners=True)
        # HIP from the ORACLE's volume, stage by stage
        n01_d, n01f_d = m._filter(x_t.to(DEV), t.to(DEV))
        out["n01"] = stats(n01f_d, n01, 1e-6)
        cost_d = m._aggregate(vol.to(DEV), n01f_d)
        out["cost"] = stats(cost_d, cost, 1e-4)
        pred3_d, _ = upsample_softmax_regress(cost_d, want_uncertainty=False, align_corners=True)
        out["pred3"] = stats(pred3_d, pred3)
        pred3_from_oracle_cost, _ = upsample_softmax_regress(cost.to(DEV), want_uncertainty=False, align_corners=True)
        out["pred3_regression_only"] = stats(pred3_from_oracle_cost, pred3)
        # refinement from the ORACLE's pred3
        p3 = pred3.unsqueeze(1)
        left = F.interpolate(fl0["finetune_feature"], [H, W], mode="bilinear", align_corners=True)
        right = F.interpolate(fr0["finetune_feature"], [H, W], mode="bilinear", align_corners=True)
        rw = P.warp(right, p3)
        cv = P.correlation_pm(left, rw, 24)
        p3f = P.mish(P.convbn2d(p3, sd, "dispupsample.0", 1, 0, 1))
        comb = torch.cat((left - rw, left, p3f, p3, cv), dim=1)
        disp = P.refinenet3(comb, p3, sd).squeeze(1)
        plans = m.prepare()
        rl, rr = m.refine_features(fl, fr, (H, W))
        out["resized_left"] = stats(rl, left, 1e-5)
        comb_d = refine_inputs(rl, rr, p3.to(DEV), plans.du_a, plans.du_b, 24)
        names = [("left_minus_warp", 0, 32), ("left", 32, 64), ("dispupsample", 64, 96), ("disp", 96, 97), ("corr", 97, 146)]
        for n, lo, hi in names:
            out["refine_inputs." + n] = stats(comb_d[:, lo:hi], comb[:, lo:hi], 1e-4)
        disp_d = plans.refinenet3(comb.to(DEV), p3.to(DEV).contiguous()).squeeze(1)
        out["refinenet3_on_oracle_inputs"] = stats(disp_d, disp)
        disp_d2 = plans.refinenet3(comb_d, p3.to(DEV).contiguous()).squeeze(1)
        out["refine_on_hip_inputs"] = stats(disp_d2, disp)
        full = m._refine(pred3_d, fl, fr)
        out["disp_end_to_end"] = stats(full, disp)
        # where are the bad pixels of the end-to-end disparity?
        d = (full.cpu() - disp).abs()[0]
        bad = d > 1e-3
        out["bad_share_by_column_band_of_78"] = [float(bad[:, i:i + 78].float().mean()) for i in range(0, W, 78)]
        out["bad_share_by_row_band_of_48"] = [float(bad[i:i + 48].float().mean()) for i in range(0, H, 48)]
        e3 = (pred3_d.cpu() - pred3).abs()[0]
        out["pred3_err_where_disp_bad"] = {"mean": float(e3[bad].mean()) if bool(bad.any()) else 0.0,
                                           "mean_elsewhere": float(e3[~bad].mean())}
        # sensitivity of the oracle's refinement to a 1e-4 px perturbation of pred3 (float64)
        sd64 = {k: (v.double() if v.is_floating_point() and not k.startswith("time_embedding") else v) for k, v in sd.items()}
        o64 = P.PCWDiffusionOracle(sd64)
        f64 = lambda dct: {k: v.double() for k, v in dct.items()}
        base = o64.refine(pred3.double(), f64(fl0), f64(fr0))
        delta = 1e-4 * torch.randn(pred3.shape, dtype=torch.float64, generator=_gen(1, "d"))
        pert = o64.refine(pred3.double() + delta, f64(fl0), f64(fr0))
        g = (pert - base).abs()
        out["oracle64_refine_sensitivity_to_1e-4"] = {"mean_out_over_mean_in": float(g.mean() / delta.abs().mean()),
                                                       "max_out": float(g.max()), "frac_out_gt_1e-3": float((g > 1e-3).float().mean())}
        out["oracle32_vs_64_refine_same_pred3"] = stats(disp, base)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "pcw_cond_stages.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
