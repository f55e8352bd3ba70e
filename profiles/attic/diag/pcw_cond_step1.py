"""Conditioned KITTI12 network at 1248x384, the teacher-forced comparison of tests/test_gpu_configs.py taken apart for
step 1: which input (HIP volume vs oracle volume), which call (first vs repeated) and which stage (cost, pred3, disp)."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import conditioned_pcw_state_dict  # noqa: E402
from diffuvolume_amd.pwcnet_ddim import PWCNet_ddim  # noqa: E402
from diffuvolume_amd.submodule import upsample_softmax_regress  # noqa: E402
from diffuvolume_amd.synth import synth_stereo_batch  # noqa: E402
from oracle import acv_oracle as A  # noqa: E402
from oracle import loop_parity as LP  # noqa: E402
from oracle import pcw_oracle as P  # noqa: E402

DEV = "cuda:0"
H, W = 384, 1248


def stats(a, b, bar=1e-3):
    d = (a.double().cpu() - b.double().cpu()).abs()
    return {"mean": float(d.mean()), "max": float(d.max()), "frac_gt_bar": float((d > bar).float().mean())}


def main():
    torch.set_num_threads(max(1, (os.cpu_count() or 2) // 2))
    sd, _ = conditioned_pcw_state_dict("pcw_conditioned_config4")
    m = PWCNet_ddim(192, True)
    m.load_state_dict(sd, strict=True)
    m = m.to(DEV).eval()
    batch = {k: v.to(DEV) for k, v in synth_stereo_batch(4, H, W, seed=0).items()}
    out = {}
    with torch.no_grad():
        fl, fr = m.feature_extraction(batch["left"]), m.feature_extraction(batch["right"])
        dl, dr = {k: v[:1] for k, v in fl.items()}, {k: v[:1] for k, v in fr.items()}
        fl0, fr0 = {k: v.cpu() for k, v in dl.items()}, {k: v.cpu() for k, v in dr.items()}
        vol_d = m.fused_volume(dl, dr)
        asd = m.encode_disparity(batch["disp"][:1])
        vol = P.fused_volume(fl0, fr0, sd)
        used = batch["used"][:1].cpu()
        orc = P.PCWDiffusionOracle(sd)
        _, _, trace = LP.oracle_trajectory(orc, vol, used, asd.cpu(), 11, fl0, fr0)
        r = trace[0]
        t = torch.full((1,), r["time"], dtype=torch.long)
        shift = A.time_shift(t, sd)[:, :, None, None]
        n01 = ((torch.clamp(r["img"] + shift, -1, 1)) + 1) / 2
        cost_o = orc.aggregate(vol * n01.unsqueeze(1).float())
        pred3_o, _ = A.upsample_softmax_regress(cost_o, 192, align_corners=True)
        out["img_dtype"] = str(r["img"].dtype)
        out["img_absmax"] = float(r["img"].abs().max())
        for name, v in (("hip_volume_first_call", vol_d), ("hip_volume_again", vol_d), ("oracle_volume", vol.to(DEV))):
            mask = r["mask_in"].to(DEV).clone()
            disp, unc, xs, xn, cost = m.ddim_step(0, v, batch["used"][:1], r["img"].to(DEV), mask, None, r["eps"].to(DEV),
                                                  r["fill"].to(DEV), dl, dr, want_cost=True)
            p3, _ = upsample_softmax_regress(cost, want_uncertainty=False, align_corners=True)
            out[name] = {"cost": stats(cost, cost_o, 1e-4), "pred3": stats(p3, pred3_o), "disp": stats(disp, r["disp"])}
        # a different x_T of the same kind (the diag that passed used its own draw)
        out["volume"] = stats(vol_d, vol, 1e-5)
        d = (vol_d.cpu() - vol).abs()
        out["volume_err_by_channel_max"] = [float(d[0, c].max()) for c in range(32)]
        out["volume_err_by_dbin_max"] = [float(d[0, :, k].max()) for k in range(48)]
    with open(os.path.join(ROOT, "gpurun_out", "pcw_cond_step1.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out))


if __name__ == "__main__":
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    main()
