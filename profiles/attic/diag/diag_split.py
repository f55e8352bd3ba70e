"""Where does the 1.3e-4 px mean distance between the HIP step and the oracle step come from?  Full size, pair 0,
DDIM step 1.  Splits it into the aggregation stack (26 conv layers -> cost) and the regression tail (trilinear x4,
softmax over 192 bins, soft-argmax) by crossing the two implementations, and measures both sides against float64.
    python tools/diag/diag_split.py            (GPU box; prints one JSON line)"""
import json
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import diffuvolume_amd as dv  # noqa: E402
from diffuvolume_amd import submodule as S  # noqa: E402
from diffuvolume_amd.synth import synth_hot_inputs, synth_state_dict  # noqa: E402
from oracle import acv_oracle as O  # noqa: E402

DEV = "cuda:0"


def stats(a, b):
    d = (a.double() - b.double()).abs()
    return {"mean": float(d.mean()), "frac_gt_1e-3": float((d > 1e-3).double().mean()), "max": float(d.max())}


def main():
    h, w = (128, 240) if len(sys.argv) < 2 else (int(sys.argv[1]), int(sys.argv[2]))
    sd = synth_state_dict(dv.ACVNet_DDIM(192, False, False).state_dict(), seed=1, logit_gain=8.0)
    x = synth_hot_inputs(1, h, w, seed=100)
    orc = O.ACVDiffusionOracle(sd)
    vol = O.attention_concat_volume(x["att"], O.build_concat_volume(x["cl"], x["cr"], 48))
    x_T = orc.encode_x_T(x["dq"])
    t = torch.full((1,), 999, dtype=torch.long)
    n01 = orc.noise_to_filter(x_T, t)
    c_o = orc.aggregate(vol * n01.unsqueeze(1))
    out = {"cost_abs_mean": float(c_o.abs().mean()), "cost_abs_max": float(c_o.abs().max())}
    d_oo, prob = O.upsample_softmax_regress(c_o, 192)
    out["unc_mean"] = float(O.disparity_uncertainty(d_oo, prob).mean())
    del prob
    d64 = O.upsample_softmax_regress(c_o.double(), 192)[0]
    out["oracle_tail_vs_fp64_tail_same_cost"] = stats(d_oo, d64)
    res = {}
    for prec in ("f32", "f32_direct"):
        S.set_default_conv_precision(prec)
        model = dv.ACVNet_DDIM(192, False, False)
        model.load_state_dict(sd, strict=True)
        model = model.to(DEV).eval()
        with torch.no_grad():
            _, n01f = model._filter(x_T.to(DEV), t.to(DEV))
            c_h = model._aggregate(vol.to(DEV), n01f)
            d_hh = S.upsample_softmax_regress(c_h, want_uncertainty=False)[0].cpu()
            d_ho = S.upsample_softmax_regress(c_o.to(DEV), want_uncertainty=False)[0].cpu()
        c_h = c_h.cpu()
        d_oh = O.upsample_softmax_regress(c_h, 192)[0]
        res[prec] = {"cost_diff": stats(c_h, c_o),
                     "hip_stack_oracle_tail_vs_oracle": stats(d_oh, d_oo),
                     "oracle_stack_hip_tail_vs_oracle": stats(d_ho, d_oo),
                     "hip_tail_vs_fp64_tail_same_cost": stats(d_ho, d64),
                     "hip_vs_oracle": stats(d_hh, d_oo)}
    S.set_default_conv_precision(None)
    out["by_precision"] = res
    print(json.dumps(out))


if __name__ == "__main__":
    main()
