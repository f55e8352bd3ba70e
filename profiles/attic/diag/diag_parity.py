"""GPU diagnostic: stage-by-stage error of the HIP aggregation stack against the CPU oracle."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import torch
import diffuvolume_amd as dv
from diffuvolume_amd.synth import _gen, synth_state_dict
from oracle import acv_oracle as O

torch.manual_seed(0)
sd = synth_state_dict(dv.ACVNet_DDIM(192).state_dict(), seed=1, logit_gain=8.0)
m = dv.ACVNet_DDIM(192); m.load_state_dict(sd); m = m.cuda().eval(); p = m.prepare()
vol = torch.rand(1, 64, 48, 16, 32, generator=_gen(31, "vol"))
sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}

def rel(a, b):
    b = b.double(); a = a.cpu().double()
    return float((a - b).abs().max() / b.abs().max()), float((a - b).abs().mean() / b.abs().mean())

with torch.no_grad():
    x32, x64, xg = vol, vol.double(), vol.cuda()
    stages = [("dres0", lambda x, s: O.conv_relu_conv(x, s, "dres0", True), lambda x: p.dres0(x)),
              ("dres1", lambda x, s: O.conv_relu_conv(x, s, "dres1", False) + x, lambda x: p.dres1(x, residual_self=True)),
              ("dres2", lambda x, s: O.hourglass(x, s, "dres2"), lambda x: p.dres2(x)),
              ("dres3", lambda x, s: O.hourglass(x, s, "dres3"), lambda x: p.dres3(x)),
              ("classif2", lambda x, s: O.conv_relu_conv(x, s, "classif2", False, bn_last=False), lambda x: p.classif2(x))]
    for name, fo, fg in stages:
        x32 = fo(x32, sd); x64 = fo(x64, sd64); xg = fg(xg)
        print(f"{name:9s} scale {float(x64.abs().max()):9.3f}  hip-vs-f64 max/mean rel {rel(xg, x64)}  cpu32-vs-f64 {rel(x32, x64)}  hip-vs-cpu32 {rel(xg, x32)}")
    d64, p64 = O.upsample_softmax_regress(x64, 192)
    d32, p32 = O.upsample_softmax_regress(x32, 192)
    dg, ug = dv.upsample_softmax_regress(xg)
    unc = O.disparity_uncertainty(d64, p64)
    for nm, d in (("hip", dg.cpu().double()), ("cpu32", d32.double())):
        e = (d - d64).abs()
        print(f"disp {nm}-vs-f64: mean {float(e.mean()):.3e} max {float(e.max()):.3e} frac>1e-3 {float((e>1e-3).double().mean()):.4f}  "
              f"mean(e/unc) {float((e/unc).mean()):.3e}; unc mean {float(unc.mean()):.2f}")
