"""GPU diagnostic: DDIM-loop error of HIP and of the fp32 oracle against a float64 oracle run."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
sys.path.insert(0, str(Path(__file__).resolve().parents[2] / "tests"))
import torch
import diffuvolume_amd as dv
from conftest import load_golden
from diffuvolume_amd.synth import NoiseTape, _gen, synth_state_dict
from oracle import acv_oracle as O

sd = synth_state_dict(dv.ACVNet_DDIM(192).state_dict(), seed=1, logit_gain=8.0)
sd64 = {k: (v.double() if v.is_floating_point() and not k.startswith("time_embedding") else v) for k, v in sd.items()}
m = dv.ACVNet_DDIM(192); m.load_state_dict(sd); m = m.cuda().eval()
g = load_golden("ddim_sample")
vol = torch.rand(1, 64, 48, 16, 32, generator=_gen(g["vol_seed"], "vol"))
o32, o64 = O.ACVDiffusionOracle(sd), O.ACVDiffusionOracle(sd64)
f32, s32 = o32.ddim_sample(vol, g["used"], g["x_T"], NoiseTape(g["tape_seed"]))
f64, s64 = o64.ddim_sample(vol.double(), g["used"].double(), g["x_T"], NoiseTape(g["tape_seed"]))
fh, sh = m.ddim_sample(vol.cuda(), g["used"].cuda(), g["x_T"].cuda(), noise=NoiseTape(g["tape_seed"]))
fh, sh = fh.cpu(), sh.cpu()
def stats(a, b):
    e = (a.double() - b.double()).abs()
    return f"mean {float(e.mean()):.2e} med {float(e.median()):.2e} p99 {float(e.flatten().quantile(0.99)):.2e} max {float(e.max()):.2e} f>1e-3 {float((e>1e-3).double().mean()):.4f}"
for i in range(1, 6):
    print(f"step {i}: hip-gold  {stats(sh[i], g['stack'][i])}")
    print(f"        o32-gold  {stats(s32[i], g['stack'][i])}")
    print(f"        hip-f64   {stats(sh[i], s64[i])}")
    print(f"        gold-f64  {stats(g['stack'][i], s64[i])}")
print("final hip-gold", stats(fh, g["final"]), " gold-f64", stats(g["final"], f64), " hip-f64", stats(fh, f64))
