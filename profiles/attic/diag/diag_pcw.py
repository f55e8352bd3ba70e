import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2])); sys.path.insert(0, str(Path(__file__).resolve().parents[2] / "tests"))
import torch, warnings
warnings.filterwarnings("ignore")
from conftest import load_golden
from diffuvolume_amd.pwcnet_ddim import PWCNet_ddim
from diffuvolume_amd.synth import NoiseTape, synth_state_dict, synth_stereo_batch
from oracle import pcw_oracle as P
sd = synth_state_dict(PWCNet_ddim(192, True).state_dict(), seed=2, logit_gain=8.0, scale={"refinenet3.conv8.weight": 0.002})
m = PWCNet_ddim(192, True); m.load_state_dict(sd); mc = PWCNet_ddim(192, True); mc.load_state_dict(sd); mc.eval()
m = m.cuda().eval()
g = load_golden("pcw_forward_eval")
batch = synth_stereo_batch(1, 64, 128, seed=g["stereo_seed"], shifts=(8,))
def rel(a, b): return float((a.cpu().double()-b.double()).abs().max() / b.double().abs().max())
with torch.no_grad():
    flc, frc = mc.feature_extraction(batch["left"]), mc.feature_extraction(batch["right"])
    flg, frg = m.feature_extraction(batch["left"].cuda()), m.feature_extraction(batch["right"].cuda())
    for k in flc: print("feat", k, tuple(flc[k].shape), rel(flg[k], flc[k]), float(flc[k].abs().max()))
    comb_o = P.fused_volume(flc, frc, sd)
    comb_h = m.fused_volume({k: v.cuda() for k, v in flc.items()}, {k: v.cuda() for k, v in frc.items()})
    print("fused volume", tuple(comb_o.shape), rel(comb_h, comb_o), float(comb_o.abs().max()))
    print("golden pred range", float(g["pred"].min()), float(g["pred"].max()))
