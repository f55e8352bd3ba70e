"""The whole hot-path step of the bench (batch 8, 960x512, 5 DDIM steps) with the F(2x2x2,3x3x3) kernel routed from 32 / 64 input
channels on or not at all, alternating in one process:  python tools/ab_step_wino3.py"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
import bench as B
import diffuvolume_amd as dv
from diffuvolume_amd import submodule as S
from diffuvolume_amd.synth import synth_state_dict

dev = torch.device("cuda", 0)
model = dv.ACVNet_DDIM(192, False, False, sampling_timesteps=5)
model.load_state_dict(synth_state_dict(model.state_dict(), seed=1, logit_gain=8.0), strict=True)
model = model.to(dev).eval()
model.prepare()
host, x = B.make_inputs(8, 128, 240, seed=100, device=dev)


def timeit(n=3):
    B.hot_path(model, x); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        B.hot_path(model, x)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


res = {}
for rep in range(3):
    for name, on, mc in (("in-plane everywhere", False, 64), ("F(2x2x2) from 64 channels", True, 64), ("F(2x2x2) from 32 channels", True, 32)):
        S.Conv3dPlan.WINO3, S.Conv3dPlan.WINO3_MIN_CIN = on, mc
        res.setdefault(name, []).append(timeit())
S.Conv3dPlan.WINO3, S.Conv3dPlan.WINO3_MIN_CIN = True, 64
for name, ts in res.items():
    print(f"{name:28s} " + " / ".join(f"{t:.2f}" for t in ts) + " ms per step", flush=True)
