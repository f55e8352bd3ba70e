"""Breakdown of the end-to-end `test_sample` equivalent (origin ACVNet -> ACVNet_DDIM.forward) at the bench size:
HIP kernel families (KernelTimer) vs everything else (PyTorch 2-D CNNs, glue).  python tools/bench_e2e.py"""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
import diffuvolume_amd as dv
from diffuvolume_amd.profiling import KernelTimer
from diffuvolume_amd.synth import _gen, synth_state_dict

dev = "cuda:0"
B, H, W = 8, 512, 960
g = _gen(7, "e2e")
left = torch.randn(B, 3, H, W, generator=g).to(dev)
right = torch.roll(left, -8, dims=-1)
origin = dv.ACVNet(192, False, False)
origin.load_state_dict(synth_state_dict(origin.state_dict(), seed=3, logit_gain=8.0), strict=True)
origin = origin.to(dev).eval()
ddim = dv.ACVNet_DDIM(192, False, False)
ddim.load_state_dict(synth_state_dict(ddim.state_dict(), seed=0, logit_gain=8.0), strict=True)
ddim = ddim.to(dev).eval()


def wall(fn, n=2):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def hip_ms(fn):
    kt = KernelTimer(); KernelTimer.active = kt
    try:
        fn()
    finally:
        KernelTimer.active = None
    s = kt.summary()
    return sum(v["total_ms"] for v in s.values()), {k: round(v["total_ms"], 2) for k, v in sorted(s.items(), key=lambda kv: -kv[1]["total_ms"])[:8]}


with torch.no_grad():
    used = origin(left, right)[-1]
    dn = torch.nn.functional.interpolate(torch.clamp(used, 0, 191).unsqueeze(1), size=(H // 4, W // 4), mode="bilinear") / 4
    t_o = wall(lambda: origin(left, right))
    t_d = wall(lambda: ddim(left, right, used, dn, None))
    t_f = wall(lambda: (ddim.feature_extraction(left), ddim.feature_extraction(right)))
    h_o, top_o = hip_ms(lambda: origin(left, right))
    h_d, top_d = hip_ms(lambda: ddim(left, right, used, dn, None))
print(f"origin ACVNet.forward  {t_o:8.1f} ms  (HIP kernels {h_o:7.1f} ms)  {top_o}")
print(f"ACVNet_DDIM.forward    {t_d:8.1f} ms  (HIP kernels {h_d:7.1f} ms)  {top_d}")
print(f"feature_extraction x2  {t_f:8.1f} ms")
