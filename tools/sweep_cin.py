"""Fit time(Cin) = fixed + per-chunk for the conv / deconv kernels: python tools/sweep_cin.py"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from diffuvolume_amd import submodule as S

dev = "cuda:0"


def t_ms(fn, iters=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


B = 8
ONLY = sys.argv[1] if len(sys.argv) > 1 else None
CINS = [int(c) for c in sys.argv[2:]]
for kind, dims, cins in (("deconv", (24, 64, 128), (8, 16, 32, 64, 128, 256)), ("conv", (48, 128, 240), (4, 8, 16, 32, 64, 128))):
    if ONLY and kind != ONLY:
        continue
    cins = CINS or cins
    for cin in cins:
        x = torch.randn(B, cin, *dims, device=dev)
        bn = tuple(torch.rand(32, device=dev) + 0.5 for _ in range(4))
        if kind == "conv":
            plan = S.Conv3dPlan(torch.randn(32, cin, 3, 3, 3, device=dev) * 0.05, bn, stride=1, act=S.ACT_RELU)
        else:
            plan = S.Deconv3dPlan(torch.randn(cin, 32, 3, 3, 3, device=dev) * 0.05, bn, act=S.ACT_RELU)
        ms = t_ms(lambda: plan(x))
        fl = 2.0 * B * 32 * cin * 27 * dims[0] * dims[1] * dims[2]
        print(f"{kind:6s} cin={cin:4d} {ms:8.3f} ms {fl / ms / 1e9:7.1f} TF", flush=True)
        del x, plan
