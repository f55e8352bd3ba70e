"""Time the 3x3x3 stride-1 layers of the bench step on the F(2x2x2,3x3x3) kernel and on the in-plane F(2x2,3x3) kernel (same
process, alternating):  python tools/ab_wino3.py"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from diffuvolume_amd import submodule as S
dev = "cuda:0"
S.Conv3dPlan.WINO3_MIN_CIN = 1      # every layer on the kernel under test


def timeit(run, n=20):
    run(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        run()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for name, c, dims in (("32->32 @48x128x240", 32, (48, 128, 240)), ("64->64 @24x64x120", 64, (24, 64, 120)),
                      ("128->128 @12x32x60", 128, (12, 32, 60))):
    x = torch.randn(8, c, *dims, device=dev)
    w = torch.randn(c, c, 3, 3, 3, device=dev) * 0.05
    bn = tuple(torch.rand(c, device=dev) + 0.5 for _ in range(4))
    plan = S.Conv3dPlan(w, bn, act=S.ACT_RELU, precision="f32")
    res = {}
    for rep in range(3):
        for flag in (True, False):
            S.Conv3dPlan.WINO3 = flag
            res.setdefault(flag, []).append(timeit(lambda: plan(x)))
    S.Conv3dPlan.WINO3 = True
    y3 = plan(x)
    S.Conv3dPlan.WINO3 = False
    y2 = plan(x)
    S.Conv3dPlan.WINO3 = True
    d = float((y3 - y2).abs().max() / y2.abs().max())
    print(f"{name}:  F(2x2x2) " + " / ".join(f"{t:.3f}" for t in res[True]) + "   F(2x2)+depth " +
          " / ".join(f"{t:.3f}" for t in res[False]) + f"  ms    max |diff| / scale {d:.2e}", flush=True)
    del x, plan
