"""A/B of the Winograd F(2x2,3x3) kernel against the direct implicit GEMM: max error vs an fp64 torch reference and
time per layer.  python tools/ab_wino.py"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
import torch.nn.functional as F
from diffuvolume_amd import submodule as S

dev = "cuda:0"
torch.manual_seed(0)


def check(b, cin, cout, dims, res=False, scale=False, act=S.ACT_RELU):
    x = torch.randn(b, cin, *dims, device=dev)
    w = torch.randn(cout, cin, 3, 3, 3, device=dev) * 0.05
    bn = tuple(torch.rand(cout, device=dev) + 0.5 for _ in range(4))
    r = torch.randn(b, cout, *dims, device=dev) if res else None
    sc = torch.rand(b, *dims, device=dev) if scale else None
    pw = S.Conv3dPlan(w, bn, act=act, precision="f32")
    pd = S.Conv3dPlan(w, bn, act=act, precision="f32_direct")
    yw, yd = pw(x, in_scale=sc, residual=r), pd(x, in_scale=sc, residual=r)
    xs = x.double() * (sc.double().unsqueeze(1) if scale else 1.0)
    ref = F.conv3d(xs, w.double(), padding=1)
    g, be, m, v = (t.double() for t in bn)
    s_ = g / torch.sqrt(v + 1e-5)
    ref = ref * s_.view(1, -1, 1, 1, 1) + (be - m * s_).view(1, -1, 1, 1, 1)
    if res:
        ref = ref + r.double()
    ref = torch.relu(ref) if act == S.ACT_RELU else ref
    ew, ed = (yw.double() - ref).abs().max().item(), (yd.double() - ref).abs().max().item()
    print(f"B{b} {cin}->{cout} {dims} res={res} scale={scale}: max err wino {ew:.2e} direct {ed:.2e} (|ref| max {ref.abs().max().item():.1f})",
          flush=True)
    return ew


def timeit(plan, x, n=5):
    plan(x); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        plan(x)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


check(1, 8, 32, (8, 8, 32))
check(2, 5, 20, (5, 7, 19), res=True)
check(1, 32, 32, (6, 10, 36), res=True, scale=True)
check(1, 64, 64, (4, 12, 16), act=S.ACT_NONE)
check(1, 3, 40, (3, 3, 3))
if "--time" in sys.argv:
    for name, cin, cout, dims in (("c32", 32, 32, (48, 128, 240)), ("c64in", 64, 32, (48, 128, 240)),
                                  ("c64", 64, 64, (24, 64, 120)), ("c128", 128, 128, (12, 32, 60))):
        x = torch.randn(8, cin, *dims, device=dev)
        w = torch.randn(cout, cin, 3, 3, 3, device=dev) * 0.05
        bn = tuple(torch.rand(cout, device=dev) + 0.5 for _ in range(4))
        fl = 2.0 * x.numel() / cin * cout * cin * 27
        for prec in ("f32", "f32_direct"):
            ms = timeit(S.Conv3dPlan(w, bn, act=S.ACT_RELU, precision=prec), x)
            print(f"{name:6s} {prec:10s} {ms:7.3f} ms  {fl / ms / 1e9:6.1f} TFLOP/s (direct-equivalent)", flush=True)
