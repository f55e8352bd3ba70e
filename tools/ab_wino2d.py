"""A/B of the 2-D Winograd kernel against the direct 2-D implicit GEMM: max error vs an fp64 torch reference, and time at
the IGEV update-block / feature-CNN sizes.  python tools/ab_wino2d.py [--time]"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
import torch.nn.functional as F
from diffuvolume_amd import submodule as S

dev = "cuda:0"
torch.manual_seed(0)


def plans(w, bias, act):
    S.Conv2dPlan.WINO_MIN_BLOCKS = 0
    pw = S.Conv2dPlan(w, None, act=act, bias=bias)
    pd = S.Conv2dPlan(w, None, act=act, bias=bias)
    pd.wino_packed = None
    return pw, pd


def check(b, cins, cout, h, w_, act=S.ACT_TANH, gates=True):
    xs = [torch.randn(b, c, h, w_, device=dev) for c in cins]
    cin = sum(cins)
    w = torch.randn(cout, cin, 3, 3, device=dev) * (2.0 / (9 * cin)) ** 0.5
    bias = torch.randn(cout, device=dev) * 0.1
    res = torch.randn(b, cout, h, w_, device=dev)
    z, hh = torch.rand(b, cout, h, w_, device=dev), torch.randn(b, cout, h, w_, device=dev)
    pw, pd = plans(w, bias, act)
    kw = dict(residual=res, blend=(z, hh)) if gates else dict(residual=res)
    yw, yd = pw(xs, **kw), pd(xs, **kw)
    ref = F.conv2d(torch.cat(xs, 1).double(), w.double(), bias.double(), 1, 1) + res.double()
    ref = {S.ACT_TANH: torch.tanh, S.ACT_SIGMOID: torch.sigmoid, S.ACT_RELU: torch.relu, S.ACT_NONE: lambda t: t}[act](ref)
    if gates:
        ref = hh.double() + z.double() * (ref - hh.double())
    ew, ed = (yw.double() - ref).abs().max().item(), (yd.double() - ref).abs().max().item()
    print(f"B{b} {cins}->{cout} {h}x{w_} act={act} gates={gates}: max err wino {ew:.2e} direct {ed:.2e}", flush=True)


def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


check(1, [8], 32, 16, 16, S.ACT_NONE, False)
check(2, [5, 7], 20, 9, 21)
check(1, [128, 128, 64, 64], 128, 24, 40, S.ACT_SIGMOID)
check(1, [3], 40, 5, 3, S.ACT_RELU, False)
check(1, [32], 32, 33, 50, S.ACT_RELU, False)
if "--time" in sys.argv:
    for name, b, cin, cout, h, w_ in (("gru04 B4", 4, 384, 128, 96, 312), ("gru04 B1", 1, 384, 128, 96, 312),
                                      ("gru08 B4", 4, 384, 128, 48, 156), ("gru08 B1", 1, 384, 128, 48, 156),
                                      ("gru16 B4", 4, 256, 128, 24, 78), ("gru16 B1", 1, 256, 128, 24, 78),
                                      ("feat32 B8", 8, 32, 32, 256, 480), ("feat128 B8", 8, 128, 128, 128, 240)):
        x = torch.randn(b, cin, h, w_, device=dev)
        w = torch.randn(cout, cin, 3, 3, device=dev) * 0.05
        pw, pd = plans(w, torch.zeros(cout, device=dev), S.ACT_RELU)
        fl = 2.0 * b * cout * h * w_ * cin * 9
        tw, td = timeit(lambda: pw(x)), timeit(lambda: pd(x))
        tt = timeit(lambda: F.relu(F.conv2d(x, w, None, 1, 1)))
        print(f"{name:11s} wino {tw:6.3f} ms ({fl / tw / 1e9:6.1f} TF)  direct {td:6.3f} ms ({fl / td / 1e9:6.1f} TF)  torch {tt:6.3f} ms", flush=True)
