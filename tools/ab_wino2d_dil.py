"""Dilated 3x3 layers of the KITTI12 refinement stack (384 x 1248, batch 4): 2-D Winograd on the dilation^2 sub-images
vs the direct implicit GEMM, error vs fp64 and time.  python tools/ab_wino2d_dil.py"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
import torch.nn.functional as F
from diffuvolume_amd import submodule as S

dev = "cuda:0"
torch.manual_seed(0)
S.Conv2dPlan.WINO_MAX_DILATION = 16
S.Conv2dPlan.WINO_MIN_BLOCKS = 0


def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for cin, cout, d in ((128, 128, 1), (128, 128, 2), (128, 128, 4), (128, 96, 8), (96, 96, 8), (96, 64, 16), (64, 64, 16)):
    b, h, w_ = 4, 384, 1248
    x = torch.randn(b, cin, h, w_, device=dev)
    w = torch.randn(cout, cin, 3, 3, device=dev) * (2.0 / (9 * cin)) ** 0.5
    pw = S.Conv2dPlan(w, None, dilation=d, act=S.ACT_RELU)
    pd = S.Conv2dPlan(w, None, dilation=d, act=S.ACT_RELU)
    pd.wino_packed = None
    yw, yd = pw(x), pd(x)
    ref = F.relu(F.conv2d(x[:1].double(), w.double(), None, 1, d, d))
    ew, ed = (yw[:1].double() - ref).abs().max().item(), (yd[:1].double() - ref).abs().max().item()
    fl = 2.0 * b * cout * h * w_ * cin * 9
    tw, td = timeit(lambda: pw(x)), timeit(lambda: pd(x))
    print(f"{cin:3d}->{cout:3d} d={d:2d}: wino {tw:6.3f} ms ({fl / tw / 1e9:6.1f} TF, err {ew:.1e})  direct {td:6.3f} ms ({fl / td / 1e9:6.1f} TF, err {ed:.1e})", flush=True)
