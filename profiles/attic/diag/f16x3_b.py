import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import torch
from diffuvolume_amd import submodule as S
from diffuvolume_amd.synth import _gen
cin, cout = 8, 32
g = _gen(18, "c")
w = torch.zeros(cout, cin, 3, 3, 3); w[:, :, 1, 1, 1] = torch.eye(cout)[:, :cin]
plan = S.Conv3dPlan(w.cuda(), None, 1, S.ACT_NONE, precision="f16x3")
for dims in [(1, 2, 4, 48), (1, 6, 8, 48)]:
    tot = 0
    for trial in range(20):
        x = (torch.rand(dims[0], cin, *dims[1:], generator=g) + 1.0) * 3        # |x| in [3,6): no tiny values
        sc = torch.rand(dims[0], *dims[1:], generator=g) * 0.8 + 0.2
        fused = plan(x.cuda(), in_scale=sc.cuda()).cpu()[:, :cin]
        want = x * sc.unsqueeze(1)
        d = (fused - want).abs() / want.abs()
        idx = torch.nonzero(d > 2e-6)
        tot += len(idx)
        for i in idx[:3].tolist():
            b, c, z, y, xx = i
            print(f"dims {dims} trial {trial} pos {i}: x={float(x[b,c,z,y,xx])!r} s={float(sc[b,z,y,xx])!r} want={float(want[b,c,z,y,xx])!r} got={float(fused[b,c,z,y,xx])!r} ratio-1={float(fused[b,c,z,y,xx]/want[b,c,z,y,xx])-1:.2e}")
    print(dims, "bad", tot)
