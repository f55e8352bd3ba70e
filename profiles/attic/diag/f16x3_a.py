import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import torch
from diffuvolume_amd import submodule as S
from diffuvolume_amd.synth import _gen
cfg = (32, 32, (1, 6, 8, 48))
cin, cout, dims = cfg
g = _gen(17, str(cfg))
x = torch.randn(dims[0], cin, *dims[1:], generator=g) * 3
w = torch.randn(cout, cin, 3, 3, 3, generator=g) * (2.0 / (27 * cin)) ** 0.5
bn = (torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g) * 0.1, torch.randn(cout, generator=g) * 0.1, torch.rand(cout, generator=g) + 0.5)
scale = torch.rand(dims[0], *dims[1:], generator=g)
res = torch.randn(dims[0], cout, *dims[1:], generator=g)
def run(use_scale, use_bn, use_res, act):
    xs = x.double() * (scale.unsqueeze(1).double() if use_scale else 1.0)
    y = torch.nn.functional.conv3d(xs, w.double(), None, 1, 1)
    if use_bn: y = torch.nn.functional.batch_norm(y, bn[2].double(), bn[3].double(), bn[0].double(), bn[1].double(), False, 0.0, 1e-5)
    if use_res: y = y + res.double()
    if act: y = torch.relu(y)
    outs = []
    for prec in ("f16x3", "f32"):
        plan = S.Conv3dPlan(w.cuda(), tuple(t.cuda() for t in bn) if use_bn else None, 1, S.ACT_RELU if act else S.ACT_NONE, precision=prec)
        o = plan(x.cuda(), in_scale=scale.cuda() if use_scale else None, residual=res.cuda() if use_res else None).cpu().double()
        e = (o - y).abs()
        outs.append((float(e.max() / y.abs().max()), float(e.mean())))
    print(f"scale={use_scale} bn={use_bn} res={use_res} act={act}: f16x3 {outs[0]}  f32 {outs[1]}")
for flags in [(0,0,0,0),(1,0,0,0),(0,1,0,0),(0,0,1,0),(0,0,0,1),(1,1,1,1)]:
    run(*flags)
