"""Random-shape checks of the kernels added / rewritten in round 4 against float64 PyTorch statements on the GPU box:
the table build (csrc/pointwise_expand.hip), the 16-byte patch stencils (csrc/patch_volume.hip), the attention-concat
volume from probabilities and its lazy handle (csrc/concat_volume.hip).   python tools/fuzz_round4.py [n_cases]"""
import random
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
import torch.nn.functional as F
import diffuvolume_amd as dv
from diffuvolume_amd import _lib, submodule as S

dev = "cuda:0"
random.seed(97)
torch.manual_seed(97)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
bad = 0


def rel(a, b):
    return float((a.double().cpu() - b.double().cpu()).abs().max() / b.double().abs().max().clamp(min=1e-20))


for i in range(n):
    # ---- table build: 1x1 convolution, Cin <= 32
    b, cin, cout = random.choice([1, 2, 3]), random.choice([1, 3, 4, 8, 13, 32]), random.choice([1, 15, 16, 17, 27, 162, 864])
    h, w = random.randint(1, 40), random.choice([1, 2, 3, 4, 7, 16, 63, 64, 65, 240])
    print("PW", i, b, cin, cout, h, w, flush=True)
    x = torch.randn(b, cin, h, w, device=dev)
    wt = torch.randn(cout, cin, 1, 1, device=dev)
    e = rel(S.PointwiseExpandPlan(wt)(x), F.conv2d(x.double(), wt.double()))
    if not e < 5e-6:
        bad += 1
        print("  BAD pointwise", e)
    # ---- patch stencils: random dilation runs, widths on both paths
    g = random.choice([3, 8, 40])
    bb, d, hh, ww = random.choice([1, 2]), random.randint(1, 4), random.randint(1, 40), random.choice([4, 8, 12, 50, 128, 132, 240, 260, 7])
    dil = []
    while len(dil) < g:
        dil += [random.choice([1, 2, 3])] * random.randint(1, g)
    dil = torch.tensor(dil[:g], dtype=torch.int32)
    print("PV", i, bb, g, d, hh, ww, dil.tolist()[:8], flush=True)
    xv = torch.randn(bb, g, d, hh, ww, device=dev)
    w1, w2 = torch.randn(g, 9, device=dev) * 0.4, torch.randn(g, 9, device=dev) * 0.4
    y = F.conv3d(xv.double(), w1.double().view(g, 1, 1, 3, 3), None, 1, (0, 1, 1), 1, g)
    ref = torch.cat([F.conv3d(y[:, c:c + 1], w2[c].double().view(1, 1, 1, 3, 3), None, 1, (0, int(dil[c]), int(dil[c])), int(dil[c]))
                     for c in range(g)], dim=1)
    e = rel(S.patch_volume(xv, w1, w2, dil.to(dev)), ref)
    if not e < 5e-6:
        bad += 1
        print("  BAD patch", e)
    # ---- attention-concat volume: tensor, handle, handle.tensor()
    c, dd = random.choice([1, 5, 8, 32]), random.choice([1, 5, 12, 48])
    hb, wb = random.randint(1, 9), random.choice([4, 8, 36, 240, 244, 37])
    print("AC", i, c, dd, hb, wb, flush=True)
    L, R = torch.randn(1, c, hb, wb, device=dev), torch.randn(1, c, hb, wb, device=dev)
    att = torch.randn(1, 1, dd, hb, wb, device=dev) * 3
    p = torch.softmax(att.double(), dim=2)
    shifted = torch.stack([F.pad(R.double(), (k, 0))[..., :wb] for k in range(dd)], dim=2)
    ref = p * torch.cat((L.double().unsqueeze(2).expand(1, c, dd, hb, wb), shifted), dim=1)
    vol = dv.build_concat_attention_volume(L, R, att, dd)
    lazy = dv.build_concat_attention_volume(L, R, att, dd, lazy=True)
    e = max(rel(vol, ref), rel(lazy.tensor(), ref))
    if not (e < 2e-6 and torch.equal(vol, lazy.tensor())):
        bad += 1
        print("  BAD concat", e, bool(torch.equal(vol, lazy.tensor())))
    # ---- stride-2 convolution: both tilings, the default choice, with and without the filter prologue, a residual
    cin, cout = random.choice([1, 4, 8, 12, 32, 64]), random.choice([16, 24, 32, 64, 72, 128])
    bs, ds, hs, ws = random.choice([1, 2, 3]), random.randint(1, 11), random.randint(1, 20), random.choice([1, 2, 7, 31, 32, 33, 63, 64, 65, 120])
    print("S2", i, bs, cin, cout, ds, hs, ws, flush=True)
    xs2 = torch.randn(bs, cin, ds, hs, ws, device=dev)
    ws2 = torch.randn(cout, cin, 3, 3, 3, device=dev) * (2.0 / (27 * cin)) ** 0.5
    bn = tuple(t.to(dev) for t in (torch.rand(cout) + 0.5, torch.randn(cout) * 0.1, torch.randn(cout) * 0.1, torch.rand(cout) + 0.5))
    flt = torch.rand(bs, ds, hs, ws, device=dev) if random.random() < 0.5 else None
    xin = xs2.double() if flt is None else xs2.double() * flt.double().unsqueeze(1)
    ref = F.batch_norm(F.conv3d(xin, ws2.double(), None, 2, 1), bn[2].double(), bn[3].double(), bn[0].double(), bn[1].double(), False, 0.0, 1e-5)
    res = torch.randn(ref.shape, device=dev) if random.random() < 0.5 else None
    ref = torch.relu(ref if res is None else ref + res.double())
    plan = S.Conv3dPlan(ws2, bn, stride=2, act=S.ACT_RELU)
    outs = []
    for pin in (1, 2, 0):                      # 2 x 4 x 32 tiles, 2 x 2 x 32 tiles, the launcher's own choice
        _lib.load().dv_conv3d_set_s2_tile(pin)
        outs.append(plan(xs2, in_scale=flt, residual=res).clone())
    e = rel(outs[0], ref)
    if not (e < 2e-5 and torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])):
        bad += 1
        print("  BAD stride2", e, bool(torch.equal(outs[0], outs[1])), bool(torch.equal(outs[0], outs[2])))
print("failures:", bad)
sys.exit(1 if bad else 0)
