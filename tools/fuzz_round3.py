"""Random-shape checks of the kernels added in round 3 against float64 PyTorch statements on the GPU box:
the rank-1 first aggregation layer (csrc/rank1_filter.hip), the ConvGRU gate pair (dv_conv2d_wino_cat_pair_f32), the
k4 s2 transposed convolution plan.   python tools/fuzz_round3.py [n_cases]"""
import random
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
import torch.nn.functional as F
import diffuvolume_amd as dv
from diffuvolume_amd import submodule as S

dev = "cuda:0"
random.seed(4321)
torch.manual_seed(4321)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
bad = 0


def rel(a, b):
    return float((a.double().cpu() - b.double().cpu()).abs().max() / b.double().abs().max().clamp(min=1e-20))


def bn(c):
    return tuple(t.to(dev) for t in (torch.rand(c) + 0.5, torch.randn(c) * 0.1, torch.randn(c) * 0.1, torch.rand(c) + 0.5))


for i in range(n):
    # ---- rank-1 first layer vs conv3d(volume * noise) in float64
    b, c, cout = random.choice([1, 2]), random.choice([4, 8, 32]), random.choice([2, 5, 16, 32])
    d, h, w = random.choice([3, 7, 12, 24, 48]), random.randint(1, 9), random.choice([1, 5, 33, 120, 240, 253, 257, 300, 420])
    print("R1", i, b, c, cout, d, h, w, flush=True)
    L, R = torch.randn(b, c, h, w, device=dev), torch.randn(b, c, h, w, device=dev)
    att = torch.randn(b, 1, d, h, w, device=dev) * 2
    noise = torch.rand(b, d, h, w, device=dev)
    wt = torch.randn(cout, 2 * c, 3, 3, 3, device=dev) * 0.1
    bnp = bn(cout)
    vol = dv.build_concat_attention_volume(L, R, att, d)
    ref = F.conv3d(vol.double() * noise.double().unsqueeze(1), wt.double(), None, 1, 1)
    ref = torch.relu(F.batch_norm(ref, bnp[2].double(), bnp[3].double(), bnp[0].double(), bnp[1].double(), False, 0.0, 1e-5))
    out = S.Rank1FilterPlan(wt, bnp, act=S.ACT_RELU)(vol, noise)
    e = rel(out, ref)
    if not e < 2e-5:
        bad += 1
        print("  BAD rank1", e)
    # ---- gate pair vs two convolutions
    S.Conv2dPlan.WINO_MIN_BLOCKS = 0 if i % 2 else 128
    hid, b2, h2, w2 = random.choice([32, 64, 128]), random.choice([1, 2, 4]), random.randint(1, 50), random.randint(1, 90)
    chans = random.choice([(hid, hid), (hid, 127, 1, hid), (hid, 40)])
    print("GP", i, hid, chans, b2, h2, w2, flush=True)
    parts = [torch.randn(b2, ch, h2, w2, device=dev) for ch in chans]
    cin = sum(chans)
    w1, w2_ = torch.randn(hid, cin, 3, 3, device=dev) * 0.05, torch.randn(hid, cin, 3, 3, device=dev) * 0.05
    b1, b2_ = torch.randn(hid, device=dev) * 0.1, torch.randn(hid, device=dev) * 0.1
    cz, cr = torch.randn(b2, hid, h2, w2, device=dev), torch.randn(b2, hid, h2, w2, device=dev)
    z, rh = S.Conv2dPairPlan((w1, b1), (w2_, b2_), S.ACT_SIGMOID)(parts, residual=(cz, cr), mul=(None, parts[0]))
    x64 = torch.cat(parts, 1).double()
    zr = torch.sigmoid(F.conv2d(x64, w1.double(), b1.double(), padding=1) + cz.double())
    rr = torch.sigmoid(F.conv2d(x64, w2_.double(), b2_.double(), padding=1) + cr.double()) * parts[0].double()
    e = max(rel(z, zr), rel(rh, rr))
    if not e < 2e-5:
        bad += 1
        print("  BAD pair", e)
    # ---- transposed convolution k4 s2 p1
    cin, co, b3, h3, w3 = random.choice([3, 32, 64]), random.choice([1, 9, 32]), random.choice([1, 2]), random.randint(1, 30), random.randint(1, 70)
    print("DC", i, cin, co, b3, h3, w3, flush=True)
    x = torch.randn(b3, cin, h3, w3, device=dev)
    wt = torch.randn(cin, co, 4, 4, device=dev) * 0.1
    bias = torch.randn(co, device=dev) * 0.1
    ref = F.conv_transpose2d(x.double(), wt.double(), bias.double(), 2, 1)
    e = rel(S.Deconv2dK4S2Plan(wt, None, bias=bias)(x), ref)
    if not e < 2e-5:
        bad += 1
        print("  BAD deconv2d", e)
print("bad cases:", bad)
sys.exit(1 if bad else 0)
