"""Random-shape checks of the kernels added / rewritten in the second half of round 5 against float64 PyTorch statements on the
GPU box: the 2-D Winograd kernel's source modes (one tensor / chunk-aligned / arbitrary splits), gate epilogues, gate pair and
K-split forms (csrc/conv2d_wino.hip), the fused geometry lookup + 1x1 convolution (csrc/geo_lookup.hip), `interp`
(csrc/update_glue.hip).   python tools/fuzz_round5.py [n_cases]"""
import random
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
import torch.nn.functional as F
from diffuvolume_amd import submodule as S
from diffuvolume_amd.geometry_ddim import Combined_Geo_Encoding_Volume, pack_lookup_conv1x1
from diffuvolume_amd.update import interp

dev = "cuda:0"
random.seed(515)
torch.manual_seed(515)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
bad = 0
ACTS = {S.ACT_NONE: lambda t: t, S.ACT_RELU: torch.relu, S.ACT_SIGMOID: torch.sigmoid, S.ACT_TANH: torch.tanh}


def rel(a, b):
    return float((a.double().cpu() - b.double().cpu()).abs().max() / b.double().abs().max().clamp(min=1e-20))


def check(tag, err, bar, *what):
    global bad
    if not err < bar:
        bad += 1
        print("BAD", tag, err, *what, flush=True)


for i in range(n):
    # ---- 2-D Winograd: sources, epilogues, K-split (small planes with many channels) and plain launches
    b = random.choice([1, 2, 3])
    nsrc = random.choice([1, 1, 2, 3, 4])
    aligned = random.random() < 0.5
    split = [8 * random.randint(1, 20) if aligned else random.choice([1, 3, 7, 8, 16, 33, 64, 127, 128]) for _ in range(nsrc)]
    cin = sum(split)
    cout = random.choice([1, 16, 32, 33, 64, 96, 128])
    h, w = random.randint(1, 50), random.choice([3, 4, 8, 15, 16, 17, 20, 39, 40, 78, 80])
    d = random.choice([1, 1, 1, 2, 4])
    act = random.choice(list(ACTS))
    print("W2D", i, b, split, cout, h, w, "dil", d, "act", act, flush=True)
    parts = [torch.randn(b, c, h, w, device=dev) for c in split]
    wt = torch.randn(cout, cin, 3, 3, device=dev) * (2.0 / (9 * cin)) ** 0.5
    bias = torch.randn(cout, device=dev) * 0.1
    res = torch.randn(b, cout, h, w, device=dev) if random.random() < 0.6 else None
    gate = random.choice(["none", "mul", "blend"])
    hh, zz = torch.randn(b, cout, h, w, device=dev), torch.rand(b, cout, h, w, device=dev)
    S.Conv2dPlan.WINO_MIN_BLOCKS = 0                          # the Winograd kernel whatever the size
    plan = S.Conv2dPlan(wt, None, dilation=d, act=act, bias=bias)
    kw = dict(residual=res)
    if gate == "mul":
        kw["mul"] = hh
    if gate == "blend":
        kw["blend"] = (zz, hh)
    out = plan(parts if nsrc > 1 else parts[0], **kw)
    ref = F.conv2d(torch.cat(parts, 1).double(), wt.double(), bias.double(), 1, d, d)
    if res is not None:
        ref = ref + res.double()
    ref = ACTS[act](ref)
    if gate == "mul":
        ref = ref * hh.double()
    if gate == "blend":
        ref = hh.double() + zz.double() * (ref - hh.double())
    check("w2d", rel(out, ref), 2e-5, b, split, cout, h, w, d, act, gate)
    if b > 1:                                                  # a shard reproduces the batch's bits
        kw1 = {k: (tuple(t[:1].contiguous() for t in v) if isinstance(v, tuple) else (None if v is None else v[:1].contiguous()))
               for k, v in kw.items()}
        one = plan([t[:1].contiguous() for t in parts] if nsrc > 1 else parts[0][:1].contiguous(), **kw1)
        if not torch.equal(one, out[:1]):
            bad += 1
            print("BAD shard", b, split, cout, h, w, d, flush=True)
    # ---- the ConvGRU gate pair
    if cout % 32 == 0 and d == 1:
        w2 = torch.randn(cout, cin, 3, 3, device=dev) * (2.0 / (9 * cin)) ** 0.5
        b2 = torch.randn(cout, device=dev) * 0.1
        pair = S.Conv2dPairPlan((wt, bias), (w2, b2), S.ACT_SIGMOID)
        r2 = torch.randn(b, cout, h, w, device=dev)
        z, rh = pair(parts, residual=(res, r2), mul=(None, hh))
        x64 = torch.cat(parts, 1).double()
        zr = torch.sigmoid(F.conv2d(x64, wt.double(), bias.double(), 1, 1) + (0 if res is None else res.double()))
        rr = torch.sigmoid(F.conv2d(x64, w2.double(), b2.double(), 1, 1) + r2.double()) * hh.double()
        check("pair z", rel(z, zr), 2e-5, b, split, cout, h, w)
        check("pair r", rel(rh, rr), 2e-5, b, split, cout, h, w)
    # ---- fused geometry lookup + 1x1 convolution
    gb, gh, gw_, gd = random.choice([1, 2]), random.randint(1, 14), random.choice([5, 16, 31, 64, 78]), random.choice([8, 13, 24, 48])
    print("GEO", i, gb, gh, gw_, gd, flush=True)
    geo = torch.randn(gb, 8, gd, gh, gw_, device=dev)
    f1, f2 = torch.randn(gb, 16, gh, gw_, device=dev), torch.randn(gb, 16, gh, gw_, device=dev)
    disp = torch.rand(gb, 1, gh, gw_, device=dev) * (gd + 12) - 6
    if random.random() < 0.5:
        disp = disp.mean() + 0.3 * torch.randn_like(disp)       # smooth field: narrow plane walk
    coords = torch.arange(gw_, dtype=torch.float32, device=dev).view(1, 1, 1, gw_).expand(gb, 1, gh, gw_).contiguous()
    noisy = torch.rand(gb, gd, gh, gw_, device=dev)
    wc, bc = torch.randn(64, 162, 1, 1, device=dev) * 0.1, torch.randn(64, device=dev) * 0.1
    fn = Combined_Geo_Encoding_Volume(f1, f2, geo)
    look = fn(disp, coords, noisy)
    fused = fn.lookup_conv1x1(disp, coords, noisy, pack_lookup_conv1x1(wc, 8), bc, S.ACT_RELU)
    check("geo", rel(fused, torch.relu(F.conv2d(look.double(), wc.double(), bc.double()))), 2e-5, gb, gh, gw_, gd)
    # ---- interp (both store widths)
    ib, ic, ih, iw = random.choice([1, 2]), random.choice([1, 5, 32]), random.randint(1, 20), random.randint(1, 40)
    oh, ow = random.randint(1, 45), random.choice([1, 3, 4, 8, 20, 78, 80])
    xi = torch.randn(ib, ic, ih, iw, device=dev)
    o = interp(xi, torch.empty(ib, 1, oh, ow, device=dev))
    # (the source coordinate is an fp32 product, as in PyTorch's fp32 kernel: a few 1e-6 of the value range against float64)
    check("interp", float((o.double() - F.interpolate(xi.double(), (oh, ow), mode="bilinear", align_corners=True)).abs().max()), 5e-5,
          ib, ic, ih, iw, oh, ow)
print("done", n, "cases,", bad, "bad")
sys.exit(1 if bad else 0)
