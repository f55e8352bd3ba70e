"""Random-shape comparison of the 2-D / 3-D convolution entry points with PyTorch's own ops on the GPU.
python tools/fuzz_kernels.py [n_cases]"""
import random
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
import torch.nn.functional as F
from diffuvolume_amd import submodule as S

dev = "cuda:0"
random.seed(1234)
torch.manual_seed(1234)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
bad = 0


def rel(a, b):
    return float((a.double() - b.double()).abs().max() / b.double().abs().max().clamp(min=1e-20))


def bn(c):
    return (torch.rand(c, device=dev) + 0.5, torch.randn(c, device=dev) * 0.1, torch.randn(c, device=dev) * 0.1, torch.rand(c, device=dev) + 0.5)


for i in range(n):
    # ---- conv2d (every other case forces the Winograd kernel onto shapes below its block threshold)
    S.Conv2dPlan.WINO_MIN_BLOCKS = 0 if i % 2 else 128
    cin, cout = random.choice([1, 3, 5, 16, 33, 64, 130]), random.choice([1, 7, 16, 32, 48, 127])
    k = random.choice([1, 3, 3])
    stride = random.choice([1, 1, 2])
    dil = 1 if (stride == 2 or k == 1) else random.choice([1, 2, 3, 4, 5, 8, 16])
    b, h, w = random.choice([1, 2, 3]), random.randint(1, 40), random.randint(1, 150)
    print('C2', i, cin, cout, k, stride, dil, b, h, w, flush=True)
    x = torch.randn(b, cin, h, w, device=dev)
    wt = torch.randn(cout, cin, k, k, device=dev) * 0.1
    bnp = bn(cout)
    ref = F.batch_norm(F.conv2d(x, wt, None, stride, dil if k == 3 else 0, dil if k == 3 else 1), bnp[2], bnp[3], bnp[0], bnp[1], False, 0.0, 1e-5)
    use_res = random.random() < 0.5
    res = torch.randn_like(ref) if use_res else None
    if use_res:
        ref = ref + res
    ref = torch.relu(ref)
    out = S.Conv2dPlan(wt, bnp, dilation=dil, act=S.ACT_RELU, stride=stride)(x, residual=res)
    torch.cuda.synchronize()
    e = rel(out, ref) if out.shape == ref.shape else 1e9
    if e > 2e-5:
        bad += 1
        print("conv2d FAIL", dict(cin=cin, cout=cout, k=k, stride=stride, dil=dil, b=b, h=h, w=w, res=use_res), e)
    # ---- conv3d
    cin, cout = random.choice([4, 8, 20, 32, 64]), random.choice([1, 8, 16, 32, 40, 64])
    k = random.choice([1, 3, 3])
    stride = 1 if (k == 1 or cout == 1) else random.choice([1, 2])
    b, d, h, w = random.choice([1, 2]), random.randint(1, 9), random.randint(1, 13), random.randint(1, 70)
    print('C3', i, cin, cout, k, stride, b, d, h, w, flush=True)
    x = torch.randn(b, cin, d, h, w, device=dev)
    wt = torch.randn(cout, cin, k, k, k, device=dev) * 0.1
    bnp = bn(cout)
    ref = F.batch_norm(F.conv3d(x, wt, None, stride, k // 2), bnp[2], bnp[3], bnp[0], bnp[1], False, 0.0, 1e-5)
    ref = torch.relu(ref)
    out = S.Conv3dPlan(wt, bnp, stride=stride, act=S.ACT_RELU, precision="f32")(x)
    torch.cuda.synchronize()
    e = rel(out, ref) if out.shape == ref.shape else 1e9
    if e > 2e-5:
        bad += 1
        print("conv3d FAIL", dict(cin=cin, cout=cout, k=k, stride=stride, b=b, d=d, h=h, w=w), e)
    # ---- deconv3d (k3 with fused redir, k4)
    cin, cout, cskip = random.choice([8, 16, 40, 64]), random.choice([8, 16, 32, 48]), random.choice([8, 16, 32, 40])
    b, d, h, w = random.choice([1, 2]), random.randint(1, 5), random.randint(1, 7), random.randint(1, 40)
    print('DC', i, cin, cout, cskip, b, d, h, w, flush=True)
    x = torch.randn(b, cin, d, h, w, device=dev)
    wt = torch.randn(cin, cout, 3, 3, 3, device=dev) * 0.1
    wr = torch.randn(cout, cskip, 1, 1, 1, device=dev) * 0.1
    skip = torch.randn(b, cskip, 2 * d, 2 * h, 2 * w, device=dev)
    b1, b2 = bn(cout), bn(cout)
    ref = torch.relu(F.batch_norm(F.conv_transpose3d(x, wt, None, 2, 1, 1), b1[2], b1[3], b1[0], b1[1], False, 0.0, 1e-5)
                     + F.batch_norm(F.conv3d(skip, wr), b2[2], b2[3], b2[0], b2[1], False, 0.0, 1e-5))
    out = S.Deconv3dPlan(wt, b1, act=S.ACT_RELU, redir=(wr, b2))(x, skip=skip)
    torch.cuda.synchronize()
    e = rel(out, ref)
    if e > 2e-5:
        bad += 1
        print("deconv+redir FAIL", dict(cin=cin, cout=cout, cskip=cskip, b=b, d=d, h=h, w=w), e)
    w4 = torch.randn(cin, cout, 4, 4, 4, device=dev) * 0.1
    ref = F.leaky_relu(F.conv_transpose3d(x, w4, None, 2, 1), 0.01)
    out = S.Deconv3dPlan(w4, None, act=S.ACT_LEAKY)(x)
    e = rel(out, ref)
    if e > 2e-5:
        bad += 1
        print("deconv k4 FAIL", dict(cin=cin, cout=cout, b=b, d=d, h=h, w=w), e)
print("cases", n, "failures", bad)

# ---- builders, patch stencils, regression tail (second pass) ----
bad2 = 0
for i in range(n):
    b, g, cpg, h, w, d = random.choice([1, 2]), random.choice([1, 4, 8]), random.choice([1, 3, 8, 12]), random.randint(1, 9), random.randint(1, 80), random.randint(1, 20)
    print('GW', i, b, g, cpg, h, w, d, flush=True)
    l, r = torch.randn(b, g * cpg, h, w, device=dev), torch.randn(b, g * cpg, h, w, device=dev)
    ref = torch.zeros(b, g, d, h, w, device=dev)
    refc = torch.zeros(b, 2 * g * cpg, d, h, w, device=dev)
    for k in range(d):
        if k < w:
            ref[:, :, k, :, k:] = (l[..., k:] * r[..., :w - k]).view(b, g, cpg, h, w - k).mean(2)
            refc[:, g * cpg:, k, :, k:] = r[..., :w - k]
        refc[:, :g * cpg, k] = l
    import diffuvolume_amd as dv
    out = dv.build_gwc_volume(l, r, d, g); torch.cuda.synchronize()
    if float((out - ref).abs().max()) > 1e-5:
        bad2 += 1; print("gwc FAIL", (b, g, cpg, h, w, d))
    outc = dv.build_concat_volume(l, r, d); torch.cuda.synchronize()
    if not torch.equal(outc, refc):
        bad2 += 1; print("concat FAIL", (b, g, cpg, h, w, d))
    # regression tail
    dd, hh, ww = random.choice([4, 12, 48]), random.randint(1, 9), random.randint(1, 40)
    print('RG', i, dd, hh, ww, flush=True)
    cost = torch.randn(b, 1, dd, hh, ww, device=dev) * 3
    disp, unc = S.upsample_softmax_regress(cost); torch.cuda.synchronize()
    p = torch.softmax(F.interpolate(cost, scale_factor=4, mode="trilinear", align_corners=False).squeeze(1), dim=1)
    kk = torch.arange(4 * dd, device=dev, dtype=torch.float32).view(1, -1, 1, 1)
    refd = (p * kk).sum(1)
    if float((disp - refd).abs().max()) > 2e-3:
        bad2 += 1; print("regress FAIL", (b, dd, hh, ww), float((disp - refd).abs().max()))
    # patch stencils
    gg, d2, h2, w2 = 40, random.randint(1, 3), random.randint(1, 40), random.randint(1, 150)
    print('PV', i, d2, h2, w2, flush=True)
    xv = torch.randn(b, gg, d2, h2, w2, device=dev)
    w1, w2_ = torch.randn(gg, 9, device=dev), torch.randn(gg, 9, device=dev)
    dil = torch.tensor([1] * 8 + [2] * 16 + [3] * 16, dtype=torch.int32, device=dev)
    y = F.conv3d(xv, w1.view(gg, 1, 1, 3, 3), None, 1, (0, 1, 1), 1, gg)
    refp = torch.cat([F.conv3d(y[:, a:b_], w2_[a:b_].view(-1, 1, 1, 3, 3), None, 1, (0, dl, dl), dl, b_ - a)
                      for a, b_, dl in ((0, 8, 1), (8, 24, 2), (24, 40, 3))], dim=1)
    outp = S.patch_volume(xv, w1, w2_, dil); torch.cuda.synchronize()
    if rel(outp, refp) > 2e-5:
        bad2 += 1; print("patch FAIL", (b, d2, h2, w2), rel(outp, refp))
print("second pass cases", n, "failures", bad2)

# ---- third pass: conv3d prologue / residual / split-fp16, refinement inputs, gated conv2d ----
from diffuvolume_amd.pwcnet_ddim import groupwise_corr_pm


def warp(x, disp):
    """Right features sampled at x - disp (bilinear, zeros outside, samples touching the border masked) -- the
    statement refine_inputs is checked against (KITTI12/models/submodule.py:137-176)."""
    b, c, h, w = x.shape
    xx = torch.arange(w, device=x.device, dtype=torch.float32).view(1, 1, 1, w).expand(b, 1, h, w)
    yy = torch.arange(h, device=x.device, dtype=torch.float32).view(1, 1, h, 1).expand(b, 1, h, w)
    gx = 2.0 * (xx - disp) / max(w - 1, 1) - 1.0
    gy = 2.0 * yy / max(h - 1, 1) - 1.0
    grid = torch.cat((gx, gy), 1).permute(0, 2, 3, 1)
    out = F.grid_sample(x, grid)
    mask = (F.grid_sample(torch.ones_like(x), grid) >= 0.999).float()
    return out * mask
bad3 = 0
for i in range(n):
    cin, cout = random.choice([8, 32, 64]), random.choice([16, 32, 64])
    b, d, h, w = random.choice([1, 2]), random.randint(1, 6), random.randint(1, 9), random.randint(1, 60)
    print('C3b', i, cin, cout, b, d, h, w, flush=True)
    x = torch.randn(b, cin, d, h, w, device=dev)
    sc = torch.rand(b, d, h, w, device=dev)
    wt = torch.randn(cout, cin, 3, 3, 3, device=dev) * 0.1
    res = torch.randn(b, cout, d, h, w, device=dev)
    bnp = bn(cout)
    ref = torch.relu(F.batch_norm(F.conv3d(x * sc.unsqueeze(1), wt, None, 1, 1), bnp[2], bnp[3], bnp[0], bnp[1], False, 0.0, 1e-5) + res)
    for prec in ("f32", "f32_direct", "f16x3"):
        out = S.Conv3dPlan(wt, bnp, stride=1, act=S.ACT_RELU, precision=prec)(x, in_scale=sc, residual=res); torch.cuda.synchronize()
        if rel(out, ref) > 2e-5:
            bad3 += 1; print("conv3d", prec, "FAIL", (cin, cout, b, d, h, w), rel(out, ref))
    c, h2, w2 = random.choice([8, 20, 32]), random.randint(1, 12), random.randint(24, 200)
    print('RI', i, c, h2, w2, flush=True)
    fl, fr = torch.randn(b, c, h2, w2, device=dev), torch.randn(b, c, h2, w2, device=dev)
    p3 = torch.rand(b, 1, h2, w2, device=dev) * 70 - 8
    da, db = torch.randn(c, device=dev) * 0.1, torch.randn(c, device=dev) * 0.1
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        frw = warp(fr, p3)
    aff = da.view(1, c, 1, 1) * p3 + db.view(1, c, 1, 1)
    refr = torch.cat((fl - frw, fl, aff * torch.tanh(F.softplus(aff)), p3, groupwise_corr_pm(fl, frw, 24)), dim=1)
    outr = S.refine_inputs(fl, fr, p3, da, db, 24); torch.cuda.synchronize()
    if float((outr - refr).abs().max()) > 3e-4:      # the sample coordinate itself carries ~W * 2^-24 px of rounding
        bad3 += 1; print("refine_inputs FAIL", (b, c, h2, w2), float((outr - refr).abs().max()))
    parts = [torch.randn(b, cc, h2, w2, device=dev) for cc in random.sample([4, 6, 16, 30, 64], random.randint(1, 4))]
    co = random.choice([16, 32, 40])
    wt2 = torch.randn(co, sum(t.shape[1] for t in parts), 3, 3, device=dev) * 0.05
    bias = torch.randn(co, device=dev) * 0.1
    hh, zz = torch.randn(b, co, h2, w2, device=dev), torch.rand(b, co, h2, w2, device=dev)
    v = torch.tanh(F.conv2d(torch.cat(parts, 1), wt2, bias, 1, 1))
    outg = S.Conv2dPlan(wt2, None, act=S.ACT_TANH, bias=bias)(parts, blend=(zz, hh)); torch.cuda.synchronize()
    if float((outg - ((1 - zz) * hh + zz * v)).abs().max()) > 2e-5:
        bad3 += 1; print("gated cat conv2d FAIL", [t.shape[1] for t in parts], co, h2, w2)
print("third pass cases", n, "failures", bad3)
