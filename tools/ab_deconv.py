"""A/B of the two launch shapes of the k3 transposed convolution (csrc/deconv3d.hip one-tile blocks vs csrc/deconv3d_pl.hip
persistent + loader waves): max error of both against a float64 torch reference on small and ragged shapes, then time per
launch at the bench sizes, alternating.   python tools/ab_deconv.py [--time] [--reps N]"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
import torch.nn.functional as F
from diffuvolume_amd import submodule as S, _lib

dev = "cuda:0"
torch.manual_seed(0)
lib = _lib.load()


def bn_fold(bn):
    g, be, m, v = (t.double() for t in bn)
    s = g / torch.sqrt(v + 1e-5)
    return s, be - m * s


def act_ref(t, act):
    if act == S.ACT_RELU:
        return torch.relu(t)
    if act == S.ACT_MISH:
        return t * torch.tanh(F.softplus(t))
    if act == S.ACT_LEAKY:
        return F.leaky_relu(t, 0.01)
    return t


def check(b, cin, cout, dims, mode="plain", cskip=None, act=S.ACT_RELU):
    x = torch.randn(b, cin, *dims, device=dev)
    w = torch.randn(cin, cout, 3, 3, 3, device=dev) * 0.05
    bn = tuple(torch.rand(cout, device=dev) + 0.5 for _ in range(4))
    odims = tuple(2 * d for d in dims)
    ref = F.conv_transpose3d(x.double(), w.double(), stride=2, padding=1, output_padding=1)
    s, sh = bn_fold(bn)
    ref = ref * s.view(1, -1, 1, 1, 1) + sh.view(1, -1, 1, 1, 1)
    kw = {}
    if mode == "redir":
        cskip = cskip or cout
        rw = torch.randn(cout, cskip, 1, 1, 1, device=dev) * 0.1
        rbn = tuple(torch.rand(cout, device=dev) + 0.5 for _ in range(4))
        plan = S.Deconv3dPlan(w, bn, act=act, redir=(rw, rbn))
        skip = torch.randn(b, cskip, *odims, device=dev)
        rs, rsh = bn_fold(rbn)
        ref = ref + F.conv3d(skip.double(), rw.double()) * rs.view(1, -1, 1, 1, 1) + rsh.view(1, -1, 1, 1, 1)
        kw = dict(skip=skip)
    else:
        plan = S.Deconv3dPlan(w, bn, act=act)
        if mode == "res":
            res = torch.randn(b, cout, *odims, device=dev)
            ref = ref + res.double()
            kw = dict(residual=res)
    ref = act_ref(ref, act)
    sup = lib.dv_deconv3d_pl_supported(cin, cout, *dims, cskip if mode == "redir" else 0)
    errs = []
    for impl in (1, 2):
        assert lib.dv_deconv3d_set_impl(impl) == 0
        y = plan(x, **kw)
        torch.cuda.synchronize()
        errs.append((y.double() - ref).abs().max().item() / max(ref.abs().max().item(), 1e-30))
    same = True
    for cap in (1, 3, 8, 13):                      # few blocks: long tile lists per block; the bits must not change
        lib.dv_deconv3d_pl_set_max_blocks(cap)
        poison = torch.full_like(y, float("nan")); del poison        # the next output lands on these bytes
        yc = plan(x, **kw)
        if not torch.equal(yc, y):
            same = False
            d = (yc - y).abs()
            nz = (d > 0).nonzero()
            print(f"   cap {cap}: max abs diff {d.max().item():.3e} at {d.argmax().item()}, {len(nz)} elements differ of {d.numel()}, first {nz[0].tolist()} last {nz[-1].tolist()}", flush=True)
    lib.dv_deconv3d_pl_set_max_blocks(0)
    lib.dv_deconv3d_set_impl(0)
    ok = max(errs) < 1e-5 and same
    if not same:
        print("   grid-size dependence!", flush=True)
    print(f"B{b} {cin}->{cout} {dims} {mode} cskip={cskip} act={act} pl_supported={sup}: rel err one-tile {errs[0]:.2e} persistent {errs[1]:.2e} "
          f"{'ok' if ok else 'FAIL'}", flush=True)
    return ok


def timeit(run, n):
    run(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        run()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


good = True
good &= check(1, 8, 32, (2, 4, 32), "plain")
good &= check(1, 16, 32, (3, 5, 36), "redir")
good &= check(2, 16, 64, (3, 6, 60), "redir", act=S.ACT_MISH)
good &= check(1, 32, 32, (5, 7, 40), "res", act=S.ACT_LEAKY)
good &= check(2, 64, 32, (4, 6, 120), "redir")
good &= check(1, 128, 64, (3, 4, 60), "redir")
good &= check(1, 24, 32, (1, 1, 4), "plain", act=S.ACT_NONE)
good &= check(3, 16, 32, (2, 3, 8), "redir", cskip=8)
good &= check(1, 64, 96, (2, 9, 44), "res")
good &= check(1, 16, 32, (7, 2, 68), "redir", cskip=4, act=S.ACT_NONE)
good &= check(1, 12, 32, (2, 4, 32), "plain")           # not supported by the persistent kernel: both runs take the one-tile path
print("ALL OK" if good else "SOME FAILED", flush=True)

if "--time" in sys.argv:
    reps = int(sys.argv[sys.argv.index("--reps") + 1]) if "--reps" in sys.argv else 3
    B = 8
    for name, cin, cout, dims, mode in (("dc64x", 64, 32, (24, 64, 120), "redir"), ("dc128x", 128, 64, (12, 32, 60), "redir"),
                                        ("dc64", 64, 32, (24, 64, 120), "plain"), ("dc128", 128, 64, (12, 32, 60), "plain"),
                                        ("dc64r", 64, 32, (24, 64, 120), "res")):
        x = torch.randn(B, cin, *dims, device=dev)
        w = torch.randn(cin, cout, 3, 3, 3, device=dev) * 0.05
        bn = tuple(torch.rand(cout, device=dev) + 0.5 for _ in range(4))
        odims = tuple(2 * d for d in dims)
        if mode == "redir":
            rw = torch.randn(cout, cout, 1, 1, 1, device=dev) * 0.1
            plan = S.Deconv3dPlan(w, bn, act=S.ACT_RELU, redir=(rw, tuple(torch.rand(cout, device=dev) + 0.5 for _ in range(4))))
            t = torch.randn(B, cout, *odims, device=dev)
            run = lambda: plan(x, skip=t)
            fl = 2.0 * x.numel() * cout * 27 + 2.0 * t.numel() * cout
        elif mode == "res":
            plan = S.Deconv3dPlan(w, bn, act=S.ACT_RELU)
            t = torch.randn(B, cout, *odims, device=dev)
            run = lambda: plan(x, residual=t)
            fl = 2.0 * x.numel() * cout * 27
        else:
            plan = S.Deconv3dPlan(w, bn, act=S.ACT_RELU)
            run = lambda: plan(x)
            fl = 2.0 * x.numel() * cout * 27
        lib.dv_deconv3d_set_impl(1); y1 = run(); lib.dv_deconv3d_set_impl(2); y2 = run()
        print(f"   {name}: persistent vs one-tile max rel diff {((y1 - y2).abs().max() / y1.abs().max()).item():.2e}", flush=True)
        del y1, y2
        out = []
        for r in range(reps):
            for impl in (1, 2):
                lib.dv_deconv3d_set_impl(impl)
                out.append((impl, timeit(run, 20)))
        lib.dv_deconv3d_set_impl(0)
        a = [f"{ms:.3f}" for i, ms in out if i == 1]
        b = [f"{ms:.3f}" for i, ms in out if i == 2]
        ma, mb = min(ms for i, ms in out if i == 1), min(ms for i, ms in out if i == 2)
        print(f"{name:7s} one-tile {' / '.join(a)} ms ({fl / ma / 1e9:.1f} TF)   persistent {' / '.join(b)} ms ({fl / mb / 1e9:.1f} TF)   "
              f"{(mb / ma - 1) * 100:+.1f} %", flush=True)
        del x, plan
