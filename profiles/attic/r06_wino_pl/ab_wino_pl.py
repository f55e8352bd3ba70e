"""A/B of the two launch shapes of the 3-D Winograd convolution (csrc/conv3d_wino.hip one-tile blocks vs csrc/conv3d_wino_pl.hip
persistent, two groups of four waves): both against a float64 torch reference on small shapes, bit-equality of the two (they
sum every output in the same order) and of capped grids, then time per launch at the bench sizes, alternating.
python tools/ab_wino_pl.py [--time] [--reps N]"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
import torch.nn.functional as F
from diffuvolume_amd import submodule as S, _lib

dev = "cuda:0"
torch.manual_seed(0)
lib = _lib.load()


def check(b, cin, cout, dims, res=False, act=S.ACT_RELU):
    x = torch.randn(b, cin, *dims, device=dev)
    w = torch.randn(cout, cin, 3, 3, 3, device=dev) * 0.05
    bn = tuple(torch.rand(cout, device=dev) + 0.5 for _ in range(4))
    r = torch.randn(b, cout, *dims, device=dev) if res else None
    plan = S.Conv3dPlan(w, bn, act=act, precision="f32")
    ref = F.conv3d(x.double(), w.double(), padding=1)
    g, be, m, v = (t.double() for t in bn)
    s_ = g / torch.sqrt(v + 1e-5)
    ref = ref * s_.view(1, -1, 1, 1, 1) + (be - m * s_).view(1, -1, 1, 1, 1)
    if res:
        ref = ref + r.double()
    ref = torch.relu(ref) if act == S.ACT_RELU else ref
    sup = lib.dv_conv3d_wino_pl_supported(cin, cout, *dims)
    outs = []
    for impl in (1, 2):
        lib.dv_conv3d_wino_set_impl(impl)
        outs.append(plan(x, residual=r))
    torch.cuda.synchronize()
    errs = [((o.double() - ref).abs().max() / ref.abs().max()).item() for o in outs]
    same = bool(torch.equal(outs[0], outs[1]))
    for cap in (1, 3, 8, 13):
        lib.dv_conv3d_wino_pl_set_max_blocks(cap)
        poison = torch.full_like(outs[1], float("nan")); del poison
        yc = plan(x, residual=r)
        if not torch.equal(yc, outs[1]):
            same = False
            d = (yc - outs[1]).abs()
            print(f"   cap {cap}: max abs diff {d.max().item():.3e}, {(d > 0).sum().item()} of {d.numel()} differ (nan: {torch.isnan(yc).sum().item()})", flush=True)
    lib.dv_conv3d_wino_pl_set_max_blocks(0)
    lib.dv_conv3d_wino_set_impl(0)
    ok = max(errs) < 1e-5 and same
    print(f"B{b} {cin}->{cout} {dims} res={res} act={act} pl_supported={sup}: rel err one-tile {errs[0]:.2e} persistent {errs[1]:.2e} "
          f"bit-equal {same} {'ok' if ok else 'FAIL'}", flush=True)
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
good &= check(1, 8, 32, (8, 8, 32))
good &= check(2, 4, 32, (5, 4, 16), res=True)
good &= check(1, 32, 32, (6, 12, 48), res=True)
good &= check(1, 64, 64, (4, 16, 24), act=S.ACT_NONE)            # 8 x 8 tiles
good &= check(1, 128, 128, (3, 32, 12), res=True)                # 4 x 16 tiles
good &= check(3, 16, 40, (9, 8, 32))                             # Cout tail (40 = 32 + 8)
good &= check(1, 32, 32, (48, 8, 16), res=True, act=S.ACT_NONE)
good &= check(1, 20, 32, (4, 4, 16))
good &= check(1, 6, 32, (4, 4, 16))                               # Cin % 4: not supported, both the one-tile kernel
good &= check(1, 8, 32, (4, 6, 16))                               # ragged H: not supported
print("ALL OK" if good else "SOME FAILED", flush=True)

if "--time" in sys.argv:
    reps = int(sys.argv[sys.argv.index("--reps") + 1]) if "--reps" in sys.argv else 3
    for name, cin, cout, dims in (("c32", 32, 32, (48, 128, 240)), ("c64", 64, 64, (24, 64, 120)), ("c128", 128, 128, (12, 32, 60))):
        x = torch.randn(8, cin, *dims, device=dev)
        w = torch.randn(cout, cin, 3, 3, 3, device=dev) * 0.05
        bn = tuple(torch.rand(cout, device=dev) + 0.5 for _ in range(4))
        plan = S.Conv3dPlan(w, bn, act=S.ACT_RELU, precision="f32")
        r = torch.randn(8, cout, *dims, device=dev)
        fl = 2.0 * x.numel() * cout * 27
        for label, run in (("", lambda: plan(x)), (" +res", lambda: plan(x, residual=r))):
            lib.dv_conv3d_wino_set_impl(1); y1 = run(); lib.dv_conv3d_wino_set_impl(2); y2 = run()
            eq = bool(torch.equal(y1, y2)); del y1, y2
            out = []
            for _ in range(reps):
                for impl in (1, 2):
                    lib.dv_conv3d_wino_set_impl(impl)
                    out.append((impl, timeit(run, 20)))
            lib.dv_conv3d_wino_set_impl(0)
            a = [f"{ms:.3f}" for i, ms in out if i == 1]; b = [f"{ms:.3f}" for i, ms in out if i == 2]
            ma, mb = min(ms for i, ms in out if i == 1), min(ms for i, ms in out if i == 2)
            print(f"{name + label:9s} one-tile {' / '.join(a)} ms ({fl / ma / 1e9 / 2.25 / 157.3:.3f} issued)   persistent {' / '.join(b)} ms "
                  f"({fl / mb / 1e9 / 2.25 / 157.3:.3f})   {(mb / ma - 1) * 100:+.1f} %   bit-equal {eq}", flush=True)
        del x, r, plan
