"""Random-shape checks of the round-6 kernels against the kernels they replace (both through the plans / the C ABI):
  * conv3d_wino3_kernel (F(2x2x2,3x3x3)) vs conv3d_wino_kernel (F(2x2,3x3) in-plane) -- channel tails, odd depth / height,
    ragged tiles in every shape, residual, activations; <= 3e-6 of the output scale between the two fp32 forms;
  * deconv3d_pl_kernel (persistent) vs deconv3d_mfma_kernel (one tile per block) -- <= 3e-6, and grid caps bit-identical.
python tools/fuzz_round6.py [cases] [seed]"""
import random
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from diffuvolume_amd import _lib
from diffuvolume_amd import submodule as S

DEV = "cuda:0"


def rel(a, b):
    return float((a - b).abs().max() / b.abs().max().clamp(min=1e-30))


def wino3_case(rng, g):
    b = rng.choice([1, 1, 2, 3])
    cin = rng.choice([1, 3, 4, 6, 8, 12, 16, 20, 32, 36, 64])
    cout = rng.choice([2, 5, 16, 31, 32, 33, 48, 64, 96])
    d, h = rng.randint(1, 9), rng.randint(1, 40)
    w = 4 * rng.randint(1, 16)
    act = rng.choice([S.ACT_NONE, S.ACT_RELU, S.ACT_LEAKY, S.ACT_MISH])
    x = torch.randn(b, cin, d, h, w, generator=g).to(DEV)
    wt = (torch.randn(cout, cin, 3, 3, 3, generator=g) * (2.0 / (27 * cin)) ** 0.5).to(DEV)
    bn = tuple(t.to(DEV) for t in (torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g) * 0.1,
                                   torch.randn(cout, generator=g) * 0.1, torch.rand(cout, generator=g) + 0.5))
    r = torch.randn(b, cout, d, h, w, generator=g).to(DEV) if rng.random() < 0.4 else None
    plan = S.Conv3dPlan(wt, bn, act=act, precision="f32")
    S.Conv3dPlan.WINO3, S.Conv3dPlan.WINO3_MIN_CIN = True, 1
    y3 = plan(x, residual=r)
    y3b = plan(x, residual=r)
    S.Conv3dPlan.WINO3 = False
    y2 = plan(x, residual=r)
    S.Conv3dPlan.WINO3 = True
    e = rel(y3, y2)
    ok = e <= 3e-6 and torch.equal(y3, y3b)
    return ok, f"wino3 b{b} {cin}->{cout} {d}x{h}x{w} act{act} res{int(r is not None)}: {e:.2e}"


def deconv_case(rng, g):
    lib = _lib.load()
    b = rng.choice([1, 2])
    cin = 8 * rng.randint(1, 12)
    cout = 32 * rng.randint(1, 3)
    d, h, w = rng.randint(1, 5), rng.randint(1, 12), 4 * rng.randint(1, 10)
    cskip = rng.choice([0, 0, 4, 8, 16])
    cskip = min(cskip, cin // 2)
    cskip -= cskip % 4
    act = rng.choice([S.ACT_NONE, S.ACT_RELU])
    x = torch.randn(b, cin, d, h, w, generator=g).to(DEV)
    wt = (torch.randn(cin, cout, 3, 3, 3, generator=g) * (2.0 / (27 * cin)) ** 0.5).to(DEV)
    bn = tuple(t.to(DEV) for t in (torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g) * 0.1,
                                   torch.randn(cout, generator=g) * 0.1, torch.rand(cout, generator=g) + 0.5))
    kw = {}
    if cskip:
        wr = (torch.randn(cout, cskip, 1, 1, 1, generator=g) * (1.0 / cskip) ** 0.5).to(DEV)
        bnr = tuple(t.to(DEV) for t in (torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g) * 0.1,
                                        torch.randn(cout, generator=g) * 0.1, torch.rand(cout, generator=g) + 0.5))
        plan = S.Deconv3dPlan(wt, bn, act=act, redir=(wr, bnr))
        kw = dict(skip=torch.randn(b, cskip, 2 * d, 2 * h, 2 * w, generator=g).to(DEV))
    else:
        plan = S.Deconv3dPlan(wt, bn, act=act)
    if not lib.dv_deconv3d_pl_supported(cin, cout, d, h, w, cskip):
        return True, "deconv skipped (unsupported shape)"
    try:
        lib.dv_deconv3d_set_impl(2)
        yp = plan(x, **kw)
        cap = rng.choice([1, 2, 3, 5, 8, 13])
        lib.dv_deconv3d_pl_set_max_blocks(cap)
        torch.empty(1 << 22, device=DEV).fill_(float("nan"))
        yc = plan(x, **kw)
        lib.dv_deconv3d_pl_set_max_blocks(0)
        lib.dv_deconv3d_set_impl(1)
        y1 = plan(x, **kw)
    finally:
        lib.dv_deconv3d_set_impl(0)
        lib.dv_deconv3d_pl_set_max_blocks(0)
    e = rel(yp, y1)
    ok = e <= 3e-6 and torch.equal(yp, yc)
    return ok, f"deconv b{b} {cin}->{cout} skip{cskip} {d}x{h}x{w} act{act} cap{cap}: {e:.2e}"


def main(cases=120, seed=6):
    rng = random.Random(seed)
    g = torch.Generator(device="cpu").manual_seed(seed)
    bad = 0
    for i in range(cases):
        ok, msg = (wino3_case if i % 3 else deconv_case)(rng, g)
        if not ok:
            bad += 1
            print("FAIL", msg, flush=True)
    print(f"fuzz_round6: {cases} cases, {bad} failures", flush=True)
    return bad


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 120
    sd = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    sys.exit(1 if main(n, sd) else 0)
