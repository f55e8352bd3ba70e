"""The F(2x2x2,3x3x3) form of the 3x3x3 stride-1 aggregation convolution (csrc/conv3d_wino3.hip; convbn_3d of the dres /
hourglass / classifier layers, SceneFlow/models/submodule.py:94-97, acv_ddim.py:60-70, :200-222) against torch's `conv3d`
on the CPU in float64 (<= 1e-5 of the layer's output scale: fp32 re-association only) and against the in-plane
F(2x2,3x3) kernel it replaces.  Also: odd sizes in every axis, channel tails, the three tile shapes, residual /
activations, a rerun and a shard of a batch reproduce the batch's bits, and a call with a filter prologue still takes the
in-plane kernel."""
import pytest
import torch

from diffuvolume_amd import _lib
from diffuvolume_amd import submodule as S
from diffuvolume_amd.synth import _gen

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(autouse=True)
def _every_layer_on_the_kernel_under_test():
    old = S.Conv3dPlan.WINO3_MIN_CIN
    S.Conv3dPlan.WINO3_MIN_CIN = 1
    yield
    S.Conv3dPlan.WINO3_MIN_CIN = old
F = torch.nn.functional
ACT = {"relu": S.ACT_RELU, "mish": S.ACT_MISH, "leaky": S.ACT_LEAKY, "none": S.ACT_NONE}


def _act(y, act):
    return {"relu": torch.relu(y), "mish": y * torch.tanh(F.softplus(y)), "leaky": F.leaky_relu(y, 0.01), "none": y}[act]


def _bn(cout, g):
    return (torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g) * 0.1,
            torch.randn(cout, generator=g) * 0.1, torch.rand(cout, generator=g) + 0.5)


# (B, Cin, Cout, D, H, W, activation, residual)
CASES = [
    (1, 32, 32, 8, 16, 32, "relu", False),        # 4 x 16 tiles (shape 0), whole tiles
    (2, 32, 32, 7, 13, 36, "none", True),         # odd depth / height, ragged columns: half-empty depth pair, partial tiles
    (1, 6, 40, 5, 9, 20, "leaky", False),         # channel tails: Cin % 4, Cout % 32
    (1, 64, 64, 6, 24, 24, "relu", True),         # 8 x 8 tiles (shape 1)
    (1, 128, 128, 4, 32, 12, "relu", False),      # 16 x 4 tiles (shape 2)
    (1, 4, 2, 2, 2, 4, "none", False),            # one chunk, one tile
    (1, 12, 32, 3, 4, 60, "mish", False),
    (3, 32, 64, 2, 8, 16, "relu", False),         # two output-channel blocks
    (1, 1, 32, 1, 1, 4, "none", False),
]


def _case(cfg, seed=97):
    b, cin, cout, d, h, w, act, res = cfg
    g = _gen(seed, str(cfg))
    x = torch.randn(b, cin, d, h, w, generator=g)
    wt = torch.randn(cout, cin, 3, 3, 3, generator=g) * (2.0 / (27 * cin)) ** 0.5
    bnd = _bn(cout, g)
    y = F.batch_norm(F.conv3d(x.double(), wt.double(), padding=1), bnd[2].double(), bnd[3].double(), bnd[0].double(),
                     bnd[1].double(), False, 0.0, 1e-5)
    r = torch.randn(b, cout, d, h, w, generator=g) if res else None
    if res:
        y = y + r.double()
    plan = S.Conv3dPlan(wt.to(DEV), tuple(t.to(DEV) for t in bnd), act=ACT[act], precision="f32")
    return plan, x.to(DEV), (None if r is None else r.to(DEV)), _act(y, act)


def _rel(a, ref):
    return float((a.cpu().double() - ref).abs().max() / ref.abs().max().clamp(min=1e-30))


@pytest.mark.parametrize("cfg", CASES, ids=[str(c) for c in CASES])
def test_wino3_vs_float64_and_the_in_plane_kernel(cfg):
    plan, x, r, ref = _case(cfg)
    assert plan.wino and S.Conv3dPlan.WINO3
    assert _lib.load().dv_conv3d_wino3_supported(*cfg[1:3], *cfg[3:6])
    assert cfg[5] % 4 == 0
    y3 = plan(x, residual=r)
    torch.cuda.synchronize()
    S.Conv3dPlan.WINO3 = False
    try:
        y2 = plan(x, residual=r)
    finally:
        S.Conv3dPlan.WINO3 = True
    e3, e2 = _rel(y3, ref), _rel(y2, ref)
    assert e3 <= 1e-5, (e3, e2)
    assert e3 <= 3 * e2 + 2e-7, (e3, e2)                 # no worse than the in-plane form (tools/probes/wino_f222_numerics.py)
    assert torch.equal(plan(x, residual=r), y3)          # a rerun: the same bits


def test_wino3_shard_of_a_batch_reproduces_the_batch():
    plan, x, r, _ = _case((3, 32, 32, 6, 12, 24, "relu", True))
    y = plan(x, residual=r)
    for i in range(3):
        assert torch.equal(plan(x[i:i + 1].contiguous(), residual=r[i:i + 1].contiguous()), y[i:i + 1])


def test_wino3_nan_and_inf_stay_local():
    plan, x, r, _ = _case((1, 8, 32, 6, 8, 16, "none", False))
    y0 = plan(x)
    x2 = x.clone()
    x2[0, 3, 2, 4, 9] = float("nan")
    y = plan(x2)
    bad = ~torch.isfinite(y)
    assert bad.any()
    # a 3x3x3 convolution spreads a NaN over its 27 neighbours; the 4x4x4 transform patches may not spread it further
    zz, yy, xx = torch.nonzero(bad[0].any(0), as_tuple=True)
    assert zz.min() >= 1 and zz.max() <= 3 and yy.min() >= 3 and yy.max() <= 5 and xx.min() >= 8 and xx.max() <= 10
    assert torch.equal(y[~bad], y0[~bad])


def test_rows_that_are_not_whole_quads_take_the_in_plane_kernel():
    lib = _lib.load()
    assert lib.dv_conv3d_wino3_supported(32, 32, 7, 13, 37) == 0
    plan, x, r, ref = _case((2, 32, 32, 7, 13, 37, "none", True))
    assert _rel(plan(x, residual=r), ref) <= 1e-5


def test_filter_prologue_takes_the_in_plane_kernel():
    plan, x, _, _ = _case((1, 8, 32, 4, 8, 16, "relu", False))
    s = torch.rand(1, 4, 8, 16, device=DEV)
    y = plan(x, in_scale=s)
    ref = plan(x * s[:, None])
    assert _rel(y, ref.cpu().double()) <= 1e-5


def test_c_abi_rejects_bad_arguments():
    lib = _lib.load()
    assert lib.dv_conv3d_wino3_packed_floats(32, 32) == 8 * 8192
    assert lib.dv_conv3d_wino3_packed_floats(5, 33) == 2 * 2 * 8192
    assert lib.dv_conv3d_wino3_supported(32, 32, 48, 128, 240) == 1
    assert lib.dv_conv3d_wino3_supported(32, 32, 512, 1024, 1024) == 0
    assert lib.dv_conv3d_wino3_f32(0, 0, 0, 0, 0, 0, 1, 4, 2, 2, 2, 32, 0, 0) != 0
