"""The polyphase minimal-filtering form of the 3x3x3 stride-2 convolution (csrc/conv3d_s2pp.hip; hourglass conv1 / conv3,
SceneFlow/models/acv_ddim.py:60,:66) against torch's fp32 convolution on the CPU and against the direct implicit-GEMM
kernel it replaces: <= 1e-5 of the layer's output scale (fp32 re-association plus one subtraction per operand)."""
import pytest
import torch

from diffuvolume_amd import _lib
from diffuvolume_amd import submodule as S
from diffuvolume_amd.synth import _gen

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def dev(t):
    return t.to(DEV)


def rel_err(a, b):
    return float((a.cpu().double() - b.double()).abs().max() / b.double().abs().max().clamp(min=1e-30))


def _case(cfg, seed=41):
    cin, cout, dims = cfg
    g = _gen(seed, str(cfg))
    x = torch.randn(dims[0], cin, *dims[1:], generator=g)
    w = torch.randn(cout, cin, 3, 3, 3, generator=g) * (2.0 / (27 * cin)) ** 0.5
    bn = (torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g) * 0.1,
          torch.randn(cout, generator=g) * 0.1, torch.rand(cout, generator=g) + 0.5)
    return x, w, bn, g


# every tile shape of a wave's 16 tiles exact and ragged (8x8 / 16x4 / 4x16 outputs), odd D / H / W (half tiles, the last
# input row / column / plane missing), channel tails on the input side (Cin % 4), two 64-channel blocks, batch > 1
# (widths that are not a multiple of 4 stay on the direct kernel: the plan falls back, the assertions are the same)
CASES = [(32, 64, (1, 8, 16, 32)), (32, 64, (2, 6, 32, 16)), (64, 128, (1, 4, 64, 8)), (5, 64, (1, 3, 5, 8)),
         (7, 64, (2, 5, 9, 36)), (33, 128, (1, 2, 18, 60)), (16, 64, (1, 7, 31, 12)), (4, 64, (1, 1, 1, 4)),
         (12, 192, (1, 4, 12, 20)), (6, 64, (3, 9, 21, 44)), (8, 64, (1, 3, 6, 7)), (8, 64, (2, 13, 70, 72))]


@pytest.mark.parametrize("cfg", CASES)
def test_s2pp_vs_torch_and_direct(cfg, monkeypatch):
    x, w, bn, g = _case(cfg)
    cout = cfg[1]
    assert _lib.load().dv_conv3d_s2pp_supported(cfg[0], cout, *cfg[2][1:]) == int(cfg[2][3] % 4 == 0)
    conv = torch.nn.functional.conv3d(x, w, None, 2, 1)
    y_ref = torch.nn.functional.batch_norm(conv, bn[2], bn[3], bn[0], bn[1], False, 0.0, 1e-5)
    plan = S.Conv3dPlan(dev(w), tuple(dev(t) for t in bn), stride=2, act=S.ACT_NONE)
    assert plan.s2pp
    out = plan(dev(x))
    assert out.shape == y_ref.shape
    assert rel_err(out, y_ref) < 1e-5
    # residual + ReLU epilogue, and no BatchNorm at all (PCW's plain Conv3d, pwcnet_ddim.py:137)
    res = torch.randn(y_ref.shape, generator=g)
    plan_r = S.Conv3dPlan(dev(w), tuple(dev(t) for t in bn), stride=2, act=S.ACT_RELU)
    assert rel_err(plan_r(dev(x), residual=dev(res)), torch.relu(y_ref + res)) < 1e-5
    assert rel_err(S.Conv3dPlan(dev(w), None, stride=2, act=S.ACT_NONE)(dev(x)), conv) < 1e-5
    # the direct kernel on the same weights (what DV_S2PP=0 and every filter-prologue call run)
    monkeypatch.setenv("DV_S2PP", "0")
    direct = S.Conv3dPlan(dev(w), tuple(dev(t) for t in bn), stride=2, act=S.ACT_NONE)
    assert not direct.s2pp
    assert rel_err(out, direct(dev(x)).cpu()) < 1e-5
    # a call WITH the filter prologue on a polyphase plan takes the direct kernel
    scale = torch.rand(x.shape[0], *x.shape[2:], generator=g)
    y2 = torch.nn.functional.batch_norm(torch.nn.functional.conv3d(x * scale.unsqueeze(1), w, None, 2, 1),
                                        bn[2], bn[3], bn[0], bn[1], False, 0.0, 1e-5)
    assert rel_err(plan(dev(x), in_scale=dev(scale)), y2) < 1e-5


def test_s2pp_c_abi_and_errors():
    """Straight through the C ABI (raw pointers, caller-allocated output) and the argument checks."""
    lib = _lib.load()
    x, w, bn, _ = _case((8, 64, (1, 4, 8, 8)), seed=43)
    xd, wd = dev(x), dev(w)
    wp = torch.empty(lib.dv_conv3d_s2pp_packed_floats(8, 64), dtype=torch.float32, device=DEV)
    out = torch.empty(1, 64, 2, 4, 4, device=DEV)
    s = _lib.stream_ptr()
    assert lib.dv_conv3d_s2pp_pack_weights_f32(wd.data_ptr(), wp.data_ptr(), 8, 64, s) == 0
    assert lib.dv_conv3d_s2pp_f32(xd.data_ptr(), wp.data_ptr(), None, None, None, out.data_ptr(), 1, 8, 4, 8, 8, 64, 0, s) == 0
    assert rel_err(out, torch.nn.functional.conv3d(x, w, None, 2, 1)) < 1e-5
    assert lib.dv_conv3d_s2pp_f32(None, wp.data_ptr(), None, None, None, out.data_ptr(), 1, 8, 4, 8, 8, 64, 0, s) != 0
    assert lib.dv_conv3d_s2pp_f32(xd.data_ptr(), wp.data_ptr(), None, None, None, out.data_ptr(), 1, 8, 0, 8, 8, 64, 0, s) != 0
    assert lib.dv_conv3d_s2pp_f32(xd.data_ptr(), wp.data_ptr(), None, None, None, out.data_ptr(), 1, 8, 4, 8, 8, 64, 99, s) != 0
    assert not lib.dv_conv3d_s2pp_supported(32, 32, 4, 8, 8) and not lib.dv_conv3d_s2pp_supported(8, 64, 4, 8, 6)
    assert lib.dv_conv3d_s2pp_f32(xd.data_ptr(), wp.data_ptr(), None, None, None, out.data_ptr(), 1, 8, 4, 8, 6, 64, 0, s) != 0


def test_s2pp_shard_invariant_and_reproducible():
    """Pairs [lo, hi) alone give the bits they have inside the batch (what the multi-GPU sharding rests on), and a rerun
    gives the same bits."""
    x, w, bn, _ = _case((32, 64, (4, 8, 16, 24)), seed=47)
    plan = S.Conv3dPlan(dev(w), tuple(dev(t) for t in bn), stride=2, act=S.ACT_RELU)
    full = plan(dev(x))
    assert torch.equal(full, plan(dev(x)))
    assert torch.equal(full[1:3], plan(dev(x[1:3].contiguous())))
    assert torch.equal(full[3:], plan(dev(x[3:].contiguous())))
