"""Parity of the HIP hot path (through the C ABI) with the CPU oracle and with the golden
vectors captured from the reference.  Tolerances: the builders are compared bit for bit
(products are rounded before the ordered sum, like the reference); fp32 GEMM-shaped layers
within 1e-5 relative of the layer's output scale (fp32 re-association only); the end-to-end
DDIM loop as "bulk within 1e-3 px, mean within 1e-4" because hard renewal masks make single
pixels chaotic."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from diffuvolume_amd import submodule as S
from diffuvolume_amd.synth import NoiseTape, _gen, synth_state_dict, synth_stereo_batch
from oracle import acv_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def dev(t):
    return t.to(DEV)


def rel_err(a, b):
    return float((a.cpu().double() - b.double()).abs().max() / b.double().abs().max().clamp(min=1e-30))


# ---------------------------------------------------------------- builders
@pytest.mark.parametrize("tag", ["small", "cpg8", "cpg12", "ragged"])
def test_builders_golden(tag):
    g = load_golden(f"builders_{tag}")
    L, R = dev(g["L"]), dev(g["R"])
    out = S.build_gwc_volume(L, R, g["maxdisp"], g["groups"]).cpu()
    assert out.shape == g["gwc"].shape
    torch.testing.assert_close(out, g["gwc"], atol=1e-7, rtol=1e-6)
    assert torch.equal(S.build_concat_volume(L, R, g["maxdisp"]).cpu(), g["concat"])
    assert torch.equal(S.build_concat_volume(L, R, g["maxdisp"], zero_left=True).cpu(), g["concat_k12"])


@pytest.mark.parametrize("shape", [(2, 320, 8, 64, 48, 40), (1, 96, 4, 312, 48, 8), (1, 24, 3, 78, 12, 2),
                                   (1, 320, 4, 240, 48, 40)])
def test_gwc_oracle(shape):
    b, c, h, w, d, g = shape
    L = torch.randn(b, c, h, w, generator=_gen(3, "L" + str(shape)))
    R = torch.randn(b, c, h, w, generator=_gen(3, "R" + str(shape)))
    ref = O.build_gwc_volume(L, R, d, g)
    out = S.build_gwc_volume(dev(L), dev(R), d, g).cpu()
    torch.testing.assert_close(out, ref, atol=1e-6, rtol=1e-6)
    assert float(out[:, :, 5, :, :5].abs().max()) == 0.0         # x < d stays exactly zero


@pytest.mark.parametrize("shape", [(2, 32, 8, 64, 48), (1, 12, 4, 312, 48), (1, 12, 3, 39, 6)])
def test_concat_oracle(shape):
    b, c, h, w, d = shape
    L = torch.randn(b, c, h, w, generator=_gen(4, "L" + str(shape)))
    R = torch.randn(b, c, h, w, generator=_gen(4, "R" + str(shape)))
    for zl in (False, True):
        assert torch.equal(S.build_concat_volume(dev(L), dev(R), d, zero_left=zl).cpu(),
                           O.build_concat_volume(L, R, d, zero_left=zl))


def test_concat_attention():
    g = load_golden("concat_attention")
    out = S.build_concat_attention_volume(dev(g["L"]), dev(g["R"]), dev(g["att"]), g["maxdisp"]).cpu()
    torch.testing.assert_close(out, g["out"], atol=1e-6, rtol=1e-5)
    L = torch.randn(1, 32, 4, 240, generator=_gen(5, "L"))
    R = torch.randn(1, 32, 4, 240, generator=_gen(5, "R"))
    att = torch.randn(1, 1, 48, 4, 240, generator=_gen(5, "a")) * 2
    ref = O.attention_concat_volume(att, O.build_concat_volume(L, R, 48))
    out = S.build_concat_attention_volume(dev(L), dev(R), dev(att), 48).cpu()
    torch.testing.assert_close(out, ref, atol=1e-6, rtol=1e-5)


def test_builder_errors():
    with pytest.raises(AssertionError):
        S.build_gwc_volume(torch.zeros(1, 6, 2, 4, device=DEV), torch.zeros(1, 6, 2, 4, device=DEV), 2, 4)
    with pytest.raises(RuntimeError):
        S.build_gwc_volume(torch.zeros(1, 8, 2, 4, device=DEV), torch.zeros(1, 8, 2, 5, device=DEV), 2, 4)
    with pytest.raises(Exception):
        S.build_gwc_volume(torch.zeros(1, 8, 2, 4), torch.zeros(1, 8, 2, 4), 2, 4)      # CPU tensors: no fallback


# ---------------------------------------------------------------- regression tail
def test_disparity_regression():
    g = load_golden("disparity_regression")
    torch.testing.assert_close(S.disparity_regression(dev(g["prob"]), 12).cpu(), g["flat"], atol=1e-5, rtol=1e-5)
    torch.testing.assert_close(S.disparity_regression(dev(g["prob"]), 12, keepdim=True).cpu(), g["keepdim"],
                               atol=1e-5, rtol=1e-5)
    with pytest.raises(AssertionError):
        S.disparity_regression(dev(g["prob"])[0], 12)


@pytest.mark.parametrize("tag", ["d12", "d48"])
def test_regression_tail_golden(tag):
    g = load_golden(f"regress_{tag}")
    for ac, sfx in ((False, ""), (True, "_ac")):
        disp, unc = S.upsample_softmax_regress(dev(g["cost"]), True, align_corners=ac)
        torch.testing.assert_close(disp.cpu(), g["disp" + sfx], atol=2e-4, rtol=1e-5)
        torch.testing.assert_close(unc.cpu(), g["unc" + sfx], atol=2e-4, rtol=1e-5)


def test_regression_tail_oracle_fullwidth():
    cost = torch.randn(1, 1, 48, 8, 240, generator=_gen(6, "c")) * 5
    disp_ref, prob = O.upsample_softmax_regress(cost, 192)
    disp, unc = S.upsample_softmax_regress(dev(cost))
    torch.testing.assert_close(disp.cpu(), disp_ref, atol=2e-4, rtol=1e-5)
    torch.testing.assert_close(unc.cpu(), O.disparity_uncertainty(disp_ref, prob), atol=2e-4, rtol=1e-5)


def test_regression_tail_extreme_costs():
    """Very peaked and very negative costs (softmax terms that underflow, a bin at -1e30): the compensated exp2 of the
    tail kernel (csrc/dv_common.h: dv_exp_le0) must give the oracle's result, not NaN."""
    g = _gen(8, "extreme")
    cost = torch.randn(1, 1, 48, 6, 20, generator=g) * 30
    cost[0, 0, 5] = -1e30
    cost[0, 0, 17, 2] = 300.0
    disp_ref, prob = O.upsample_softmax_regress(cost, 192)
    disp, unc = S.upsample_softmax_regress(dev(cost))
    assert bool(torch.isfinite(disp).all()) and bool(torch.isfinite(unc).all())
    torch.testing.assert_close(disp.cpu(), disp_ref, atol=5e-4, rtol=1e-5)
    torch.testing.assert_close(unc.cpu(), O.disparity_uncertainty(disp_ref, prob), atol=5e-4, rtol=1e-5)


# ---------------------------------------------------------------- conv layers
def _bn_tuple(sd, p):
    return (sd[p + ".weight"], sd[p + ".bias"], sd[p + ".running_mean"], sd[p + ".running_var"])


# "f32" runs the 3x3x3 stride-1 layers on the Winograd kernel (csrc/conv3d_wino.hip), "f32_direct" everything on the
# direct implicit GEMM (csrc/conv3d.hip); both are held to the same bars
PRECISIONS = ["f32", "f32_direct"]


@pytest.mark.parametrize("precision", PRECISIONS)
@pytest.mark.parametrize("tag", ["c3s1", "c3s2", "c1s1", "c3s1_wide", "c3s1_one"])
def test_conv_golden(tag, precision):
    from diffuvolume_amd.acv_ddim import _cb3
    g = load_golden(f"layer_{tag}")
    k, s = g["k"], g["stride"]
    if precision == "f32_direct" and not (k == 3 and s == 1 and g["cout"] > 1):
        pytest.skip("same kernel as 'f32' for this layer")
    sd = synth_state_dict(_cb3(g["cin"], g["cout"], k, s, (k - 1) // 2).state_dict(), seed=g["seed"])
    plan = S.Conv3dPlan(dev(sd["0.weight"]), tuple(dev(t) for t in _bn_tuple(sd, "1")), stride=s, act=S.ACT_NONE,
                        precision=precision)
    y = plan(dev(g["x"]))
    assert y.shape == g["y"].shape
    assert rel_err(y, g["y"]) < 1e-5
    plan = S.Conv3dPlan(dev(sd["0.weight"]), tuple(dev(t) for t in _bn_tuple(sd, "1")), stride=s, act=S.ACT_RELU,
                        precision=precision)
    assert rel_err(plan(dev(g["x"])), g["y_relu"]) < 1e-5


@pytest.mark.parametrize("cfg", [(64, 32, 3, 1, (1, 8, 8, 48)), (32, 32, 3, 1, (2, 5, 7, 44)),
                                 (32, 64, 3, 2, (1, 8, 12, 40)), (64, 64, 3, 1, (1, 6, 8, 24)),
                                 (64, 128, 3, 2, (1, 8, 8, 24)), (128, 128, 3, 1, (1, 4, 8, 12)),
                                 (40, 32, 3, 1, (1, 4, 4, 32)), (32, 1, 3, 1, (1, 8, 8, 32)),
                                 (32, 32, 1, 1, (1, 4, 8, 16)), (64, 64, 1, 1, (1, 4, 6, 18)),
                                 (32, 32, 3, 1, (1, 4, 5, 30)), (8, 16, 3, 2, (1, 8, 9, 13))])
@pytest.mark.parametrize("precision", PRECISIONS)
def test_conv_oracle(cfg, precision):
    cin, cout, k, s, dims = cfg
    if precision == "f32_direct" and not (k == 3 and s == 1 and cout > 1):
        pytest.skip("same kernel as 'f32' for this layer")
    g = _gen(7, str(cfg))
    x = torch.randn(dims[0], cin, *dims[1:], generator=g)
    w = torch.randn(cout, cin, k, k, k, generator=g) * (2.0 / (k ** 3 * cin)) ** 0.5
    bn = (torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g) * 0.1,
          torch.randn(cout, generator=g) * 0.1, torch.rand(cout, generator=g) + 0.5)
    y_ref = torch.nn.functional.batch_norm(torch.nn.functional.conv3d(x, w, None, s, (k - 1) // 2),
                                           bn[2], bn[3], bn[0], bn[1], False, 0.0, 1e-5)
    plan = S.Conv3dPlan(dev(w), tuple(dev(t) for t in bn), stride=s, act=S.ACT_NONE, precision=precision)
    assert rel_err(plan(dev(x)), y_ref) < 1e-5
    # fused prologue scale, residual and ReLU (acv_ddim.py:260-262); stride 2 too: it has its own instantiation with the
    # prologue since round 4 (the one without it is what the networks launch)
    if s == 1 or k == 3:
        scale = torch.rand(dims[0], *dims[1:], generator=g)
        res = torch.randn(y_ref.shape, generator=g)
        y2 = torch.nn.functional.batch_norm(torch.nn.functional.conv3d(x * scale.unsqueeze(1), w, None, s, (k - 1) // 2),
                                            bn[2], bn[3], bn[0], bn[1], False, 0.0, 1e-5)
        plan = S.Conv3dPlan(dev(w), tuple(dev(t) for t in bn), stride=s, act=S.ACT_RELU, precision=precision)
        out = plan(dev(x), in_scale=dev(scale), residual=dev(res))
        assert rel_err(out, torch.relu(y2 + res)) < 1e-5


@pytest.mark.parametrize("cfg", [(5, (1, 3, 5, 7), False), (33, (2, 6, 9, 70), True), (32, (1, 4, 8, 130), False),
                                 (1, (1, 1, 1, 1), False), (40, (1, 7, 3, 65), True)])
def test_conv_single_channel_head_edges(cfg):
    """The Cout == 1 classifier head (acv_ddim.py:214/:222) on its vector-ALU kernel (round 2: buffer-load staging,
    double-buffered brick): channel counts that are not a multiple of anything, ragged D / H / W (tiles are
    4 x 8 x 64), the volume * noise prologue and a residual."""
    cin, dims, extras = cfg
    g = _gen(23, str(cfg))
    x = torch.randn(dims[0], cin, *dims[1:], generator=g)
    w = torch.randn(1, cin, 3, 3, 3, generator=g) * (2.0 / (27 * cin)) ** 0.5
    scale = torch.rand(dims[0], *dims[1:], generator=g) if extras else None
    res = torch.randn(dims[0], 1, *dims[1:], generator=g) if extras else None
    y = torch.nn.functional.conv3d(x if scale is None else x * scale.unsqueeze(1), w, None, 1, 1)
    if res is not None:
        y = torch.relu(y + res)
    plan = S.Conv3dPlan(dev(w), None, stride=1, act=S.ACT_RELU if extras else S.ACT_NONE)
    out = plan(dev(x), in_scale=None if scale is None else dev(scale), residual=None if res is None else dev(res))
    assert out.shape == y.shape
    assert float((out.cpu() - y).abs().max()) <= 1e-5 * max(1.0, float(y.abs().max()))


@pytest.mark.parametrize("cfg", [(5, 20, (2, 5, 7, 19)), (3, 40, (1, 3, 3, 3)), (33, 33, (1, 1, 2, 17)),
                                 (4, 32, (1, 9, 5, 6)), (7, 70, (1, 2, 9, 33)), (12, 32, (1, 4, 4, 16)),
                                 # planes that select each tile shape of a wave's 16 Winograd tiles (least padding wins):
                                 # 8x8 outputs exact / ragged, 16x4 outputs exact / ragged, 4x16 outputs ragged
                                 (8, 32, (1, 5, 8, 120)), (6, 20, (2, 3, 7, 37)), (8, 64, (1, 4, 16, 60)),
                                 (5, 32, (1, 6, 30, 10)), (4, 16, (1, 3, 19, 9))])
def test_conv_winograd_edges(cfg):
    """Winograd kernel on shapes that leave every kind of partial tile: odd H / W (half 2x2 output tiles), depth not a
    multiple of the 4 planes of a block, channel tails on both sides (Cin % 4, Cout % 32), rows that are not
    16-byte aligned (scalar store path), with residual and the `volume * noise` prologue."""
    cin, cout, dims = cfg
    g = _gen(31, str(cfg))
    x = torch.randn(dims[0], cin, *dims[1:], generator=g)
    w = torch.randn(cout, cin, 3, 3, 3, generator=g) * (2.0 / (27 * cin)) ** 0.5
    scale = torch.rand(dims[0], *dims[1:], generator=g)
    res = torch.randn(dims[0], cout, *dims[1:], generator=g)
    ref = torch.nn.functional.leaky_relu(
        torch.nn.functional.conv3d(x.double() * scale.double().unsqueeze(1), w.double(), None, 1, 1) + res.double(), 0.01)
    plan = S.Conv3dPlan(dev(w), None, stride=1, act=S.ACT_LEAKY, precision="f32")
    assert plan.wino
    out = plan(dev(x), in_scale=dev(scale), residual=dev(res))
    assert rel_err(out, ref.float()) < 1e-5
    direct = S.Conv3dPlan(dev(w), None, stride=1, act=S.ACT_LEAKY, precision="f32_direct")(dev(x), in_scale=dev(scale),
                                                                                           residual=dev(res))
    torch.testing.assert_close(out, direct, atol=2e-5, rtol=1e-5)


def test_conv_winograd_c_abi():
    """dv_conv3d_wino_* straight through ctypes (the drop-in boundary), against torch's fp64 convolution."""
    from diffuvolume_amd import _lib
    lib = _lib.load()
    g = _gen(32, "abi")
    x, w = torch.randn(1, 8, 5, 6, 20, generator=g), torch.randn(16, 8, 3, 3, 3, generator=g) * 0.1
    xd, wd = dev(x), dev(w)
    n = lib.dv_conv3d_wino_packed_floats(8, 16)
    assert n == 2 * 1 * 6144
    wp = torch.empty(n, device=DEV)
    out = torch.empty(1, 16, 5, 6, 20, device=DEV)
    assert lib.dv_conv3d_wino_pack_weights_f32(wd.data_ptr(), wp.data_ptr(), 8, 16, _lib.stream_ptr()) == 0
    rc = lib.dv_conv3d_wino_f32(xd.data_ptr(), wp.data_ptr(), None, None, None, None, out.data_ptr(), 1, 8, 5, 6, 20, 16,
                                S.ACT_NONE, _lib.stream_ptr())
    assert rc == 0
    ref = torch.nn.functional.conv3d(x.double(), w.double(), None, 1, 1).float()
    assert rel_err(out, ref) < 1e-5
    # argument checks: null pointers and empty shapes are refused, nothing is launched
    assert lib.dv_conv3d_wino_f32(None, wp.data_ptr(), None, None, None, None, out.data_ptr(), 1, 8, 5, 6, 20, 16, 0,
                                  _lib.stream_ptr()) != 0
    assert lib.dv_conv3d_wino_f32(xd.data_ptr(), wp.data_ptr(), None, None, None, None, out.data_ptr(), 0, 8, 5, 6, 20, 16,
                                  0, _lib.stream_ptr()) != 0


@pytest.mark.parametrize("precision", PRECISIONS)
@pytest.mark.parametrize("act", [S.ACT_MISH, S.ACT_LEAKY])
def test_conv_activations(act, precision):
    g = _gen(8, "act")
    x = torch.randn(1, 16, 4, 6, 16, generator=g)
    w = torch.randn(16, 16, 3, 3, 3, generator=g) * 0.1
    y = torch.nn.functional.conv3d(x, w, None, 1, 1)
    ref = y * torch.tanh(torch.nn.functional.softplus(y)) if act == S.ACT_MISH else torch.nn.functional.leaky_relu(y, 0.01)
    out = S.Conv3dPlan(dev(w), None, stride=1, act=act, precision=precision)(dev(x))
    torch.testing.assert_close(out.cpu(), ref, atol=2e-5, rtol=1e-4)


def test_deconv_golden():
    g = load_golden("layer_deconv")
    m = torch.nn.Sequential(torch.nn.ConvTranspose3d(16, 8, 3, padding=1, output_padding=1, stride=2, bias=False),
                            torch.nn.BatchNorm3d(8))
    sd = synth_state_dict(m.state_dict(), seed=g["seed"])
    plan = S.Deconv3dPlan(dev(sd["0.weight"]), tuple(dev(t) for t in _bn_tuple(sd, "1")), act=S.ACT_NONE)
    y = plan(dev(g["x"]))
    assert y.shape == g["y"].shape and rel_err(y, g["y"]) < 1e-5


@pytest.mark.parametrize("cfg", [(128, 64, (1, 3, 8, 12)), (64, 32, (2, 4, 6, 20)), (16, 8, (1, 2, 3, 7))])
def test_deconv_oracle(cfg):
    cin, cout, dims = cfg
    g = _gen(9, str(cfg))
    x = torch.randn(dims[0], cin, *dims[1:], generator=g)
    w = torch.randn(cin, cout, 3, 3, 3, generator=g) * (2.0 / (27 * cin)) ** 0.5
    bn = (torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g) * 0.1,
          torch.randn(cout, generator=g) * 0.1, torch.rand(cout, generator=g) + 0.5)
    up = torch.nn.functional.conv_transpose3d(x, w, None, 2, 1, 1)
    y = torch.nn.functional.batch_norm(up, bn[2], bn[3], bn[0], bn[1], False, 0.0, 1e-5)
    res = torch.randn(y.shape, generator=g)
    plan = S.Deconv3dPlan(dev(w), tuple(dev(t) for t in bn), act=S.ACT_RELU)
    assert rel_err(plan(dev(x), residual=dev(res)), torch.relu(y + res)) < 1e-5


@pytest.mark.parametrize("tag", ["nopad", "pad", "padw"])
def test_window_attention_golden(tag):
    from diffuvolume_amd.acv_ddim import _WindowAttention
    g = load_golden(f"layer_attention_{tag}")
    sd = synth_state_dict(_WindowAttention(128, 16).state_dict(), seed=g["seed"])
    y = S.window_attention(dev(g["x"]), dev(sd["qkv_3d.weight"]), dev(sd["qkv_3d.bias"]),
                           dev(sd["final1x1.weight"]), dev(sd["final1x1.bias"]), heads=16)
    assert rel_err(y, g["y"]) < 2e-5


def test_hourglass_golden():
    from diffuvolume_amd.acv_ddim import Hourglass, _HourglassPlan
    g = load_golden("layer_hourglass")
    hg = Hourglass(32)
    hg.load_state_dict(synth_state_dict(hg.state_dict(), seed=g["seed"]))
    hg = hg.to(DEV).eval()
    with torch.no_grad():
        y = _HourglassPlan(hg)(dev(g["x"]))
    assert rel_err(y, g["y"]) < 2e-5


# ---------------------------------------------------------------- diffusion loop
@pytest.fixture(scope="module")
def model(acv_state_dict):
    from diffuvolume_amd import ACVNet_DDIM
    m = ACVNet_DDIM(192, False, False)
    m.load_state_dict(acv_state_dict, strict=True)
    return m.to(DEV).eval()


def _volume(seed, b=1, h=16, w=32):
    return torch.rand(b, 64, 48, h, w, generator=_gen(seed, "vol"))


def test_encoder_golden(model):
    g = load_golden("encoder_schedule")
    assert torch.equal(model.encode_disparity(dev(g["disp_q"])).cpu(), g["x_T"])
    assert model._time_pairs() == [(999, 799), (799, 599), (599, 399), (399, 199), (199, -1)]
    # float64 schedule recomputed on this host: libm cos() may differ in the last bit between CPUs
    torch.testing.assert_close(model.alphas_cumprod.cpu(), g["alphas_cumprod"], rtol=1e-13, atol=0)


def test_time_shift_golden(model):
    g = load_golden("time_shift")
    torch.testing.assert_close(model.time_embedding.shift(dev(g["t"])).cpu(), g["shift"], atol=1e-6, rtol=1e-5)


def _f64_state_dict(sd):
    """Conv / attention weights in float64 (the time MLP stays fp32: its output is an input here)."""
    return {k: (v.double() if v.is_floating_point() and not k.startswith("time_embedding") else v)
            for k, v in sd.items()}


def test_aggregation_cost_parity(model, acv_state_dict):
    """Per-kernel bar: the whole 26-layer aggregation stack stays within 1e-5 (relative to the
    cost scale) of the fp32 oracle, and is as close to a float64 evaluation as the oracle is."""
    g = load_golden("model_predictions")
    vol = _volume(g["vol_seed"])
    orc, orc64 = O.ACVDiffusionOracle(acv_state_dict), O.ACVDiffusionOracle(_f64_state_dict(acv_state_dict))
    n01 = orc.noise_to_filter(g["x_T"], g["t"])
    c32 = orc.aggregate(vol * n01.unsqueeze(1))
    c64 = orc64.aggregate(vol.double() * n01.unsqueeze(1).double())
    with torch.no_grad():
        _, n01f = model._filter(dev(g["x_T"]), dev(g["t"]))
        torch.testing.assert_close(n01f.cpu(), n01, atol=1e-6, rtol=0)   # time MLP runs on the GPU here
        ch = model._aggregate(dev(vol), dev(n01)).cpu()
    assert rel_err(ch, c32) < 1e-5
    e_hip = float((ch.double() - c64).abs().mean())
    e_ref = float((c32.double() - c64).abs().mean())
    assert e_hip < 3 * e_ref, (e_hip, e_ref)


def test_model_predictions_golden(model, acv_state_dict):
    """One volume-filter step against the reference's own fp32 output AND against a float64 evaluation of the same step
    (oracle/acv_oracle.py in float64): north-star bars -- disparity within 1e-3 px, EPE within 1e-4.
    The soft-argmax amplifies cost error by the spread of the distribution, |d disp| <= unc * max|d cost|; with random
    weights the spread is large, and the reference's fp32 output is itself up to 9.5e-4 px from the float64 value.  So:
      * against float64 (the value both fp32 paths approximate): EVERY pixel within 1e-3 px, and a mean error no larger
        than 1.1 x the reference's own (measured: reference 6.8e-5 mean / 9.5e-4 max; HIP 6.8e-5 / 7.0e-4);
      * against the reference's fp32 output, i.e. between two fp32 evaluations whose errors can have opposite signs:
        mean < 2e-4, 99th percentile < 1e-3, no pixel beyond 2e-3, at most 0.25 % of the pixels beyond 1e-3 (measured
        8.6e-5 / 5.7e-4 / 1.5e-3 / 0.11 %; with the in-plane Winograd kernel in every layer 8.4e-5 / 5.4e-4 / 1.4e-3 /
        0.085 % -- the F(2x2x2) kernel of the 64- and 128-channel layers is the more accurate one against float64,
        tests/test_gpu_wino3.py, but lands on the other side of the reference on two more of the 8 192 pixels)."""
    g = load_golden("model_predictions")
    pn, xs, pred, handle = model.model_predictions(dev(_volume(g["vol_seed"])), dev(g["x_T"]), dev(g["t"]))
    assert pn.dtype == torch.float64 and xs.dtype == torch.float32
    d = (pred.cpu() - g["pred"]).abs()
    assert float(d.mean()) < 2e-4 and float(d.max()) < 2e-3 and float((d > 1e-3).float().mean()) <= 2.5e-3, \
        (float(d.mean()), float(d.max()), float((d > 1e-3).float().mean()))
    orc64 = O.ACVDiffusionOracle(_f64_state_dict(acv_state_dict))
    _, _, d64, _ = orc64.model_predictions(_volume(g["vol_seed"]).double(), g["x_T"], g["t"])
    e_hip, e_ref = (pred.cpu().double() - d64).abs(), (g["pred"].double() - d64).abs()
    assert float(e_ref.max()) < 1e-3                      # (the fixture: the reference itself is inside the bar)
    assert float(e_hip.max()) < 1e-3, float(e_hip.max())
    assert float(e_hip.mean()) <= 1.1 * float(e_ref.mean()), (float(e_hip.mean()), float(e_ref.mean()))
    gt = g["used0"].reshape(g["pred"].shape)             # EPE against the fixture's origin disparity
    assert abs(float((pred.cpu() - gt).abs().mean()) - float((g["pred"] - gt).abs().mean())) < 1e-4
    p99 = float(d.flatten().quantile(0.99))
    assert p99 < 1e-3, p99
    du = (handle.uncertainty.cpu() - g["unc"]).abs()
    assert float(du.mean()) < 1e-3
    # the two-hot re-encoding moves with the disparity (weights) and flips bins only where floor() flips
    same = ((xs.cpu() - g["x_start"]).abs() < 1e-3).all(dim=1)
    assert float(same.float().mean()) > 0.99
    sel = same.unsqueeze(1).expand_as(pn)
    torch.testing.assert_close(pn.cpu()[sel], g["pred_noise"][sel], atol=1e-6, rtol=0)


def _assert_loop_contract(model, sd, vol, used, x_T, seed, gt=None, traj_epe=1e-3, traj_mean=1e-2):
    """The north-star bars on the 5-step loop (|d disp| <= 1e-3 px on 99.9 % of the pixels, |EPE_hip - EPE_oracle|
    < 1e-4 px), asserted where they are well defined (oracle/loop_parity.py): every step from the oracle's own state
    (teacher forced) and HIP's own state under the oracle's renewal decisions (decision forced); a free-run step may
    leave the bar only after a renewal decision has come out differently.  The pixel bar is 1e-3 px wherever the
    reference itself is confident (uncertainty < 3 px, acv_ddim.py:330) and scales with the spread of the
    distribution elsewhere (`frac_gt_bar`, loop_parity._stats): with these untrained weights the soft-argmax sits on
    a ~50 px wide distribution and amplifies the last bit of the fp32 cost 17x more than a trained network does
    (measured split: profiles/attic/diag/diag_split.py).  EPE is asserted unscaled.  Returns the report."""
    from oracle import loop_parity as LP
    gt = used if gt is None else gt                     # fixtures without ground truth: EPE against `used`
    orc = O.ACVDiffusionOracle(sd)
    final_o, stack_o, trace = LP.oracle_trajectory(orc, vol, used, x_T, seed)
    vol_d, used_d = dev(vol), dev(used)
    npx = stack_o[0].numel()
    bar = max(LP.BAR_FRAC, 1.0 / npx)                  # "99.9 %" of a fixture smaller than 1000 px means one pixel
    tf = LP.teacher_forced(model, trace, vol_d, used_d, used, gt)
    df = LP.decision_forced(model, trace, vol_d, used_d, x_T, gt)
    fr = LP.free_run(model, trace, stack_o, final_o, vol_d, used_d, x_T, gt, seed)
    for s in tf:                                       # per-step function parity: pixel bar + EPE bar
        assert s["frac_gt_bar"] <= bar, s
        assert s["epe_delta"] < LP.BAR_EPE, s
        if "x_next_mean_abs_where_decisions_agree" in s:
            assert s["x_next_mean_abs_where_decisions_agree"] < 1e-4, s
    # trajectory parity: on its own state HIP's 1e-4-px differences re-enter the next step through the two-hot
    # weights, and for the flat distributions of untrained weights the step map is expansive (measured here: the share
    # of pixels beyond 1e-3 px grows 1 % -> 9 % over five steps with ZERO decision flips).  The contract's EPE bar is
    # asserted on the trajectory at the BASELINE size, where it is a statement about 491 520 pixels
    # (tests/test_gpu_fullsize.py::test_fullsize_oracle_5step: |dEPE| <= 5e-5 at every step, free run included); on
    # these 8 192-pixel fixtures the trajectory is held to a divergence bound only (`traj_epe`, `traj_mean`)
    flips = sum(s["flips_mask_zero"] for s in fr["steps"])
    for s in df + (fr["steps"] if flips == 0 else []):
        assert s["epe_delta"] < traj_epe, s
        assert s["mean_abs_px"] < traj_mean, s
    if flips == 0:
        assert fr["final"]["epe_delta"] < traj_epe, fr["final"]
    return {"teacher_forced": tf, "decision_forced": df, "free_run": fr, "flips": flips}


def _teacher_forced_vs_fp64(model, sd, vol, used, x_T, seed):
    """Per step, from the state of a float64 run of the oracle: mean |d disp| of (HIP, fp32 oracle) against the
    float64 step.  Separates arithmetic error from the loop's decision chaos."""
    from oracle import loop_parity as LP
    orc, orc64 = O.ACVDiffusionOracle(sd), O.ACVDiffusionOracle(_f64_state_dict(sd))
    _, _, trace = LP.oracle_trajectory(orc64, vol.double(), used.double(), x_T, seed)
    vol_d, used_d = dev(vol), dev(used)
    out = []
    for i, r in enumerate(trace):
        mask = dev(r["mask_in"]).clone()
        eps = None if r["eps"] is None else dev(r["eps"])
        fill = None if r["fill"] is None else dev(r["fill"])
        disp_h = model.ddim_step(i, vol_d, used_d, dev(r["img"]), mask, None, eps, fill)[0].cpu()
        t = torch.full((vol.shape[0],), r["time"], dtype=torch.long)
        disp_o = orc.model_predictions(vol, r["img"], t)[2]
        out.append((float((disp_h.double() - r["disp"]).abs().mean()), float((disp_o.double() - r["disp"]).abs().mean())))
    return out


def test_ddim_sample_golden(model, acv_state_dict):
    """Five-step loop.  (1) against the reference's own outputs (golden stack): the median pixel within 1e-4 px at
    every step and step 1 -- before any decision has been fed back -- within the contract; (2) the contract bars on
    all five steps against the oracle (which test_oracle_golden.py pins to that same golden stack to 99.9 %);
    (3) arithmetic distance to a float64 evaluation, step by step from the same state: HIP is as close as the
    fp32 oracle."""
    g = load_golden("ddim_sample")
    vol = _volume(g["vol_seed"])
    final, stack = model.ddim_sample(dev(vol), dev(g["used"]), dev(g["x_T"]), noise=NoiseTape(g["tape_seed"]))
    assert stack.shape == g["stack"].shape
    assert torch.equal(stack[0].cpu(), g["stack"][0])
    d = (stack.cpu() - g["stack"]).abs()
    for i in range(1, 6):
        assert float(d[i].median()) < 1e-4, (i, float(d[i].median()))
    # (step 1 is test_model_predictions_golden's step: two fp32 evaluations, each within 1e-3 px of the float64 one)
    assert float(d[1].mean()) < 2e-4 and float(d[1].max()) < 2e-3 and float((d[1] > 1e-3).float().mean()) <= 2.5e-3, \
        (float(d[1].mean()), float(d[1].max()), float((d[1] > 1e-3).float().mean()))
    rep = _assert_loop_contract(model, acv_state_dict, vol, g["used"], g["x_T"], g["tape_seed"])
    print("ddim_sample fixture:", rep)
    for i, (e_h, e_o) in enumerate(_teacher_forced_vs_fp64(model, acv_state_dict, vol, g["used"], g["x_T"], g["tape_seed"])):
        assert e_h < 1.5 * e_o + 2e-5, (i + 1, e_h, e_o)


def test_ddim_sample_vs_oracle_batch2(model, acv_state_dict):
    """B=2 (different images per batch entry) with injected noise, same bars as above."""
    vol = _volume(32, b=2, h=8, w=16)
    used = torch.rand(2, 32, 64, generator=_gen(32, "u")) * 150 + 10
    orc = O.ACVDiffusionOracle(acv_state_dict)
    dq = torch.nn.functional.interpolate(used.unsqueeze(1), size=(8, 16), mode="bilinear") / 4
    x_T = orc.encode_x_T(dq)
    _assert_loop_contract(model, acv_state_dict, vol, used, x_T, 5)
    for i, (e_h, e_o) in enumerate(_teacher_forced_vs_fp64(model, acv_state_dict, vol, used, x_T, 5)):
        assert e_h < 1.5 * e_o + 2e-5, (i + 1, e_h, e_o)


def test_forward_golden(model, acv_state_dict):
    """ACVNet_DDIM.forward (eval) against the reference's output: median within 1e-4 px; the loop inside is then
    held to the contract bars on the very volume forward() built (feature CNN + attention branch + builders on
    HIP), so a difference in the front end cannot hide behind the loop's decision chaos."""
    g = load_golden("forward_eval")
    batch = synth_stereo_batch(1, 64, 128, seed=g["stereo_seed"], shifts=(8,))
    # forward() draws from the device RNG; inject the tape through ddim_sample
    tape = NoiseTape(g["tape_seed"])
    keep = model.ddim_sample
    seen = {}

    def spy(v, u, a, **kw):
        seen["vol"], seen["x_T"] = v, a
        return keep(v, u, a, noise=tape)

    model.ddim_sample = spy
    try:
        pred = model(dev(batch["left"]), dev(batch["right"]), dev(batch["used"]), dev(batch["disp"]), None)[0]
    finally:
        del model.ddim_sample
    d = (pred.cpu() - g["pred"]).abs()
    assert float(d.median()) < 1e-4, float(d.median())
    vol = seen["vol"].tensor() if hasattr(seen["vol"], "tensor") else seen["vol"]     # forward() passes the factor handle
    rep = _assert_loop_contract(model, acv_state_dict, vol.cpu(), batch["used"], seen["x_T"].cpu(),
                                g["tape_seed"], gt=batch["gt"])
    if rep["flips"] == 0:                      # no decision differs: the whole forward meets the contract vs the reference
        assert float((d > 1e-3).float().mean()) <= 1e-3
        gt, m = batch["gt"], (batch["gt"] > 0) & (batch["gt"] < 192)
        assert abs(float((pred.cpu() - gt).abs()[m].mean()) - float((g["pred"] - gt).abs()[m].mean())) < 1e-4


# ---------------------------------------------------------------- metrics
def test_metrics_golden():
    from diffuvolume_amd import metrics as M
    g = load_golden("metrics")
    out = M.batch_metrics(dev(g["est"]), dev(g["gt"]), dev(g["mask"]))
    for k in M.NAMES:
        assert abs(float(out[k]) - g[k]) < 1e-6, k
    assert abs(float(M.EPE_metric(dev(g["est"]), dev(g["gt"]), dev(g["mask"]))) - g["EPE"]) < 1e-6
    g = load_golden("metrics_all_skipped")
    assert float(M.batch_metrics(dev(g["est"]), dev(g["gt"]), dev(g["mask"]))["EPE"]) == 0.0


# ---------------------------------------------------------------- IGEV geometry lookup
def test_igev_geo_filter_lookup_golden():
    from diffuvolume_amd.geometry_ddim import Combined_Geo_Encoding_Volume
    g = load_golden("igev_geo_lookup")
    fn = Combined_Geo_Encoding_Volume(dev(g["f1"]), dev(g["f2"]), dev(g["geo"]), num_levels=2, radius=4)
    out = fn(dev(g["disp"]), dev(g["coords"]), dev(g["noisy"]))
    assert out.shape == g["out"].shape
    torch.testing.assert_close(out.cpu(), g["out"], atol=2e-5, rtol=1e-5)   # the corr GEMM sums channels in another order


def test_igev_geo_filter_lookup_oracle_kitti_size():
    from diffuvolume_amd.geometry_ddim import Combined_Geo_Encoding_Volume
    from oracle import igev_oracle as IO
    b, c, d, h, w = 1, 8, 48, 12, 78                      # 1/4 of a KITTI row band; W odd/4
    gen = _gen(62, "igev")
    geo = torch.randn(b, c, d, h, w, generator=gen)
    f1, f2 = torch.randn(b, 24, h, w, generator=gen), torch.randn(b, 24, h, w, generator=gen)
    disp = torch.rand(b, 1, h, w, generator=gen) * 47
    coords = torch.arange(w, dtype=torch.float32).view(1, 1, 1, w).expand(b, 1, h, w).contiguous()
    noisy = torch.rand(b, d, h, w, generator=gen)
    ref = IO.geo_filter_lookup(geo, f1, f2, disp, coords, noisy)
    out = Combined_Geo_Encoding_Volume(dev(f1), dev(f2), dev(geo))(dev(disp), dev(coords), dev(noisy))
    torch.testing.assert_close(out.cpu(), ref, atol=3e-5, rtol=1e-5)


@pytest.mark.parametrize("d", [48, 13])
def test_igev_geo_filter_lookup_window_edges(d):
    """The lookup copies the 24 disparity entries around a pixel's d into a private window (csrc/geo_lookup.hip): exact
    integers (where the reference's float sample position may land on either side of the integer), disparities at and
    beyond both ends of the range, and an odd number of disparities (one entry past the pooled pairs)."""
    from diffuvolume_amd.geometry_ddim import Combined_Geo_Encoding_Volume
    from oracle import igev_oracle as IO
    b, c, h, w = 1, 8, 6, 40
    gen = _gen(64, f"edges{d}")
    geo = torch.randn(b, c, d, h, w, generator=gen)
    f1, f2 = torch.randn(b, 16, h, w, generator=gen), torch.randn(b, 16, h, w, generator=gen)
    disp = torch.rand(b, 1, h, w, generator=gen) * (d + 12) - 6                       # -6 .. d + 6
    disp[:, :, 0] = torch.arange(w, dtype=torch.float32) - 4                           # integers, also out of range
    disp[:, :, 1] = torch.arange(w, dtype=torch.float32) * 0.5 + 1e-6
    disp[:, :, 2] = torch.nextafter(torch.arange(w, dtype=torch.float32), torch.tensor(-1.0))   # just below integers
    coords = torch.arange(w, dtype=torch.float32).view(1, 1, 1, w).expand(b, 1, h, w).contiguous()
    noisy = torch.rand(b, d, h, w, generator=gen)
    ref = IO.geo_filter_lookup(geo, f1, f2, disp, coords, noisy)
    out = Combined_Geo_Encoding_Volume(dev(f1), dev(f2), dev(geo))(dev(disp), dev(coords), dev(noisy))
    torch.testing.assert_close(out.cpu(), ref, atol=3e-5, rtol=1e-5)


@pytest.mark.parametrize("cfg", [(2, 12, 78, 48), (1, 6, 40, 13), (3, 5, 21, 48)])
def test_igev_geo_lookup_fused_with_its_1x1_convolution(cfg):
    """`dv_geo_filter_lookup_conv1x1_f32` (csrc/geo_lookup.hip): lookup + BasicMotionEncoder.convc1 + bias + ReLU in one
    kernel (KITTI15/core/update.py:79,:89) against the 1x1 convolution of the materialised lookup in float64; ragged pixel
    counts (partial waves / blocks), disparities beyond both ends, an odd D; the result of an item does not depend on its batch."""
    from diffuvolume_amd.geometry_ddim import Combined_Geo_Encoding_Volume, pack_lookup_conv1x1
    from diffuvolume_amd.submodule import ACT_RELU
    b, h, w, d = cfg
    c = 8
    gen = _gen(65, f"fusedlookup{cfg}")
    geo = torch.randn(b, c, d, h, w, generator=gen)
    f1, f2 = torch.randn(b, 16, h, w, generator=gen), torch.randn(b, 16, h, w, generator=gen)
    disp = torch.rand(b, 1, h, w, generator=gen) * (d + 12) - 6
    coords = torch.arange(w, dtype=torch.float32).view(1, 1, 1, w).expand(b, 1, h, w).contiguous()
    noisy = torch.rand(b, d, h, w, generator=gen)
    wt = torch.randn(64, 162, 1, 1, generator=gen) * 0.1
    bias = torch.randn(64, generator=gen) * 0.1
    fn = Combined_Geo_Encoding_Volume(dev(f1), dev(f2), dev(geo))
    look = fn(dev(disp), dev(coords), dev(noisy))
    ref = torch.relu(torch.nn.functional.conv2d(look.double(), dev(wt).double(), dev(bias).double()))
    wp = pack_lookup_conv1x1(dev(wt), c)
    out = fn.request(dev(disp), dev(coords), dev(noisy)).conv1x1(wp, dev(bias), ACT_RELU)
    assert out.shape == ref.shape
    assert float((out.double() - ref).abs().max()) <= 2e-5 * max(1.0, float(ref.abs().max()))
    assert torch.equal(fn.request(dev(disp), dev(coords), dev(noisy)).materialize(), look)
    if b > 1:                                              # shard invariance: item 1 alone
        fn1 = Combined_Geo_Encoding_Volume(dev(f1[1:2]), dev(f2[1:2]), dev(geo[1:2]))
        one = fn1.lookup_conv1x1(dev(disp[1:2]), dev(coords[1:2]), dev(noisy[1:2].contiguous()), wp, dev(bias), ACT_RELU)
        assert torch.equal(one, out[1:2])


@pytest.mark.parametrize("shape", [(1, 24, 3, 78, 78), (2, 96, 2, 40, 40), (1, 7, 2, 17, 33), (1, 130, 1, 16, 21),
                                   (1, 4, 1, 5, 2)])
def test_igev_allpairs_corr_oracle(shape):
    """corr() and its pooled level (geometry_ddim.py:72-80, :28-30) on the MFMA kernel vs the oracle's einsum:
    ragged W1 / W2 (partial 16-wide tiles, odd W2: the pooled level drops the last column), C not a multiple of 4."""
    from diffuvolume_amd.geometry_ddim import Combined_Geo_Encoding_Volume as G
    from oracle import igev_oracle as IO
    b, c, h, w1, w2 = shape
    gen = _gen(63, str(shape))
    f1, f2 = torch.randn(b, c, h, w1, generator=gen), torch.randn(b, c, h, w2, generator=gen)
    ref = IO.all_pairs_corr(f1.double(), f2.double())                          # [b,h,w1,w2]
    ref1 = torch.nn.functional.avg_pool2d(ref.reshape(b * h * w1, 1, 1, w2), [1, 2], stride=[1, 2]).reshape(b, h, w1, w2 // 2)
    c0, c1 = G._corr_levels(dev(f1), dev(f2))
    assert c0.shape == (b, h, w1, w2) and c1.shape == (b, h, w1, w2 // 2)
    bar = 2e-6 * float(ref.abs().max())
    assert float((c0.cpu().double() - ref).abs().max()) <= bar
    assert float((c1.cpu().double() - ref1).abs().max()) <= bar
    full = G.corr(dev(f1), dev(f2))                                            # the reference's static method
    assert full.shape == (b, h, w1, 1, w2) and torch.equal(full.reshape(b, h, w1, w2), c0)
    # pooled level formed from the accumulators == pooling the stored level
    pooled = torch.nn.functional.avg_pool2d(c0.reshape(b * h * w1, 1, 1, w2), [1, 2], stride=[1, 2]).reshape(b, h, w1, w2 // 2)
    assert torch.equal(pooled, c1)


# ---------------------------------------------------------------- split-fp16 (hi/lo) MFMA convolution
@pytest.mark.parametrize("cfg", [(32, 32, (1, 6, 8, 48)), (64, 32, (2, 5, 7, 44)), (40, 32, (1, 4, 4, 32)),
                                 (32, 16, (1, 3, 9, 30)), (8, 32, (1, 2, 4, 96))])
def test_conv_f16x3_oracle(cfg):
    """x*w ~= hi*hi' + hi*lo' + lo*hi' on the fp16 matrix cores must stay at the fp32 bar (1e-5)."""
    cin, cout, dims = cfg
    g = _gen(17, str(cfg))
    x = torch.randn(dims[0], cin, *dims[1:], generator=g) * 3
    w = torch.randn(cout, cin, 3, 3, 3, generator=g) * (2.0 / (27 * cin)) ** 0.5
    bn = (torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g) * 0.1,
          torch.randn(cout, generator=g) * 0.1, torch.rand(cout, generator=g) + 0.5)
    scale = torch.rand(dims[0], *dims[1:], generator=g)
    res = torch.randn(dims[0], cout, *dims[1:], generator=g)
    y64 = torch.nn.functional.batch_norm(
        torch.nn.functional.conv3d(x.double() * scale.unsqueeze(1).double(), w.double(), None, 1, 1),
        bn[2].double(), bn[3].double(), bn[0].double(), bn[1].double(), False, 0.0, 1e-5)
    ref = torch.relu(y64 + res.double())
    bnd = tuple(dev(t) for t in bn)
    out16 = S.Conv3dPlan(dev(w), bnd, stride=1, act=S.ACT_RELU, precision="f16x3")(dev(x), in_scale=dev(scale), residual=dev(res))
    out32 = S.Conv3dPlan(dev(w), bnd, stride=1, act=S.ACT_RELU, precision="f32")(dev(x), in_scale=dev(scale), residual=dev(res))
    e16, e32 = rel_err(out16, ref), rel_err(out32, ref)
    assert e16 < 1e-5, (e16, e32)
    assert e16 < 3 * e32 + 1e-7, (e16, e32)      # as close to float64 as the exact-fp32 MFMA kernel


@pytest.mark.skipif(__import__("os").environ.get("DV_FULL_PARITY") != "1",
                    reason="the opt-in split-fp16 path (not the contract's arithmetic; its layer test above stays): DV_FULL_PARITY=1")
def test_ddim_loop_with_split_fp16_convs(acv_state_dict):
    """The whole 5-step loop with the 3x3x3 stride-1 convs on the split-fp16 MFMA kernel: same bars as
    the exact-fp32 build (median within 1e-4 px of the reference, distance to float64 comparable)."""
    from diffuvolume_amd import ACVNet_DDIM
    S.set_default_conv_precision("f16x3")
    try:
        m = ACVNet_DDIM(192, False, False)
        m.load_state_dict(acv_state_dict, strict=True)
        m = m.to(DEV).eval()
        m.prepare()
        assert m._plans.dres0.b.split and m._plans.dres1.a.split and not m._plans.dres2.conv1.split
    finally:
        S.set_default_conv_precision(None)
    g = load_golden("ddim_sample")
    vol = _volume(g["vol_seed"])
    final, stack = m.ddim_sample(dev(vol), dev(g["used"]), dev(g["x_T"]), noise=NoiseTape(g["tape_seed"]))
    d = (stack.cpu() - g["stack"]).abs()
    for i in range(1, 6):
        assert float(d[i].median()) < 1e-4, (i, float(d[i].median()))
    assert float(d[1].mean()) < 2e-4 and float((d[1] > 1e-3).float().mean()) < 1e-2     # opt-in mode, step 1 vs golden
    _assert_loop_contract(m, acv_state_dict, vol, g["used"], g["x_T"], g["tape_seed"])
    for i, (e_h, e_o) in enumerate(_teacher_forced_vs_fp64(m, acv_state_dict, vol, g["used"], g["x_T"], g["tape_seed"])):
        assert e_h < 2.5 * e_o + 2e-5, (i + 1, e_h, e_o)


def test_origin_acvnet_forward_golden():
    """SceneFlow/models/acv.py eval forward (the network that supplies `used`) on the same HIP kernels."""
    from diffuvolume_amd import ACVNet
    g = load_golden("acv_origin_forward")
    m = ACVNet(192, False, False)
    m.load_state_dict(synth_state_dict(m.state_dict(), seed=3, logit_gain=8.0), strict=True)
    m = m.to(DEV).eval()
    batch = synth_stereo_batch(2, 64, 128, seed=g["stereo_seed"], shifts=(8, 20))
    pred = m(dev(batch["left"]), dev(batch["right"]))[-1]
    d = (pred.cpu() - g["pred"]).abs()
    assert pred.shape == g["pred"].shape
    assert float(d.median()) < 1e-4 and float(d.mean()) < 1e-3, (float(d.median()), float(d.mean()))


def test_origin_acvnet_attention_only_golden():
    """ACVNet(192, attn_weights_only=True): the regression of the attention logits (acv.py:246-252), same state_dict."""
    from diffuvolume_amd import ACVNet
    g = load_golden("acv_origin_forward")
    m = ACVNet(192, True, False)
    m.load_state_dict(synth_state_dict(m.state_dict(), seed=3, logit_gain=8.0), strict=True)
    m = m.to(DEV).eval()
    batch = synth_stereo_batch(2, 64, 128, seed=g["stereo_seed"], shifts=(8, 20))
    out = m(dev(batch["left"]), dev(batch["right"]))
    assert len(out) == 1 and out[0].shape == g["pred_attention"].shape
    d = (out[0].cpu() - g["pred_attention"]).abs()
    assert float(d.median()) < 1e-4 and float(d.mean()) < 1e-3, (float(d.median()), float(d.mean()))
    assert float((out[0].cpu() - g["pred"]).abs().mean()) > 1e-2          # and it is not the full network's output


def test_masked_x_T_golden(model):
    """`mask_gt` given (acv_ddim.py:415-417): x_T bit for bit against what the reference's forward hands to ddim_sample,
    and forward() passes the mask through."""
    g = load_golden("acv_xT_masked")
    x = model.encode_disparity(dev(g["disp"]), dev(g["mask_gt"]))
    assert torch.equal(x.cpu(), g["x_T_masked"])
    assert torch.equal(model.encode_disparity(dev(g["disp"])).cpu(), g["x_T"])
    assert torch.equal(model.encode_disparity(dev(g["disp"]), g["mask_gt"].bool()).cpu(), g["x_T_masked"])    # bool mask, host tensor
    batch = synth_stereo_batch(2, 64, 128, seed=43, shifts=(8, 20))
    keep, seen = model.ddim_sample, {}

    def spy(v, u, a, **kw):
        seen["x_T"] = a
        return u, None

    model.ddim_sample = spy
    try:
        model(dev(batch["left"]), dev(batch["right"]), dev(batch["used"]), dev(g["disp"]), dev(g["mask_gt"]))
    finally:
        del model.ddim_sample
    assert keep is not None and torch.equal(seen["x_T"].cpu(), g["x_T_masked"])


def test_split_fp16_range_guard():
    """Activations beyond the fp16 range must not pass silently: the kernel raises a device flag and the
    wrapper turns it into an error (no quiet garbage, no quiet fallback)."""
    from diffuvolume_amd import DiffuVolumeError
    w = torch.randn(32, 32, 3, 3, 3) * 0.05
    plan = S.Conv3dPlan(dev(w), None, 1, S.ACT_NONE, precision="f16x3")
    x = torch.randn(1, 32, 2, 4, 48)
    S.split_overflow_flag(DEV).zero_()
    plan(dev(x))
    S.check_split_overflow(DEV)                       # in range: no error
    x[0, 3, 1, 2, 7] = 1e6
    plan(dev(x))
    with pytest.raises(DiffuVolumeError):
        S.check_split_overflow(DEV)
    S.check_split_overflow(DEV)                       # flag was cleared


@pytest.mark.parametrize("cfg", [(64, 64, (1, 4, 8, 24)), (128, 128, (1, 4, 4, 12)), (16, 48, (1, 3, 5, 20))])
def test_conv_f16x3_wide_layers(cfg):
    """Cout > 32: the grid splits the output channels into 32-wide slices."""
    cin, cout, dims = cfg
    g = _gen(19, str(cfg))
    x = torch.randn(dims[0], cin, *dims[1:], generator=g)
    w = torch.randn(cout, cin, 3, 3, 3, generator=g) * (2.0 / (27 * cin)) ** 0.5
    ref = torch.nn.functional.conv3d(x.double(), w.double(), None, 1, 1)
    out = S.Conv3dPlan(dev(w), None, 1, S.ACT_NONE, precision="f16x3")(dev(x))
    assert rel_err(out, ref) < 2e-6


@pytest.mark.parametrize("cfg", [(48, 32, (1, 3, 4, 10)), (32, 16, (2, 2, 5, 13)), (16, 8, (1, 4, 6, 32))])
def test_deconv_k4_oracle(cfg):
    """ConvTranspose3d(4, stride 2, padding 1) + BN + LeakyReLU (IGEV hourglass, igev_stereo_ddim.py:44-51)."""
    cin, cout, dims = cfg
    g = _gen(21, str(cfg))
    x = torch.randn(dims[0], cin, *dims[1:], generator=g)
    w = torch.randn(cin, cout, 4, 4, 4, generator=g) * (2.0 / (64 * cin)) ** 0.5
    bn = (torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g) * 0.1,
          torch.randn(cout, generator=g) * 0.1, torch.rand(cout, generator=g) + 0.5)
    up = torch.nn.functional.conv_transpose3d(x, w, None, 2, 1)
    y = torch.nn.functional.leaky_relu(torch.nn.functional.batch_norm(up, bn[2], bn[3], bn[0], bn[1], False, 0.0, 1e-5), 0.01)
    plan = S.Deconv3dPlan(dev(w), tuple(dev(t) for t in bn), act=S.ACT_LEAKY)
    out = plan(dev(x))
    assert out.shape == y.shape and rel_err(out, y) < 1e-5


@pytest.mark.parametrize("cfg", [
    # cin, cout, k, dilation, (B, H, W), act, residual
    (115, 128, 3, 1, (1, 16, 96), "mish", False),
    (128, 128, 3, 2, (2, 11, 70), "mish", False),      # ragged rows and columns
    (128, 128, 3, 4, (1, 24, 128), "mish", False),
    (128, 96, 3, 8, (1, 24, 72), "mish", False),
    (96, 96, 3, 8, (1, 19, 64), "none", True),
    (64, 64, 3, 16, (1, 40, 130), "none", True),       # dilation larger than the tile, W % 4 != 0
    (128, 96, 1, 1, (2, 9, 68), "none", False),
    (32, 1, 3, 1, (1, 12, 80), "none", False),
    (20, 24, 3, 3, (1, 10, 33), "relu", False),
])
def test_conv2d_oracle(cfg):
    """convbn (+Mish, + BasicBlock residual) of the KITTI12 refinement stack (submodule.py:21-24, :192-215)."""
    cin, cout, k, dil, dims, act, use_res = cfg
    g = _gen(31, str(cfg))
    x = torch.randn(dims[0], cin, dims[1], dims[2], generator=g)
    w = torch.randn(cout, cin, k, k, generator=g) * (2.0 / (k * k * cin)) ** 0.5
    bn = (torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g) * 0.1,
          torch.randn(cout, generator=g) * 0.1, torch.rand(cout, generator=g) + 0.5)
    res = torch.randn(dims[0], cout, dims[1], dims[2], generator=g) if use_res else None
    y = torch.nn.functional.conv2d(x, w, None, 1, dil if k == 3 else 0, dil if k == 3 else 1)
    y = torch.nn.functional.batch_norm(y, bn[2], bn[3], bn[0], bn[1], False, 0.0, 1e-5)
    if res is not None:
        y = y + res
    y = {"mish": lambda t: t * torch.tanh(torch.nn.functional.softplus(t)), "relu": torch.relu, "none": lambda t: t}[act](y)
    plan = S.Conv2dPlan(dev(w), tuple(dev(t) for t in bn), dilation=dil,
                        act={"mish": S.ACT_MISH, "relu": S.ACT_RELU, "none": S.ACT_NONE}[act])
    out = plan(dev(x), residual=None if res is None else dev(res))
    assert out.shape == y.shape and rel_err(out, y) < 1e-5


def test_builder_edge_shapes():
    """Empty batch (the reference returns empty volumes), a single pixel, and maxdisp wider than the image."""
    e = S.build_gwc_volume(torch.zeros(0, 8, 2, 4, device=DEV), torch.zeros(0, 8, 2, 4, device=DEV), 3, 4)
    assert tuple(e.shape) == (0, 4, 3, 2, 4)
    assert tuple(S.build_concat_volume(torch.zeros(0, 8, 2, 4, device=DEV), torch.zeros(0, 8, 2, 4, device=DEV), 3).shape) == (0, 16, 3, 2, 4)
    assert tuple(S.disparity_regression(torch.zeros(0, 5, 2, 4, device=DEV), 5).shape) == (0, 2, 4)
    z = lambda *sh: torch.zeros(*sh, device=DEV)
    assert tuple(S.build_concat_attention_volume(z(0, 8, 2, 4), z(0, 8, 2, 4), z(0, 1, 3, 2, 4), 3).shape) == (0, 16, 3, 2, 4)
    lazy = S.build_concat_attention_volume(z(0, 8, 2, 4), z(0, 8, 2, 4), z(0, 1, 3, 2, 4), 3, lazy=True)
    assert tuple(lazy.shape) == (0, 16, 3, 2, 4) and tuple(lazy.tensor().shape) == (0, 16, 3, 2, 4)
    g = _gen(141, "edge")
    for shape, d, groups in (((1, 8, 1, 1), 4, 2), ((2, 16, 3, 5), 12, 4)):      # W < maxdisp: mostly zero wedge
        l, r = torch.randn(*shape, generator=g), torch.randn(*shape, generator=g)
        torch.testing.assert_close(S.build_gwc_volume(dev(l), dev(r), d, groups).cpu(), O.build_gwc_volume(l, r, d, groups),
                                   atol=1e-6, rtol=1e-6)
        assert torch.equal(S.build_concat_volume(dev(l), dev(r), d).cpu(), O.build_concat_volume(l, r, d))
        assert torch.equal(S.build_concat_volume(dev(l), dev(r), d, zero_left=True).cpu(),
                           O.build_concat_volume(l, r, d, zero_left=True))


@pytest.mark.parametrize("cfg", [
    # cin, cout, cskip, (B, D, H, W), act
    (64, 32, 32, (1, 4, 6, 32), "relu"),
    (128, 64, 64, (2, 3, 5, 20), "relu"),          # ragged x tile, two cout blocks
    (64, 32, 32, (1, 2, 3, 36), "mish"),
    (32, 32, 32, (1, 3, 4, 10), "relu"),           # W % 4 != 0 -> two-launch fallback inside the plan
    (16, 32, 32, (1, 2, 2, 8), "relu"),            # more skip chunks than input chunks -> fallback
])
def test_deconv_fused_redir(cfg):
    """F.relu(BN(ConvTranspose3d(x)) + BN(Conv3d_1x1x1(skip))) -- the hourglass tail (acv_ddim.py:81-92) in one launch."""
    cin, cout, cskip, dims, act = cfg
    g = _gen(51, str(cfg))
    x = torch.randn(dims[0], cin, *dims[1:], generator=g)
    skip = torch.randn(dims[0], cskip, *(2 * d for d in dims[1:]), generator=g)
    w = torch.randn(cin, cout, 3, 3, 3, generator=g) * (2.0 / (27 * cin)) ** 0.5
    wr = torch.randn(cout, cskip, 1, 1, 1, generator=g) * (1.0 / cskip) ** 0.5
    bn = lambda: (torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g) * 0.1,
                  torch.randn(cout, generator=g) * 0.1, torch.rand(cout, generator=g) + 0.5)
    bnd, bnr = bn(), bn()
    f = torch.nn.functional
    y = f.batch_norm(f.conv_transpose3d(x, w, None, 2, 1, 1), bnd[2], bnd[3], bnd[0], bnd[1], False, 0.0, 1e-5) \
        + f.batch_norm(f.conv3d(skip, wr), bnr[2], bnr[3], bnr[0], bnr[1], False, 0.0, 1e-5)
    y = torch.relu(y) if act == "relu" else y * torch.tanh(f.softplus(y))
    plan = S.Deconv3dPlan(dev(w), tuple(dev(t) for t in bnd), act=S.ACT_RELU if act == "relu" else S.ACT_MISH,
                          redir=(dev(wr), tuple(dev(t) for t in bnr)))
    out = plan(dev(x), skip=dev(skip))
    assert out.shape == y.shape and rel_err(out, y) < 1e-5


@pytest.mark.parametrize("cfg", [
    # cin, cout, k, (B, H, W), act, residual
    (3, 32, 3, (2, 32, 96), "relu", False),        # the first layer of the feature CNN (acv_ddim.py:19)
    (32, 64, 3, (1, 17, 70), "relu", False),       # odd sizes: output (H-1)/2+1
    (32, 64, 1, (1, 16, 130), "none", False),      # 1x1 stride-2 `downsample`
    (64, 128, 3, (1, 24, 64), "none", True),
])
def test_conv2d_stride2_oracle(cfg):
    cin, cout, k, dims, act, use_res = cfg
    g = _gen(61, str(cfg))
    x = torch.randn(dims[0], cin, dims[1], dims[2], generator=g)
    w = torch.randn(cout, cin, k, k, generator=g) * (2.0 / (k * k * cin)) ** 0.5
    bn = (torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g) * 0.1,
          torch.randn(cout, generator=g) * 0.1, torch.rand(cout, generator=g) + 0.5)
    y = torch.nn.functional.conv2d(x, w, None, 2, 1 if k == 3 else 0)
    y = torch.nn.functional.batch_norm(y, bn[2], bn[3], bn[0], bn[1], False, 0.0, 1e-5)
    res = torch.randn(y.shape, generator=g) if use_res else None
    if res is not None:
        y = y + res
    y = torch.relu(y) if act == "relu" else y
    plan = S.Conv2dPlan(dev(w), tuple(dev(t) for t in bn), act=S.ACT_RELU if act == "relu" else S.ACT_NONE, stride=2)
    out = plan(dev(x), residual=None if res is None else dev(res))
    assert out.shape == y.shape and rel_err(out, y) < 1e-5


def test_feature_cnn_matches_the_reference_class():
    """FeatureExtraction (acv_ddim.py:14-53) through the fused 2-D kernels against the output of the REFERENCE's
    `feature_extraction` on the same weights and image (tests/golden/feature_cnns.npz, oracle/make_golden_features.py)."""
    import diffuvolume_amd as dv
    g = load_golden("feature_cnns")
    fe = dv.ACVNet_DDIM(192, False, False).feature_extraction
    fe.load_state_dict(synth_state_dict(fe.state_dict(), seed=g["seed"]), strict=True)
    fe = fe.to(DEV).eval()
    with torch.no_grad():
        y = fe(dev(g["x"]))["gwc_feature"]
    assert y.shape == g["acv_gwc_feature"].shape == (1, 320, 8, 16)
    assert rel_err(y, g["acv_gwc_feature"]) < 2e-5


@pytest.mark.parametrize("shape", [(1, 40, 3, 20, 140), (2, 40, 2, 33, 50), (1, 40, 1, 128, 240), (2, 40, 2, 17, 12),
                                   (1, 40, 2, 16, 128)])
def test_patch_volume_vs_pytorch_depthwise(shape):
    """patch + patch_l1/l2/l3 (acv_ddim.py:181-188, :377-381) fused, vs the nn.Conv3d modules themselves: widths on the
    16-byte path (multiples of 4: one partial 128-column tile, the bench plane, a plane narrower than the halo, exactly
    one tile) and the element-wise path (W = 50)."""
    from diffuvolume_amd.submodule import patch_volume
    g = _gen(81, str(shape))
    x = torch.randn(*shape, generator=g)
    conv = lambda c, d: torch.nn.Conv3d(c, c, (1, 3, 3), 1, (0, d, d), d, groups=c, bias=False)
    patch, l1, l2, l3 = conv(40, 1), conv(8, 1), conv(16, 2), conv(16, 3)
    for m in (patch, l1, l2, l3):
        m.weight.data = torch.randn(m.weight.shape, generator=g) * 0.4
    with torch.no_grad():
        y = patch(x)
        ref = torch.cat((l1(y[:, :8]), l2(y[:, 8:24]), l3(y[:, 24:40])), dim=1)
    w1 = patch.weight.detach().reshape(40, 9)
    w2 = torch.cat([m.weight.detach().reshape(-1, 9) for m in (l1, l2, l3)])
    dil = torch.tensor([1] * 8 + [2] * 16 + [3] * 16, dtype=torch.int32)
    out = patch_volume(dev(x), dev(w1), dev(w2), dev(dil))
    torch.testing.assert_close(out.cpu(), ref, atol=2e-5, rtol=1e-5)
    # the element-wise kernel through the device-table entry point gives the same function
    out2 = torch.empty_like(out)
    from diffuvolume_amd import _lib
    xd, w1d, w2d, dd = dev(x), dev(w1), dev(w2), dev(dil)
    _lib.check(_lib.load().dv_patch_volume_f32(xd.data_ptr(), w1d.data_ptr(), w2d.data_ptr(), dd.data_ptr(), out2.data_ptr(),
                                               *shape, _lib.stream_ptr()), "dv_patch_volume_f32")
    torch.testing.assert_close(out2.cpu(), ref, atol=2e-5, rtol=1e-5)


def test_single_channel_head_is_stride1_only():
    """The Cout == 1 layer has its own packing / kernel (stride 1); other strides are refused, not mis-read."""
    from diffuvolume_amd._lib import DiffuVolumeError
    plan = S.Conv3dPlan(torch.randn(1, 8, 3, 3, 3, device=DEV), None, stride=2, act=S.ACT_NONE, precision="f32")
    with pytest.raises(DiffuVolumeError):
        plan(torch.randn(1, 8, 4, 4, 8, device=DEV))


@pytest.mark.parametrize("steps", [2, 20])
def test_ddim_sample_other_step_counts(acv_state_dict, steps):
    """sampling_timesteps is a constructor argument here (hard-coded 5 / 3 / 2 in the reference; BASELINE config 5
    asks for 20): time pairs, sigma/c coefficients and the ensemble weights follow the same rules for any S."""
    import diffuvolume_amd as dv
    cof = [0.5] + [0.0] * (steps - 1) + [0.5]
    m = dv.ACVNet_DDIM(192, False, False, sampling_timesteps=steps, ensemble_cof=cof)
    m.load_state_dict(acv_state_dict, strict=True)
    m = m.to(DEV).eval()
    vol = _volume(33, b=1, h=8, w=16)
    used = torch.rand(1, 32, 64, generator=_gen(33, "u")) * 150 + 10
    sd64 = _f64_state_dict(acv_state_dict)
    orc, orc64 = O.ACVDiffusionOracle(acv_state_dict, sampling_timesteps=steps, cof=cof), O.ACVDiffusionOracle(sd64, sampling_timesteps=steps, cof=cof)
    dq = torch.nn.functional.interpolate(used.unsqueeze(1), size=(8, 16), mode="bilinear") / 4
    x_T = orc.encode_x_T(dq)
    from oracle import loop_parity as LP
    final_o, stack_o, trace = LP.oracle_trajectory(orc, vol, used, x_T, 7)
    assert len(trace) == steps and stack_o.shape[0] == steps + 1
    bar = max(LP.BAR_FRAC, 1.0 / stack_o[0].numel())
    for s in LP.teacher_forced(m, trace, dev(vol), dev(used), used, used):
        assert s["frac_gt_bar"] <= bar and s["epe_delta"] < LP.BAR_EPE, s
    for s in LP.decision_forced(m, trace, dev(vol), dev(used), x_T, used):
        assert s["epe_delta"] < 1e-3 and s["mean_abs_px"] < 1e-2, s          # divergence bound (see _assert_loop_contract)
    with torch.no_grad():
        fh, sh = m.ddim_sample(dev(vol), dev(used), dev(x_T), noise=NoiseTape(7))
    assert sh.shape[0] == steps + 1
    assert float((sh[1].cpu() - stack_o[1]).abs().max()) < 2e-3          # before any decision is fed back


# ---------------------------------------------------------------- 2-D Winograd kernel (csrc/conv2d_wino.hip)
@pytest.mark.parametrize("cfg", [((8,), 32, 16, 16, S.ACT_NONE), ((5, 7), 20, 9, 21, S.ACT_TANH),
                                 ((32, 16, 8, 8), 48, 24, 40, S.ACT_SIGMOID), ((3,), 40, 5, 3, S.ACT_RELU),
                                 ((33,), 33, 17, 50, S.ACT_MISH), ((16,), 1, 20, 36, S.ACT_NONE)])
def test_conv2d_winograd(cfg, monkeypatch):
    """The 2-D Winograd kernel forced onto shapes with partial tiles (odd H / W, rows that are not 16-byte aligned,
    channel tails on both sides, 1..4 concatenated sources), with bias, residual, `mul` and the GRU blend, against
    torch's fp64 convolution and against the direct 2-D kernel."""
    cins, cout, h, w_, act = cfg
    monkeypatch.setattr(S.Conv2dPlan, "WINO_MIN_BLOCKS", 0)
    g = _gen(41, str(cfg))
    b = 2
    xs = [torch.randn(b, c, h, w_, generator=g) for c in cins]
    cin = sum(cins)
    w = torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (9 * cin)) ** 0.5
    bias = torch.randn(cout, generator=g) * 0.1
    res, m = torch.randn(b, cout, h, w_, generator=g), torch.randn(b, cout, h, w_, generator=g)
    z, hh = torch.rand(b, cout, h, w_, generator=g), torch.randn(b, cout, h, w_, generator=g)
    y = torch.nn.functional.conv2d(torch.cat(xs, 1).double(), w.double(), bias.double(), 1, 1) + res.double()
    y = {S.ACT_NONE: lambda t: t, S.ACT_RELU: torch.relu, S.ACT_TANH: torch.tanh, S.ACT_SIGMOID: torch.sigmoid,
         S.ACT_MISH: lambda t: t * torch.tanh(torch.nn.functional.softplus(t))}[act](y)
    ref = hh.double() + z.double() * (y * m.double() - hh.double())
    plan = S.Conv2dPlan(dev(w), None, act=act, bias=dev(bias))
    assert plan.wino_packed is not None
    out = plan([dev(t) for t in xs], residual=dev(res), mul=dev(m), blend=(dev(z), dev(hh)))
    torch.testing.assert_close(out.cpu().double(), ref, atol=2e-5, rtol=1e-5)
    plan.wino_packed = None                       # the same plan on the direct kernel
    direct = plan([dev(t) for t in xs], residual=dev(res), mul=dev(m), blend=(dev(z), dev(hh)))
    torch.testing.assert_close(out, direct, atol=2e-5, rtol=1e-5)


@pytest.mark.parametrize("cfg", [((8,), 32, 2, 16, 16, S.ACT_NONE), ((5, 7), 20, 3, 19, 41, S.ACT_TANH),
                                 ((16, 8), 48, 4, 37, 70, S.ACT_RELU), ((12,), 33, 8, 40, 52, S.ACT_SIGMOID),
                                 ((6,), 16, 16, 40, 130, S.ACT_NONE), ((4,), 8, 5, 7, 9, S.ACT_MISH)])
def test_conv2d_winograd_dilated(cfg, monkeypatch):
    """Dilated 3x3 layers on the Winograd kernel: a dilation-d convolution is d*d dilation-1 problems on the sub-sampled
    images (ry + d*Y, rx + d*X).  Shapes where the sub-images differ in size (H, W not multiples of d), are smaller than
    a tile, or the dilation exceeds the image; every epilogue option; against torch's fp64 convolution and the direct kernel."""
    cins, cout, dil, h, w_, act = cfg
    monkeypatch.setattr(S.Conv2dPlan, "WINO_MIN_BLOCKS", 0)
    monkeypatch.setattr(S.Conv2dPlan, "WINO_MAX_DILATION", 16)       # the kernel supports 1..16; the plan uses it up to 4
    g = _gen(43, str(cfg))
    b = 2
    xs = [torch.randn(b, c, h, w_, generator=g) for c in cins]
    cin = sum(cins)
    w = torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (9 * cin)) ** 0.5
    bias = torch.randn(cout, generator=g) * 0.1
    res, m = torch.randn(b, cout, h, w_, generator=g), torch.randn(b, cout, h, w_, generator=g)
    z, hh = torch.rand(b, cout, h, w_, generator=g), torch.randn(b, cout, h, w_, generator=g)
    y = torch.nn.functional.conv2d(torch.cat(xs, 1).double(), w.double(), bias.double(), 1, dil, dil) + res.double()
    y = {S.ACT_NONE: lambda t: t, S.ACT_RELU: torch.relu, S.ACT_TANH: torch.tanh, S.ACT_SIGMOID: torch.sigmoid,
         S.ACT_MISH: lambda t: t * torch.tanh(torch.nn.functional.softplus(t))}[act](y)
    ref = hh.double() + z.double() * (y * m.double() - hh.double())
    plan = S.Conv2dPlan(dev(w), None, dilation=dil, act=act, bias=dev(bias))
    assert plan.wino_packed is not None
    out = plan([dev(t) for t in xs], residual=dev(res), mul=dev(m), blend=(dev(z), dev(hh)))
    torch.testing.assert_close(out.cpu().double(), ref, atol=2e-5, rtol=1e-5)
    plan.wino_packed = None                       # the same plan on the direct kernel
    direct = plan([dev(t) for t in xs], residual=dev(res), mul=dev(m), blend=(dev(z), dev(hh)))
    torch.testing.assert_close(out, direct, atol=2e-5, rtol=1e-5)


def test_conv2d_winograd_c_abi():
    from diffuvolume_amd import _lib
    import ctypes
    lib = _lib.load()
    g = _gen(42, "abi2d")
    x, w = torch.randn(1, 12, 18, 36, generator=g), torch.randn(24, 12, 3, 3, generator=g) * 0.1
    xd, wd = dev(x), dev(w)
    n = lib.dv_conv2d_wino_packed_floats(12, 24)
    assert n == 2 * 1 * 4096
    wp, out = torch.empty(n, device=DEV), torch.empty(1, 24, 18, 36, device=DEV)
    assert lib.dv_conv2d_wino_pack_weights_f32(wd.data_ptr(), wp.data_ptr(), 12, 24, _lib.stream_ptr()) == 0
    ptrs, chans = (ctypes.c_void_p * 1)(xd.data_ptr()), (ctypes.c_int * 1)(12)
    rc = lib.dv_conv2d_wino_cat_f32(ptrs, chans, 1, wp.data_ptr(), None, None, None, None, None, None, out.data_ptr(),
                                    1, 18, 36, 24, S.ACT_NONE, _lib.stream_ptr())
    assert rc == 0
    assert rel_err(out, torch.nn.functional.conv2d(x.double(), w.double(), None, 1, 1).float()) < 1e-5
    # refused: blend_z without blend_h, zero inputs, null output
    assert lib.dv_conv2d_wino_cat_f32(ptrs, chans, 1, wp.data_ptr(), None, None, None, None, out.data_ptr(), None,
                                      out.data_ptr(), 1, 18, 36, 24, 0, _lib.stream_ptr()) != 0
    assert lib.dv_conv2d_wino_cat_f32(ptrs, chans, 0, wp.data_ptr(), None, None, None, None, None, None, out.data_ptr(),
                                      1, 18, 36, 24, 0, _lib.stream_ptr()) != 0
    assert lib.dv_conv2d_wino_cat_f32(ptrs, chans, 1, wp.data_ptr(), None, None, None, None, None, None, None,
                                      1, 18, 36, 24, 0, _lib.stream_ptr()) != 0


def test_precision_switch_selects_kernels(monkeypatch):
    """'f32' (default) puts the 3x3x3 stride-1 layers and the 3x3 dilation-1 2-D layers on the Winograd kernels,
    'f32_direct' keeps every layer on the direct implicit GEMMs; strided / 1x1 / single-channel layers are direct either
    way (dilated 3x3 2-D layers run the Winograd kernel on their sub-sampled images)."""
    w3, w2 = torch.randn(32, 16, 3, 3, 3, device=DEV), torch.randn(32, 16, 3, 3, device=DEV)
    S.set_default_conv_precision(None)
    monkeypatch.delenv("DV_CONV_PRECISION", raising=False)
    assert S.Conv3dPlan(w3).wino and S.Conv2dPlan(w2).wino_packed is not None
    assert not S.Conv3dPlan(w3, stride=2).wino
    assert not S.Conv3dPlan(torch.randn(1, 16, 3, 3, 3, device=DEV)).wino
    assert not S.Conv3dPlan(torch.randn(32, 16, 1, 1, 1, device=DEV)).wino
    assert S.Conv2dPlan(w2, dilation=2).wino_packed is not None and S.Conv2dPlan(w2, stride=2).wino_packed is None
    try:
        S.set_default_conv_precision("f32_direct")
        assert not S.Conv3dPlan(w3).wino and S.Conv2dPlan(w2).wino_packed is None
    finally:
        S.set_default_conv_precision(None)
    monkeypatch.setenv("DV_CONV_PRECISION", "f32_direct")
    assert not S.Conv3dPlan(w3).wino


def test_conv_single_channel_head_z_march():
    """The z-marching form of the Cout == 1 head (csrc/conv3d.hip, conv3d_c1z_kernel: a 16 x 64 tile, a segment of 3 / 6 / 12
    output planes per block, three rotating accumulator sets) takes over for volumes of >= 64 tiles per batch item; here the
    test hook dv_conv3d_set_c1z forces it onto ragged shapes: depth that is not a multiple of the segment, several segments,
    partial tiles in y and x, unaligned rows, channel counts 1 .. 33, residual + ReLU.  The segment length follows the batch
    size (block count) in production, so it must not change a single bit: the three pinned lengths have to agree exactly --
    that is what keeps a shard of a batch bit-identical to the batch (tests/test_gpu_fullsize.py) although the launches
    differ."""
    from diffuvolume_amd import _lib
    lib = _lib.load()
    outs = {}
    try:
        for zs in (3, 6, 12):
            assert lib.dv_conv3d_set_c1z(1, zs) == 0
            for cin, dims, extras in [(5, (1, 3, 5, 7), False), (33, (2, 13, 9, 70), True), (32, (1, 25, 18, 130), False),
                                      (1, (1, 1, 1, 1), False), (32, (1, 12, 16, 64), True), (7, (1, 27, 33, 67), True)]:
                g = _gen(29, str((cin, dims)))
                x = torch.randn(dims[0], cin, *dims[1:], generator=g)
                w = torch.randn(1, cin, 3, 3, 3, generator=g) * (2.0 / (27 * cin)) ** 0.5
                res = torch.randn(dims[0], 1, *dims[1:], generator=g) if extras else None
                y = torch.nn.functional.conv3d(x.double(), w.double(), None, 1, 1)
                if res is not None:
                    y = torch.relu(y + res.double())
                plan = S.Conv3dPlan(dev(w), None, stride=1, act=S.ACT_RELU if extras else S.ACT_NONE)
                out = plan(dev(x), residual=None if res is None else dev(res))
                err = float((out.cpu().double() - y).abs().max())
                assert out.shape == y.shape and err <= 1e-5 * max(1.0, float(y.abs().max())), (cin, dims, err)
                outs.setdefault((cin, dims), []).append(out.cpu())
        assert lib.dv_conv3d_set_c1z(1, 5) != 0
    finally:
        lib.dv_conv3d_set_c1z(0, 0)
    for key, o in outs.items():
        assert torch.equal(o[0], o[1]) and torch.equal(o[0], o[2]), key


@pytest.mark.parametrize("shape", [(2, 64, 128, 8, 12, 60), (1, 32, 64, 6, 9, 37), (1, 16, 64, 4, 8, 64)])
def test_stride2_tilings_give_the_same_bits(shape, monkeypatch):
    """The DIRECT stride-2 convolution picks 2 x 4 x 32 or 2 x 2 x 32 output tiles from the size of ONE batch item
    (csrc/conv3d.hip); both sum every output chunk by chunk, tap by tap, so the choice may not change a bit -- what lets a
    shard of a batch reproduce the batch (section 6 of DESIGN.md).  Also against F.conv3d.  (DV_S2PP=0: the layers the
    polyphase kernel takes since round 5 would not reach the direct kernel otherwise.)"""
    from diffuvolume_amd import _lib
    monkeypatch.setenv("DV_S2PP", "0")
    b, cin, cout, d, h, w = shape
    g = _gen(191, str(shape))
    x = torch.randn(b, cin, d, h, w, generator=g)
    wt = torch.randn(cout, cin, 3, 3, 3, generator=g) * (2.0 / (27 * cin)) ** 0.5
    plan = S.Conv3dPlan(dev(wt), None, stride=2, act=S.ACT_RELU)
    assert not plan.s2pp
    outs = {}
    try:
        for mode, tile in ((1, "big"), (2, "small")):
            assert _lib.load().dv_conv3d_set_s2_tile(mode) == 0
            outs[tile] = plan(dev(x)).clone()
    finally:
        _lib.load().dv_conv3d_set_s2_tile(0)
    assert torch.equal(outs["big"], outs["small"])
    ref = torch.relu(torch.nn.functional.conv3d(x.double(), wt.double(), None, 2, 1)).float()
    assert rel_err(outs["small"], ref) < 1e-5
    assert torch.equal(plan(dev(x)), outs["big"])              # and whichever the launcher picks by itself
