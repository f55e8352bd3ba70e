"""Pin the CPU oracle (oracle/acv_oracle.py) to the reference: every function is checked
against vectors produced by the imported reference (oracle/make_golden.py)."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from diffuvolume_amd.synth import NoiseTape, _gen, synth_state_dict, synth_stereo_batch
from oracle import acv_oracle as O


def close(a, b, atol=0.0, rtol=0.0):
    assert a.shape == b.shape, (a.shape, b.shape)
    assert a.dtype == b.dtype, (a.dtype, b.dtype)
    torch.testing.assert_close(a, b, atol=atol, rtol=rtol)


@pytest.mark.parametrize("tag", ["small", "cpg8", "cpg12", "ragged"])
def test_builders(tag):
    g = load_golden(f"builders_{tag}")
    close(O.build_gwc_volume(g["L"], g["R"], g["maxdisp"], g["groups"]), g["gwc"])
    close(O.build_concat_volume(g["L"], g["R"], g["maxdisp"]), g["concat"])
    close(O.build_concat_volume(g["L"], g["R"], g["maxdisp"], zero_left=True), g["concat_k12"])


def test_builder_edge_cases():
    L, R = torch.randn(1, 4, 2, 3), torch.randn(1, 4, 2, 3)
    v = O.build_gwc_volume(L, R, 5, 2)                       # maxdisp > W: planes d >= W stay zero
    assert v.shape == (1, 2, 5, 2, 3) and float(v[:, :, 3:].abs().max()) == 0.0
    with pytest.raises(AssertionError):
        O.build_gwc_volume(torch.randn(1, 6, 2, 3), torch.randn(1, 6, 2, 3), 2, 4)


def test_concat_attention():
    g = load_golden("concat_attention")
    close(O.attention_concat_volume(g["att"], O.build_concat_volume(g["L"], g["R"], g["maxdisp"])), g["out"])


def test_disparity_regression():
    g = load_golden("disparity_regression")
    close(O.disparity_regression(g["prob"], 12), g["flat"])
    close(O.disparity_regression(g["prob"], 12, keepdim=True), g["keepdim"])
    with pytest.raises(AssertionError):
        O.disparity_regression(g["prob"][0], 12)


@pytest.mark.parametrize("tag", ["d12", "d48"])
def test_regression_tail(tag):
    g = load_golden(f"regress_{tag}")
    d = g["cost"].shape[2]
    for ac, sfx in ((False, ""), (True, "_ac")):
        disp, prob = O.upsample_softmax_regress(g["cost"], 4 * d, align_corners=ac)
        close(disp, g["disp" + sfx])
        close(O.disparity_uncertainty(disp, prob), g["unc" + sfx])


def test_encoder_and_schedule():
    g = load_golden("encoder_schedule")
    orc = O.ACVDiffusionOracle({})
    close(orc.encode_x_T(g["disp_q"]), g["x_T"])
    # known answers: 0.0->{0:1}; 3.25->{3:.75,4:.25}; 46.9->{46:.1,47:.9}; 47.0, 47.75 -> {47:1}
    th = O.encode_two_hot(g["disp_q"]).view(48, -1)
    assert th[0, 0] == 1 and th[:, 0].sum() == 1
    assert th[3, 1] == 0.75 and th[4, 1] == 0.25
    np.testing.assert_allclose(th[46:, 2].numpy(), [0.1, 0.9], atol=1e-5)
    assert th[47, 3] == 1 and th[47, 4] == 1 and th[:47, 4].sum() == 0
    close(orc.alphas_cumprod, g["alphas_cumprod"], rtol=1e-13)        # libm cos() may differ by an ulp across CPUs
    close(orc.sqrt_recip_alphas_cumprod, g["sqrt_recip"], rtol=1e-13)
    close(orc.sqrt_recipm1_alphas_cumprod, g["sqrt_recipm1"], rtol=1e-13)
    assert abs(float(orc.sqrt_recip_alphas_cumprod[999]) - 20291.17) < 0.01
    for s in (2, 3, 5, 20):
        pairs = O.ddim_time_pairs(1000, s)
        assert [p[0] for p in pairs] + [pairs[-1][1]] == g[f"times_{s}"].tolist()
    assert O.ddim_time_pairs(1000, 5) == [(999, 799), (799, 599), (599, 399), (399, 199), (199, -1)]


def test_masked_x_T():
    """acv_ddim.py:415-417, the `mask_gt` variant of x_T (None at every call site, but part of the signature): the
    reference's own forward recorded what it hands to ddim_sample (oracle/make_golden_xT_masked.py)."""
    g = load_golden("acv_xT_masked")
    orc = O.ACVDiffusionOracle({})
    close(orc.encode_x_T(g["disp"]), g["x_T"])
    close(orc.encode_x_T(g["disp"], g["mask_gt"]), g["x_T_masked"])
    assert 0.2 < float((g["x_T_masked"] != g["x_T"]).float().mean()) < 0.5          # the mask does something


def test_time_shift(acv_state_dict):
    g = load_golden("time_shift")
    shift = O.time_shift(g["t"], acv_state_dict)
    close(shift, g["shift"])
    close(g["noisy"] + shift[:, :, None, None], g["out"])


@pytest.mark.parametrize("tag", ["c3s1", "c3s2", "c1s1", "c3s1_wide", "c3s1_one"])
def test_conv_layers(tag):
    from diffuvolume_amd.acv_ddim import _cb3
    g = load_golden(f"layer_{tag}")
    k, s = g["k"], g["stride"]
    sd = synth_state_dict(_cb3(g["cin"], g["cout"], k, s, (k - 1) // 2).state_dict(), seed=g["seed"])
    y = O.convbn_3d(g["x"], {"L." + n: v for n, v in sd.items()}, "L", s, (k - 1) // 2)
    close(y, g["y"])
    close(torch.relu(y), g["y_relu"])


def _deconv_sd(seed):
    m = torch.nn.Sequential(torch.nn.ConvTranspose3d(16, 8, 3, padding=1, output_padding=1, stride=2, bias=False),
                            torch.nn.BatchNorm3d(8))
    return synth_state_dict(m.state_dict(), seed=seed)


def test_deconv_layer():
    g = load_golden("layer_deconv")
    sd = _deconv_sd(g["seed"])
    up = torch.nn.functional.conv_transpose3d(g["x"], sd["0.weight"], None, 2, 1, 1)
    close(O._bn(up, {"b." + k[2:]: v for k, v in sd.items() if k.startswith("1.")}, "b"), g["y"])


@pytest.mark.parametrize("tag", ["nopad", "pad", "padw"])
def test_attention_block(tag):
    from diffuvolume_amd.acv_ddim import _WindowAttention
    g = load_golden(f"layer_attention_{tag}")
    sd = synth_state_dict(_WindowAttention(128, 16).state_dict(), seed=g["seed"])
    y = O.attention_block(g["x"], {"a." + k: v for k, v in sd.items()}, "a")
    close(y, g["y"], atol=2e-6, rtol=1e-5)      # same algebra, different matmul blocking


def test_hourglass():
    from diffuvolume_amd.acv_ddim import Hourglass
    g = load_golden("layer_hourglass")
    sd = synth_state_dict(Hourglass(32).state_dict(), seed=g["seed"])
    y = O.hourglass(g["x"], {"h." + k: v for k, v in sd.items()}, "h")
    close(y, g["y"], atol=2e-5, rtol=1e-4)


def _volume(seed, b=1, h=16, w=32):
    return torch.rand(b, 64, 48, h, w, generator=_gen(seed, "vol"))


def test_model_predictions(acv_state_dict):
    g = load_golden("model_predictions")
    orc = O.ACVDiffusionOracle(acv_state_dict)
    pn, xs, pred, prob = orc.model_predictions(_volume(g["vol_seed"]), g["x_T"], g["t"])
    assert pn.dtype == torch.float64 and xs.dtype == torch.float32
    close(pred, g["pred"], atol=2e-3, rtol=0)
    frac = float(((pred - g["pred"]).abs() > 1e-4).float().mean())
    assert frac < 2e-3, frac
    close(O.disparity_uncertainty(pred, prob), g["unc"], atol=5e-3, rtol=1e-3)
    same = (xs == g["x_start"]).all(dim=1)
    assert float(same.float().mean()) > 0.99      # two-hot bins move only where floor() flips
    torch.testing.assert_close(pn[same.unsqueeze(1).expand_as(pn)],
                               g["pred_noise"][same.unsqueeze(1).expand_as(pn)], atol=1e-9, rtol=1e-9)


def test_ddim_sample(acv_state_dict):
    g = load_golden("ddim_sample")
    assert g["state_dtypes"].tolist() == ["torch.float32"] + ["torch.float64"] * 4
    orc = O.ACVDiffusionOracle(acv_state_dict)
    final, stack = orc.ddim_sample(_volume(g["vol_seed"]), g["used"], g["x_T"], NoiseTape(g["tape_seed"]))
    assert stack.shape == g["stack"].shape
    d = (stack - g["stack"]).abs()
    # hard renewal masks make single pixels chaotic under fp re-association:
    # judge the bulk tightly and the tail loosely
    assert float((d > 1e-3).float().mean()) < 1e-3, float((d > 1e-3).float().mean())
    assert float((final - g["final"]).abs().mean()) < 1e-4


def test_metrics():
    g = load_golden("metrics")
    m = O.image_metrics(g["est"], g["gt"], g["mask"].bool())
    for k in ("EPE", "D1", "Thres1", "Thres2", "Thres3"):
        assert abs(float(m[k]) - g[k]) < 1e-6, k
    g = load_golden("metrics_all_skipped")
    assert float(O.image_metrics(g["est"], g["gt"], g["mask"].bool())["EPE"]) == 0.0 == g["EPE"]


def test_igev_geo_filter_lookup():
    """KITTI15/core/geometry_ddim.py:33-69 (noise-filtered geometry lookup, 2 levels, radius 4)."""
    from oracle import igev_oracle as IO
    g = load_golden("igev_geo_lookup")
    out = IO.geo_filter_lookup(g["geo"], g["f1"], g["f2"], g["disp"], g["coords"], g["noisy"])
    assert out.shape == g["out"].shape == (2, 162, 5, 24)
    torch.testing.assert_close(out, g["out"], atol=2e-6, rtol=1e-6)
