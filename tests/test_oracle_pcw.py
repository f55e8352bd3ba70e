"""Pin the KITTI12 (PCWNet + DiffuVolume) oracle to vectors produced by the imported reference."""
import pytest
import torch

from conftest import load_golden
from diffuvolume_amd.synth import NoiseTape, _gen, synth_state_dict
from oracle import pcw_oracle as P


@pytest.fixture(scope="module")
def pcw_sd():
    from diffuvolume_amd.pwcnet_ddim import PWCNet_ddim
    return synth_state_dict(PWCNet_ddim(192, True).state_dict(), seed=2, logit_gain=8.0,
                            scale={"refinenet3.conv8.weight": 0.002})


def _inputs(seed, b=1, h=16, w=32):
    vol = torch.rand(b, 32, 48, h, w, generator=_gen(seed, "vol"))
    fl = {"finetune_feature": torch.randn(b, 32, h, w, generator=_gen(seed, "fl"))}
    fr = {"finetune_feature": torch.randn(b, 32, h, w, generator=_gen(seed, "fr"))}
    return vol, fl, fr


def test_layers():
    from diffuvolume_amd.pwcnet_ddim import Hourglass, HourglassUp
    g = load_golden("pcw_layers")
    sd = {"h." + k: v for k, v in synth_state_dict(Hourglass(32).state_dict(), seed=71).items()}
    torch.testing.assert_close(P.hourglass_mish(g["hg_x"], sd, "h"), g["hg_y"], atol=1e-5, rtol=1e-5)
    sd = {"u." + k: v for k, v in synth_state_dict(HourglassUp(32).state_dict(), seed=72).items()}
    torch.testing.assert_close(P.hourglassup(g["hu_x"], g["hu_f4"], g["hu_f5"], g["hu_f6"], sd, "u"), g["hu_y"],
                               atol=1e-5, rtol=1e-5)


def test_model_predictions(pcw_sd):
    g = load_golden("pcw_model_predictions")
    vol, fl, fr = _inputs(g["seed"])
    pn, xs, disp, prob = P.PCWDiffusionOracle(pcw_sd).model_predictions(vol, g["x_t"], g["t"], fl, fr)
    d = (disp - g["disp"]).abs()
    assert float(d.mean()) < 1e-4 and float((d > 1e-3).float().mean()) < 5e-3
    from oracle import acv_oracle as A
    assert float((A.disparity_uncertainty(disp, prob) - g["unc"]).abs().mean()) < 1e-3
    same = ((xs - g["x_start"]).abs() < 1e-3).all(dim=1)
    assert float(same.float().mean()) > 0.99


def test_ddim_sample(pcw_sd):
    g = load_golden("pcw_ddim_sample")
    vol, fl, fr = _inputs(g["seed"])
    final, stack = P.PCWDiffusionOracle(pcw_sd).ddim_sample(vol, g["used"], g["asd"], fl, fr, NoiseTape(g["tape_seed"]))
    assert stack.shape == g["stack"].shape == (4, 1, 64, 128)
    d = (stack - g["stack"]).abs()
    for i in range(1, 4):
        assert float(d[i].median()) < 1e-4, (i, float(d[i].median()))
    assert float((final - g["final"]).abs().mean()) < 1e-3


def test_conditioned_network_at_the_raw_bars():
    """The conditioned network (oracle/calibrate.py: BatchNorm buffers = statistics of the data, classifier gain 0.5,
    refinement head x 0.2): the oracle meets the IMPORTED REFERENCE's outputs at the contract's raw bars on every
    step, and a float64 evaluation of the oracle shows the network is well conditioned (fp32 within 1e-3 px of float64
    on every pixel) -- which is what entitles the GPU tests to assert the raw bars on it."""
    from conftest import conditioned_pcw_state_dict
    from oracle import acv_oracle as A
    sd, g = conditioned_pcw_state_dict("pcw_conditioned_fixture")
    vol, fl, fr = _inputs(g["seed"])
    orc = P.PCWDiffusionOracle(sd)
    pn, xs, disp, prob = orc.model_predictions(vol, g["x_t"], g["t"], fl, fr)
    d = (disp - g["disp"]).abs()
    assert float((d > 1e-3).float().mean()) <= 1e-3 and float(d.mean()) < 1e-4, (float(d.mean()), float(d.max()))
    assert float((A.disparity_uncertainty(disp, prob) - g["unc"]).abs().mean()) < 1e-3
    final, stack = orc.ddim_sample(vol, g["used"], g["asd"], fl, fr, NoiseTape(g["tape_seed"]))
    for i in range(1, 4):
        di = (stack[i] - g["stack"][i]).abs()
        assert float((di > 1e-3).float().mean()) <= 1e-3, (i, float(di.mean()), float(di.max()))
        assert abs(float((stack[i] - g["used"]).abs().mean()) - float((g["stack"][i] - g["used"]).abs().mean())) < 1e-4
    assert float(((final - g["final"]).abs() > 1e-3).float().mean()) <= 1e-3
    # conditioning: float64 weights and activations
    sd64 = {k: (v.double() if v.is_floating_point() and not k.startswith("time_embedding") else v) for k, v in sd.items()}
    f64 = lambda feats: {k: v.double() for k, v in feats.items()}
    d64 = P.PCWDiffusionOracle(sd64).model_predictions(vol.double(), g["x_t"].double(), g["t"], f64(fl), f64(fr))[2]
    e = (disp.double() - d64).abs()
    assert float(e.max()) < 1e-3 and float(e.mean()) < 1e-4, (float(e.mean()), float(e.max()))
    # the refinement head is not idle on this network: it moves the disparity by O(1) px
    pred3 = A.upsample_softmax_regress(orc.aggregate(vol * (((torch.clamp(g["x_t"] + A.time_shift(g["t"], sd)[:, :, None, None], -1, 1)) + 1) / 2).unsqueeze(1)), 192, align_corners=True)[0]
    assert float((disp - pred3).abs().mean()) > 0.5
