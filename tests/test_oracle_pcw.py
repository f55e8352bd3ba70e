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
