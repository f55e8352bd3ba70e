"""KITTI15 flavour: IGEV's once-per-pair cost-volume front on the HIP kernels vs the reference's golden vectors
(igev_stereo_ddim.py:24-91, :377-383) and the oracle."""
import pytest
import torch

from conftest import load_golden
from diffuvolume_amd.synth import _gen
from oracle import igev_oracle as I
from test_igev_volume_oracle import igev_inputs, volume_state_dict

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def dev(t):
    return t.to(DEV)


def rel_err(a, b):
    return float((a.cpu().double() - b.double()).abs().max() / b.double().abs().max().clamp(min=1e-30))


@pytest.fixture(scope="module")
def model():
    from diffuvolume_amd.igev_stereo_ddim import IGEVCostVolume
    g = load_golden("igev_volume")
    m = IGEVCostVolume()
    m.load_state_dict(volume_state_dict(g), strict=True)
    return m.to(DEV).eval()


@pytest.mark.parametrize("shape", [(2, 8, 5, 6, 10), (1, 32, 4, 8, 64), (1, 48, 3, 5, 7)])
def test_feature_gate(shape):
    from diffuvolume_amd.submodule import feature_gate
    g = _gen(95, str(shape))
    cv = torch.randn(*shape, generator=g)
    logit = torch.randn(shape[0], shape[1], shape[3], shape[4], generator=g) * 3
    ref = torch.sigmoid(logit).unsqueeze(2) * cv
    out = feature_gate(dev(cv), dev(logit))
    torch.testing.assert_close(out.cpu(), ref, atol=1e-6, rtol=1e-6)
    buf = dev(cv).clone()
    assert feature_gate(buf, dev(logit), inplace=True).data_ptr() == buf.data_ptr()
    torch.testing.assert_close(buf.cpu(), ref, atol=1e-6, rtol=1e-6)


def test_softmax_regress():
    from diffuvolume_amd.submodule import softmax_regress
    from oracle import acv_oracle as A
    cost = torch.randn(2, 1, 48, 7, 9, generator=_gen(96, "c")) * 5
    ref = A.disparity_regression(torch.softmax(cost.squeeze(1), 1), 48)
    torch.testing.assert_close(softmax_regress(dev(cost)).cpu(), ref, atol=2e-5, rtol=1e-6)


def test_hourglass_golden(model):
    g = load_golden("igev_volume")
    x = torch.randn(2, 8, 16, 16, 24, generator=_gen(int(g["hg_seed"]), "x"))
    _, _, feats = igev_inputs(int(g["hg_seed"]), 2, 16, 24)
    with torch.no_grad():
        y = model.cost_agg(dev(x), [dev(f) for f in feats])
    assert rel_err(y, g["hg_y"]) < 2e-5


def test_front_golden(model):
    g = load_golden("igev_volume")
    ml, mr, feats = igev_inputs(int(g["front_seed"]), 1, 8, 32)
    with torch.no_grad():
        geo, init = model(dev(ml), dev(mr), [dev(f) for f in feats])
    assert rel_err(geo, g["geo"]) < 2e-5
    assert init.shape == g["init_disp"].shape
    assert float((init.cpu() - g["init_disp"]).abs().max()) < 2e-3


def test_front_vs_oracle_ragged(model):
    """Width not a multiple of the MFMA row tile, batch 2; compared with the CPU oracle on the same weights."""
    g = load_golden("igev_volume")
    sd = volume_state_dict(g)
    ml, mr, feats = igev_inputs(97, 2, 16, 40)
    ref_geo, ref_init = I.igev_cost_volume(ml, mr, feats, sd)
    with torch.no_grad():
        geo, init = model(dev(ml), dev(mr), [dev(f) for f in feats])
    assert rel_err(geo, ref_geo) < 2e-5
    assert float((init.cpu() - ref_init).abs().max()) < 2e-3
