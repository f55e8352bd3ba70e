"""The first aggregation layer of a DiffuVolume step on the FACTORS of its input (csrc/rank1_filter.hip,
submodule.Rank1FilterPlan): relu(bn(conv3d(volume * noise))) with volume = softmax(att) * concat(L, R) (SceneFlow/models/
acv_ddim.py:260, :200-203, :388-390) against the oracle's statement of exactly that -- the 64-channel volume built,
multiplied by the filter and convolved by F.conv3d on the CPU -- and against the generic HIP convolution with the filter
prologue; ragged widths, a channel count that is not a multiple of 4, fewer than 48 disparities, more than 256 columns."""
import pytest
import torch
import torch.nn.functional as F

import diffuvolume_amd as dv
from diffuvolume_amd import submodule as S
from diffuvolume_amd.synth import _gen
from oracle import acv_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def rel(a, b):
    return float((a.cpu().double() - b.double()).abs().max() / b.double().abs().max().clamp(min=1e-30))


@pytest.mark.parametrize("shape", [(2, 32, 32, 48, 8, 24), (1, 32, 32, 48, 5, 37), (1, 8, 6, 12, 4, 9), (1, 16, 32, 24, 3, 300),
                                   (2, 32, 30, 48, 6, 16)])
def test_rank1_filter_layer_vs_oracle_and_generic_conv(shape):
    b, c, cout, d, h, w = shape
    g = _gen(211, str(shape))
    L, R = torch.randn(b, c, h, w, generator=g), torch.randn(b, c, h, w, generator=g)
    att = torch.randn(b, 1, d, h, w, generator=g) * 2
    noise = torch.rand(b, d, h, w, generator=g)
    wt = torch.randn(cout, 2 * c, 3, 3, 3, generator=g) * (2.0 / (27 * cout)) ** 0.5
    bn = (torch.rand(cout, generator=g) * 0.4 + 0.8, torch.randn(cout, generator=g) * 0.1,
          torch.randn(cout, generator=g) * 0.1, torch.rand(cout, generator=g) + 0.5)
    # the reference's statement (acv_ddim.py:388-390, :260, dres0[0]) on the CPU
    vol = O.attention_concat_volume(att, O.build_concat_volume(L, R, d))
    y = F.conv3d(vol * noise.unsqueeze(1), wt, None, 1, 1)
    ref = torch.relu(F.batch_norm(y, bn[2], bn[3], bn[0], bn[1], False, 0.0, 1e-5))
    # HIP: the volume with its factors, then the layer on the factors
    vol_d = dv.build_concat_attention_volume(L.to(DEV), R.to(DEV), att.to(DEV), d)
    assert rel(vol_d, vol) < 1e-6 and hasattr(vol_d, "_dv_factors")
    assert rel(vol_d._dv_factors[0], torch.softmax(att, dim=2)[:, 0]) < 1e-6
    plan = S.Rank1FilterPlan(wt.to(DEV), tuple(t.to(DEV) for t in bn), act=S.ACT_RELU)
    assert plan.applies(vol_d)
    out = plan(vol_d, noise.to(DEV))
    assert out.shape == ref.shape
    assert rel(out, ref) < 1e-5, rel(out, ref)
    assert torch.equal(plan(vol_d, noise.to(DEV)), out)                 # tables cached on the volume: same bits
    # the generic path (filter prologue of the 3-D convolution kernel) on the same volume
    gen = S.Conv3dPlan(wt.to(DEV), tuple(t.to(DEV) for t in bn), stride=1, act=S.ACT_RELU)
    assert rel(gen(vol_d, in_scale=noise.to(DEV)), ref) < 1e-5
    # a volume without factors (any other tensor) does not take the fast path
    assert not plan.applies(vol_d.clone())


def test_model_takes_the_factored_layer_and_agrees_with_the_generic_one(acv_state_dict):
    """ACVNet_DDIM.model_predictions on a volume that carries its factors vs the same volume stripped of them: the two
    first-layer paths give the same step (cost within 1e-5 of its scale, disparity within the contract)."""
    from diffuvolume_amd import acv_ddim as AD
    m = dv.ACVNet_DDIM(192, False, False)
    m.load_state_dict(acv_state_dict, strict=True)
    m = m.to(DEV).eval()
    g = _gen(212, "m")
    b, h, w = 1, 16, 32
    L, R = torch.randn(b, 32, h, w, generator=g).to(DEV), torch.randn(b, 32, h, w, generator=g).to(DEV)
    att = (torch.randn(b, 1, 48, h, w, generator=g) * 2).to(DEV)
    vol = dv.build_concat_attention_volume(L, R, att, 48)
    x_t = torch.randn(b, 48, h, w, generator=g).to(DEV)
    t = torch.full((b,), 999, dtype=torch.long, device=DEV)
    assert m.prepare().dres0_rank1 is not None and m.prepare().dres0_rank1.applies(vol)
    _, xs1, d1, _ = m.model_predictions(vol, x_t, t)
    _, xs2, d2, _ = m.model_predictions(vol.clone(), x_t, t)                # no factors: generic convolution
    dd = (d1 - d2).abs()
    assert float((dd > 1e-3).float().mean()) <= 1e-3 and float(dd.mean()) < 1e-4, (float(dd.mean()), float(dd.max()))
    assert AD.RANK1_FILTER is True
