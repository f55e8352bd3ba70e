"""CPU: the oracle's IGEV cost-volume front (hourglass(8), FeatureAtt, classifier + regression) against golden
vectors produced by the reference's own modules (oracle/make_golden_igev_volume.py)."""
import torch

from conftest import load_golden
from diffuvolume_amd.synth import _gen, synth_state_dict
from oracle import igev_oracle as I


def igev_inputs(seed, b, h, w, shift=3):
    ml = torch.randn(b, 96, h, w, generator=_gen(seed, "ml"))
    mr = torch.roll(ml, -shift, dims=-1) + 0.1 * torch.randn(b, 96, h, w, generator=_gen(seed, "mr"))
    feats = [torch.randn(b, c, h // s, w // s, generator=_gen(seed, f"feat{i}"))
             for i, (c, s) in enumerate(((96, 1), (64, 2), (192, 4), (160, 8)))]
    return ml, mr, feats


def volume_state_dict(g):
    from diffuvolume_amd.igev_stereo_ddim import IGEVCostVolume
    return synth_state_dict(IGEVCostVolume().state_dict(), seed=int(g["sd_seed"]), logit_gain=float(g["logit_gain"]))


def test_state_dict_names_match_reference_layout():
    from diffuvolume_amd.igev_stereo_ddim import IGEVCostVolume
    keys = set(IGEVCostVolume().state_dict().keys())
    for k in ("corr_stem.conv.weight", "corr_stem.bn.running_var", "corr_feature_att.feat_att.0.conv.weight",
              "corr_feature_att.feat_att.1.bias", "cost_agg.conv1.0.conv.weight", "cost_agg.conv3_up.conv.weight",
              "cost_agg.agg_0.2.bn.weight", "cost_agg.feature_att_up_16.feat_att.0.bn.running_mean",
              "cost_agg.conv1_up.conv.weight", "classifier.weight"):
        assert k in keys, k
    sd = IGEVCostVolume().state_dict()
    assert tuple(sd["cost_agg.conv3_up.conv.weight"].shape) == (48, 32, 4, 4, 4)
    assert tuple(sd["cost_agg.agg_0.0.conv.weight"].shape) == (32, 64, 1, 1, 1)


def test_hourglass_oracle_matches_reference():
    g = load_golden("igev_volume")
    sd = volume_state_dict(g)
    x = torch.randn(2, 8, 16, 16, 24, generator=_gen(int(g["hg_seed"]), "x"))
    _, _, feats = igev_inputs(int(g["hg_seed"]), 2, 16, 24)
    y = I.igev_hourglass(x, feats, sd)
    torch.testing.assert_close(y, g["hg_y"], atol=1e-6, rtol=1e-5)


def test_front_oracle_matches_reference():
    g = load_golden("igev_volume")
    sd = volume_state_dict(g)
    ml, mr, feats = igev_inputs(int(g["front_seed"]), 1, 8, 32)
    geo, init = I.igev_cost_volume(ml, mr, feats, sd)
    torch.testing.assert_close(geo, g["geo"], atol=1e-6, rtol=1e-5)
    torch.testing.assert_close(init, g["init_disp"], atol=1e-4, rtol=1e-5)
