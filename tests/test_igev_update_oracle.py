"""CPU: the oracle's restatement of IGEV's update block against golden vectors from the reference class
(oracle/make_golden_igev_update.py), and the product module's parameter names."""
import types

import torch

from conftest import load_golden
from diffuvolume_amd.synth import _gen, synth_state_dict
from oracle import igev_oracle as I

ARGS = types.SimpleNamespace(corr_levels=2, corr_radius=4, n_gru_layers=3, n_downsample=2)


def update_inputs(seed, b, h, w):
    dims = [(h, w), (h // 2, w // 2), (h // 4, w // 4)]
    net = [torch.tanh(torch.randn(b, 128, hh, ww, generator=_gen(seed, f"net{i}"))) for i, (hh, ww) in enumerate(dims)]
    inp = [[torch.randn(b, 128, hh, ww, generator=_gen(seed, f"inp{i}{j}")) * 0.5 for j in range(3)]
           for i, (hh, ww) in enumerate(dims)]
    corr = torch.randn(b, 162, h, w, generator=_gen(seed, "corr"))
    disp = torch.rand(b, 1, h, w, generator=_gen(seed, "disp")) * 40
    return net, inp, corr, disp


def update_state_dict(seed):
    from diffuvolume_amd.update import BasicMultiUpdateBlock
    return synth_state_dict(BasicMultiUpdateBlock(ARGS, hidden_dims=[128, 128, 128]).state_dict(), seed=seed)


def test_parameter_names_match_reference_layout():
    sd = update_state_dict(1)
    assert len(sd) == 34
    for k, shape in (("encoder.convc1.weight", (64, 162, 1, 1)), ("encoder.convd1.weight", (64, 1, 7, 7)),
                     ("encoder.conv.bias", (127,)), ("gru04.convz.weight", (128, 384, 3, 3)),
                     ("gru08.convq.weight", (128, 384, 3, 3)), ("gru16.convr.weight", (128, 256, 3, 3)),
                     ("disp_head.conv2.weight", (1, 256, 3, 3)), ("mask_feat_4.0.weight", (32, 128, 3, 3))):
        assert tuple(sd[k].shape) == shape, k


def test_update_block_oracle_matches_reference():
    g = load_golden("igev_update")
    sd = update_state_dict(int(g["sd_seed"]))
    net, inp, corr, disp = update_inputs(int(g["in_seed"]), 1, 16, 24)
    n1, mask1, d1 = I.update_block(sd, net, inp, corr, disp)
    n2, mask2, d2 = I.update_block(sd, n1, inp, corr, disp + d1)
    for i in range(3):
        torch.testing.assert_close(n1[i], g[f"net1_{i}"], atol=2e-6, rtol=1e-5)
        torch.testing.assert_close(n2[i], g[f"net2_{i}"], atol=5e-6, rtol=1e-5)
    torch.testing.assert_close(mask1, g["mask1"], atol=1e-5, rtol=1e-5)
    torch.testing.assert_close(d1, g["delta1"], atol=1e-4, rtol=1e-5)
    torch.testing.assert_close(d2, g["delta2"], atol=2e-4, rtol=1e-5)
    slow = I.update_block(sd, net, inp, None, None, iter04=False, iter08=False, update=False)
    torch.testing.assert_close(slow[2], g["slow_net2"], atol=2e-6, rtol=1e-5)
