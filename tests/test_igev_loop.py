"""IGEV DDIM loop: oracle vs the reference's own methods (CPU), HIP vs golden (GPU)."""
import pytest
import torch

from conftest import load_golden
from diffuvolume_amd.synth import NoiseTape, _gen, synth_state_dict, toy_update_block, toy_upsample_disp


def _setup(g):
    from diffuvolume_amd.igev_stereo_ddim import DynamicHead180
    b, c, d, h, w = 1, 8, 48, 8, 24
    s = g["seed"]
    geo = torch.randn(b, c, d, h, w, generator=_gen(s, "geo"))
    f1, f2 = torch.randn(b, 16, h, w, generator=_gen(s, "f1")), torch.randn(b, 16, h, w, generator=_gen(s, "f2"))
    init = torch.rand(b, 1, h, w, generator=_gen(s, "init")) * 40
    head = DynamicHead180()
    head.load_state_dict(synth_state_dict(head.state_dict(), seed=g["head_seed"]), strict=True)
    return geo, f1, f2, init, head.eval()


def test_oracle_matches_reference_methods():
    from oracle import igev_oracle as IO
    g = load_golden("igev_loop")
    geo, f1, f2, init, head = _setup(g)
    sd = head.state_dict()
    torch.testing.assert_close(IO.head180_shift(g["shift_t"], sd), g["shifts"], atol=1e-6, rtol=1e-5)
    with torch.no_grad():
        torch.testing.assert_close(head.shift(g["shift_t"]), g["shifts"], atol=1e-6, rtol=1e-5)   # host-side head
    orc = IO.IGEVLoopOracle(sd, toy_update_block, toy_upsample_disp, geo, f1, f2)
    t = torch.full((1,), 999, dtype=torch.long)
    pn, xs, pred, c1 = orc.model_predictions(init, init, 3, g["x_t"], t)
    torch.testing.assert_close(pred, g["pred"], atol=1e-4, rtol=1e-5)
    torch.testing.assert_close(c1, g["coords1"], atol=1e-4, rtol=1e-5)
    assert float(((xs - g["x_start"]).abs() < 1e-3).all(dim=1).float().mean()) > 0.99
    final = orc.ddim_sample(init, init, 3, g["used"], g["asd"], NoiseTape(g["tape_seed"]))
    d = (final - g["final"]).abs()
    assert float(d.median()) < 1e-4 and float(d.mean()) < 1e-2, (float(d.median()), float(d.mean()))


@pytest.mark.gpu
def test_hip_loop_matches_reference_methods():
    from diffuvolume_amd.geometry_ddim import Combined_Geo_Encoding_Volume
    from diffuvolume_amd.igev_stereo_ddim import IGEVDiffusionLoop
    g = load_golden("igev_loop")
    geo, f1, f2, init, head = _setup(g)
    dev = "cuda:0"
    head = head.to(dev)
    geo_fn = Combined_Geo_Encoding_Volume(f1.to(dev), f2.to(dev), geo.to(dev), radius=4, num_levels=2)
    loop = IGEVDiffusionLoop(head, toy_update_block, toy_upsample_disp, n_gru_layers=3, slow_fast_gru=False)
    i = init.to(dev)
    t = torch.full((1,), 999, dtype=torch.long, device=dev)
    pn, xs, pred, c1 = loop.model_predictions(i, i, None, 3, [None], [None], geo_fn, g["x_t"].to(dev), t, None)
    torch.testing.assert_close(pred.cpu(), g["pred"], atol=2e-4, rtol=1e-5)
    same = ((xs.cpu() - g["x_start"]).abs() < 1e-3).all(dim=1)
    assert float(same.float().mean()) > 0.99
    sel = same.unsqueeze(1).expand_as(pn)
    torch.testing.assert_close(pn.cpu()[sel], g["pred_noise"][sel], atol=5e-6, rtol=0)   # time MLP runs on the GPU
    final = loop.ddim_sample(i, i, None, 3, [None], [None], geo_fn, g["used"].to(dev), g["asd"].to(dev), None,
                             noise=NoiseTape(g["tape_seed"]))
    d = (final.cpu() - g["final"]).abs()
    # the contract's bars against the reference's own output (measured: mean 1.6e-6 px, max 1.5e-5 px, |dEPE| 5e-9)
    assert float(d.max()) < 1e-3 and float(d.mean()) < 1e-4, (float(d.max()), float(d.mean()))
    u = g["used"].reshape(g["final"].shape)
    assert abs(float((final.cpu() - u).abs().mean()) - float((g["final"] - u).abs().mean())) < 1e-4
