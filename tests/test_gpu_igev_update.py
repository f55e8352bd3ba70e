"""KITTI15 flavour: IGEV's recurrent update block (ConvGRU x3 scales, motion encoder, heads) on the 2-D HIP
convolution with the gate arithmetic fused, vs the reference's golden vectors and the oracle."""
import pytest
import torch
import torch.nn.functional as F

from conftest import load_golden
from diffuvolume_amd import submodule as S
from diffuvolume_amd.synth import _gen
from oracle import igev_oracle as I
from test_igev_update_oracle import ARGS, update_inputs, update_state_dict

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def dev(t):
    return t.to(DEV)


def rel_err(a, b):
    return float((a.cpu().double() - b.double()).abs().max() / b.double().abs().max().clamp(min=1e-30))


def make_block(seed):
    from diffuvolume_amd.update import BasicMultiUpdateBlock
    m = BasicMultiUpdateBlock(ARGS, hidden_dims=[128, 128, 128])
    m.load_state_dict(update_state_dict(seed), strict=True)
    return m.to(DEV).eval()


@pytest.mark.parametrize("act", ["sigmoid", "tanh"])
def test_conv2d_gate_epilogues(act):
    """act(conv + bias + residual) [* mul] [blended with (z, h)] -- the ConvGRU arithmetic of update.py:36-39."""
    g = _gen(111, act)
    x = torch.randn(2, 40, 11, 70, generator=g)
    w = torch.randn(32, 40, 3, 3, generator=g) * 0.05
    bias, res = torch.randn(32, generator=g) * 0.1, torch.randn(2, 32, 11, 70, generator=g)
    h, z = torch.randn(2, 32, 11, 70, generator=g), torch.rand(2, 32, 11, 70, generator=g)
    fn = torch.sigmoid if act == "sigmoid" else torch.tanh
    v = fn(torch.nn.functional.conv2d(x, w, bias, 1, 1) + res)
    plan = S.Conv2dPlan(dev(w), None, act=S.ACT_SIGMOID if act == "sigmoid" else S.ACT_TANH, bias=dev(bias))
    torch.testing.assert_close(plan(dev(x), residual=dev(res)).cpu(), v, atol=2e-6, rtol=1e-5)
    torch.testing.assert_close(plan(dev(x), residual=dev(res), mul=dev(h)).cpu(), v * h, atol=4e-6, rtol=1e-5)
    torch.testing.assert_close(plan(dev(x), residual=dev(res), blend=(dev(z), dev(h))).cpu(), (1 - z) * h + z * v,
                               atol=4e-6, rtol=1e-5)


def test_update_block_golden():
    g = load_golden("igev_update")
    m = make_block(int(g["sd_seed"]))
    net, inp, corr, disp = update_inputs(int(g["in_seed"]), 1, 16, 24)
    dnet, dinp = [dev(t) for t in net], [[dev(t) for t in l] for l in inp]
    n1, mask1, d1 = m([t.clone() for t in dnet], dinp, dev(corr), dev(disp))
    n1 = [t.clone() for t in n1]
    n2, mask2, d2 = m([t.clone() for t in n1], dinp, dev(corr), dev(disp) + d1)
    for i in range(3):
        assert rel_err(n1[i], g[f"net1_{i}"]) < 1e-5 and rel_err(n2[i], g[f"net2_{i}"]) < 2e-5
    assert rel_err(mask1, g["mask1"]) < 1e-5 and rel_err(mask2, g["mask2"]) < 2e-5
    assert rel_err(d1, g["delta1"]) < 1e-5 and rel_err(d2, g["delta2"]) < 3e-5
    slow = m([t.clone() for t in dnet], dinp, iter04=False, iter08=False, update=False)
    assert rel_err(slow[2], g["slow_net2"]) < 1e-5


def test_update_block_vs_oracle_kitti_aspect_batch2():
    """1/4 resolution of a 1248-wide frame is 312 columns (not a multiple of the 64-column tile), batch 2."""
    from diffuvolume_amd.synth import synth_state_dict  # noqa: F401
    m = make_block(121)
    sd = update_state_dict(121)
    net, inp, corr, disp = update_inputs(122, 2, 24, 312)
    ref_net, ref_mask, ref_delta = I.update_block(sd, net, inp, corr, disp)
    n, mask, delta = m([dev(t) for t in net], [[dev(t) for t in l] for l in inp], dev(corr), dev(disp))
    for i in range(3):
        assert rel_err(n[i], ref_net[i]) < 1e-5
    assert rel_err(mask, ref_mask) < 1e-5 and rel_err(delta, ref_delta) < 1e-5


def test_ddim_loop_with_the_real_update_block():
    """IGEVDiffusionLoop.model_predictions / ddim_sample (igev_stereo_ddim.py:226-359) with the HIP update block and
    the HIP filtered lookup, against the oracle's loop driven by the oracle's update block: the whole per-pair IGEV
    iteration (filter -> lookup -> motion encoder -> 3 ConvGRUs -> heads -> two-hot -> DDIM update)."""
    from diffuvolume_amd.geometry_ddim import Combined_Geo_Encoding_Volume
    from diffuvolume_amd.igev_stereo_ddim import DynamicHead180, IGEVDiffusionLoop
    from diffuvolume_amd.synth import NoiseTape, synth_state_dict, toy_upsample_disp
    from diffuvolume_amd.update import BasicMultiUpdateBlock
    b, h, w = 1, 16, 24
    sd = update_state_dict(131)
    sd["disp_head.conv2.weight"] = sd["disp_head.conv2.weight"] * 0.05      # keep the per-iteration step ~1 bin
    m = BasicMultiUpdateBlock(ARGS, hidden_dims=[128, 128, 128])
    m.load_state_dict(sd, strict=True)
    m = m.to(DEV).eval()
    head = DynamicHead180()
    head.load_state_dict(synth_state_dict(head.state_dict(), seed=132), strict=True)
    head = head.eval()
    net, inp, _, _ = update_inputs(133, b, h, w)
    geo = torch.randn(b, 8, 48, h, w, generator=_gen(134, "geo"))
    f1, f2 = torch.randn(b, 16, h, w, generator=_gen(134, "f1")), torch.randn(b, 16, h, w, generator=_gen(134, "f2"))
    init = torch.rand(b, 1, h, w, generator=_gen(134, "init")) * 40
    used = torch.nn.functional.interpolate(init * 4, scale_factor=4, mode="bilinear") + 1.5
    asd = torch.rand(b, 48, h, w, generator=_gen(134, "asd")) * 2 - 1
    x_t = torch.randn(b, 48, h, w, generator=_gen(134, "xt"))
    t = torch.full((b,), 999, dtype=torch.long)

    orc = I.IGEVLoopOracle(head.state_dict(), lambda n, i, c, f, **kw: I.update_block(sd, n, i, c, f),
                           toy_upsample_disp, geo, f1, f2, net_list=net, inp_list=inp)
    _, xs_ref, pred_ref, c1_ref = orc.model_predictions(init, init, 3, x_t, t)
    orc = I.IGEVLoopOracle(head.state_dict(), lambda n, i, c, f, **kw: I.update_block(sd, n, i, c, f),
                           toy_upsample_disp, geo, f1, f2, net_list=net, inp_list=inp)
    final_ref = orc.ddim_sample(init, init, 2, used, asd, NoiseTape(135))

    geo_fn = Combined_Geo_Encoding_Volume(dev(f1), dev(f2), dev(geo), radius=4, num_levels=2)
    loop = IGEVDiffusionLoop(head.to(DEV), m, toy_upsample_disp, n_gru_layers=3, slow_fast_gru=False)
    dinp = [[dev(x) for x in l] for l in inp]
    _, xs, pred, c1 = loop.model_predictions(dev(init), dev(init), None, 3, [dev(x) for x in net], dinp, geo_fn,
                                             dev(x_t), dev(t), None)
    torch.testing.assert_close(c1.cpu(), c1_ref, atol=2e-3, rtol=1e-4)
    torch.testing.assert_close(pred.cpu(), pred_ref, atol=8e-3, rtol=1e-4)
    final = loop.ddim_sample(dev(init), dev(init), None, 2, [dev(x) for x in net], dinp, geo_fn, dev(used), dev(asd),
                             None, noise=NoiseTape(135))
    d = (final.cpu() - final_ref).abs()
    assert float(d.median()) < 2e-3 and float(d.mean()) < 5e-2, (float(d.median()), float(d.mean()))


def test_ddim_loop_20_steps_teacher_forced_vs_oracle():
    """BASELINE config 5 runs 20 DDIM steps (the reference hard-codes 2, igev_stereo_ddim.py:124): the IGEV loop with
    the REAL update block (HIP 2-D convolutions with fused gates) and the HIP filtered lookup over all 20 steps against
    oracle/igev_oracle.py driven by the oracle's own update block -- every step teacher forced (HIP from the oracle's
    img / mask / coords1 / hidden states entering that step, igev_stereo_ddim.py:306-351), at the contract's raw bars:
    |d disp| <= 1e-3 px on 99.9 % of the pixels and |EPE_hip - EPE_oracle| < 1e-4 (against `used`), then the free run's
    ensemble output."""
    from diffuvolume_amd.geometry_ddim import Combined_Geo_Encoding_Volume
    from diffuvolume_amd.igev_stereo_ddim import DynamicHead180, IGEVDiffusionLoop
    from diffuvolume_amd.synth import NoiseTape, synth_state_dict, toy_upsample_disp
    from diffuvolume_amd.update import BasicMultiUpdateBlock
    b, h, w, steps, iters = 1, 16, 24, 20, 4
    cof = (0.6,) + (0.0,) * (steps - 2) + (0.1, 0.3)
    sd = update_state_dict(141)
    sd["disp_head.conv2.weight"] = sd["disp_head.conv2.weight"] * 0.05      # keep the per-iteration step ~1 bin
    m = BasicMultiUpdateBlock(ARGS, hidden_dims=[128, 128, 128])
    m.load_state_dict(sd, strict=True)
    m = m.to(DEV).eval()
    head = DynamicHead180()
    head.load_state_dict(synth_state_dict(head.state_dict(), seed=142), strict=True)
    head = head.eval()
    net, inp, _, _ = update_inputs(143, b, h, w)
    geo = torch.randn(b, 8, 48, h, w, generator=_gen(144, "geo"))
    f1, f2 = torch.randn(b, 16, h, w, generator=_gen(144, "f1")), torch.randn(b, 16, h, w, generator=_gen(144, "f2"))
    init = torch.rand(b, 1, h, w, generator=_gen(144, "init")) * 40
    used = torch.nn.functional.interpolate(init * 4, scale_factor=4, mode="bilinear") + 1.5
    asd = torch.rand(b, 48, h, w, generator=_gen(144, "asd")) * 2 - 1
    orc = I.IGEVLoopOracle(head.state_dict(), lambda n, i, c, f, **kw: I.update_block(sd, n, i, c, f), toy_upsample_disp,
                           geo, f1, f2, sampling_timesteps=steps, cof=cof, net_list=net, inp_list=inp)
    trace = []
    final_ref = orc.ddim_sample(init, init, iters, used, asd, NoiseTape(145), trace=trace)
    assert len(trace) == steps and [r["time"] for r in trace][:3] == [999, 949, 899] and trace[-1]["time_next"] == -1

    geo_fn = Combined_Geo_Encoding_Volume(dev(f1), dev(f2), dev(geo), radius=4, num_levels=2)
    loop = IGEVDiffusionLoop(head.to(DEV), m, toy_upsample_disp, n_gru_layers=3, slow_fast_gru=False,
                             sampling_timesteps=steps, ensemble_cof=cof)
    dinp = [[dev(x) for x in l] for l in inp]
    u2 = used.reshape(b, 4 * h, 4 * w)
    worst = {"frac": 0.0, "epe": 0.0, "max": 0.0}
    for i, r in enumerate(trace):
        mask = dev(r["mask_in"]).clone()
        eps = None if r["eps"] is None else dev(r["eps"])
        fill = None if r["fill"] is None else dev(r["fill"]).double().contiguous()
        pred, xs, xn, c1, nets = loop.ddim_step(i, dev(init), dev(r["coords1_in"]), None, iters,
                                                [dev(t) for t in r["nets_in"]], dinp, geo_fn, dev(used), dev(r["img"]),
                                                mask, None, eps, fill, None)
        d = (pred.cpu() - r["disp"].reshape(b, 4 * h, 4 * w)).abs()
        frac = float((d > 1e-3).float().mean())
        epe = abs(float((pred.cpu() - u2).abs().mean()) - float((r["disp"].reshape(b, 4 * h, 4 * w) - u2).abs().mean()))
        worst = {"frac": max(worst["frac"], frac), "epe": max(worst["epe"], epe), "max": max(worst["max"], float(d.max()))}
        assert frac <= 1e-3 and epe < 1e-4, (i + 1, frac, epe, float(d.max()))
        torch.testing.assert_close(c1.cpu(), r["coords1_out"], atol=1e-3, rtol=1e-5)
        for a_, b_ in zip(nets, r["nets_out"]):
            assert rel_err(a_, b_) < 1e-4
        assert float((mask.cpu() - r["mask_out"]).abs().max()) <= 1.0
        same = ((xs.cpu() - r["x_start"]).abs() < 1e-2).all(dim=1)
        assert float(same.float().mean()) > 0.99
        if xn is not None:
            agree = ((mask.cpu() == 0) == (r["mask_out"] == 0)) & same
            dx = (xn.cpu() - r["img_next"]).abs()[agree.unsqueeze(1).expand_as(r["img_next"])]
            assert float(dx.mean()) < 1e-4
    print("igev 20-step teacher forced, worst over steps:", worst)
    final = loop.ddim_sample(dev(init), dev(init), None, iters, [dev(x) for x in net], dinp, geo_fn, dev(used), dev(asd),
                             None, noise=NoiseTape(145))
    d = (final.cpu() - final_ref).abs()
    assert abs(float((final.cpu() - u2).abs().mean()) - float((final_ref - u2).abs().mean())) < 1e-3
    assert float(d.median()) < 2e-3, (float(d.median()), float(d.mean()))


def test_ddim_step_teacher_forced_at_config5_size():
    """One DDIM step of BASELINE config 5 at its real geometry -- 1248x384 (quarter resolution 96 x 312: 4.875 tiles of
    64 columns), 32 GRU iterations through the real update block, the filtered geometry lookup at every iteration --
    against oracle/igev_oracle.py (igev_stereo_ddim.py:226-292), at the contract's raw bars on all 479 232 pixels:
    |d disp| <= 1e-3 px on 99.9 % of them, |EPE_hip - EPE_oracle| < 1e-4.  Steps t = 999 (float32 state) and t = 949
    (a float64 state, hidden states carried over).  ~1 min of host CPU."""
    from diffuvolume_amd.geometry_ddim import Combined_Geo_Encoding_Volume
    from diffuvolume_amd.igev_stereo_ddim import DynamicHead180, IGEVDiffusionLoop
    from diffuvolume_amd.synth import synth_state_dict, toy_upsample_disp
    from diffuvolume_amd.update import BasicMultiUpdateBlock
    b, h, w, steps, iters = 1, 96, 312, 20, 32
    cof = (0.6,) + (0.0,) * (steps - 2) + (0.1, 0.3)
    sd = update_state_dict(161)
    sd["disp_head.conv2.weight"] = sd["disp_head.conv2.weight"] * 0.05      # keep the per-iteration step ~1 bin
    m = BasicMultiUpdateBlock(ARGS, hidden_dims=[128, 128, 128])
    m.load_state_dict(sd, strict=True)
    m = m.to(DEV).eval()
    head = DynamicHead180()
    head.load_state_dict(synth_state_dict(head.state_dict(), seed=162), strict=True)
    head = head.eval()
    net, inp, _, _ = update_inputs(163, b, h, w)
    geo = torch.randn(b, 8, 48, h, w, generator=_gen(164, "geo"))
    f1, f2 = torch.randn(b, 16, h, w, generator=_gen(164, "f1")), torch.randn(b, 16, h, w, generator=_gen(164, "f2"))
    init = torch.rand(b, 1, h, w, generator=_gen(164, "init")) * 40
    used = torch.nn.functional.interpolate(init * 4, scale_factor=4, mode="bilinear") + 1.5
    u2 = used.reshape(b, 4 * h, 4 * w)
    orc = I.IGEVLoopOracle(head.state_dict(), lambda n, i, c, f, **kw: I.update_block(sd, n, i, c, f), toy_upsample_disp,
                           geo, f1, f2, sampling_timesteps=steps, cof=cof, net_list=net, inp_list=inp)
    geo_fn = Combined_Geo_Encoding_Volume(dev(f1), dev(f2), dev(geo), radius=4, num_levels=2)
    loop = IGEVDiffusionLoop(head.to(DEV), m, toy_upsample_disp, n_gru_layers=3, slow_fast_gru=False,
                             sampling_timesteps=steps, ensemble_cof=cof)
    dinp = [[dev(x) for x in l] for l in inp]
    coords1 = init
    # (default run: t = 999; the float64-state step t = 949 under DV_FULL_PARITY=1 -- the free run below feeds a float64
    # state into its second step at this size in every run)
    full = __import__("os").environ.get("DV_FULL_PARITY") == "1"
    for time, dtype in ((999, torch.float32), (949, torch.float64))[:2 if full else 1]:
        x_t = torch.randn(b, 48, h, w, generator=_gen(165, f"xt{time}")).to(dtype)
        t = torch.full((b,), time, dtype=torch.long)
        nets_in = list(orc.net_list)
        _, xs_ref, pred_ref, c1_ref = orc.model_predictions(init, coords1, iters, x_t, t)
        _, xs, pred, c1 = loop.model_predictions(dev(init), dev(coords1), None, iters, [dev(x) for x in nets_in], dinp,
                                                 geo_fn, dev(x_t), dev(t), None)
        d = (pred.cpu() - pred_ref).abs().reshape(b, 4 * h, 4 * w)
        frac = float((d > 1e-3).float().mean())
        epe = abs(float((pred.cpu().reshape(u2.shape) - u2).abs().mean()) - float((pred_ref.reshape(u2.shape) - u2).abs().mean()))
        print(f"config-5 size, t = {time}: mean {float(d.mean()):.2e} px, max {float(d.max()):.2e} px, "
              f"share beyond 1e-3 px {frac:.2e}, |dEPE| {epe:.2e}")
        assert frac <= 1e-3 and epe < 1e-4, (time, frac, epe, float(d.max()))
        torch.testing.assert_close(c1.cpu(), c1_ref, atol=1e-3, rtol=1e-5)
        coords1 = c1_ref


def test_free_run_at_config5_size_vs_oracle():
    """The WHOLE loop, not one step: `ddim_sample` of BASELINE config 5's geometry (1248x384, quarter resolution 96 x 312) run
    freely -- 2 DDIM steps (the reference's own default, igev_stereo_ddim.py:124) x 4 GRU iterations through the real update
    block, the renewal mask and the refill fed back, the ensemble of [used, step 1, step 2] -- against oracle/igev_oracle.py
    driven by the same noise tape (igev_stereo_ddim.py:294-359).  Bars: the contract's on the final output over all
    479 232 pixels (<= 1e-3 px on 99.9 % of them, |EPE_hip - EPE_oracle| < 1e-4 against `used`), and the renewal decisions
    of the two runs may differ on at most 0.1 % of the pixels.  ~25 s of host CPU."""
    from diffuvolume_amd.geometry_ddim import Combined_Geo_Encoding_Volume
    from diffuvolume_amd.igev_stereo_ddim import DynamicHead180, IGEVDiffusionLoop
    from diffuvolume_amd.synth import NoiseTape, synth_state_dict, toy_upsample_disp
    from diffuvolume_amd.update import BasicMultiUpdateBlock
    b, h, w, steps, iters = 1, 96, 312, 2, 4
    sd = update_state_dict(171)
    sd["disp_head.conv2.weight"] = sd["disp_head.conv2.weight"] * 0.05      # keep the per-iteration step ~1 bin
    m = BasicMultiUpdateBlock(ARGS, hidden_dims=[128, 128, 128])
    m.load_state_dict(sd, strict=True)
    m = m.to(DEV).eval()
    head = DynamicHead180()
    head.load_state_dict(synth_state_dict(head.state_dict(), seed=172), strict=True)
    head = head.eval()
    net, inp, _, _ = update_inputs(173, b, h, w)
    geo = torch.randn(b, 8, 48, h, w, generator=_gen(174, "geo"))
    f1, f2 = torch.randn(b, 16, h, w, generator=_gen(174, "f1")), torch.randn(b, 16, h, w, generator=_gen(174, "f2"))
    init = torch.rand(b, 1, h, w, generator=_gen(174, "init")) * 40
    used = torch.nn.functional.interpolate(init * 4, scale_factor=4, mode="bilinear") + 1.5
    asd = torch.rand(b, 48, h, w, generator=_gen(174, "asd")) * 2 - 1
    orc = I.IGEVLoopOracle(head.state_dict(), lambda n, i, c, f, **kw: I.update_block(sd, n, i, c, f), toy_upsample_disp,
                           geo, f1, f2, sampling_timesteps=steps, net_list=net, inp_list=inp)
    trace = []
    final_ref = orc.ddim_sample(init, init, iters, used, asd, NoiseTape(175), trace=trace)
    assert len(trace) == steps and [r["time"] for r in trace] == [999, 499] and trace[-1]["time_next"] == -1
    geo_fn = Combined_Geo_Encoding_Volume(dev(f1), dev(f2), dev(geo), radius=4, num_levels=2)
    loop = IGEVDiffusionLoop(head.to(DEV), m, toy_upsample_disp, n_gru_layers=3, slow_fast_gru=False, sampling_timesteps=steps)
    dinp = [[dev(x) for x in l] for l in inp]
    final = loop.ddim_sample(dev(init), dev(init), None, iters, [dev(x) for x in net], dinp, geo_fn, dev(used), dev(asd), None,
                             noise=NoiseTape(175))
    u2 = used.reshape(b, 4 * h, 4 * w)
    d = (final.cpu().reshape(u2.shape) - final_ref.reshape(u2.shape)).abs()
    frac = float((d > 1e-3).float().mean())
    epe = abs(float((final.cpu().reshape(u2.shape) - u2).abs().mean()) - float((final_ref.reshape(u2.shape) - u2).abs().mean()))
    print(f"config-5 size free run, 2 steps x 4 iterations: mean {float(d.mean()):.2e} px, max {float(d.max()):.2e} px, "
          f"share beyond 1e-3 px {frac:.2e}, |dEPE| {epe:.2e}")
    assert frac <= 1e-3 and epe < 1e-4, (frac, epe, float(d.max()))


@pytest.mark.parametrize("shape", [(2, 5, 24, 78), (1, 3, 7, 9), (1, 2, 1, 5), (2, 4, 12, 39)])
def test_update_glue_kernels_vs_torch(shape):
    """csrc/update_glue.hip: `pool2x`, `interp` (bilinear, align_corners=True) and the single-input-channel 7x7 `convd1`
    of the motion encoder (KITTI15/core/update.py:86-102) against PyTorch's own operators on the CPU."""
    from diffuvolume_amd import update as U
    b, c, h, w = shape
    g = _gen(151, str(shape))
    x = torch.randn(b, c, h, w, generator=g)
    torch.testing.assert_close(U.pool2x(dev(x)).cpu(), F.avg_pool2d(x, 3, stride=2, padding=1), atol=1e-6, rtol=1e-6)
    for size in ((2 * h, 2 * w), (2 * h - 1, 2 * w + 1), (h, w), (1, 1)):
        dest = torch.empty(b, c, *size)
        ref = F.interpolate(x, size, mode="bilinear", align_corners=True)
        torch.testing.assert_close(U.interp(dev(x), dev(dest)).cpu(), ref, atol=1e-6, rtol=1e-6)
    enc = U.BasicMotionEncoder(ARGS)
    with torch.no_grad():
        enc.convd1.weight.copy_(torch.randn(enc.convd1.weight.shape, generator=g) * 0.1)
        enc.convd1.bias.copy_(torch.randn(enc.convd1.bias.shape, generator=g) * 0.1)
        d = torch.rand(b, 1, h, w, generator=g) * 40
        ref = F.relu(enc.convd1.double()(d.double())).float()
        enc = enc.float().to(DEV)
        out = enc._convd1(dev(d))
    assert float((out.cpu() - ref).abs().max()) <= 2e-6 * max(1.0, float(ref.abs().max()))


def test_conv2d_over_virtual_concatenation():
    """conv(torch.cat([a, b, c], 1)) without the copy; chunk boundaries fall inside sources (20 + 6 + 38 channels)."""
    g = _gen(151, "cat")
    parts = [torch.randn(2, c, 9, 70, generator=g) for c in (20, 6, 38)]
    w = torch.randn(32, 64, 3, 3, generator=g) * 0.05
    bias = torch.randn(32, generator=g) * 0.1
    ref = torch.relu(torch.nn.functional.conv2d(torch.cat(parts, 1), w, bias, 1, 1))
    plan = S.Conv2dPlan(dev(w), None, act=S.ACT_RELU, bias=dev(bias))
    torch.testing.assert_close(plan([dev(t) for t in parts]).cpu(), ref, atol=2e-6, rtol=1e-5)
    torch.testing.assert_close(plan(dev(torch.cat(parts, 1))).cpu(), ref, atol=2e-6, rtol=1e-5)


@pytest.mark.parametrize("cfg", [((128, 128, 128), 128, 24, 78, "sigmoid"), ((128, 256), 128, 48, 156, "tanh"),
                                 ((64,), 32, 12, 40, "relu"), ((40, 24), 48, 9, 33, "none")])
def test_conv2d_ksplit_small_launches(cfg):
    """Launches too small to fill the chip (one IGEV pair at 1/8 and 1/16 resolution: the GRU convolutions of
    KITTI15/core/update.py:33-40) are split over the input channels (`dv_conv2d_cat_ksplit_f32`): same fused epilogue
    (bias, residual, sigmoid / tanh, `mul`, the GRU blend), same result as the one-block-per-tile kernel and as
    PyTorch's convolution, bit-reproducible (the slices are added in a fixed order)."""
    from diffuvolume_amd import _lib
    from diffuvolume_amd import submodule as S
    chans, cout, h, w, act = cfg
    cin = sum(chans)
    g = _gen(57, str(cfg))
    xs = [torch.randn(1, c, h, w, generator=g) for c in chans]
    wt = torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (9 * cin)) ** 0.5
    bias = torch.randn(cout, generator=g) * 0.1
    res, mul = torch.randn(1, cout, h, w, generator=g) * 0.2, torch.rand(1, cout, h, w, generator=g)
    z, hh = torch.rand(1, cout, h, w, generator=g), torch.randn(1, cout, h, w, generator=g)
    lib = _lib.load()
    ks = lib.dv_conv2d_auto_kslices(1, cin, h, w, cout, 3, 1)
    # (the factor is a function of ONE batch item since round 5 -- it fixes the summation order, and a shard of a batch has
    # to reproduce the batch's bits -- so the batch argument does not move it)
    assert ks > 1 and lib.dv_conv2d_auto_kslices(8, cin, h, w, cout, 3, 1) == ks
    assert lib.dv_conv2d_auto_kslices(1, cin, 192, 624, cout, 3, 1) == 1
    fn = {"sigmoid": torch.sigmoid, "tanh": torch.tanh, "relu": torch.relu, "none": lambda t: t}[act]
    y = fn(torch.nn.functional.conv2d(torch.cat(xs, 1), wt, bias, 1, 1) + res) * mul
    y = hh + z * (y - hh)
    plan = S.Conv2dPlan(wt.to(DEV), None, act={"sigmoid": S.ACT_SIGMOID, "tanh": S.ACT_TANH, "relu": S.ACT_RELU,
                                               "none": S.ACT_NONE}[act], bias=bias.to(DEV))
    plan.wino_packed = None                              # force the direct kernel (these sizes stay off Winograd anyway)
    args = ([t.to(DEV) for t in xs],)
    kw = dict(residual=res.to(DEV), mul=mul.to(DEV), blend=(z.to(DEV), hh.to(DEV)))
    out = plan(*args, **kw)
    assert float((out.cpu() - y).abs().max() / y.abs().max()) < 1e-5
    assert torch.equal(plan(*args, **kw), out)


@pytest.mark.parametrize("shape", [(2, 40, 72), (1, 96, 312), (3, 23, 50)])
def test_gru_gate_pair_launch_equals_the_two_convolutions(shape):
    """ConvGRU's convz / convr (update.py:33-35) read the same [h | x]: Conv2dPairPlan runs them as one Winograd launch.
    Every output element is computed by the same instruction sequence as in a single Winograd launch: bit-equal."""
    b, h, w = shape
    g = _gen(31, "gate_pair")
    parts = [dev(torch.randn(b, c, h, w, generator=g)) for c in (128, 127, 1, 128)]
    w1, w2 = (dev(torch.randn(128, 384, 3, 3, generator=g) * 0.03) for _ in range(2))
    b1, b2 = (dev(torch.randn(128, generator=g) * 0.1) for _ in range(2))
    cz, cr = (dev(torch.randn(b, 128, h, w, generator=g)) for _ in range(2))
    pair = S.Conv2dPairPlan((w1, b1), (w2, b2), S.ACT_SIGMOID)
    z, rh = pair(parts, residual=(cz, cr), mul=(None, parts[0]))
    z1 = S.Conv2dPlan(w1, None, act=S.ACT_SIGMOID, bias=b1)(parts, residual=cz)
    rh1 = S.Conv2dPlan(w2, None, act=S.ACT_SIGMOID, bias=b2)(parts, residual=cr, mul=parts[0])
    from diffuvolume_amd import _lib
    ks = [_lib.load().dv_conv2d_wino_auto_kslices(384, h, w, c, 1) for c in (256, 128)]
    if -(-h // 16) * -(-w // 16) * 4 >= S.Conv2dPlan.WINO_MIN_BLOCKS and ks[0] == ks[1]:
        assert torch.equal(z, z1) and torch.equal(rh, rh1)                       # (per batch item) the single launches are Winograd too
    else:                                                     # they ran the direct kernel / another K-split: another order of summation
        assert rel_err(z, z1.cpu()) < 1e-5 and rel_err(rh, rh1.cpu()) < 1e-5
    ref = torch.sigmoid(F.conv2d(torch.cat(parts, 1).double().cpu(), w2.double().cpu(), b2.double().cpu(), padding=1)
                        + cr.double().cpu()) * parts[0].double().cpu()
    assert rel_err(rh, ref) < 1e-5


@pytest.mark.parametrize("cfg", [(2, 32, 32, 24, 78, True), (1, 64, 9, 48, 156, False), (2, 12, 5, 7, 9, False)])
def test_deconv2d_k4s2_as_parity_convolutions(cfg):
    """ConvTranspose2d(k 4, s 2, p 1) [+ BN + LeakyReLU | + bias] of IGEV's spx heads (igev_stereo_ddim.py:110-112,
    :209-217) on the 3x3 kernels + pixel shuffle, against PyTorch's own transposed convolution in float64."""
    b, cin, cout, h, w, with_bn = cfg
    g = _gen(57, "deconv2d")
    x = torch.randn(b, cin, h, w, generator=g)
    wt = torch.randn(cin, cout, 4, 4, generator=g) * 0.1
    if with_bn:
        bn = (torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g) * 0.1,
              torch.randn(cout, generator=g) * 0.1, torch.rand(cout, generator=g) + 0.5)
        ref = F.conv_transpose2d(x.double(), wt.double(), None, 2, 1)
        ref = F.leaky_relu(F.batch_norm(ref, bn[2].double(), bn[3].double(), bn[0].double(), bn[1].double(), False, 0.0, 1e-5), 0.01)
        plan = S.Deconv2dK4S2Plan(dev(wt), tuple(dev(t) for t in bn), act=S.ACT_LEAKY)
    else:
        bias = torch.randn(cout, generator=g) * 0.1
        ref = F.conv_transpose2d(x.double(), wt.double(), bias.double(), 2, 1)
        plan = S.Deconv2dK4S2Plan(dev(wt), None, bias=dev(bias))
    out = plan(dev(x))
    assert tuple(out.shape) == (b, cout, 2 * h, 2 * w)
    assert rel_err(out, ref) < 1e-5


@pytest.mark.parametrize("chans", [(128, 127, 1, 128), (1, 1, 1, 1), (7, 1), (9, 8, 7, 6), (1, 30, 1), (3,), (8, 8, 8, 8),
                                   (5, 1, 1, 9)])
@pytest.mark.parametrize("kernel", ["wino", "direct", "ksplit"])
def test_conv2d_source_queue_edge_cases(chans, kernel, monkeypatch):
    """The position in the virtual concatenation is running scalar state in the 2-D kernels (address of the next channel
    plane, channels left in the source, a queue of the sources to come; conv2d_wino.hip / conv2d.hip).  Its corner cases:
    one-channel sources (IGEV hands the GRU `[h 128 | motion 127 | disp 1 | interp 128]`, KITTI15/core/update.py:126-142),
    several source switches inside one 8-channel chunk, four sources, a channel count that is not a multiple of the chunk,
    a second batch item (every source has its own batch stride), and K-slices that start in the middle of a source."""
    from diffuvolume_amd import _lib
    cin, cout = sum(chans), 32
    b, h, w = (2, 24, 40) if kernel != "ksplit" else (1, 12, 20)
    g = _gen(311, f"{chans}{kernel}")
    xs = [torch.randn(b, c, h, w, generator=g) for c in chans]
    wt = torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (9 * cin)) ** 0.5
    bias = torch.randn(cout, generator=g) * 0.1
    ref = torch.tanh(F.conv2d(torch.cat(xs, 1).double(), wt.double(), bias.double(), 1, 1)).float()
    monkeypatch.setattr(S.Conv2dPlan, "WINO_MIN_BLOCKS", 0)
    monkeypatch.setattr(S.Conv2dPlan, "KSPLIT", kernel == "ksplit")
    plan = S.Conv2dPlan(dev(wt), None, act=S.ACT_TANH, bias=dev(bias))
    if kernel != "wino":
        plan.wino_packed = None
    if kernel == "ksplit" and cin >= 128:                # (the short concatenations stay on one slice)
        assert _lib.load().dv_conv2d_auto_kslices(b, cin, h, w, cout, 3, 1) > 1
    out = plan([dev(t) for t in xs])
    assert float((out.cpu() - ref).abs().max()) <= 1e-5 * max(1.0, float(ref.abs().max()))    # the layer-level bar (fp32 re-association)
    # the same through one materialised tensor, and with the sources as views into it (unequal base alignment)
    cat = dev(torch.cat(xs, 1))
    assert torch.equal(plan(cat), out)


def test_front_kernels_vs_torch():
    """csrc/igev_front.hip: the few-input-channel KxK convolution (7x7 stride-2 stem of the context encoder, RGB stems) with
    folded BatchNorm, and InstanceNorm2d + activation, against float64 torch statements; plus `hip_sequential` on a stem
    (conv + InstanceNorm + LeakyReLU, conv + InstanceNorm + ReLU) and a transposed-convolution head."""
    import torch.nn as nn
    from diffuvolume_amd import _lib
    from diffuvolume_amd import igev_stereo_ddim as I
    g = _gen(61, "front")
    for cin, cout, k, st, h, w in ((3, 64, 7, 2, 37, 50), (3, 32, 3, 2, 33, 47), (1, 16, 5, 1, 9, 20), (4, 70, 7, 1, 18, 17)):
        conv = nn.Conv2d(cin, cout, k, st, k // 2).to(DEV)
        bn = nn.BatchNorm2d(cout).to(DEV).eval()
        with torch.no_grad():
            bn.running_mean.normal_(0, 0.3); bn.running_var.uniform_(0.5, 1.5); bn.weight.uniform_(0.5, 1.5); bn.bias.normal_(0, 0.2)
        x = torch.randn(2, cin, h, w, generator=g).to(DEV)
        with torch.no_grad():
            ref = torch.relu(bn.double()(conv.double()(x.double())))
            conv.float(); bn.float()
            out = I.hip_conv2d(conv, x, bn, S.ACT_RELU)
        assert out.shape == ref.shape and float((out.double() - ref).abs().max() / ref.abs().max()) < 1e-5, (cin, cout, k, st)
    x = (torch.randn(3, 5, 21, 34, generator=g) * 3 + 1).to(DEV)
    ref = torch.nn.functional.leaky_relu(torch.nn.functional.instance_norm(x.double()), 0.01)
    out = I.instance_norm_act(x.clone(), S.ACT_LEAKY)
    assert float((out.double() - ref).abs().max()) < 1e-5
    stem = nn.Sequential(I.BasicConv_IN(3, 32, kernel_size=3, stride=2, padding=1), nn.Conv2d(32, 32, 3, 1, 1, bias=False),
                         nn.InstanceNorm2d(32), nn.ReLU()).to(DEV).eval()
    head = nn.Sequential(nn.ConvTranspose2d(32, 9, kernel_size=4, stride=2, padding=1)).to(DEV).eval()
    img = torch.randn(2, 3, 40, 56, generator=g).to(DEV)
    with torch.no_grad():
        a = I.hip_sequential(head, I.hip_sequential(stem, img))
        stem.double(); head.double()
        ref = head(stem[3](stem[2](stem[1](torch.nn.functional.leaky_relu(stem[0].IN(stem[0].conv(img.double())), 0.01)))))
    assert a.shape == ref.shape and float((a.double() - ref).abs().max() / ref.abs().max()) < 2e-5
    with pytest.raises(_lib.DiffuVolumeError):
        I.hip_conv2d(nn.Conv2d(8, 8, 5, padding=2).to(DEV), torch.zeros(1, 8, 8, 8, device=DEV))


@pytest.mark.parametrize("cfg", [((128, 128), 24, 78, 4), ((128, 128, 128), 12, 40, 2), ((130, 61, 3), 24, 78, 1), ((256,), 20, 36, 3)])
def test_winograd_ksplit_small_launches(cfg):
    """The K-split form of the Winograd launches (csrc/conv2d_wino.hip, `dv_conv2d_wino_cat_ksplit_f32` /
    `..._pair_ksplit_f32`: IGEV's ConvGRU at 1/16 resolution, KITTI15/core/update.py:26-40): every source mode of the virtual
    concatenation, the gate pair and the candidate with its blend epilogue against float64; the slices are summed in a fixed
    order (rerun bit-equal) and their number depends on one batch item only (a shard reproduces the batch's bits)."""
    from diffuvolume_amd import _lib
    split, h, w, b = cfg
    cin = sum(split)
    g = _gen(91, "wksplit")
    parts = [dev(torch.randn(b, c, h, w, generator=g)) for c in split]
    w1, w2, wq = (dev(torch.randn(128, cin, 3, 3, generator=g) * (2.0 / (9 * cin)) ** 0.5) for _ in range(3))
    b1, b2, bq = (dev(torch.randn(128, generator=g) * 0.1) for _ in range(3))
    cz, cr, cq = (dev(torch.randn(b, 128, h, w, generator=g)) for _ in range(3))
    hh = dev(torch.randn(b, 128, h, w, generator=g))
    lib = _lib.load()
    assert lib.dv_conv2d_wino_auto_kslices(cin, h, w, 256, 1) > 1 and lib.dv_conv2d_wino_auto_kslices(cin, h, w, 128, 1) > 1
    pair = S.Conv2dPairPlan((w1, b1), (w2, b2), S.ACT_SIGMOID)
    pq = S.Conv2dPlan(wq, None, act=S.ACT_TANH, bias=bq)
    z, rh = pair(parts, residual=(cz, cr), mul=(None, hh))
    q = pq(parts, residual=cq, blend=(z, hh))
    x = torch.cat(parts, 1).double().cpu()
    conv = lambda wt, bt: F.conv2d(x, wt.double().cpu(), bt.double().cpu(), padding=1)
    zr = torch.sigmoid(conv(w1, b1) + cz.double().cpu())
    rr = torch.sigmoid(conv(w2, b2) + cr.double().cpu()) * hh.double().cpu()
    assert rel_err(z, zr) < 1e-5 and rel_err(rh, rr) < 1e-5
    qr = hh.double().cpu() + z.double().cpu() * (torch.tanh(conv(wq, bq) + cq.double().cpu()) - hh.double().cpu())
    assert rel_err(q, qr) < 1e-5
    z2, rh2 = pair(parts, residual=(cz, cr), mul=(None, hh))
    assert torch.equal(z, z2) and torch.equal(rh, rh2) and torch.equal(q, pq(parts, residual=cq, blend=(z, hh)))
    if b > 1:
        sl = slice(b - 1, b)
        z1, rh1 = pair([t[sl].contiguous() for t in parts], residual=(cz[sl].contiguous(), cr[sl].contiguous()),
                       mul=(None, hh[sl].contiguous()))
        assert torch.equal(z1, z[sl]) and torch.equal(rh1, rh[sl])


@pytest.mark.parametrize("act", ["sigmoid", "tanh"])
def test_gate_nonlinearities_accuracy(act):
    """dv_sigmoid / dv_tanh (csrc/dv_common.h: hardware exponential + reciprocal, tanh by its odd series below 1/8) against
    float64 over the whole range, through an identity 1x1 convolution (exact in fp32): relative error <= 1e-6 where the
    function is not saturated, no cancellation near 0, saturation to +-1 / 0 / 1, NaN stays NaN."""
    c = 32
    mags = torch.cat([torch.logspace(-30, -6, 64), torch.logspace(-6, 2, 4000), torch.tensor([0.0, 0.124999, 0.125, 0.125001, 88.0, 1e4, 3e38])])
    v = torch.cat([mags, -mags])
    n = v.numel()
    h, w = 16, -(-n // 16)
    flat = torch.zeros(h * w)
    flat[:n] = v
    x = flat.view(1, 1, h, w).repeat(1, c, 1, 1).contiguous()
    wt = torch.eye(c).view(c, c, 1, 1)
    plan = S.Conv2dPlan(dev(wt), None, act=S.ACT_SIGMOID if act == "sigmoid" else S.ACT_TANH)
    out = plan(dev(x)).cpu().double()[0, 3].reshape(-1)[:n]
    ref = torch.sigmoid(v.double()) if act == "sigmoid" else torch.tanh(v.double())
    rel = ((out - ref).abs() / ref.abs().clamp(min=1e-300))
    ok = ref.abs() > 1e-30                                # (sigmoid of a large negative argument underflows to 0 like the reference)
    assert float(rel[ok].max()) < 1e-6, (act, float(rel[ok].max()), float(v[ok][rel[ok].argmax()]))
    assert float((out - ref).abs().max()) < 2e-7
    xn = torch.full((1, c, 4, 4), float("nan"))
    assert bool(torch.isnan(plan(dev(xn))).all())


def test_update_block_side_stream_is_bit_identical():
    """BasicMultiUpdateBlock runs the motion encoder on a side stream beside gru16 / gru08 (update.py, OVERLAP): the same
    kernels on the same inputs -- three chained iterations are bit-equal to the one-stream run, with a tensor and with a
    fused-lookup request as `corr`."""
    from diffuvolume_amd.update import BasicMultiUpdateBlock
    from diffuvolume_amd.geometry_ddim import Combined_Geo_Encoding_Volume
    m = make_block(77)
    b, h, w = 2, 24, 80
    net, inp, corr, disp = update_inputs(78, b, h, w)
    g = _gen(79, "side")
    geo_fn = Combined_Geo_Encoding_Volume(dev(torch.randn(b, 16, h, w, generator=g)), dev(torch.randn(b, 16, h, w, generator=g)),
                                          dev(torch.randn(b, 8, 48, h, w, generator=g)))
    coords = dev(torch.arange(w, dtype=torch.float32).view(1, 1, 1, w).expand(b, 1, h, w).contiguous())
    noisy = dev(torch.rand(b, 48, h, w, generator=g))

    def run(overlap, use_request):
        BasicMultiUpdateBlock.OVERLAP = overlap
        nl = [dev(t).clone() for t in net]
        il = [[dev(t) for t in l] for l in inp]
        d = dev(disp).clone()
        outs = []
        for _ in range(3):
            c = geo_fn.request(d, coords + d, noisy) if use_request else dev(corr)
            nl, _, delta = m(nl, il, c, d, mask=False)
            d = d + delta
            outs.append(delta.clone())
        torch.cuda.synchronize()
        return [t.clone() for t in nl] + outs

    keep = BasicMultiUpdateBlock.OVERLAP
    try:
        for use_request in (False, True):
            a, c = run(False, use_request), run(True, use_request)
            assert all(torch.equal(x, y) for x, y in zip(a, c))
    finally:
        BasicMultiUpdateBlock.OVERLAP = keep
