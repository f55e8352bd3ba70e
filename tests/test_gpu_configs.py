"""BASELINE.json config 4 at its real size on one GPU: KITTI12 PCWNet + DiffuVolume, 1248x384, maxdisp 192, batch 4,
3 DDIM steps.  Quarter resolution is 312 x 96, so every 16-wide tile row ends in a partial tile, and the coarser
volumes are 156 / 78 / 39 wide (the last one takes the generic gwc path).  Checked against the CPU oracle on pair 0
(builders, fused volume, every DDIM step from the oracle's state, the free run), against PyTorch's own fp32
convolution for a full-resolution dilated refinement layer, and for batch-shard invariance.  ~1 min of host CPU."""
import pytest
import torch
import torch.nn.functional as F

from diffuvolume_amd import submodule as S
from diffuvolume_amd.synth import NoiseTape, _gen, synth_state_dict, synth_stereo_batch
from oracle import acv_oracle as A
from oracle import loop_parity as LP
from oracle import pcw_oracle as P

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
B, H, W = 4, 384, 1248


def rel(a, b):
    return float((a.cpu().double() - b.double()).abs().max() / b.double().abs().max().clamp(min=1e-30))


_SETUPS = {}


def _setup(kind):
    """kind 'diagnostic': the synthetic weights with random BatchNorm statistics and the calibration factors of the
    pcw_forward_eval fixture (untrained residual stacks otherwise reach 1e9; logits still reach +-2600).  kind
    'conditioned': oracle/calibrate.py -- BatchNorm buffers = statistics of pair 0 of this very batch (stored in
    tests/golden/pcw_conditioned_config4.npz), classifier gain 0.5, refinement head x 0.2."""
    if kind in _SETUPS:
        return _SETUPS[kind]
    from diffuvolume_amd.pwcnet_ddim import PWCNet_ddim
    from conftest import conditioned_pcw_state_dict, load_golden
    if kind == "conditioned":
        sd, _ = conditioned_pcw_state_dict("pcw_conditioned_config4")
    else:
        g = load_golden("pcw_forward_eval")
        scale = {str(k): float(v) for k, v in zip(g["scale_keys"].tolist(), g["scale_vals"].tolist())}
        sd = synth_state_dict(PWCNet_ddim(192, True).state_dict(), seed=2, logit_gain=8.0, scale=scale)
    m = PWCNet_ddim(192, True)
    m.load_state_dict(sd, strict=True)
    m = m.to(DEV).eval()
    batch = {k: v.to(DEV) for k, v in synth_stereo_batch(B, H, W, seed=0).items()}
    with torch.no_grad():
        fl = m.feature_extraction(batch["left"])
        fr = m.feature_extraction(batch["right"])
    _SETUPS[kind] = (m, sd, batch, fl, fr)
    return _SETUPS[kind]


@pytest.fixture(scope="module")
def setup():
    return _setup("diagnostic")


def one(feats, i):
    return {k: v[i:i + 1].cpu() for k, v in feats.items()}


def test_builders_at_kitti_widths(setup):
    m, sd, batch, fl, fr = setup
    for i, (div, w) in enumerate(((4, 312), (8, 156), (16, 78), (32, 39)), start=1):
        L, R = fl[f"gw{i}"], fr[f"gw{i}"]
        assert L.shape[-1] == w and L.shape[-2] == H // div
        vol = S.build_gwc_volume(L, R, 192 // div, 40)
        ref = A.build_gwc_volume(L[:1].cpu(), R[:1].cpu(), 192 // div, 40)
        assert rel(vol[:1], ref) < 2e-6            # products rounded before the ordered sum, as the reference does
        cl, cr = fl[f"concat_feature{i}"], fr[f"concat_feature{i}"]
        cat = S.build_concat_volume(cl, cr, 192 // div, zero_left=True)
        assert torch.equal(cat[:1].cpu(), A.build_concat_volume(cl[:1].cpu(), cr[:1].cpu(), 192 // div, zero_left=True))


@pytest.mark.parametrize("kind", ["conditioned", pytest.param("diagnostic", marks=pytest.mark.skipif(
    __import__("os").environ.get("DV_FULL_PARITY") != "1",
    reason="the unconditioned diagnostic network (recorded in profiles/r04_parity_config4_diagnostic.json): DV_FULL_PARITY=1"))])
def test_fused_volume_and_loop_vs_oracle(kind):
    """Pair 0 at 1248x384 against the oracle: the fused volume, every DDIM step from the oracle's state, the free run.

    'conditioned' (verdict r3 #1b): the contract as written, RAW -- |d disp| <= 1e-3 px on 99.9 % of all 479 232
    pixels and |dEPE| < 1e-4 per teacher-forced step (all inputs the oracle's) and, when no hard decision of the
    reference function came out differently, for the free run on HIP's own volume.
    'diagnostic': the unconditioned network of rounds 1-3 (logits +-2600; the fp32 ORACLE is beyond 1e-3 px of its own
    float64 evaluation on 15-19 % of the pixels, profiles/r03_parity_config4_fp64_triangulation.json): figures
    recorded, EPE bar and a defect bound (a real defect shows up at 1e-2..1 px) asserted; its float64 triangulation
    runs under DV_FULL_PARITY=1."""
    import json
    import os
    m, sd, batch, fl, fr = _setup(kind)
    raw = kind == "conditioned"
    full = os.environ.get("DV_FULL_PARITY") == "1"
    fl0, fr0 = one(fl, 0), one(fr, 0)
    with torch.no_grad():
        vol_d = m.fused_volume({k: v[:1] for k, v in fl.items()}, {k: v[:1] for k, v in fr.items()})
        asd = m.encode_disparity(batch["disp"][:1])
    vol = P.fused_volume(fl0, fr0, sd)
    assert vol_d.shape == vol.shape == (1, 32, 48, 96, 312)
    assert rel(vol_d, vol) < 2e-5
    used, gt = batch["used"][:1].cpu(), batch["gt"][:1].cpu()
    orc = P.PCWDiffusionOracle(sd)
    final_o, stack_o, trace = LP.oracle_trajectory(orc, vol, used, asd.cpu(), 11, fl0, fr0)
    dl, dr = {k: v[:1] for k, v in fl.items()}, {k: v[:1] for k, v in fr.items()}
    # teacher forced = every input of the step is the oracle's, the volume included (its own parity is the assertion
    # above); the free run is the chain on HIP's own volume
    vol_in = vol.to(DEV) if raw else vol_d
    tf = LP.teacher_forced(m, trace, vol_in, batch["used"][:1], used, gt, features_left=dl, features_right=dr)
    fr_ = LP.free_run(m, trace, stack_o, final_o, vol_d, batch["used"][:1], asd, gt, 11, dl, dr)
    keys = ("step", "mean_abs_px", "frac_gt_1e-3", "max_px", "epe_delta", "unc_mean_px", "flips_mask_zero")
    report = {"network": kind, "teacher_forced": [{k: s.get(k) for k in keys} for s in tf],
              "free_run": [{k: s.get(k) for k in keys} for s in fr_["steps"]], "free_run_final": fr_["final"]}
    print(json.dumps(report))
    flips = sum(s["flips_mask_zero"] for s in fr_["steps"])
    # The step taken apart at the one hard decision inside it (the 0.999 validity threshold of `warp`, submodule.py:
    # 170-174; oracle/loop_parity.py::pcw_teacher_forced_split): the 3-D stack + regression and the 2-D refinement are each
    # held to the RAW bars as functions of the oracle's inputs; the composite step is held to them whenever the two
    # validity masks agree (one flipped pixel moves 2 % of this image by up to 0.1 px through the +-61-pixel receptive
    # field of the dilated stack -- in the reference exactly as here).
    split = LP.pcw_teacher_forced_split(m, trace, vol_in, batch["used"][:1], dl, dr) if raw else []
    report["teacher_forced_split"] = split
    print(json.dumps(split))
    for s in split:
        assert s["pred3"]["frac_gt_1e-3"] <= LP.BAR_FRAC and s["pred3"]["mean_abs_px"] < 2e-4, s
        assert s["refine"]["frac_gt_1e-3"] <= LP.BAR_FRAC and s["refine"]["mean_abs_px"] < 1e-4, s
    warp_flips = sum(s["warp_mask_flips"] for s in split)
    for i, s in enumerate(tf):
        assert s["epe_delta"] < LP.BAR_EPE, s
        assert s["mean_abs_px"] < LP.BAR_PX, s
        if raw and split[i]["warp_mask_flips"] == 0:
            assert s["frac_gt_1e-3"] <= LP.BAR_FRAC, s            # the contract's pixel figure: all pixels, unscaled
        elif raw:      # a validity decision differs: the disturbance is bounded by its receptive field (123 x 123 pixels)
            assert s["frac_gt_1e-3"] <= split[i]["warp_mask_flips"] * 123 * 123 / float(H * W) + LP.BAR_FRAC and s["max_px"] < 0.5, s
        else:
            assert s["frac_gt_1e-3"] < 0.05 and s["max_px"] < 0.5, s
    # the same steps against a float64 evaluation of the reference's function (weights and activations, 3-D stack and
    # 2-D refinement), RAW figures: HIP no further from float64 than the fp32 reference path is (the float64 oracle
    # costs ~40 s of CPU per step)
    # Default run: ONE float64 step (step 1) so that every driver run shows HIP as close to float64 as the fp32 oracle
    # is, on this network too; DV_FULL_PARITY=1: all three.
    tri = None
    if True:
        sd64 = {k: (v.double() if v.is_floating_point() and not k.startswith("time_embedding") else v) for k, v in sd.items()}
        f64 = lambda feats: {k: v.double() for k, v in feats.items()}
        tri = LP.teacher_forced_vs_fp64(m, orc, P.PCWDiffusionOracle(sd64), trace if full else trace[:1], vol, vol_in,
                                        batch["used"][:1], oracle_args=(f64(fl0), f64(fr0)), features_left=dl,
                                        features_right=dr)
        print(json.dumps(tri))
        for s in tri:
            h, o = s["hip_vs_fp64"], s["oracle32_vs_fp64"]
            assert h["mean_abs_px"] <= 1.5 * o["mean_abs_px"] + 2e-5, s
            assert h["frac_gt_1e-3"] <= 1.5 * o["frac_gt_1e-3"] + 1e-3, s
    if raw:    # the chain on HIP's own volume: its validity decisions against the oracle's
        chain = LP.pcw_teacher_forced_split(m, trace, vol_d, batch["used"][:1], dl, dr)
        warp_flips += sum(s["warp_mask_flips"] for s in chain)
        report["chain_on_hip_volume"] = chain
    os.makedirs("gpurun_out", exist_ok=True)
    with open(f"gpurun_out/parity_config4_{kind}.json", "w") as f:
        json.dump(dict(report, fp64_triangulation=tri, warp_mask_flips=warp_flips), f, indent=1)
    if flips == 0:
        assert fr_["final"]["epe_delta"] < LP.BAR_EPE, fr_["final"]
        if raw and warp_flips == 0:
            assert fr_["final"]["frac_gt_1e-3"] <= LP.BAR_FRAC, fr_["final"]
        for s in fr_["steps"]:
            assert s["epe_delta"] < LP.BAR_EPE and s["mean_abs_px"] < 2e-3, s
            if raw and warp_flips == 0:
                assert s["frac_gt_1e-3"] <= LP.BAR_FRAC, s


def test_batch_of_four_and_shard_invariance(setup):
    m, sd, batch, fl, fr = setup

    def run(lo, hi):
        tape = NoiseTape(21)

        def draw(kind, shape, dtype):
            return tape(kind, (B,) + tuple(shape[1:]), dtype)[lo:hi]

        sl = lambda d: {k: v[lo:hi] for k, v in d.items()}
        with torch.no_grad():
            vol = m.fused_volume(sl(fl), sl(fr))
            asd = m.encode_disparity(batch["disp"][lo:hi])
            return m.ddim_sample(vol, batch["used"][lo:hi], asd, sl(fl), sl(fr), noise=draw)[0]

    full = run(0, B)
    assert full.shape == (B, H, W) and bool(torch.isfinite(full).all())
    assert torch.equal(run(0, B), full)                        # bit-reproducible
    for lo, hi in ((2, 3), (0, 2)):
        assert torch.equal(run(lo, hi), full[lo:hi]), (lo, hi)  # what rank r of an N-GPU run computes for its slice


@pytest.mark.parametrize("dil", [1, 4, 16])
def test_fullres_refinement_conv_vs_torch(dil):
    """One 32 -> 32 layer of refinenet_version3 (KITTI12/models/pwcnet_ddim.py:251-306) at 384 x 1248 against
    F.conv2d on the GPU (MIOpen, fp32): partial tiles in x (1248 = 19.5 x 64) and the banded staging of d = 16."""
    g = _gen(5, f"rf{dil}")
    x = torch.randn(1, 32, H, W, generator=g).to(DEV)
    w = (torch.randn(32, 32, 3, 3, generator=g) * 0.06).to(DEV)
    plan = S.Conv2dPlan(w, None, dilation=dil, act=S.ACT_NONE)
    ref = F.conv2d(x, w, None, 1, dil, dil)
    assert rel(plan(x), ref.cpu()) < 1e-5
