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
                                   (2, 32, 30, 48, 6, 16), (2, 8, 7, 12, 4, 12), (3, 16, 1, 16, 3, 20)])
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
    assert rel(vol_d, vol) < 1e-6 and S.volume_factors(vol_d) is not None
    assert rel(S.volume_factors(vol_d).p_att, torch.softmax(att, dim=2)[:, 0]) < 1e-6
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
    # (odd Cout -- shapes 6 and 7 -- has a one-channel last pair: a stray second-channel store would land in the next
    # batch item's channel 0 and show up in `rel(out, ref)` above)
    # the factor handle alone (nothing of the volume written) gives the same bits, and materialises the same tensor
    lazy = dv.build_concat_attention_volume(L.to(DEV), R.to(DEV), att.to(DEV), d, lazy=True)
    assert isinstance(lazy, dv.AttentionConcatVolume) and tuple(lazy.shape) == tuple(vol.shape) and lazy._tensor is None
    assert plan.applies(lazy) and torch.equal(plan(lazy, noise.to(DEV)), out) and lazy._tensor is None
    assert torch.equal(lazy.tensor(), vol_d)


def _layer(seed, b=2, c=32, cout=32, d=48, h=6, w=40):
    g = _gen(seed, "stale")
    L, R = torch.randn(b, c, h, w, generator=g), torch.randn(b, c, h, w, generator=g)
    att = torch.randn(b, 1, d, h, w, generator=g) * 2
    noise = torch.rand(b, d, h, w, generator=g)
    wt = torch.randn(cout, 2 * c, 3, 3, 3, generator=g) * (2.0 / (27 * cout)) ** 0.5
    bn = (torch.rand(cout, generator=g) * 0.4 + 0.8, torch.randn(cout, generator=g) * 0.1,
          torch.randn(cout, generator=g) * 0.1, torch.rand(cout, generator=g) + 0.5)

    def ref(vol):
        y = F.conv3d(vol * noise.unsqueeze(1), wt, None, 1, 1)
        return torch.relu(F.batch_norm(y, bn[2], bn[3], bn[0], bn[1], False, 0.0, 1e-5))
    return L, R, att, noise, wt, bn, ref


def _first_layer(model_like_plan, generic, vol, noise):
    """ACVNet_DDIM._aggregate's choice (acv_ddim.py: rank-1 layer when the volume's factors are valid)."""
    return model_like_plan(vol, noise) if model_like_plan.applies(vol) else generic(vol, in_scale=noise)


def test_volume_edited_in_place_falls_back_to_the_generic_layer():
    """A caller that scales / masks the volume it got from build_concat_attention_volume (``vol.mul_(m)``,
    ``vol[:, :, 0] = 0``) must get the convolution of the EDITED tensor: the ride-along factors are void."""
    L, R, att, noise, wt, bn, ref = _layer(301)
    plan = S.Rank1FilterPlan(wt.to(DEV), tuple(t.to(DEV) for t in bn), act=S.ACT_RELU)
    gen = S.Conv3dPlan(wt.to(DEV), tuple(t.to(DEV) for t in bn), stride=1, act=S.ACT_RELU)
    for edit in ("mul_", "setitem", "copy_"):
        vol_d = dv.build_concat_attention_volume(L.to(DEV), R.to(DEV), att.to(DEV), 48)
        assert plan.applies(vol_d)
        m = (torch.rand(1, 1, 48, 6, 40, generator=_gen(302, edit)) + 0.5).to(DEV)
        if edit == "mul_":
            vol_d.mul_(m)
        elif edit == "setitem":
            vol_d[:, :, 0] = 0
        else:
            vol_d.copy_(vol_d * m)
        assert not plan.applies(vol_d) and S.volume_factors(vol_d) is None
        out = _first_layer(plan, gen, vol_d, noise.to(DEV))
        want = ref(vol_d.cpu())                                          # the oracle's statement on the MUTATED tensor
        assert rel(out, want) < 1e-5, (edit, rel(out, want))
        stale = plan(dv.build_concat_attention_volume(L.to(DEV), R.to(DEV), att.to(DEV), 48), noise.to(DEV))
        assert rel(stale, want) > 1e-3                                   # (the old factors would have been wrong)


def test_feature_buffers_reused_after_the_build_do_not_reach_the_factors():
    """A caller that overwrites its feature buffers for the next pair between the build and ``ddim_sample`` (the
    rank-1 tables are built lazily, on the first step): the factors are private copies taken at build time."""
    L, R, att, noise, wt, bn, ref = _layer(303)
    plan = S.Rank1FilterPlan(wt.to(DEV), tuple(t.to(DEV) for t in bn), act=S.ACT_RELU)
    Ld, Rd = L.to(DEV), R.to(DEV)
    for lazy in (False, True):
        Ld.copy_(L), Rd.copy_(R)
        vol_d = dv.build_concat_attention_volume(Ld, Rd, att.to(DEV), 48, lazy=lazy)
        want = ref(O.attention_concat_volume(att, O.build_concat_volume(L, R, 48)))
        Ld.normal_(), Rd.zero_()                                         # next pair's features land in the same buffers
        Ld.data.mul_(3.0)                                                # (`.data` edits do not even bump _version)
        assert plan.applies(vol_d)
        out = plan(vol_d, noise.to(DEV))
        assert rel(out, want) < 1e-5, (lazy, rel(out, want))


def test_ddim_sample_on_a_mutated_volume_matches_the_oracle(acv_state_dict):
    """The public reference API ``ddim_sample(volume, used, asd)`` on a volume edited in place: the whole loop must be
    the loop of the EDITED tensor (oracle on the mutated volume), and the lazy handle must agree with the tensor."""
    from oracle import loop_parity as LP
    m = dv.ACVNet_DDIM(192, False, False)
    m.load_state_dict(acv_state_dict, strict=True)
    m = m.to(DEV).eval()
    g = _gen(304, "loop")
    b, h, w = 1, 8, 16
    L, R = torch.randn(b, 32, h, w, generator=g), torch.randn(b, 32, h, w, generator=g)
    att = torch.randn(b, 1, 48, h, w, generator=g) * 2
    used = torch.rand(b, 4 * h, 4 * w, generator=g) * 100 + 10
    dq = F.interpolate(used.unsqueeze(1), scale_factor=0.25, mode="bilinear") / 4
    mask = (torch.rand(1, 1, 48, h, w, generator=g) > 0.3).float()
    orc = O.ACVDiffusionOracle(acv_state_dict, sampling_timesteps=5)
    vol_ref = O.attention_concat_volume(att, O.build_concat_volume(L, R, 48)) * mask
    x_T = orc.encode_x_T(dq)
    final_o, stack_o, trace = LP.oracle_trajectory(orc, vol_ref, used, x_T, 5)
    vol_d = dv.build_concat_attention_volume(L.to(DEV), R.to(DEV), att.to(DEV), 48)
    vol_d.mul_(mask.to(DEV))
    tf = LP.teacher_forced(m, trace, vol_d, used.to(DEV), used, used)
    assert all(s["frac_gt_bar"] <= LP.BAR_FRAC and s["mean_abs_px"] < 2e-4 for s in tf), tf
    # and the unedited volume: handle and tensor take the same (rank-1) path, bit for bit
    t1 = LP.teacher_forced(m, trace, dv.build_concat_attention_volume(L.to(DEV), R.to(DEV), att.to(DEV), 48), used.to(DEV), used, used)
    t2 = LP.teacher_forced(m, trace, dv.build_concat_attention_volume(L.to(DEV), R.to(DEV), att.to(DEV), 48, lazy=True), used.to(DEV), used, used)
    assert [s["mean_abs_px"] for s in t1] == [s["mean_abs_px"] for s in t2]
    assert t1[0]["mean_abs_px"] > 10 * tf[0]["mean_abs_px"]             # (the unmasked loop is a different function)


@pytest.mark.parametrize("shape", [(2, 32, 864, 16, 60), (1, 8, 162, 5, 7), (3, 30, 27, 3, 21), (1, 32, 864, 128, 240)])
def test_table_convolution_vs_torch(shape):
    """The per-pair table build (csrc/pointwise_expand.hip): a 1x1 convolution from <= 32 input channels to 27 * Cout
    output channels against F.conv2d -- pixel counts that are not multiples of 4 / 64, channel counts that are not
    multiples of 4 / 16, the bench size."""
    b, cin, cout, h, w = shape
    g = _gen(221, str(shape))
    x = torch.randn(b, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, 1, 1, generator=g) * cin ** -0.5
    ref = F.conv2d(x[:1].double(), wt.double()).float() if h * w > 10000 else F.conv2d(x, wt)
    out = S.PointwiseExpandPlan(wt.to(DEV))(x.to(DEV))
    assert out.shape == (b, cout, h, w)
    assert rel(out[:ref.shape[0]], ref) < 2e-6, rel(out[:ref.shape[0]], ref)


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
    _, xs0, d0, _ = m.model_predictions(dv.build_concat_attention_volume(L, R, att, 48, lazy=True), x_t, t)
    assert torch.equal(d0, m.model_predictions(vol, x_t, t)[2])            # the handle is the same path
    _, xs1, d1, _ = m.model_predictions(vol, x_t, t)
    _, xs2, d2, _ = m.model_predictions(vol.clone(), x_t, t)                # no factors: generic convolution
    dd = (d1 - d2).abs()
    assert float((dd > 1e-3).float().mean()) <= 1e-3 and float(dd.mean()) < 1e-4, (float(dd.mean()), float(dd.max()))
    assert AD.RANK1_FILTER is True


def test_factor_handle_without_the_rank1_layer_is_materialised(acv_state_dict):
    """With the rank-1 layer switched off (the split-fp16 precision does that too) the factor handle is materialised once
    and the generic first layer runs on the tensor: same result as handing over the tensor."""
    from diffuvolume_amd import acv_ddim as AD
    g = _gen(231, "off")
    b, h, w = 1, 8, 20
    L, R = torch.randn(b, 32, h, w, generator=g).to(DEV), torch.randn(b, 32, h, w, generator=g).to(DEV)
    att = (torch.randn(b, 1, 48, h, w, generator=g) * 2).to(DEV)
    x_t = torch.randn(b, 48, h, w, generator=g).to(DEV)
    t = torch.full((b,), 999, dtype=torch.long, device=DEV)
    AD.RANK1_FILTER = False
    try:
        m = dv.ACVNet_DDIM(192, False, False)
        m.load_state_dict(acv_state_dict, strict=True)
        m = m.to(DEV).eval()
        assert m.prepare().dres0_rank1 is None
        handle = dv.build_concat_attention_volume(L, R, att, 48, lazy=True)
        d_handle = m.model_predictions(handle, x_t, t)[2]
        assert handle._tensor is not None
        d_tensor = m.model_predictions(dv.build_concat_attention_volume(L, R, att, 48), x_t, t)[2]
        assert torch.equal(d_handle, d_tensor)
    finally:
        AD.RANK1_FILTER = True
