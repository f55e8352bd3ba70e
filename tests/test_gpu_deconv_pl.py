"""The persistent loader-wave form of the k3 transposed convolution (csrc/deconv3d_pl.hip; hourglass conv5 / conv6 with the
fused `redir`, SceneFlow/models/acv_ddim.py:74-92; KITTI12/models/pwcnet_ddim.py:131-205) against torch's fp32
`conv_transpose3d` on the CPU and against the one-tile kernel it replaces: <= 1e-5 of the layer's output scale (fp32
re-association only).  Also: the bits do not depend on how many blocks walk the tile list (test hook), a rerun and a shard
of a batch reproduce the batch's bits, shapes it does not take fall back, and the C ABI."""
import pytest
import torch

from diffuvolume_amd import _lib
from diffuvolume_amd import submodule as S
from diffuvolume_amd.synth import _gen

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
F = torch.nn.functional


def dev(t):
    return t.to(DEV)


def rel_err(a, b):
    return float((a.cpu().double() - b.double()).abs().max() / b.double().abs().max().clamp(min=1e-30))


def _bn(cout, g):
    return (torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g) * 0.1,
            torch.randn(cout, generator=g) * 0.1, torch.rand(cout, generator=g) + 0.5)


def _act(y, act):
    return {"relu": torch.relu(y), "mish": y * torch.tanh(F.softplus(y)), "leaky": F.leaky_relu(y, 0.01), "none": y}[act]


ACT = {"relu": S.ACT_RELU, "mish": S.ACT_MISH, "leaky": S.ACT_LEAKY, "none": S.ACT_NONE}


def _case(cfg, seed=71):
    cin, cout, cskip, dims, act, mode = cfg
    g = _gen(seed, str(cfg))
    x = torch.randn(dims[0], cin, *dims[1:], generator=g)
    w = torch.randn(cin, cout, 3, 3, 3, generator=g) * (2.0 / (27 * cin)) ** 0.5
    bnd = _bn(cout, g)
    odims = tuple(2 * d for d in dims[1:])
    y = F.batch_norm(F.conv_transpose3d(x, w, None, 2, 1, 1), bnd[2], bnd[3], bnd[0], bnd[1], False, 0.0, 1e-5)
    kw = {}
    if mode == "redir":
        wr = torch.randn(cout, cskip, 1, 1, 1, generator=g) * (1.0 / cskip) ** 0.5
        bnr = _bn(cout, g)
        skip = torch.randn(dims[0], cskip, *odims, generator=g)
        y = y + F.batch_norm(F.conv3d(skip, wr), bnr[2], bnr[3], bnr[0], bnr[1], False, 0.0, 1e-5)
        plan = S.Deconv3dPlan(dev(w), tuple(dev(t) for t in bnd), act=ACT[act], redir=(dev(wr), tuple(dev(t) for t in bnr)))
        kw = dict(skip=dev(skip))
    else:
        plan = S.Deconv3dPlan(dev(w), tuple(dev(t) for t in bnd), act=ACT[act])
        if mode == "res":
            res = torch.randn(dims[0], cout, *odims, generator=g)
            y = y + res
            kw = dict(residual=dev(res))
    return plan, dev(x), kw, _act(y, act)


# cin, cout, cskip, (B, D, H, W), activation, mode.  Exact and ragged x tiles (W % 32), odd H (half tiles), D = 1, one to three
# 32-channel output blocks (pairs = the channel blocks of a brick, or x neighbours), 1 .. 16 chunks, fewer skip chunks than
# input chunks, every activation, the plain / residual / fused-redir epilogues
CASES = [
    (64, 32, 32, (1, 4, 6, 32), "relu", "redir"),
    (128, 64, 64, (2, 3, 5, 20), "relu", "redir"),
    (64, 32, 32, (1, 2, 3, 36), "mish", "redir"),
    (64, 32, 32, (2, 4, 6, 120), "relu", "redir"),          # config 2's conv6 geometry along x (120 = 3.75 tiles)
    (128, 64, 64, (1, 3, 4, 60), "relu", "redir"),          # conv5's
    (16, 32, 8, (3, 2, 3, 8), "relu", "redir"),
    (16, 32, 4, (1, 7, 2, 68), "none", "redir"),            # one skip chunk, two input chunks
    (8, 32, 0, (1, 2, 4, 32), "relu", "plain"),
    (24, 32, 0, (1, 1, 1, 4), "none", "plain"),
    (64, 96, 0, (1, 2, 9, 44), "leaky", "plain"),           # three channel blocks: the last pair is half empty
    (32, 32, 0, (1, 5, 7, 40), "leaky", "res"),
    (64, 96, 0, (1, 2, 9, 44), "relu", "res"),
]


@pytest.mark.parametrize("cfg", CASES)
def test_persistent_vs_torch_and_one_tile(cfg):
    lib = _lib.load()
    cin, cout, cskip, dims, act, mode = cfg
    assert lib.dv_deconv3d_pl_supported(cin, cout, *dims[1:], cskip) == 1
    plan, x, kw, y_ref = _case(cfg)
    try:
        assert lib.dv_deconv3d_set_impl(2) == 0
        out = plan(x, **kw)
        assert out.shape == y_ref.shape
        assert rel_err(out, y_ref) < 1e-5
        # the grid must not matter: every output is summed in the same order whichever block computes its tile
        for cap in (1, 3, 8, 13):
            assert lib.dv_deconv3d_pl_set_max_blocks(cap) == 0
            poison = torch.full_like(out, float("nan"))
            del poison                                     # the next output lands on these bytes
            assert torch.equal(plan(x, **kw), out), f"grid of {cap} blocks changes the result"
        lib.dv_deconv3d_pl_set_max_blocks(0)
        assert lib.dv_deconv3d_set_impl(1) == 0
        one_tile = plan(x, **kw)
        assert rel_err(one_tile, y_ref) < 1e-5
        assert float((out - one_tile).abs().max() / one_tile.abs().max()) < 1e-5
        if mode != "redir":                                # same summation order without the skip k-steps
            assert torch.equal(out, one_tile)
    finally:
        lib.dv_deconv3d_pl_set_max_blocks(0)
        lib.dv_deconv3d_set_impl(0)


def test_default_choice_and_fallbacks():
    """impl 0: the persistent kernel where it takes the shape and there is no residual tensor; otherwise one-tile blocks."""
    lib = _lib.load()
    assert lib.dv_deconv3d_pl_supported(64, 32, 24, 64, 120, 32) == 1 and lib.dv_deconv3d_pl_supported(128, 64, 12, 32, 60, 64) == 1
    assert lib.dv_deconv3d_pl_supported(12, 32, 2, 4, 32, 0) == 0          # Cin % 8
    assert lib.dv_deconv3d_pl_supported(16, 16, 2, 4, 32, 0) == 0          # Cout % 32
    assert lib.dv_deconv3d_pl_supported(16, 32, 2, 4, 78, 0) == 0          # W % 4 (KITTI12's 78-wide level)
    assert lib.dv_deconv3d_pl_supported(16, 32, 2, 4, 32, 32) == 0         # more skip chunks than input chunks
    assert lib.dv_deconv3d_pl_supported(16, 32, 2, 4, 32, 6) == 0          # Cskip % 4
    assert lib.dv_deconv3d_set_impl(3) != 0 and lib.dv_deconv3d_pl_set_max_blocks(-1) != 0
    for cfg in [(12, 32, 0, (1, 2, 4, 32), "relu", "plain"), (16, 32, 0, (1, 2, 3, 10), "relu", "plain"),
                (32, 32, 32, (1, 3, 4, 10), "relu", "redir"), (16, 48, 0, (1, 2, 3, 8), "none", "res")]:
        plan, x, kw, y_ref = _case(cfg, seed=73)
        outs = []
        for impl in (0, 1, 2):
            lib.dv_deconv3d_set_impl(impl)
            outs.append(plan(x, **kw))
        lib.dv_deconv3d_set_impl(0)
        assert rel_err(outs[0], y_ref) < 1e-5
        assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])   # unsupported: all three are the one-tile kernel


def test_shard_invariant_and_reproducible():
    """Pairs [lo, hi) alone give the bits they have inside the batch (what the multi-GPU sharding rests on; the kernel choice
    never looks at the batch size), and a rerun gives the same bits."""
    cfg = (64, 32, 32, (4, 3, 6, 40), "relu", "redir")
    plan, x, kw, _ = _case(cfg, seed=79)
    full = plan(x, **kw)
    assert torch.equal(full, plan(x, **kw))
    for lo, hi in ((1, 3), (3, 4)):
        assert torch.equal(full[lo:hi], plan(x[lo:hi].contiguous(), skip=kw["skip"][lo:hi].contiguous()))


def test_c_abi():
    """Straight through the C ABI (raw pointers, caller-allocated output): both entry points reach the persistent kernel."""
    lib = _lib.load()
    g = _gen(83, "abi")
    cin, cout, d, h, w = 16, 32, 2, 4, 36
    x = torch.randn(1, cin, d, h, w, generator=g)
    wt = torch.randn(cin, cout, 3, 3, 3, generator=g) * 0.1
    bias = torch.randn(cout, generator=g)
    skip = torch.randn(1, 8, 2 * d, 2 * h, 2 * w, generator=g)
    rw = torch.randn(cout, 8, generator=g) * 0.2
    xd, wd, bd, sd, rd = dev(x), dev(wt), dev(bias), dev(skip), dev(rw)
    wp = torch.empty(lib.dv_deconv3d_packed_floats(cin, cout), dtype=torch.float32, device=DEV)
    out = torch.empty(1, cout, 2 * d, 2 * h, 2 * w, device=DEV)
    s = _lib.stream_ptr()
    assert lib.dv_deconv3d_pack_weights_f32(wd.data_ptr(), wp.data_ptr(), cin, cout, s) == 0
    y = F.conv_transpose3d(x, wt, None, 2, 1, 1) + bias.view(1, -1, 1, 1, 1)
    try:
        lib.dv_deconv3d_set_impl(2)
        assert lib.dv_deconv3d_k3s2_f32(xd.data_ptr(), wp.data_ptr(), None, bd.data_ptr(), None, out.data_ptr(), 1, cin, d, h, w,
                                        cout, S.ACT_NONE, s) == 0
        assert rel_err(out, y) < 1e-5
        assert lib.dv_deconv3d_k3s2_redir_f32(xd.data_ptr(), wp.data_ptr(), bd.data_ptr(), sd.data_ptr(), rd.data_ptr(),
                                              out.data_ptr(), 1, cin, d, h, w, cout, 8, S.ACT_RELU, s) == 0
        assert rel_err(out, torch.relu(y + F.conv3d(skip, rw.view(cout, 8, 1, 1, 1)))) < 1e-5
    finally:
        lib.dv_deconv3d_set_impl(0)
