"""Host-side logic that needs no GPU: state_dict compatibility, schedule, sharding, synthetic
data determinism, metric bookkeeping, and the refusal to run the hot path on the CPU."""
import math

import pytest
import torch

from conftest import load_golden
from diffuvolume_amd import ACVNet_DDIM, DiffuVolumeError, __models__
from diffuvolume_amd import distributed as D
from diffuvolume_amd import metrics as M
from diffuvolume_amd.synth import NoiseTape, synth_state_dict, synth_stereo_batch


@pytest.fixture(scope="module")
def model():
    return ACVNet_DDIM(192, False, False)


def test_registry_and_constructor(model):
    assert __models__["acvnet_ddim"] is ACVNet_DDIM
    assert model.num_groups == 40 and model.concat_channels == 32 and model.sampling_timesteps == 5
    with pytest.raises(ValueError):
        ACVNet_DDIM(128)
    with pytest.raises(ValueError):
        ACVNet_DDIM(192, sampling_timesteps=20)                    # needs its own ensemble weights
    m20 = ACVNet_DDIM(192, sampling_timesteps=20, ensemble_cof=[0.5] + [0.0] * 19 + [0.5])
    assert len(m20._time_pairs()) == 20 and m20._time_pairs()[0] == (999, 949)


def test_state_dict_layout(model):
    sd = model.state_dict()
    assert len(sd) == 579                                               # SURVEY section 5 (checkpoint row)
    assert sum(p.numel() for p in model.parameters()) == 7229216        # README.md:108 "7.23 M"
    assert tuple(sd["dres0.0.0.weight"].shape) == (32, 64, 3, 3, 3)
    assert tuple(sd["dres2.attention_block.qkv_3d.weight"].shape) == (384, 128)
    assert tuple(sd["dres2.conv5.0.weight"].shape) == (128, 64, 3, 3, 3)  # ConvTranspose3d: [Cin, Cout, ...]
    assert tuple(sd["time_embedding.time_mlp.1.weight"].shape) == (192, 48)
    for name in ("betas", "alphas_cumprod", "sqrt_recip_alphas_cumprod", "posterior_mean_coef2"):
        assert sd[name].dtype == torch.float64 and sd[name].shape == (1000,)
    # DataParallel checkpoints carry a "module." prefix (SceneFlow/main.py:118-121)
    wrapped = {"module." + k: v for k, v in sd.items()}
    fresh = ACVNet_DDIM(192)
    fresh.load_state_dict({k[len("module."):]: v for k, v in wrapped.items()}, strict=True)


def test_schedule_constants(model):
    g = load_golden("encoder_schedule")
    torch.testing.assert_close(model.alphas_cumprod, g["alphas_cumprod"], rtol=1e-13, atol=0)
    torch.testing.assert_close(model.sqrt_recip_alphas_cumprod, g["sqrt_recip"], rtol=1e-13, atol=0)
    assert model._time_pairs() == [(999, 799), (799, 599), (599, 399), (399, 199), (199, -1)]
    for t, val in ((999, 2.4288e-9), (799, 0.094046), (599, 0.340810), (399, 0.647478), (199, 0.898706), (0, 0.999959)):
        assert math.isclose(float(model.alphas_cumprod[t]), val, rel_tol=2e-4)


def test_time_embedding_matches_reference(model, acv_state_dict):
    g = load_golden("time_shift")
    m = ACVNet_DDIM(192)
    m.load_state_dict(acv_state_dict, strict=True)
    with torch.no_grad():
        torch.testing.assert_close(m.time_embedding.shift(g["t"]), g["shift"], atol=1e-6, rtol=1e-5)
        torch.testing.assert_close(m.time_embedding(g["noisy"], g["t"]), g["out"], atol=1e-6, rtol=1e-5)


def test_feature_extractor_shapes(model):
    """The parameter containers are plain nn.Modules (1/4 resolution, 64 + 128 + 128 channels); the module's own
    forward runs on the HIP kernels and refuses CPU tensors like the rest of the path."""
    model.eval()
    fe = model.feature_extraction
    with torch.no_grad():
        l2 = fe.layer2(fe.layer1(fe.firstconv(torch.zeros(1, 3, 32, 64))))
        l4 = fe.layer4(fe.layer3(l2))
    assert tuple(l2.shape) == (1, 64, 8, 16) and tuple(l4.shape) == (1, 128, 8, 16)
    with pytest.raises(DiffuVolumeError):
        fe(torch.zeros(1, 3, 32, 64))


def test_hot_path_refuses_cpu(model):
    """There is no CPU / eager fallback: a model that is not on the GPU fails loudly."""
    model.eval()
    with pytest.raises(DiffuVolumeError):
        model(torch.zeros(1, 3, 64, 128), torch.zeros(1, 3, 64, 128), torch.zeros(1, 64, 128),
              torch.zeros(1, 1, 16, 32))
    with pytest.raises(DiffuVolumeError):
        from diffuvolume_amd import build_gwc_volume
        build_gwc_volume(torch.zeros(1, 8, 2, 4), torch.zeros(1, 8, 2, 4), 2, 4)
    model.train()
    with pytest.raises(NotImplementedError):
        model(torch.zeros(1, 3, 64, 128), torch.zeros(1, 3, 64, 128), torch.zeros(1, 64, 128),
              torch.zeros(1, 1, 16, 32))
    model.eval()


def test_update_glue_refuses_cpu_and_dispatches_autograd_explicitly():
    """IGEV's update-block glue (update.py:96-102): a CPU tensor raises (no silent F.avg_pool2d / F.interpolate fallback);
    only a tensor that asks for gradients takes the differentiable torch expression."""
    from diffuvolume_amd import update as U
    x = torch.zeros(1, 2, 8, 8)
    with pytest.raises(DiffuVolumeError):
        U.pool2x(x)
    with pytest.raises(DiffuVolumeError):
        U.interp(x, torch.zeros(1, 2, 16, 16))
    with pytest.raises(TypeError):
        U._hip_ok(_cuda_like(x.double()), "probe")
    xg = x.clone().requires_grad_(True)
    assert U.pool2x(xg).requires_grad and U.interp(xg, torch.zeros(1, 2, 16, 16)).shape[-1] == 16


class _CudaLike(torch.Tensor):
    """A CPU tensor that claims to be on the GPU (dtype check of `_hip_ok` without a device)."""
    @property
    def is_cuda(self):
        return True


def _cuda_like(t):
    return t.as_subclass(_CudaLike)


def test_product_code_never_imports_the_oracle():
    import pathlib
    root = pathlib.Path(__file__).resolve().parents[1] / "diffuvolume_amd"
    for f in root.rglob("*.py"):
        text = f.read_text()
        assert "import oracle" not in text and "from oracle" not in text, f


def test_synth_is_deterministic(model):
    a = synth_state_dict(model.state_dict(), seed=3)
    b = synth_state_dict(ACVNet_DDIM(192).state_dict(), seed=3)   # independent of construction RNG
    assert all(torch.equal(a[k], b[k]) for k in a)
    c = synth_state_dict(model.state_dict(), seed=4)
    assert not torch.equal(a["dres0.0.0.weight"], c["dres0.0.0.weight"])
    assert float(a["dres0.0.1.running_var"].min()) >= 0.5           # non-trivial, positive BN statistics
    t1, t2 = NoiseTape(9), NoiseTape(9)
    assert torch.equal(t1("eps", (2, 3), torch.float32), t2("eps", (2, 3), torch.float32))
    assert not torch.equal(t1("eps", (2, 3), torch.float64), t2("fill", (2, 3), torch.float64))
    s = synth_stereo_batch(2, 32, 64, seed=1)
    assert s["disp"].shape == (2, 1, 8, 16) and float(s["used"].max()) <= 191


@pytest.mark.parametrize("n,world", [(64, 8), (8, 1), (10, 4), (3, 8), (0, 2)])
def test_shard_range_partitions(n, world):
    spans = [D.shard_range(n, r, world) for r in range(world)]
    assert spans[0][0] == 0 and spans[-1][1] == n
    assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
    sizes = [hi - lo for lo, hi in spans]
    assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        D.shard_range(4, 4, 4)


def test_per_image_metric_rules():
    """The reference's per-image semantics applied to the kernel's sums (metrics.py:22-41)."""
    sums = torch.tensor([[100., 200., 50., 10., 40., 20., 10., 0.],      # ratio 0.5  -> kept
                         [5., 200., 50., 1., 1., 1., 1., 0.],             # ratio 0.025 -> skipped
                         [100., 100., 0., 0., 0., 0., 0., 0.]],           # perfect image
                        dtype=torch.float64)
    vals, keep = M.per_image_values(sums)
    assert keep.tolist() == [True, False, True]
    assert vals[0].tolist() == [0.5, 0.1, 0.4, 0.2, 0.1]
    # two batches: the first is `sums` above (images 0 and 2 kept: EPE 0.5 and 0 -> 0.25), the second one image of EPE 3
    acc = M.MetricAccumulator("cpu")
    acc.update_sums(sums)
    acc.update_sums(torch.tensor([[10.0, 10.0, 30.0, 3.0, 2.0, 3.0, 4.0, 0.0]], dtype=torch.float64))
    assert acc.table().shape == (2, 6) and acc.table()[0, 5] == 2.0
    out = acc.reduce()
    assert abs(out["EPE"] - (0.5 * (sums[0, 2] / sums[0, 0] + sums[2, 2] / sums[2, 0]) + 3.0) / 2) < 1e-12   # AverageMeterDict: mean over batches
    assert abs(out["D1"] - ((0.1 + float(sums[2, 3] / sums[2, 0])) / 2 + 0.3) / 2) < 1e-12


def test_origin_and_pcw_state_dict_layouts():
    from diffuvolume_amd import ACVNet, PWCNet_ddim
    assert len(ACVNet(192).state_dict()) == 561                        # acvnet: no schedule buffers / time MLP
    pcw = PWCNet_ddim(192, True)
    sd = pcw.state_dict()
    assert len(sd) == 905 and sum(p.numel() for p in pcw.parameters()) == 35994800
    assert tuple(sd["combine1.conv7.0.weight"].shape) == (128, 128, 3, 3, 3)
    assert pcw._time_pairs() == [(999, 665), (665, 332), (332, -1)]
    # SceneFlow/models/__init__.py + KITTI12/models/__init__.py:5-9
    assert sorted(__models__) == ["acvnet", "acvnet_ddim", "gwcnet-g", "gwcnet-gc", "pwc_ddimgc"]
    assert len(__models__["gwcnet-gc"](192).state_dict()) == 887 and len(__models__["gwcnet-g"](192).state_dict()) == 859


def test_rendezvous_port_below_the_ephemeral_range():
    """distributed.free_port: a rendezvous port that no outgoing connection can be handed in the meantime."""
    import socket
    from diffuvolume_amd.distributed import free_port
    ports = {free_port() for _ in range(8)}
    assert all(20000 <= p < 30000 for p in ports)
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", next(iter(ports))))          # still free


def test_one_place_for_switches(monkeypatch):
    """The native library reads no environment variable (kernel choices are pinned through dv_*_set_* hooks), and every
    DV_* variable the Python side reads is listed in diffuvolume_amd/_env.py -- the list bench.py reports overrides from."""
    import re
    from pathlib import Path
    from diffuvolume_amd import _env
    root = Path(__file__).resolve().parents[1]
    for f in (root / "diffuvolume_amd" / "csrc").glob("*.hip"):
        assert "getenv" not in f.read_text(), f.name
    read = set()
    for f in list((root / "diffuvolume_amd").glob("*.py")) + [root / "bench.py"]:
        read |= set(re.findall(r"environ(?:\.get|\.setdefault)?\(\s*[\"'](DV_[A-Z0-9_]+)[\"']", f.read_text()))
        read |= set(re.findall(r"environ\[\s*[\"'](DV_[A-Z0-9_]+)[\"']\s*\]", f.read_text()))
    assert read <= set(_env.KNOBS), read - set(_env.KNOBS)
    for k in _env.KNOBS:
        monkeypatch.delenv(k, raising=False)
    assert _env.overrides() == {}
    monkeypatch.setenv("DV_S2PP", "0")
    assert _env.overrides() == {"DV_S2PP": "0"}
