"""Golden vectors for the origin IGEV-Stereo drop-in: the REFERENCE class itself (KITTI15/core/igev_stereo.py:91-221)
constructed here with `timm.create_model` stubbed to return synth.StubMobileNetV2 (the pretrained MobileNetV2 is not
available offline), loaded with synthetic weights, run through its own eval `forward` in both modes
(test_mode=True: evaluate_stereo.py:88; test_mode=False: initial disparity + every iteration's prediction).  Also checks
that the reference's state_dict and this build's have the same keys / shapes.
Build container only:  PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden_igev_origin.py"""
import contextlib
import io
import sys
import types
import warnings
from pathlib import Path

import numpy as np
import torch

REPO = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(REPO))
from diffuvolume_amd.synth import StubMobileNetV2, _gen, synth_state_dict  # noqa: E402

warnings.filterwarnings("ignore")
torch.Tensor.cuda = lambda self, *a, **k: self
timm = types.ModuleType("timm")
timm.create_model = lambda *a, **k: StubMobileNetV2()
sys.modules["timm"] = timm
oe = types.ModuleType("opt_einsum")
oe.contract = torch.einsum
sys.modules.setdefault("opt_einsum", oe)
sys.path.insert(0, "/root/reference/KITTI15")
import core.igev_stereo as R  # noqa: E402

ARGS = dict(hidden_dims=[128, 128, 128], n_gru_layers=3, n_downsample=2, corr_levels=2, corr_radius=4,
            slow_fast_gru=False, max_disp=192, mixed_precision=False, corr_implementation="reg", shared_backbone=False)
SEED, ITERS = 61, 5
SCALE = {"update_block.disp_head.conv2.weight": 0.05, "update_block.disp_head.conv2.bias": 0.0, "classifier.weight": 20.0}


def inputs(h=64, w=128):
    g = _gen(SEED, "igev_origin")
    img1 = torch.rand(1, 3, h, w, generator=g) * 255
    return img1, torch.roll(img1, -6, dims=-1)


def main():
    from diffuvolume_amd.igev_stereo import IGEVStereo
    from diffuvolume_amd.igev_stereo_ddim import Feature
    args = types.SimpleNamespace(**ARGS)
    with contextlib.redirect_stdout(io.StringIO()):
        ref = R.IGEVStereo(args).eval()
    mine = IGEVStereo(args, feature=Feature(StubMobileNetV2()))
    rs, ms = ref.state_dict(), mine.state_dict()
    assert list(rs.keys()) == list(ms.keys()), (set(rs) ^ set(ms))
    assert all(rs[k].shape == ms[k].shape and rs[k].dtype == ms[k].dtype for k in rs)
    sd = synth_state_dict(ms, seed=SEED, scale=SCALE)
    ref.load_state_dict(sd, strict=True)
    img1, img2 = inputs()
    with torch.no_grad():
        pred = ref(img1, img2, iters=ITERS, test_mode=True)
        init_disp, preds = ref(img1, img2, iters=ITERS, test_mode=False)
    assert len(preds) == ITERS and float((preds[-1] - pred).abs().max()) < 1e-4
    out = REPO / "tests/golden/igev_origin.npz"
    np.savez_compressed(out, seed=SEED, iters=ITERS, n_keys=len(rs), pred=pred.numpy(), init_disp=init_disp.numpy(),
                        preds=torch.cat(preds).numpy(), scale_keys=np.array(list(SCALE)), scale_vals=np.array(list(SCALE.values())))
    print(out.name, tuple(pred.shape), "range", float(pred.min()), float(pred.max()), "| init", tuple(init_disp.shape),
          "| iteration steps", [round(float((preds[i] - preds[i - 1]).abs().mean()), 3) for i in range(1, ITERS)], "|", len(rs), "keys")


if __name__ == "__main__":
    main()
