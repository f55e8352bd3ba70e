"""Golden vectors for IGEV's once-per-pair cost-volume front, from the imported REFERENCE modules
(KITTI15/core/igev_stereo_ddim.py `hourglass` :24-91; core/submodule.py BasicConv / FeatureAtt /
build_gwc_volume / disparity_regression) wired as IGEVStereo_ddim.forward :377-383 does.  The full class cannot
be constructed here (timm pretrained backbone), so the volume-side modules are instantiated on their own
under the reference's attribute names.  Build container only:
    PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden_igev_volume.py"""
import sys
import types
import warnings
from pathlib import Path

import numpy as np
import torch
import torch.nn.functional as F
from torch import nn

REPO = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(REPO))
from diffuvolume_amd.synth import _gen, synth_state_dict  # noqa: E402

warnings.filterwarnings("ignore")
sys.modules.setdefault("timm", types.ModuleType("timm"))
oe = types.ModuleType("opt_einsum")
oe.contract = torch.einsum
sys.modules.setdefault("opt_einsum", oe)
sys.path.insert(0, "/root/reference/KITTI15")
import core.igev_stereo_ddim as R  # noqa: E402
from core.submodule import BasicConv, FeatureAtt, build_gwc_volume, disparity_regression  # noqa: E402


class VolumeSide(nn.Module):
    """igev_stereo_ddim.py:196-199."""

    def __init__(self):
        super().__init__()
        self.corr_stem = BasicConv(8, 8, is_3d=True, kernel_size=3, stride=1, padding=1)
        self.corr_feature_att = FeatureAtt(8, 96)
        self.cost_agg = R.hourglass(8)
        self.classifier = nn.Conv3d(8, 1, 3, 1, 1, bias=False)


def igev_inputs(seed, b, h, w, shift=3):
    """match features (96 ch, 1/4 res) with a correlation ridge + the 4-level left feature pyramid."""
    ml = torch.randn(b, 96, h, w, generator=_gen(seed, "ml"))
    mr = torch.roll(ml, -shift, dims=-1) + 0.1 * torch.randn(b, 96, h, w, generator=_gen(seed, "mr"))
    feats = [torch.randn(b, c, h // s, w // s, generator=_gen(seed, f"feat{i}"))
             for i, (c, s) in enumerate(((96, 1), (64, 2), (192, 4), (160, 8)))]
    return ml, mr, feats


def main():
    m = VolumeSide().eval()
    sd = synth_state_dict(m.state_dict(), seed=91, logit_gain=60.0)
    m.load_state_dict(sd, strict=True)
    out = {}
    with torch.no_grad():
        # (1) hourglass(8) alone on a small ragged-ish volume
        x = torch.randn(2, 8, 16, 16, 24, generator=_gen(92, "x"))
        _, _, feats = igev_inputs(92, 2, 16, 24)
        out["hg_y"] = m.cost_agg(x, feats).numpy()
        # (2) the whole front: gwc -> stem -> att -> hourglass -> classifier -> softmax -> regression
        ml, mr, feats = igev_inputs(93, 1, 8, 32)
        gwc = build_gwc_volume(ml, mr, 192 // 4, 8)
        gwc = m.corr_feature_att(m.corr_stem(gwc), feats[0])
        geo = m.cost_agg(gwc, feats)
        prob = F.softmax(m.classifier(geo).squeeze(1), dim=1)
        init = disparity_regression(prob, 192 // 4)
        out["geo"], out["init_disp"] = geo.numpy(), init.numpy()
    np.savez_compressed(REPO / "tests/golden/igev_volume.npz", sd_seed=91, logit_gain=60.0, hg_seed=92, front_seed=93,
                        **out)
    print({k: v.shape for k, v in out.items()}, float(init.min()), float(init.max()), float(np.abs(out["geo"]).max()))


if __name__ == "__main__":
    main()
