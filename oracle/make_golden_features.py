"""Golden vectors of the two 2-D feature CNNs from the imported reference classes: SceneFlow
`feature_extraction` (models/acv_ddim.py:14-53) and KITTI12 `feature_extraction(concat_feature=True)`
(models/pwcnet_ddim.py:12-128), eval mode, synthetic weights, a 32x64 image.  Build container only:
    PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden_features.py"""
import os
import sys
import warnings
from pathlib import Path

import numpy as np
import torch

REPO = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(REPO))
from diffuvolume_amd.synth import _gen, synth_state_dict  # noqa: E402

warnings.filterwarnings("ignore")
OUT = REPO / "tests" / "golden"


def ref_models(flavour):
    for k in [k for k in sys.modules if k == "models" or k.startswith("models.")]:
        del sys.modules[k]
    sys.path[:] = [p for p in sys.path if not p.startswith("/root/reference/")]
    sys.path.insert(0, f"/root/reference/{flavour}")
    os.chdir(f"/root/reference/{flavour}")


def main():
    torch.set_num_threads(8)
    torch.Tensor.cuda = lambda self, *a, **k: self
    x = torch.randn(1, 3, 32, 64, generator=_gen(93, "img")) * 0.05
    arrays = {"x": x.numpy(), "seed": 6}
    # SceneFlow
    ref_models("SceneFlow")
    from models.acv_ddim import feature_extraction as RefACV
    import diffuvolume_amd as dv
    mine = dv.ACVNet_DDIM(192, False, False).feature_extraction
    ref = RefACV().eval()
    assert list(ref.state_dict().keys()) == list(mine.state_dict().keys())
    ref.load_state_dict(synth_state_dict(mine.state_dict(), seed=6), strict=True)
    with torch.no_grad():
        y = ref(x)
    arrays["acv_gwc_feature"] = y["gwc_feature"].numpy()
    print("  acv gwc_feature", tuple(y["gwc_feature"].shape), float(y["gwc_feature"].abs().mean()))
    # KITTI12
    ref_models("KITTI12")
    from models.pwcnet_ddim import feature_extraction as RefPCW
    from diffuvolume_amd.pwcnet_ddim import FeatureExtraction
    mine = FeatureExtraction(True, 12)
    ref = RefPCW(concat_feature=True, concat_feature_channel=12).eval()
    assert list(ref.state_dict().keys()) == list(mine.state_dict().keys())
    ref.load_state_dict(synth_state_dict(mine.state_dict(), seed=6), strict=True)
    with torch.no_grad():
        y = ref(x)
    for k, v in y.items():
        arrays[f"pcw_{k}"] = v.numpy()
        print(f"  pcw {k}", tuple(v.shape), float(v.abs().mean()))
    np.savez_compressed(OUT / "feature_cnns.npz", **arrays)
    print(f"  feature_cnns.npz  {(OUT / 'feature_cnns.npz').stat().st_size / 1024:.1f} KB")


if __name__ == "__main__":
    main()
