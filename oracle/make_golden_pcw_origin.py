"""Golden vector of the ORIGIN PCWNet (KITTI12/models/pwcnet.py:310-507; registry names gwcnet-g / gwcnet-gc,
models/__init__.py:5-9) from the imported reference class: state_dict keys checked one by one against this build's
`PWCNet`, eval forward on a 64x128 synthetic pair recorded.  Build container only:
    PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden_pcw_origin.py"""
import os
import sys
import warnings
from pathlib import Path

import numpy as np
import torch

REPO = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(REPO))
from diffuvolume_amd.synth import synth_state_dict, synth_stereo_batch  # noqa: E402

warnings.filterwarnings("ignore")
OUT = REPO / "tests" / "golden"


def main():
    torch.set_num_threads(8)
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.Tensor.get_device = lambda self: self.device                 # KITTI12 warp() (submodule.py:146)
    sys.path.insert(0, "/root/reference/KITTI12")
    os.chdir("/root/reference/KITTI12")
    from models import __models__ as REF_MODELS
    from diffuvolume_amd.pwcnet_ddim import PWCNet
    batch = synth_stereo_batch(1, 64, 128, seed=75, shifts=(8,))
    arrays = {"stereo_seed": 75}
    for name, concat in (("gwcnet-gc", True), ("gwcnet-g", False)):
        ref = REF_MODELS[name](192).eval()
        mine = PWCNet(192, concat)
        rk, mk = list(ref.state_dict().items()), list(mine.state_dict().items())
        assert [k for k, _ in rk] == [k for k, _ in mk], "state_dict keys / order differ from the reference class"
        assert all(a.shape == b.shape and a.dtype == b.dtype for (_, a), (_, b) in zip(rk, mk))
        sd = synth_state_dict(mine.state_dict(), seed=4, logit_gain=8.0, scale={"refinenet3.conv8.weight": 0.002})
        ref.load_state_dict(sd, strict=True)
        tag = name.replace("-", "_")
        arrays[f"{tag}_n_keys"] = len(rk)
        if not concat:
            # the reference's gwcnet-g cannot run: `hourglassup` is built for 64-channel volumes (40 groups + 2 x 12
            # concat channels, pwcnet.py:137-160) and its feature CNN returns `finetune_feature` only with
            # concat_feature=True (:122-129).  Recorded so that the test can hold this build to the same behaviour.
            try:
                with torch.no_grad():
                    ref(batch["left"], batch["right"])
                raise SystemExit("gwcnet-g forward unexpectedly worked in the reference")
            except (KeyError, RuntimeError) as e:
                print(f"  {name}: {len(rk)} keys, eval forward raises {type(e).__name__} in the reference")
                arrays[f"{tag}_forward_error"] = type(e).__name__
            continue
        with torch.no_grad():
            # untrained residual stacks blow the 2-D features up: calibrate the last 1x1 conv of every feature head to unit
            # output scale (factors stored with the fixture and re-applied by the test), as make_golden_pcw.py does
            feats = ref.feature_extraction(batch["left"])
            heads = {"gw1": "layer11.2.weight", "gw2": "gw2.2.weight", "gw3": "gw3.2.weight", "gw4": "gw4.2.weight"}
            if concat:
                heads.update(concat_feature1="lastconv.2.weight", concat_feature2="concat2.2.weight",
                             concat_feature3="concat3.2.weight", concat_feature4="concat4.2.weight")
            fscale = {"feature_extraction." + k: 1.0 / float(feats[f].std()) for f, k in heads.items()}
            fscale["feature_extraction.layer_refine.0.0.weight"] = 1.0 / float(feats["finetune_feature"].abs().mean() + 1)
            fscale["refinenet3.conv8.weight"] = 0.002
            sd = synth_state_dict(mine.state_dict(), seed=4, logit_gain=8.0, scale=fscale)
            ref.load_state_dict(sd, strict=True)
            fin, p3 = ref(batch["left"], batch["right"])
        print(f"  {name}: {len(rk)} keys, disp_finetune range {float(fin[0].min()):.2f}..{float(fin[0].max()):.2f}, "
              f"pred3 {float(p3[0].min()):.2f}..{float(p3[0].max()):.2f}")
        arrays.update({f"{tag}_disp_finetune": fin[0].numpy(), f"{tag}_pred3": p3[0].numpy(),
                       f"{tag}_scale_keys": np.array(list(fscale.keys())),
                       f"{tag}_scale_vals": np.array(list(fscale.values()), dtype=np.float64)})
    np.savez_compressed(OUT / "pcw_origin_forward.npz", **arrays)
    print(f"  pcw_origin_forward.npz  {(OUT / 'pcw_origin_forward.npz').stat().st_size / 1024:.1f} KB")


if __name__ == "__main__":
    main()
