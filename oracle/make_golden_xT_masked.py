"""Golden vector for the masked x_T of ACVNet_DDIM.forward (SceneFlow/models/acv_ddim.py:403-419 with `mask_gt` given --
None at every call site of the reference, but part of the signature): the reference's own forward is run on a small pair
with `ddim_sample` replaced by a spy that records the `disp_volume_final` it is handed.
Build container only:  PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden_xT_masked.py"""
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
torch.Tensor.cuda = lambda self, *a, **k: self
sys.path.insert(0, "/root/reference/SceneFlow")
os.chdir("/root/reference/SceneFlow")
from models import __models__ as REF_MODELS  # noqa: E402
from diffuvolume_amd import ACVNet_DDIM  # noqa: E402

ref = REF_MODELS["acvnet_ddim"](192, False, False).eval()
ref.load_state_dict(synth_state_dict(ACVNet_DDIM(192).state_dict(), seed=3, logit_gain=8.0), strict=True)
batch = synth_stereo_batch(2, 64, 128, seed=43, shifts=(8, 20))
g = torch.Generator().manual_seed(5)
disp = batch["disp"].clone()
disp[0, 0, 0, :4] = torch.tensor([0.0, 46.9, 47.0, 47.75])             # the encoder's edge cases
mask_gt = (torch.rand(2, 1, 16, 32, generator=g) > 0.3).float()
seen = {}


def spy(volume, used, asd):
    seen["x_T"] = asd.clone()
    return used, None


ref.ddim_sample = spy
with torch.no_grad():
    ref(batch["left"], batch["right"], batch["used"], disp, mask_gt)
    x_masked = seen["x_T"]
    ref(batch["left"], batch["right"], batch["used"], disp, None)
    x_plain = seen["x_T"]
np.savez_compressed(REPO / "tests/golden/acv_xT_masked.npz", disp=disp.numpy(), mask_gt=mask_gt.numpy(),
                    x_T_masked=x_masked.numpy(), x_T=x_plain.numpy())
print("acv_xT_masked.npz", tuple(x_masked.shape), float((x_masked != x_plain).float().mean()))
