"""Golden vector for the origin ACVNet eval forward (SceneFlow/models/acv.py:168-260) from the imported
reference.  Build container only:  PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden_acv_origin.py"""
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
from diffuvolume_amd import ACVNet  # noqa: E402

ref = REF_MODELS["acvnet"](192, False, False).eval()
ref.load_state_dict(synth_state_dict(ACVNet(192).state_dict(), seed=3, logit_gain=8.0), strict=True)
batch = synth_stereo_batch(2, 64, 128, seed=42, shifts=(8, 20))
with torch.no_grad():
    pred = ref(batch["left"], batch["right"])[-1]
# the same weights through the attention-only variant (acv.py:246-252: the regression of the attention logits)
ref_att = REF_MODELS["acvnet"](192, True, False).eval()
ref_att.load_state_dict(ref.state_dict(), strict=True)
with torch.no_grad():
    pred_att = ref_att(batch["left"], batch["right"])[-1]
np.savez_compressed(REPO / "tests/golden/acv_origin_forward.npz", stereo_seed=42, pred=pred.numpy(), pred_attention=pred_att.numpy())
print("acv_origin_forward.npz", tuple(pred.shape), float(pred.min()), float(pred.max()), float(pred_att.min()), float(pred_att.max()))
