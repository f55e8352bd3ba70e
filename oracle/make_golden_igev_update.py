"""Golden vectors for IGEV's recurrent update block from the imported REFERENCE class
(KITTI15/core/update.py BasicMultiUpdateBlock), two consecutive iterations.  Build container only:
    PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden_igev_update.py"""
import sys
import types
import warnings
from pathlib import Path

import numpy as np
import torch

REPO = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(REPO))
from diffuvolume_amd.synth import _gen, synth_state_dict  # noqa: E402

warnings.filterwarnings("ignore")
sys.modules.setdefault("timm", types.ModuleType("timm"))
oe = types.ModuleType("opt_einsum")
oe.contract = torch.einsum
sys.modules.setdefault("opt_einsum", oe)
sys.path.insert(0, "/root/reference/KITTI15")
from core.update import BasicMultiUpdateBlock  # noqa: E402

ARGS = dict(corr_levels=2, corr_radius=4, n_gru_layers=3, n_downsample=2)


def update_inputs(seed, b, h, w):
    """hidden states / context terms at 1/4, 1/8, 1/16 + correlation features + disparity."""
    dims = [(h, w), (h // 2, w // 2), (h // 4, w // 4)]
    net = [torch.tanh(torch.randn(b, 128, hh, ww, generator=_gen(seed, f"net{i}"))) for i, (hh, ww) in enumerate(dims)]
    inp = [[torch.randn(b, 128, hh, ww, generator=_gen(seed, f"inp{i}{j}")) * 0.5 for j in range(3)]
           for i, (hh, ww) in enumerate(dims)]
    corr = torch.randn(b, 162, h, w, generator=_gen(seed, "corr"))
    disp = torch.rand(b, 1, h, w, generator=_gen(seed, "disp")) * 40
    return net, inp, corr, disp


def main():
    m = BasicMultiUpdateBlock(types.SimpleNamespace(**ARGS), hidden_dims=[128, 128, 128]).eval()
    sd = synth_state_dict(m.state_dict(), seed=101)
    m.load_state_dict(sd, strict=True)
    net, inp, corr, disp = update_inputs(102, 1, 16, 24)
    out = {}
    with torch.no_grad():
        n1, mask1, d1 = m([t.clone() for t in net], inp, corr, disp)
        n1 = [t.clone() for t in n1]
        n2, mask2, d2 = m([t.clone() for t in n1], inp, corr, disp + d1)
        slow = m([t.clone() for t in net], inp, iter04=False, iter08=False, update=False)     # the slow_fast_gru call
    for i in range(3):
        out[f"net1_{i}"], out[f"net2_{i}"] = n1[i].numpy(), n2[i].numpy()
    out.update(mask1=mask1.numpy(), delta1=d1.numpy(), mask2=mask2.numpy(), delta2=d2.numpy(), slow_net2=slow[2].numpy())
    np.savez_compressed(REPO / "tests/golden/igev_update.npz", sd_seed=101, in_seed=102, **out)
    print({k: v.shape for k, v in out.items()}, float(np.abs(out["delta1"]).max()))


if __name__ == "__main__":
    main()
