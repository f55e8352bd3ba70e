import os
import sys
from pathlib import Path

import numpy as np
import pytest
import torch

ROOT = Path(__file__).resolve().parents[1]
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

GOLDEN = ROOT / "tests" / "golden"

# torch's own GPU convolutions (MIOpen) appear in the GPU suite only as REFERENCES for a handful of full-size property tests:
# its exhaustive solver search costs tens of seconds per new shape, the fast find mode a fraction of that (set before MIOpen
# is initialised; the product path never calls MIOpen)
os.environ.setdefault("MIOPEN_FIND_MODE", "FAST")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def load_golden(name):
    """tests/golden/<name>.npz -> dict of torch tensors (numpy scalars stay python numbers)."""
    out = {}
    with np.load(GOLDEN / f"{name}.npz", allow_pickle=False) as z:
        for k in z.files:
            a = z[k]
            if a.dtype.kind in "US":
                out[k] = a
            elif a.ndim == 0:
                out[k] = a.item()
            else:
                out[k] = torch.from_numpy(a.copy())
    return out


@pytest.fixture(scope="session")
def golden():
    return load_golden


@pytest.fixture(scope="session")
def acv_state_dict():
    """The synthetic ACVNet_DDIM weights the golden vectors were produced with (seed 1, gain 8)."""
    from diffuvolume_amd.acv_ddim import ACVNet_DDIM
    from diffuvolume_amd.synth import synth_state_dict
    return synth_state_dict(ACVNet_DDIM(192, False, False).state_dict(), seed=1, logit_gain=8.0)


def conditioned_pcw_state_dict(name):
    """The CONDITIONED KITTI12 weights of oracle/calibrate.py: the synthetic PWCNet_ddim state dict (seed 2) with the
    classifier gain / refinement-head factor the fixture names and the BatchNorm statistics stored in
    tests/golden/<name>.npz (written by oracle/make_golden_pcw_conditioned.py)."""
    from diffuvolume_amd.pwcnet_ddim import PWCNet_ddim
    from diffuvolume_amd.synth import synth_state_dict
    from oracle import calibrate as C
    g = load_golden(name)
    sd = synth_state_dict(PWCNet_ddim(192, True).state_dict(), seed=2, logit_gain=float(g["gain"]),
                          scale={"refinenet3.conv8.weight": float(g["head"])})
    stats = C.unpack(g["bn_keys"], g["bn_vals"], g["bn_lens"])
    assert set(stats) <= set(sd)
    sd.update(stats)
    return sd, g
