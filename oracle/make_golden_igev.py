"""Golden vectors for the IGEV geometry lookup, from the imported reference
(KITTI15/core/geometry_ddim.py).  Run in the build container only:
    PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden_igev.py"""
import sys
import types
from pathlib import Path

import numpy as np
import torch

REPO = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(REPO))
from diffuvolume_amd.synth import _gen  # noqa: E402

REF = Path("/root/reference/KITTI15")
sys.modules.setdefault("timm", types.ModuleType("timm"))
oe = types.ModuleType("opt_einsum")
oe.contract = torch.einsum
sys.modules.setdefault("opt_einsum", oe)
sys.path.insert(0, str(REF))
from core.geometry_ddim import Combined_Geo_Encoding_Volume  # noqa: E402


def rnd(key, *shape):
    return torch.randn(*shape, generator=_gen(61, key))


def main():
    b, c, d, h, w, cf = 2, 8, 48, 5, 24, 16
    geo = rnd("geo", b, c, d, h, w)
    f1, f2 = rnd("f1", b, cf, h, w), rnd("f2", b, cf, h, w)
    disp = torch.rand(b, 1, h, w, generator=_gen(61, "disp")) * 50 - 2        # some taps fall off both ends
    coords = torch.arange(w, dtype=torch.float32).view(1, 1, 1, w).expand(b, 1, h, w).contiguous()
    noisy = torch.rand(b, d, h, w, generator=_gen(61, "noisy"))
    fn = Combined_Geo_Encoding_Volume(f1, f2, geo, num_levels=2, radius=4)
    out = fn(disp, coords, noisy)
    np.savez_compressed(REPO / "tests/golden/igev_geo_lookup.npz", geo=geo.numpy(), f1=f1.numpy(), f2=f2.numpy(),
                        disp=disp.numpy(), coords=coords.numpy(), noisy=noisy.numpy(), out=out.numpy())
    print("igev_geo_lookup.npz", tuple(out.shape))


if __name__ == "__main__":
    main()
