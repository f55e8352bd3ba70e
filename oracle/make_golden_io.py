"""Golden vectors for diffuvolume_amd/data_io.py: synthetic PFM files (grey little-endian, grey big-endian, colour)
read by the REFERENCE's own pfm_imread (SceneFlow/datasets/data_io.py:32-66, imported here with a torchvision stub)
-> tests/golden/pfm_*.pfm (data files) + tests/golden/pfm_io.npz (what the reference returned).
Run in the build container only:  PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden_io.py"""
import importlib.util
import sys
import types
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
GOLD = ROOT / "tests" / "golden"
REF = Path("/root/reference/SceneFlow/datasets/data_io.py")


def write_pfm(path, arr, little=True, scale=1.0):
    arr = np.asarray(arr, dtype=np.float32)
    magic = b"PF\n" if arr.ndim == 3 else b"Pf\n"
    h, w = arr.shape[:2]
    with open(path, "wb") as f:
        f.write(magic)
        f.write(f"{w} {h}\n".encode())
        f.write(f"{-scale if little else scale}\n".encode())
        f.write(arr[::-1].astype("<f4" if little else ">f4").tobytes())


def main():
    tv = types.ModuleType("torchvision")
    tv.transforms = types.ModuleType("torchvision.transforms")
    sys.modules["torchvision"], sys.modules["torchvision.transforms"] = tv, tv.transforms
    spec = importlib.util.spec_from_file_location("ref_data_io", REF)
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    rng = np.random.default_rng(7)
    grey = (rng.random((5, 7)) * 190).astype(np.float32)
    colour = rng.standard_normal((4, 6, 3)).astype(np.float32)
    write_pfm(GOLD / "pfm_grey_le.pfm", grey, little=True, scale=1.0)
    write_pfm(GOLD / "pfm_grey_be.pfm", grey, little=False, scale=2.5)
    write_pfm(GOLD / "pfm_colour_le.pfm", colour, little=True, scale=1.0)
    out = {}
    for tag in ("grey_le", "grey_be", "colour_le"):
        data, scale = ref.pfm_imread(str(GOLD / f"pfm_{tag}.pfm"))
        out[tag] = np.ascontiguousarray(data, dtype=np.float32)
        out[tag + "_scale"] = np.float64(scale)
    lst = GOLD / "pfm_list.txt"
    lst.write_text("a/left.png a/right.png a/disp.pfm  \nb/left.png b/right.png b/disp.pfm\n")
    out["lines"] = np.array(ref.read_all_lines(str(lst)))
    np.savez(GOLD / "pfm_io.npz", **out)
    print("wrote", sorted(p.name for p in GOLD.glob("pfm_*")))


if __name__ == "__main__":
    main()
