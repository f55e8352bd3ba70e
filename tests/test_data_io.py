"""PFM / list-file IO and the evaluation crop against what the reference's own reader returned for the same files
(tests/golden/pfm_*.pfm + pfm_io.npz, written by oracle/make_golden_io.py)."""
import numpy as np
import pytest

from conftest import GOLDEN
from diffuvolume_amd import data_io as IO


@pytest.mark.parametrize("tag", ["grey_le", "grey_be", "colour_le"])
def test_pfm_matches_reference_reader(tag):
    g = np.load(GOLDEN / "pfm_io.npz")
    data, scale = IO.pfm_imread(str(GOLDEN / f"pfm_{tag}.pfm"))
    assert data.shape == g[tag].shape and scale == float(g[tag + "_scale"])
    assert np.array_equal(np.ascontiguousarray(data, dtype=np.float32), g[tag])
    if data.ndim == 2:
        d = IO.load_disp(str(GOLDEN / f"pfm_{tag}.pfm"))
        assert d.dtype == np.float32 and d.flags["C_CONTIGUOUS"] and np.array_equal(d, g[tag])


def test_list_file_and_errors(tmp_path):
    g = np.load(GOLDEN / "pfm_io.npz")
    assert IO.read_all_lines(str(GOLDEN / "pfm_list.txt")) == g["lines"].tolist()
    bad = tmp_path / "x.pfm"
    bad.write_bytes(b"P6\n3 3\n255\n")
    with pytest.raises(Exception, match="Not a PFM file"):
        IO.pfm_imread(str(bad))
    bad.write_bytes(b"Pf\n3x3\n-1\n")
    with pytest.raises(Exception, match="Malformed PFM header"):
        IO.pfm_imread(str(bad))


def test_eval_crop_is_bottom_right_960x512():
    h, w = 540, 960
    disp = np.arange(h * w, dtype=np.float32).reshape(h, w)
    left = np.stack([disp, disp + 1, disp + 2])            # CHW
    l, r, d = IO.eval_crop(left, left, disp)
    assert d.shape == (512, 960) and l.shape == (3, 512, 960)
    assert d[0, 0] == disp[28, 0] and l[1, -1, -1] == disp[-1, -1] + 1
    hwc = left.transpose(1, 2, 0)
    assert IO.eval_crop(hwc, hwc, disp)[0].shape == (512, 960, 3)
    img = (np.arange(2 * 3 * 3) % 255).astype(np.uint8).reshape(2, 3, 3)
    x = IO.normalize_image(img)
    assert x.shape == (3, 2, 3) and abs(float(x[0, 0, 0]) - (0 / 255 - 0.485) / 0.229) < 1e-6
