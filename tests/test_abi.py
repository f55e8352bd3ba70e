"""The C-ABI library loads and exports every symbol include/diffuvolume_hip.h declares
(no compute calls: this runs without a GPU)."""
import ctypes
import re
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]


def declared_symbols():
    text = (ROOT / "include" / "diffuvolume_hip.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(dv_[a-z0-9_]+)\s*\(", text)))


@pytest.fixture(scope="module")
def built():
    from diffuvolume_amd import _build
    return _build.build()


def test_header_symbols_exported(built):
    lib = ctypes.CDLL(str(built))
    names = declared_symbols()
    assert len(names) >= 19
    for n in names:
        assert hasattr(lib, n), f"{n} declared in the header but not exported"


def test_binding_table_matches_header(built):
    from diffuvolume_amd import _lib
    assert sorted(_lib.SIGNATURES) == declared_symbols()
    lib = _lib.load()
    assert lib.dv_version() == 100
    assert lib.dv_error_string(0) == b"ok"
    assert b"NULL" in lib.dv_error_string(-1)


def test_argument_validation_without_gpu(built):
    """Bad arguments are rejected before anything touches the device."""
    from diffuvolume_amd import _lib
    lib = _lib.load()
    assert lib.dv_gwc_volume_f32(None, None, None, 1, 8, 2, 4, 2, 4, None) == -1
    assert lib.dv_conv3d_packed_floats(32, 32, 3) == 32 * 27 * 32
    assert lib.dv_conv3d_packed_floats(40, 1, 3) == 40 * 27          # single-channel head: raw weights
    assert lib.dv_conv3d_packed_floats(32, 32, 5) == 0
    assert lib.dv_deconv3d_packed_floats(128, 64) == 128 * 27 * 64


def test_code_object_is_gfx950(built):
    data = Path(built).read_bytes()
    assert b"gfx950" in data and b"gfx90a" not in data and b"sm_" not in data[:0]


def test_missing_library_is_loud(monkeypatch, tmp_path):
    from diffuvolume_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "_LIB_PATH", tmp_path / "nope.so")
    with pytest.raises(_lib.DiffuVolumeError):
        _lib.load()
