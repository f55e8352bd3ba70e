"""diffuvolume_amd -- MI355X-native DiffuVolume hot path (cost volumes, DDIM volume filter,
3-D hourglass aggregation, regression) behind the reference's Python API.  The arithmetic
lives in hand-written gfx950 HIP kernels (csrc/, C ABI in include/diffuvolume_hip.h)."""
from ._lib import DiffuVolumeError, lib_path, load
from .submodule import (build_concat_attention_volume, build_concat_volume, build_gwc_volume,
                        disparity_regression, upsample_softmax_regress)
from .acv_ddim import ACVNet_DDIM, __models__

__all__ = ["ACVNet_DDIM", "__models__", "build_gwc_volume", "build_concat_volume",
           "build_concat_attention_volume", "disparity_regression", "upsample_softmax_regress",
           "DiffuVolumeError", "lib_path", "load"]
__version__ = "0.1.0"
