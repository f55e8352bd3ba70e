"""diffuvolume_amd -- MI355X-native DiffuVolume hot path (cost volumes, DDIM volume filter,
3-D hourglass aggregation, regression) behind the reference's Python API.  The arithmetic
lives in hand-written gfx950 HIP kernels (csrc/, C ABI in include/diffuvolume_hip.h)."""
from ._lib import DiffuVolumeError, lib_path, load
from .submodule import (AttentionConcatVolume, build_concat_attention_volume, build_concat_volume, build_gwc_volume,
                        disparity_regression, upsample_softmax_regress)
from .acv_ddim import ACVNet, ACVNet_DDIM, __models__
from .pwcnet_ddim import PWCNet, PWCNet_G, PWCNet_GC, PWCNet_ddim

__all__ = ["ACVNet", "ACVNet_DDIM", "PWCNet", "PWCNet_ddim", "__models__", "build_gwc_volume", "build_concat_volume",
           "build_concat_attention_volume", "AttentionConcatVolume", "disparity_regression", "upsample_softmax_regress",
           "DiffuVolumeError", "lib_path", "load"]
# KITTI12/models/__init__.py:5-9: the origin network under both registry names and the DiffuVolume flavour
__models__ = dict(__models__, **{"gwcnet-g": PWCNet_G, "gwcnet-gc": PWCNet_GC,
                                 "pwc_ddimgc": lambda d: PWCNet_ddim(d, use_concat_volume=True)})
__version__ = "0.1.0"
