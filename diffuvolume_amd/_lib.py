"""ctypes binding of libdiffuvolume_hip.so (the C ABI in include/diffuvolume_hip.h).

There is deliberately no CPU / eager fallback: if the library is missing or a
kernel reports an error the call raises.  ``import torch`` happens first so the
library resolves libamdhip64 against the HIP runtime PyTorch already loaded
(one runtime, one set of streams).
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, Structure, c_char_p, c_double, c_float, c_int, c_size_t, c_void_p
from pathlib import Path

import torch  # noqa: F401  (must precede the dlopen below)

# DV_LIB_PATH: load another build of the same ABI (A/B of compiler flags / kernel variants); default = the in-tree .so
_LIB_PATH = Path(os.environ.get("DV_LIB_PATH") or Path(__file__).resolve().parent / "libdiffuvolume_hip.so")
_lib = None


class DvDdimCoef(Structure):
    """struct dv_ddim_coef (include/diffuvolume_hip.h)."""
    _fields_ = [("sqrt_recip_alpha", c_double), ("sqrt_recipm1_alpha", c_double),
                ("sqrt_alpha_next", c_double), ("c", c_double), ("sigma", c_double),
                ("dif_thr", c_float), ("unc_thr", c_float), ("cof", c_float), ("last", c_int),
                ("clamp_max", c_float), ("ens_dif_thr", c_float)]


P = c_void_p
I = c_int
# name -> (restype, argtypes); mirrors include/diffuvolume_hip.h one to one
SIGNATURES = {
    "dv_version": (c_int, []),
    "dv_error_string": (c_char_p, [I]),
    "dv_gwc_volume_f32": (c_int, [P, P, P, I, I, I, I, I, I, P]),
    "dv_concat_volume_f32": (c_int, [P, P, P, I, I, I, I, I, I, P]),
    "dv_concat_attn_volume_f32": (c_int, [P, P, P, P, I, I, I, I, I, P]),
    "dv_concat_prob_volume_f32": (c_int, [P, P, P, P, I, I, I, I, I, P]),
    "dv_pointwise_expand_packed_floats": (c_size_t, [I, I]),
    "dv_pointwise_expand_pack_weights_f32": (c_int, [P, P, I, I, P]),
    "dv_pointwise_expand_f32": (c_int, [P, P, P, I, I, I, I, P]),
    "dv_softmax_d_f32": (c_int, [P, P, I, I, I, P]),
    "dv_mul_f32": (c_int, [P, P, P, c_size_t, P]),
    "dv_conv3d_rank1_filter_f32": (c_int, [P, P, P, P, P, P, I, I, I, I, I, I, P]),
    "dv_noise_prepare_f32": (c_int, [P, P, P, I, I, I, P]),
    "dv_noise_prepare_f64": (c_int, [P, P, P, P, I, I, I, P]),
    "dv_conv3d_packed_floats": (c_size_t, [I, I, I]),
    "dv_conv3d_pack_weights_f32": (c_int, [P, P, I, I, I, P]),
    "dv_conv3d_f32": (c_int, [P, P, P, P, P, P, P, I, I, I, I, I, I, I, I, I, P]),
    "dv_conv3d_set_s2_tile": (c_int, [I]),
    "dv_conv3d_set_c1z": (c_int, [I, I]),
    "dv_conv3d_f16x3_packed_bytes": (c_size_t, [I, I]),
    "dv_conv3d_f16x3_pack_weights": (c_int, [P, P, I, I, P]),
    "dv_conv3d_f16x3_f32": (c_int, [P, P, P, P, P, P, P, P, I, I, I, I, I, I, I, P]),
    "dv_conv3d_wino_packed_floats": (c_size_t, [I, I]),
    "dv_conv3d_wino_pack_weights_f32": (c_int, [P, P, I, I, P]),
    "dv_conv3d_wino_f32": (c_int, [P, P, P, P, P, P, P, I, I, I, I, I, I, I, P]),
    "dv_conv3d_wino3_packed_floats": (c_size_t, [I, I]),
    "dv_conv3d_wino3_supported": (c_int, [I, I, I, I, I]),
    "dv_conv3d_wino3_pack_weights_f32": (c_int, [P, P, I, I, P]),
    "dv_conv3d_wino3_f32": (c_int, [P, P, P, P, P, P, I, I, I, I, I, I, I, P]),
    "dv_conv3d_s2pp_supported": (c_int, [I, I, I, I, I]),
    "dv_conv3d_s2pp_packed_floats": (c_size_t, [I, I]),
    "dv_conv3d_s2pp_pack_weights_f32": (c_int, [P, P, I, I, P]),
    "dv_conv3d_s2pp_f32": (c_int, [P, P, P, P, P, P, I, I, I, I, I, I, I, P]),
    "dv_deconv3d_packed_floats": (c_size_t, [I, I]),
    "dv_deconv3d_pack_weights_f32": (c_int, [P, P, I, I, P]),
    "dv_deconv3d_k3s2_f32": (c_int, [P, P, P, P, P, P, I, I, I, I, I, I, I, P]),
    "dv_deconv3d_pl_supported": (c_int, [I, I, I, I, I, I]),
    "dv_deconv3d_set_impl": (c_int, [I]),
    "dv_deconv3d_pl_set_max_blocks": (c_int, [I]),
    "dv_deconv3d_k4_packed_floats": (c_size_t, [I, I]),
    "dv_deconv3d_k4_pack_weights_f32": (c_int, [P, P, I, I, P]),
    "dv_deconv3d_k4s2_f32": (c_int, [P, P, P, P, P, P, I, I, I, I, I, I, I, P]),
    "dv_conv2d_packed_floats": (c_size_t, [I, I, I, I]),
    "dv_conv2d_pack_weights_f32": (c_int, [P, P, I, I, I, I, P]),
    "dv_conv2d_f32": (c_int, [P, P, P, P, P, P, I, I, I, I, I, I, I, I, P]),
    "dv_conv2d_gated_f32": (c_int, [P, P, P, P, P, P, P, P, P, I, I, I, I, I, I, I, I, P]),
    "dv_conv2d_cat_f32": (c_int, [P, P, I, P, P, P, P, P, P, P, P, I, I, I, I, I, I, I, P]),
    "dv_conv2d_auto_kslices": (c_int, [I, I, I, I, I, I, I]),
    "dv_conv2d_cat_ksplit_f32": (c_int, [P, P, I, P, P, P, P, P, P, P, P, P, I, I, I, I, I, I, I, I, P]),
    "dv_conv2d_s2_f32": (c_int, [P, P, P, P, P, P, I, I, I, I, I, I, I, P]),
    "dv_conv2d_wino_packed_floats": (c_size_t, [I, I]),
    "dv_conv2d_wino_pack_weights_f32": (c_int, [P, P, I, I, P]),
    "dv_conv2d_wino_cat_f32": (c_int, [P, P, I, P, P, P, P, P, P, P, P, I, I, I, I, I, P]),
    "dv_conv2d_wino_cat_pair_f32": (c_int, [P, P, I, P, P, P, P, P, P, P, P, P, I, I, I, I, I, I, P]),
    "dv_conv2d_wino_dil_cat_f32": (c_int, [P, P, I, P, P, P, P, P, P, P, P, I, I, I, I, I, I, P]),
    "dv_refine_inputs_f32": (c_int, [P, P, P, P, P, P, I, I, I, I, I, P]),
    "dv_space_to_batch2_f32": (c_int, [P, P, I, I, I, I, P]),
    "dv_conv2d_wino_auto_kslices": (c_int, [I, I, I, I, I]),
    "dv_conv2d_wino_cat_ksplit_f32": (c_int, [P, P, I, P, P, P, P, P, P, P, P, P, I, I, I, I, I, I, I, P]),
    "dv_conv2d_wino_cat_pair_ksplit_f32": (c_int, [P, P, I, P, P, P, P, P, P, P, P, P, P, I, I, I, I, I, I, I, P]),
    "dv_conv2d_wino_s2b_f32": (c_int, [P, P, I, P, P, P, P, I, I, I, I, I, P]),
    "dv_batch_to_space_f32": (c_int, [P, P, I, I, I, I, I, P]),
    "dv_softmax_regress_f32": (c_int, [P, P, I, I, I, I, P]),
    "dv_feature_gate_f32": (c_int, [P, P, P, I, I, I, I, I, P]),
    "dv_deconv3d_k3s2_redir_f32": (c_int, [P, P, P, P, P, P, I, I, I, I, I, I, I, I, P]),
    "dv_patch_volume_f32": (c_int, [P, P, P, P, P, I, I, I, I, I, P]),
    "dv_patch_volume_runs_f32": (c_int, [P, P, P, P, P, I, I, I, I, I, I, P, P, P, P]),
    "dv_window_attn3d_f32": (c_int, [P, P, P, P, P, P, I, I, I, I, I, I, P]),
    "dv_upsample_softmax_regress_f32": (c_int, [P, P, P, I, I, I, I, I, P]),
    "dv_upsample_softmax_uncertainty_f32": (c_int, [P, P, P, I, I, I, I, I, P]),
    "dv_disparity_regression_f32": (c_int, [P, P, I, I, I, I, P]),
    "dv_encode_two_hot_f32": (c_int, [P, P, I, I, I, P]),
    "dv_ddim_step": (c_int, [P, P, P, P, P, P, P, P, P, P, P, P, P, P, I, I, I, I, POINTER(DvDdimCoef), P]),
    "dv_context_upsample_f32": (c_int, [P, P, P, I, I, I, c_float, I, P]),
    "dv_allpairs_corr_f32": (c_int, [P, P, P, P, I, I, I, I, I, P]),
    "dv_conv2d_1in_f32": (c_int, [P, P, P, P, I, I, I, I, I, I, P]),
    "dv_resize_bilinear_ac_f32": (c_int, [P, P, I, I, I, I, I, P]),
    "dv_avg_pool3s2_f32": (c_int, [P, P, I, I, I, P]),
    "dv_conv2d_fewin_f32": (c_int, [P, P, P, P, P, P, I, I, I, I, I, I, I, I, P]),
    "dv_instance_norm_act_f32": (c_int, [P, P, I, I, ctypes.c_float, I, P]),
    "dv_geo_filter_lookup_f32": (c_int, [P, P, P, P, P, P, P, I, I, I, I, I, I, I, P]),
    "dv_geo_lookup_conv1x1_packed_floats": (c_size_t, [I]),
    "dv_geo_lookup_conv1x1_pack_weights_f32": (c_int, [P, P, I, P]),
    "dv_geo_filter_lookup_conv1x1_f32": (c_int, [P, P, P, P, P, P, P, P, P, I, I, I, I, I, I, I, I, I, P]),
    "dv_masked_metrics_f32": (c_int, [P, P, P, P, I, I, P]),
}


class DiffuVolumeError(RuntimeError):
    pass


def lib_path() -> Path:
    return _LIB_PATH


def load() -> ctypes.CDLL:
    """Load (once) and type the shared library; raise loudly when it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not _LIB_PATH.exists():
        raise DiffuVolumeError(
            f"{_LIB_PATH} is missing: build it with `python -m diffuvolume_amd._build` "
            "(there is no CPU fallback for the DiffuVolume hot path)")
    lib = ctypes.CDLL(os.fspath(_LIB_PATH), mode=ctypes.RTLD_GLOBAL)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError => ABI mismatch, also loud
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(code: int, what: str) -> None:
    if code != 0:
        msg = load().dv_error_string(code).decode()
        raise DiffuVolumeError(f"{what} failed with code {code}: {msg}")


def stream_ptr() -> int:
    """Raw hipStream_t of PyTorch's current stream on the current device."""
    return torch.cuda.current_stream().cuda_stream


def ptr(t) -> int:
    return 0 if t is None else t.data_ptr()
