"""ctypes binding of libdiffsim_amd.so (the C ABI declared in include/diffsim_amd.h).

There is NO fallback: if the shared library is missing or a call fails, an exception is raised.
"""
from __future__ import annotations

import ctypes as C
import os

# PyTorch-ROCm owns the device memory and streams this library works on, and its wheel bundles its
# own HIP runtime.  Import it BEFORE dlopen()ing libdiffsim_amd.so so the library's libamdhip64
# dependency binds to the runtime torch already loaded; loaded the other way round the process
# ends up with two HIP runtimes and the second one sees no device.
import torch  # noqa: F401

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libdiffsim_amd.so")

DSIM_F32, DSIM_BF16, DSIM_F16 = 0, 1, 2
TAP = {"down_blocks": 0, "mid_blocks": 1, "up_blocks": 2}
MAX_LEVELS = 4
FUSE_FF, FUSE_LNPROJ, FUSE_TAPQKV, FUSE_ALL = 1, 2, 4, 7        # dsim_unet_set_fusion bits


class DsimError(RuntimeError):
    pass


class UNetCfgC(C.Structure):
    _fields_ = [
        ("in_channels", C.c_int32), ("n_levels", C.c_int32),
        ("block_out_channels", C.c_int32 * MAX_LEVELS),
        ("down_has_attn", C.c_int32 * MAX_LEVELS), ("up_has_attn", C.c_int32 * MAX_LEVELS),
        ("layers_per_block", C.c_int32), ("num_heads", C.c_int32),
        ("cross_attention_dim", C.c_int32), ("norm_num_groups", C.c_int32),
        ("norm_eps", C.c_float), ("sample_size", C.c_int32), ("ctx_len", C.c_int32),
        ("compute_dtype", C.c_int32), ("tap_block", C.c_int32), ("tap_layer", C.c_int32),
        ("tap_attn", C.c_int32), ("tap_tfm", C.c_int32),
        ("heads_per_level", C.c_int32 * MAX_LEVELS), ("depth_per_level", C.c_int32 * MAX_LEVELS),
        ("addition_embed", C.c_int32), ("addition_time_embed_dim", C.c_int32), ("pooled_dim", C.c_int32),
    ]


class VAECfgC(C.Structure):
    _fields_ = [
        ("in_channels", C.c_int32), ("latent_channels", C.c_int32), ("n_levels", C.c_int32),
        ("block_out_channels", C.c_int32 * MAX_LEVELS), ("layers_per_block", C.c_int32),
        ("norm_num_groups", C.c_int32), ("compute_dtype", C.c_int32),
    ]


class DiTCfgC(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("input_size", "patch_size", "in_channels", "hidden_size", "depth", "num_heads",
                                          "mlp_ratio", "num_classes", "freq_dim", "compute_dtype", "tap_layer")]


# every symbol include/diffsim_amd.h declares: (restype, argtypes)
_vp, _i, _f, _sz = C.c_void_p, C.c_int, C.c_float, C.c_size_t
ABI_VERSION = 7          # include/diffsim_amd.h DSIM_ABI_VERSION (tests/test_host.py checks the header against it)
SYMBOLS = {
    "dsim_version": (_i, []),
    "dsim_strerror": (C.c_char_p, [_i]),
    "dsim_device_count": (_i, []),
    "dsim_unet_create": (_i, [C.POINTER(UNetCfgC), C.POINTER(_vp)]),
    "dsim_unet_destroy": (None, [_vp]),
    "dsim_unet_load_weight": (_i, [_vp, C.c_char_p, _vp, _i, C.POINTER(C.c_int64), _i]),
    "dsim_unet_finalize": (_i, [_vp, _vp]),
    "dsim_unet_set_timestep": (_i, [_vp, _i, _vp]),
    "dsim_unet_set_conditioning": (_i, [_vp, _i, _vp, _vp, _vp]),
    "dsim_unet_workspace_bytes": (_sz, [_vp, _i]),
    "dsim_unet_qkv": (_i, [_vp, _vp, _vp, _f, _f, _vp, _i, _vp, _vp, _vp, _vp, _sz, _vp]),
    "dsim_unet_tap_shape": (_i, [_vp, C.POINTER(_i), C.POINTER(_i), C.POINTER(_i)]),
    "dsim_unet_set_tap": (_i, [_vp, _i, _i, _i, _i]),
    "dsim_unet_set_sample_size": (_i, [_vp, _i]),
    "dsim_unet_set_cfg_dedup": (_i, [_vp, _i]),
    "dsim_unet_set_fusion": (_i, [_vp, _i]),
    "dsim_unet_profile": (_i, [_vp, _i]),
    "dsim_unet_profile_count": (_i, [_vp]),
    "dsim_unet_profile_get": (_i, [_vp, _i, C.c_char_p, _i, C.POINTER(C.c_double), C.POINTER(C.c_double),
                                   C.POINTER(C.c_double)]),
    "dsim_vae_create": (_i, [C.POINTER(VAECfgC), C.POINTER(_vp)]),
    "dsim_vae_destroy": (None, [_vp]),
    "dsim_vae_load_weight": (_i, [_vp, C.c_char_p, _vp, _i, C.POINTER(C.c_int64), _i]),
    "dsim_vae_finalize": (_i, [_vp, _vp]),
    "dsim_vae_workspace_bytes": (_sz, [_vp, _i, _i]),
    "dsim_vae_encode": (_i, [_vp, _vp, _i, _i, _vp, _vp, _sz, _vp]),
    "dsim_vae_profile": (_i, [_vp, _i]),
    "dsim_vae_profile_count": (_i, [_vp]),
    "dsim_vae_profile_get": (_i, [_vp, _i, C.c_char_p, _i, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "dsim_image_preprocess": (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    "dsim_latent_sample": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _f, _i, _vp]),
    "dsim_dit_create": (_i, [C.POINTER(DiTCfgC), C.POINTER(_vp)]),
    "dsim_dit_destroy": (None, [_vp]),
    "dsim_dit_load_weight": (_i, [_vp, C.c_char_p, _vp, _i, C.POINTER(C.c_int64), _i]),
    "dsim_dit_finalize": (_i, [_vp, _vp]),
    "dsim_dit_set_conditioning": (_i, [_vp, _i, _i, _i, _vp]),
    "dsim_dit_set_attention": (_i, [_vp, _i]),
    "dsim_dit_set_tap": (_i, [_vp, _i]),
    "dsim_dit_profile": (_i, [_vp, _i]),
    "dsim_dit_profile_count": (_i, [_vp]),
    "dsim_dit_profile_get": (_i, [_vp, _i, C.c_char_p, _i, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "dsim_dit_workspace_bytes": (_sz, [_vp, _i]),
    "dsim_dit_qkv": (_i, [_vp, _vp, _vp, _f, _f, _i, _vp, _vp, _vp, _vp, _sz, _vp]),
    "dsim_pair_score_workspace_bytes": (_sz, [_i, _i, _i, _i, _i]),
    "dsim_pair_score": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _sz, _vp]),
    "dsim_pair_score_status": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _sz, _vp]),
    "dsim_op_linear": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "dsim_op_conv3x3": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "dsim_op_groupnorm": (_i, [_vp, _i, _vp, _i, _vp, _vp, _vp, _i, _i, _i, _f, _i, _i, _vp]),
    "dsim_op_layernorm": (_i, [_vp, _vp, _vp, _vp, _i, _i, _f, _i, _vp]),
    "dsim_op_attention": (_i, [_vp, _i, _vp, _vp, _i, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "dsim_op_attention_fp8": (_i, [_vp, _i, _vp, _vp, _i, _vp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "dsim_op_ff_fused": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _f, _vp]),
    "dsim_op_ln_linear": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _f, _vp]),
}

_lib = None


def lib() -> C.CDLL:
    """Load the HIP extension; raise loudly if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise DsimError(
                f"{LIB_PATH} not found: the HIP extension is not built "
                "(run `python -m diffsim_amd.build`); there is no CPU fallback")
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(L, name)          # AttributeError if the .so lacks a declared symbol
            fn.restype = res
            fn.argtypes = args
        if L.dsim_version() != ABI_VERSION:
            raise DsimError(f"ABI version mismatch: library {L.dsim_version()}, bindings {ABI_VERSION}")
        _lib = L
    return _lib


def check(status: int, what: str = "") -> None:
    if status != 0:
        msg = lib().dsim_strerror(status).decode()
        raise DsimError(f"{what}: {msg} (status {status})")
