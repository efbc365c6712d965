"""ctypes binding of ``csrc/librawformer_hip.so`` (the C ABI in ``include/rawformer_hip.h``).

There is no fallback: if the library is missing or a call fails, a ``RuntimeError`` is raised
(the reference's own error convention is Python exceptions).
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RF_LIB_PATH") or os.path.join(_HERE, "csrc", "librawformer_hip.so")  # override: diagnostic builds

RF_VARIANT_FLCA = 0
RF_VARIANT_PLAIN = 1
RF_VARIANT_TRUECOLOR = 2


class RfConfig(C.Structure):
    _fields_ = [("dim", C.c_int32), ("heads", C.c_int32 * 4), ("inp_channels", C.c_int32),
                ("out_channels", C.c_int32), ("ffn_expansion", C.c_int32), ("variant", C.c_int32),
                ("branch_lrelu", C.c_int32), ("clamp_io", C.c_int32), ("flca_levels", C.c_int32)]


_vp, _i, _f, _sz = C.c_void_p, C.c_int, C.c_float, C.c_size_t
# rf_allreduce_fn (include/rawformer_hip.h): void (*)(void* user, float* buf, size_t n, int op, void* stream); buf arrives as an integer address
ALLREDUCE_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p)
_psz = C.POINTER(C.c_size_t)
# rf_grad_ready_fn: void (*)(void* user, size_t offset, size_t count, void* stream)
GRAD_READY_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p)

# name -> (restype, argtypes); mirrors include/rawformer_hip.h one to one
SIGNATURES = {
    "rf_last_error": (C.c_char_p, []),
    "rf_version": (_i, []),
    "rf_profile_begin": (_i, []),
    "rf_profile_end": (_i, [C.c_char_p, _sz]),
    "rf_create": (_i, [C.POINTER(RfConfig), C.POINTER(_vp)]),
    "rf_destroy": (None, [_vp]),
    "rf_param_count": (_i, [_vp]),
    "rf_param_info": (_i, [_vp, _i, C.POINTER(C.c_char_p), C.POINTER(C.c_int64 * 4), C.POINTER(_i)]),
    "rf_set_param": (_i, [_vp, C.c_char_p, _vp, C.POINTER(C.c_int64), _i]),
    "rf_packed_bytes": (_i, [_vp, _psz]),
    "rf_pack_params": (_i, [_vp, _vp, _sz, _vp]),
    "rf_workspace_bytes": (_i, [_vp, _i, _i, _i, _psz]),
    "rf_forward": (_i, [_vp, _vp, _vp, _vp, _sz, _i, _i, _i, _i, _vp]),
    "rf_forward_stage": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _sz, _i, _i, _i, _vp]),
    "rf_set_shard": (_i, [_vp, _i, _i, _i, ALLREDUCE_FN, _vp]),
    "rf_flat_param_floats": (_i, [_vp, _psz]),
    "rf_flat_offset": (_i, [_vp, _i, _psz]),
    "rf_train_workspace_bytes": (_i, [_vp, _i, _i, _i, _psz]),
    "rf_set_grad_ready": (_i, [_vp, GRAD_READY_FN, _vp]),
    "rf_grad_range_count": (_i, [_vp, C.POINTER(_i)]),
    "rf_grad_range": (_i, [_vp, _i, _psz, _psz]),
    "rf_train_step": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _i, _i, _i, _i, _f, _vp]),
    "rf_adam_step": (_i, [_vp, _vp, _vp, _vp, _sz, _f, _f, _f, _f, _f, _i, _i, _f, _vp]),
    "rf_pixel_unshuffle2": (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    "rf_pixel_shuffle2": (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    "rf_dwt_haar": (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    "rf_idwt_haar": (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    "rf_dwt_custom": (_i, [_vp, _vp, C.POINTER(C.c_float), _i, _i, _i, _i, _i, _vp]),
    "rf_idwt_custom": (_i, [_vp, _vp, C.POINTER(C.c_float), _i, _i, _i, _i, _i, _vp]),
    "rf_haar_dwt": (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    "rf_layernorm2d": (_i, [_vp, _vp, _vp, _vp, _f, _i, _i, _i, _i, _vp]),
    "rf_conv1x1_scratch_bytes": (_i, [_i, _i, _psz]),
    "rf_conv1x1": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "rf_dwconv3x3": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "rf_conv3x3_scratch_bytes": (_i, [_i, _i, _psz]),
    "rf_conv3x3": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "rf_convT2x2_scratch_bytes": (_i, [_i, _i, _psz]),
    "rf_convT2x2": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "rf_chan_attn_scratch_bytes": (_i, [_i, _i, _i, _i, _i, _psz]),
    "rf_chan_attn": (_i, [_vp] * 10 + [_i] * 5 + [_vp]),
    "rf_transformer_block_scratch_bytes": (_i, [_i, _i, _i, _i, _i, _i, _psz]),
    "rf_transformer_block": (_i, [_vp, _vp, C.POINTER(_vp), _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "rf_flca_scratch_bytes": (_i, [_i, _i, _i, _i, _psz]),
    "rf_flca": (_i, [_vp, _vp, _vp, C.POINTER(_vp), _vp, _i, _i, _i, _i, _vp]),
    "rf_guidance_scratch_bytes": (_i, [_i, _i, _i, _psz]),
    "rf_flca_guidance": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "rf_to_uint8_hwc": (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    "rf_u8_sse": (_i, [_vp, _vp, _vp, _i, _sz, _vp]),
    "rf_u8_channel_sums": (_i, [_vp, _vp, _i, _i, _sz, _vp]),
    "rf_sid_pack": (_i, [_vp, _vp, _i, _i, _i, _i, _i, C.c_double, _i, _vp]),
    "rf_token_attn": (_i, [_vp, _vp, _vp, _vp, C.c_longlong, C.c_longlong, _i, _i, _i, _i, _f, _vp]),
    "rf_bayer_luma_scratch_bytes": (_i, [_i, _i, _i, _psz]),
    "rf_bayer_luma": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "rf_luma_film_scratch_bytes": (_i, [_i, _i, _i, _psz]),
    "rf_luma_film": (_i, [_vp, _vp, _vp, C.c_longlong, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "rf_dwgate3x3": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "rf_dwconv5x5": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "rf_rfft2_polar_scratch_bytes": (_i, [_i, _i, _i, _psz]),
    "rf_rfft2_polar": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "rf_polar_irfft2": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "rf_feb_scratch_bytes": (_i, [_i, _i, _i, _i, _psz]),
    "rf_feb": (_i, [_vp, _vp, C.POINTER(_vp), _vp, _i, _i, _i, _i, _vp]),
    "rf_ffab_scratch_bytes": (_i, [_i, _i, _i, _i, _psz]),
    "rf_ffab": (_i, [_vp, _vp, C.POINTER(_vp), _vp, _i, _i, _i, _i, _vp]),
    "rf_affine_clamp_add": (_i, [_vp, _vp, _vp, _sz, _f, _f, _f, _f, _vp]),
    "rf_upcat_scratch_bytes": (_i, [_i, _psz]),
    "rf_upcat": (_i, [_vp] * 8 + [_i, _i, _i, _i, _vp]),
}

_lib = None


def load() -> C.CDLL:
    """Load the HIP library (once).  Raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} not found: build it with `python -m bayer_low_light_image_enhancement_amd.build` "
                "(hipcc, gfx950).  There is no CPU fallback for the RawFormer HIP path.")
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)   # AttributeError here = header and library out of sync
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = load().rf_last_error()
        raise RuntimeError(f"{what} failed ({rc}): {msg.decode() if msg else 'unknown error'}")
