"""ctypes binding of libtgs_hip.so (C ABI: include/tgs.h).  Fails loudly if the library is absent."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# TGS_LIB_PATH: developer override for same-box A/B runs of two builds (tools/ab.py); the default is
# the in-tree build
LIB_PATH = os.environ.get("TGS_LIB_PATH") or os.path.join(_HERE, "lib", "libtgs_hip.so")

SPLAT_FLOATS = 12
PARTIAL_FLOATS = 12
GROUP = 256
BLOCK = 16


class TgsCamera(C.Structure):
    _fields_ = [("viewmat", C.c_float * 16), ("fx", C.c_float), ("fy", C.c_float),
                ("cx", C.c_float), ("cy", C.c_float), ("W", C.c_int32), ("H", C.c_int32),
                ("near_plane", C.c_float), ("pix_center", C.c_float), ("bg", C.c_float * 3),
                ("glob_scale", C.c_float)]


class TgsLossSpec(C.Structure):
    _fields_ = [("gt_rgb", C.c_void_p), ("gt_depth", C.c_void_p), ("uncertainty", C.c_void_p),
                ("l1_weight", C.c_float), ("depth_weight", C.c_float),
                ("uncertainty_weight", C.c_float), ("eps", C.c_float)]


class TgsRasterOpts(C.Structure):
    """Per-call choice of the compositing kernels' forms (tgs.h); -1 = the process-wide default."""
    _fields_ = [("k6_blocks", C.c_int32), ("k6_split", C.c_int32), ("k7_front_to_back", C.c_int32),
                ("k7_quad", C.c_int32), ("k7_quad_min_walk", C.c_int32), ("k7_blocks", C.c_int32)]


class TgsAdamSpec(C.Structure):
    _fields_ = [("lr_means", C.c_float), ("lr_scales", C.c_float), ("lr_quats", C.c_float),
                ("lr_opac", C.c_float), ("lr_sh_dc", C.c_float), ("lr_sh_rest", C.c_float),
                ("beta1", C.c_float), ("beta2", C.c_float), ("eps", C.c_float),
                ("bias_corr1", C.c_float), ("bias_corr2", C.c_float), ("device_bias_corr", C.c_void_p)]


_P = C.c_void_p
_I = C.c_int
# name -> (restype, argtypes); must list every symbol declared in include/tgs.h
SIGNATURES = {
    "tgs_version": (C.c_int, []),
    "tgs_last_error": (C.c_char_p, []),
    "tgs_calib_fma_stream": (C.c_int, [_I, _P, C.POINTER(C.c_int64), _P]),
    "tgs_num_groups": (C.c_int, [_I]),
    "tgs_num_tiles": (C.c_int, [_I, _I]),
    "tgs_tile_order_len": (C.c_int, [_I, _I]),
    "tgs_num_bands": (C.c_int, [_I, _I]),
    "tgs_band_tiles": (C.c_int, [_I, _I, _I, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "tgs_tile_counter_len": (C.c_int, [_I, _I]),
    "tgs_tile_start_len": (C.c_int64, [_I, _I]),
    "tgs_sort_scratch_bytes": (C.c_size_t, [C.c_int64]),
    "tgs_project_fwd": (C.c_int, [C.POINTER(TgsCamera), _I, _P, _P, _P, _P, _P, _I, _I, _P, _P, _P, _P]),
    "tgs_sh_fwd": (C.c_int, [_I, _I, _I, _P, _P, _P, _P]),
    "tgs_sh_bwd": (C.c_int, [_I, _I, _I, _P, _P, _P, _P]),
    "tgs_bin_sort": (C.c_int, [C.POINTER(TgsCamera), _I, _P, _P, _P, C.c_int64, _P, _P, _P, C.c_int64, _P, _P, _P, C.c_int32, _P]),
    "tgs_project_bin_sort": (C.c_int, [C.POINTER(TgsCamera), _I, _P, _P, _P, _P, _P, _I, _I, _P, _P, _P, _P, C.c_int64, _P, _P,
                                       _P, C.c_int64, _P, _P, _P, C.c_int32, _P]),
    "tgs_project_bin_sort_colors": (C.c_int, [C.POINTER(TgsCamera), _I, _P, _P, _P, _P, _P, _I, _I, _P, _P, _P, _P, C.c_int64, _P,
                                              _P, _P, C.c_int64, _P, _P, _P, C.c_int32, _P, _P, C.c_int32, _P]),
    "tgs_rasterize_fwd": (C.c_int, [C.POINTER(TgsCamera), _P, _P, _P, C.c_int64, _P, _P, _P, _P, _P, _P, _P, C.POINTER(TgsRasterOpts), _P]),
    "tgs_set_raster_variant": (C.c_int, [_I, _I]),
    "tgs_set_k7_scan": (C.c_int, [_I, _I]),
    "tgs_set_k7_quad": (C.c_int, [_I, _I]),
    "tgs_set_k6_split": (C.c_int, [_I]),
    "tgs_set_long_run": (C.c_int, [_I]),
    "tgs_set_k6_split_shape": (C.c_int, [_I, _I]),
    "tgs_slot_ok_len": (C.c_size_t, [_I, _I, C.c_int64]),
    "tgs_rasterize_bwd": (C.c_int, [C.POINTER(TgsCamera)] + [_P] * 4 + [C.c_int64] + [_P] * 8 + [C.POINTER(TgsLossSpec), _P, _P, _P, C.POINTER(TgsRasterOpts), _P]),
    "tgs_rasterize_bwd_band": (C.c_int, [C.POINTER(TgsCamera)] + [_P] * 4 + [C.c_int64] + [_P] * 8 + [C.POINTER(TgsLossSpec), _P, _P, _I, _P, C.POINTER(TgsRasterOpts), _P]),
    "tgs_reduce_partials": (C.c_int, [_I, _P, _P, C.POINTER(TgsCamera), _P, _P, _P]),
    "tgs_project_bwd": (C.c_int, [C.POINTER(TgsCamera), _I, _P, _P, _P, _P, _P, _I, _I] + [_P] * 12),
    "tgs_project_bwd_adam": (C.c_int, [C.POINTER(TgsCamera), _I, _I, _I, _P, _P, _P, C.POINTER(TgsAdamSpec), _P, _P, _P, _P, _P, _P]),
    "tgs_project_bwd_adam_next": (C.c_int, [C.POINTER(TgsCamera), _I, _I, _I, _P, _P, _P, C.POINTER(TgsAdamSpec), _P, _P, _P,
                                            _P, _P, C.POINTER(TgsCamera), _P, _P, C.c_int32, _P]),
    "tgs_project_bwd_adam_next_front": (C.c_int, [C.POINTER(TgsCamera), _I, _I, _I, _P, _P, _P, C.POINTER(TgsAdamSpec), _P, _P, _P,
                                                  _P, _P, C.POINTER(TgsCamera), _P, _P, C.c_int32, _P, _P, _P, _P, C.c_int64,
                                                  _P, _P, _P, _I, _P]),
    "tgs_adam_geom_project_next": (C.c_int, [C.POINTER(TgsCamera), _I, _I, _I, _P, _P, _P, _P, C.POINTER(TgsAdamSpec), C.c_float,
                                             _P, _P, C.c_int32, _P, _P, _P, _P, C.c_int64, _P, _P, _P, _I, _P]),
    "tgs_adam_sh_gathered_geom_project_next": (C.c_int, [C.POINTER(TgsCamera), _I, _I, _I, _I, _P, _P, _I, C.POINTER(C.c_int32),
                                                         C.POINTER(C.c_void_p), _P, _P, C.POINTER(TgsAdamSpec), C.c_float, _P, _P,
                                                         C.c_int32, _P, _P, _P, _P, C.c_int64, _P, _P, _P, _I, _P]),
    "tgs_project_bin_sort_front": (C.c_int, [C.POINTER(TgsCamera), _I, _P, _P, _P, _P, _P, _I, _I, _P, _P, _P, _P, C.c_int64, _P,
                                             _P, _P, C.c_int64, _P, _P, _P, C.c_int32, _P, C.c_int32, C.POINTER(TgsCamera), _P, _P, _P]),
    "tgs_front_can_clear_next": (C.c_int, [_I, _I, _I]),
    "tgs_project_bwd_color": (C.c_int, [C.POINTER(TgsCamera), _I, _P, _P, _P, _P, _P, _I, _I] + [_P] * 11),
    "tgs_project_bwd_color_rows": (C.c_int, [C.POINTER(TgsCamera), _I, _I, _I, _P, _P, _P, _P, _P, _I, _I] + [_P] * 11),
    "tgs_adam_step_sh_gathered_rows": (C.c_int, [_I, _I, _I, _I, _I, _I, _P, _P, _P, _P, C.POINTER(TgsAdamSpec), C.c_float, _P, _P]),
    "tgs_dp_agree_overflow": (C.c_int, [_I, _I, _P, _P, _P, _P]),
    "tgs_adam_step_sh_gathered": (C.c_int, [_I, _I, _I, _I, _P, _P, _P, _P, C.POINTER(TgsAdamSpec), C.c_float, _P, _P]),
    "tgs_store_small": (C.c_int, [_P, C.POINTER(C.c_float), _I, _P]),
    "tgs_adam_step": (C.c_int, [_I, _I, _P, _P, _P, _P, C.POINTER(TgsAdamSpec), C.c_float, C.c_int64, C.c_int64, _P, _P]),
    "tgs_ssim_fwd_bwd": (C.c_int, [_I, _I, _P, _P, C.c_float, _P, _P, _P, _P]),
    "tgs_ssim_fwd_bwd_rows": (C.c_int, [_I, _I, _P, _P, C.c_float, _P, _I, _P, _P, _I, _I, _I, _I, _P]),
    "tgs_peer_alloc": (C.c_int, [C.c_size_t, _I, C.POINTER(C.c_void_p), C.POINTER(C.c_ubyte), C.POINTER(C.c_int)]),
    "tgs_peer_open": (C.c_int, [C.POINTER(C.c_ubyte), C.POINTER(C.c_void_p)]),
    "tgs_peer_close": (C.c_int, [_P]),
    "tgs_peer_free": (C.c_int, [_P]),
    "tgs_peer_push": (C.c_int, [_I, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), _P, C.c_size_t, C.c_int32, _P, _P]),
    "tgs_peer_scatter": (C.c_int, [_I, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), _P, C.c_size_t, C.c_size_t, C.c_int32, _P, _P]),
    "tgs_peer_reduce_push": (C.c_int, [_I, C.POINTER(C.c_void_p), _I, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_size_t,
                                       C.c_int32, _P, _P]),
    "tgs_peer_signal": (C.c_int, [_I, C.POINTER(C.c_void_p), C.c_int32, _P]),
    "tgs_peer_wait": (C.c_int, [_I, C.POINTER(C.c_void_p), C.c_int32, _P, C.c_float, _P, _P, _P]),
}

_lib = None


def load():
    """Load the shared library (once).  No fallback: a missing library is an error."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or touch_gs_amd/csrc/build.sh (hipcc --offload-arch=gfx950). touch_gs_amd has no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = load().tgs_last_error().decode(errors="replace")
        raise RuntimeError(f"libtgs_hip {what} failed ({rc}): {msg}")


def ptr(t):
    """Device pointer of a contiguous CUDA/HIP tensor (None -> NULL)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("touch_gs_amd ops need device (HIP) tensors; there is no CPU path")
    if not t.is_contiguous():
        raise RuntimeError("touch_gs_amd ops need contiguous tensors")
    if t.numel() == 0:  # empty tensors have a NULL data pointer; the C ABI wants a valid (unused) one
        return _dummy(t.device).data_ptr()
    return t.data_ptr()


_DUMMY = {}


def _dummy(device):
    import torch
    key = str(device)
    if key not in _DUMMY:
        _DUMMY[key] = torch.zeros(64, dtype=torch.float32, device=device)
    return _DUMMY[key]
