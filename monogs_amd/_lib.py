"""ctypes binding of libmonogs_raster.so (the C ABI declared in include/monogs_raster.h).

There is no fallback: if the library is missing, stale or fails to load, importing the product
path raises.  The library is built in-tree (monogs_amd/lib/) by ``__graft_entry__.build()`` or
``make -C monogs_amd/csrc``.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MGS_LIB_PATH") or os.path.join(_HERE, "lib", "libmonogs_raster.so")   # (override: kernel experiments)
ABI_VERSION = 9

c_float_p = C.c_void_p   # device pointers travel as integers (tensor.data_ptr())


class MgsCamera(C.Structure):
    _fields_ = [
        ("image_height", C.c_int32), ("image_width", C.c_int32),
        ("tanfovx", C.c_float), ("tanfovy", C.c_float), ("scale_modifier", C.c_float),
        ("sh_degree", C.c_int32), ("sh_coeffs", C.c_int32), ("scale_dim", C.c_int32), ("flags", C.c_int32),
        ("bg", C.c_void_p), ("viewmatrix", C.c_void_p), ("projmatrix", C.c_void_p),
        ("projmatrix_raw", C.c_void_p), ("campos", C.c_void_p),
    ]


class MgsTiming(C.Structure):
    _fields_ = [(n, C.c_float) for n in (
        "preprocess_ms", "depth_sort_ms", "scan_ms", "duplicate_ms", "sort_ms", "ranges_ms", "blend_fwd_ms",
        "blend_bwd_ms", "geom_bwd_ms")]

    def as_dict(self):
        return {n: float(getattr(self, n)) for n, _ in self._fields_}


# symbol -> (restype, argtypes); exactly the declarations of include/monogs_raster.h
SIGNATURES = {
    "mgs_abi_version": (C.c_int, []),
    "mgs_last_error": (C.c_char_p, []),
    "mgs_geometry_bytes": (C.c_size_t, [C.c_int32]),
    "mgs_image_bytes": (C.c_size_t, [C.c_int32, C.c_int32]),
    "mgs_binning_bytes": (C.c_size_t, [C.c_uint64, C.c_int32, C.c_int32]),
    "mgs_backward_bytes": (C.c_size_t, [C.c_int32]),
    "mgs_forward_preprocess": (C.c_int, [C.POINTER(MgsCamera), C.c_int32] + [C.c_void_p] * 7
                               + [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_uint64), C.c_void_p, C.POINTER(C.c_uint32),
                                  C.POINTER(MgsTiming), C.c_void_p]),
    "mgs_forward_capacity": (C.c_int, [C.POINTER(MgsCamera), C.c_int32] + [C.c_void_p] * 7 + [C.c_void_p] * 3 + [C.c_uint64]
                             + [C.c_void_p] * 7 + [C.POINTER(MgsTiming), C.c_void_p]),
    "mgs_forward_render": (C.c_int, [C.POINTER(MgsCamera), C.c_int32, C.c_uint64] + [C.c_void_p] * 8
                           + [C.POINTER(MgsTiming), C.c_void_p]),
    "mgs_debug_set_radix_spin_limit": (C.c_int, [C.c_uint32]),
    "mgs_debug_set_option": (C.c_int, [C.c_char_p, C.c_int64]),
    "mgs_debug_sort_temp_bytes": (C.c_size_t, [C.c_uint64, C.c_int32]),
    "mgs_debug_sort_pairs": (C.c_int, [C.c_void_p] * 4 + [C.c_uint64, C.c_int32, C.c_void_p, C.c_void_p]),
    "mgs_forward_render_capacity": (C.c_int, [C.POINTER(MgsCamera), C.c_int32, C.c_uint64] + [C.c_void_p] * 8
                                    + [C.POINTER(MgsTiming), C.c_void_p]),
    "mgs_backward": (C.c_int, [C.POINTER(MgsCamera), C.c_int32, C.c_uint64] + [C.c_void_p] * 7
                     + [C.c_void_p] * 4 + [C.c_void_p] * 2 + [C.c_void_p] * 9
                     + [C.c_void_p, C.c_int32, C.POINTER(MgsTiming), C.c_void_p]),
    "mgs_backward_tau": (C.c_void_p, [C.c_void_p, C.c_int32]),
    "mgs_debug_blend_stats": (C.c_int, [C.POINTER(MgsCamera), C.c_int32, C.c_uint64] + [C.c_void_p] * 5),
    "mgs_debug_valu_ceiling": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p]),
    "mgs_debug_set_blend_events": (C.c_int, [C.c_void_p] * 4),
    "mgs_mark_visible": (C.c_int, [C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mgs_loss_scratch_bytes": (C.c_size_t, []),
    "mgs_loss_forward": (C.c_int, [C.c_int32] * 4 + [C.c_float] + [C.c_void_p] * 9 + [C.c_void_p, C.c_void_p, C.c_void_p]),
    "mgs_loss_backward": (C.c_int, [C.c_int32] * 4 + [C.c_float] + [C.c_void_p] * 9
                          + [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mgs_loss_grads": (C.c_int, [C.c_int32] * 4 + [C.c_float] + [C.c_void_p] * 9 + [C.c_void_p] * 4),
    "mgs_backproject": (C.c_int, [C.c_int32] * 3 + [C.c_void_p] * 6 + [C.c_float] * 4 + [C.c_void_p] * 6),
    "mgs_camera_setup": (C.c_int, [C.c_void_p] * 7),
    "mgs_pose_step": (C.c_int, [C.c_void_p] * 12 + [C.c_int32] + [C.c_float] * 7
                      + [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p] + [C.c_void_p] * 4 + [C.c_void_p]),
    "mgs_pose_step_batch": (C.c_int, [C.c_int32, C.POINTER(C.c_void_p)] + [C.c_float] * 7 + [C.c_int32, C.c_void_p]),
    "mgs_adam_step": (C.c_int, [C.c_int32] + [C.c_void_p] * 6 + [C.c_double] * 3 + [C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mgs_window_stats": (C.c_int, [C.c_int32, C.c_int32] + [C.c_void_p] * 6 + [C.c_int32, C.c_void_p, C.c_void_p]),
    "mgs_window_apply": (C.c_int, [C.c_int32] + [C.c_void_p] * 7),
    "mgs_lr_schedule_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_double, C.c_double, C.c_int32, C.c_double, C.c_int32, C.c_void_p]),
    "mgs_densify_stats": (C.c_int, [C.c_int32] + [C.c_void_p] * 6),
    "mgs_sum_buffers": (C.c_int, [C.c_int32, C.POINTER(C.c_void_p), C.c_void_p, C.c_uint64, C.c_void_p]),
    "mgs_activate_forward": (C.c_int, [C.c_int32, C.c_int32] + [C.c_void_p] * 7),
    "mgs_activate_backward": (C.c_int, [C.c_int32, C.c_int32] + [C.c_void_p] * 10),
    "mgs_knn_scratch_bytes": (C.c_size_t, [C.c_int32]),
    "mgs_dist2_knn": (C.c_int, [C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
}

_lib = None


class MonoGSNativeError(RuntimeError):
    pass


def load():
    """Load (once) and return the ctypes handle.  Raises if the HIP library is unavailable."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise MonoGSNativeError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C monogs_amd/csrc`.  There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the .so is stale
        fn.restype = res
        fn.argtypes = args
    if lib.mgs_abi_version() != ABI_VERSION:
        raise MonoGSNativeError(f"ABI mismatch: library {lib.mgs_abi_version()} vs binding {ABI_VERSION}; rebuild")
    # measurement knobs for A/B runs of whole programs (tools/, bench.py): MGS_DEBUG_OPTIONS="name=value,name=value"
    # -> mgs_debug_set_option at load time (read here, once; nothing on the launch path consults the environment)
    for kv in filter(None, os.environ.get("MGS_DEBUG_OPTIONS", "").split(",")):
        k, _, v = kv.partition("=")
        if lib.mgs_debug_set_option(k.strip().encode(), int(v)) != 0:
            raise MonoGSNativeError(f"MGS_DEBUG_OPTIONS: {lib.mgs_last_error().decode()}")
    _lib = lib
    return lib


def check(rc: int, what: str):
    if rc != 0:
        msg = load().mgs_last_error().decode("utf-8", "replace")
        # argument errors mirror the upstream extension's Python exceptions
        raise Exception(msg) if rc == 1 else MonoGSNativeError(f"{what}: {msg}")
