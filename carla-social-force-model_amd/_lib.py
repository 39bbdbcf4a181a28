"""ctypes binding of libsfm_hip.so (C ABI: include/sfm_hip.h).

The library is the product: there is no CPU fallback.  ``load()`` raises ``SfmLibraryError`` when the
shared object has not been built (``python -c 'import __graft_entry__ as g; g.build()'`` or
``make -C carla-social-force-model_amd/csrc``), and every compute entry point raises when no MI355X is
visible.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SFM_LIB_PATH") or os.path.join(PKG_DIR, "libsfm_hip.so")   # (the override is for A/B builds of the kernels)
ABI_VERSION = 5
SINCE = {"sfm_step_packed": 4, "sfm_set_dynamic_obstacles_packed": 4, "sfm_step_records": 5}      # entry points younger than ABI 3: an OLDER build named by SFM_LIB_PATH (A/B of builds) may lack them

FORCE_NAMES = ("acceleration_force", "pedestrian_force", "border_force",
               "static_obstacle_force", "dynamic_obstacle_force")
FORCE_TOTAL = 5

TICK_INTEGRATE = 1
TICK_REDRAW_WAYPOINTS = 2
TICK_RECORD_FORCES = 4


class SfmLibraryError(RuntimeError):
    """libsfm_hip.so is missing, stale, or a call into it failed."""


class SfmInteraction(C.Structure):
    _fields_ = [("lambda_", C.c_float), ("A", C.c_float), ("gamma", C.c_float), ("n", C.c_float),
                ("n_prime", C.c_float), ("epsilon", C.c_float), ("perception_threshold", C.c_float)]


class SfmParamsC(C.Structure):
    _fields_ = [("use_ped_radius", C.c_int32), ("max_speed_factor", C.c_float), ("tau", C.c_float),
                ("step_length", C.c_float), ("enabled", C.c_int32 * 5),
                ("pedestrian", SfmInteraction), ("border_a", C.c_float), ("border_b", C.c_float),
                ("static_obstacle", SfmInteraction), ("dynamic_obstacle", SfmInteraction)]


# float* / int32_t* / uint8_t* / uint32_t* parameters are typed void* here and receive plain addresses
# (ndarray.ctypes.data): a typed ctypes cast costs ~8 us per array, and the drop-in tick passes twenty of them
_F = _I = _U8 = _U32 = C.c_void_p
_H = C.c_void_p

# name -> (restype, argtypes): every symbol include/sfm_hip.h declares
SYMBOLS = {
    "sfm_abi_version": (C.c_int, []),
    "sfm_create": (C.c_int, [C.POINTER(SfmParamsC), C.c_int, C.POINTER(_H)]),
    "sfm_destroy": (C.c_int, [_H]),
    "sfm_set_params": (C.c_int, [_H, C.POINTER(SfmParamsC)]),
    "sfm_set_stream": (C.c_int, [_H, C.c_void_p]),
    "sfm_set_borders": (C.c_int, [_H, C.c_int, _I, _F, _F, _F, _F, _F]),
    "sfm_set_static_obstacles": (C.c_int, [_H, C.c_int, _I, _F, _F, _F, _F]),
    "sfm_set_dynamic_obstacles": (C.c_int, [_H, C.c_int, _I, _F, _F, _F, _F, _F, _F]),
    "sfm_set_dynamic_boxes": (C.c_int, [_H, C.c_int, _I, _F, _F, _F, _F, _F, _F, _F, _F]),
    "sfm_download_dynamic_obstacles": (C.c_int, [_H, _F, _F, _F, _F]),
    "sfm_upload_state": (C.c_int, [_H, C.c_int, _F, _F, _F, _F, _F, _F, _F, _F, _F, _F, _U8]),
    "sfm_set_shard": (C.c_int, [_H, C.c_int, C.c_int]),
    "sfm_set_waypoint_stream": (C.c_int, [_H, C.c_uint32, C.c_float, C.c_float]),
    "sfm_set_mode_fsm": (C.c_int, [_H, C.c_int, _U8, _F, _F, _F, _F, _F, _I, _F, _F, _U8, C.c_int, C.c_float, _F]),
    "sfm_download_modes": (C.c_int, [_H, _U8, _F, _I]),
    "sfm_tick": (C.c_int, [_H, C.c_uint32]),
    "sfm_run": (C.c_int, [_H, C.c_int, C.c_uint32]),
    "sfm_run_recorded": (C.c_int, [_H, C.c_int, C.c_uint32, C.c_int, _F, C.c_int, C.POINTER(C.c_int)]),
    "sfm_download_velocities": (C.c_int, [_H, _F, _F, _F]),
    "sfm_step_packed": (C.c_int, [_H, C.c_int, _F, _F, C.c_uint32, _F]),
    "sfm_step_records": (C.c_int, [_H, C.c_int, C.c_void_p, C.c_int64, _I, _U8, C.c_float, C.c_uint32, _F, C.POINTER(C.c_int32)]),
    "sfm_set_dynamic_obstacles_packed": (C.c_int, [_H, C.c_int, _I, _F, _F]),
    "sfm_download_state": (C.c_int, [_H, _F, _F, _F, _F, _F, _F, _F, _F]),
    "sfm_download_forces": (C.c_int, [_H, C.c_int, _F, _F, _F]),
    "sfm_get_arrived": (C.c_int, [_H, C.c_float, _U8]),
    "sfm_download_draw_counts": (C.c_int, [_H, _U32]),
    "sfm_packed_state_ptr": (C.c_void_p, [_H, C.POINTER(C.c_int)]),
    "sfm_packed_z_ptr": (C.c_void_p, [_H]),
    "sfm_row_data_ptr": (C.c_void_p, [_H, C.c_int, C.POINTER(C.c_int)]),
    "sfm_resort": (C.c_int, [_H]),
    "sfm_last_error": (C.c_char_p, [_H]),
    "sfm_get_timing": (C.c_int, [_H, C.POINTER(C.c_float), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "sfm_set_timing": (C.c_int, [_H, C.c_int]),
    "sfm_profile_dominant_kernel": (C.c_int, [_H, C.c_int, C.POINTER(C.c_float)]),
    "sfm_kernel_variant": (C.c_char_p, [_H]),
    "sfm_get_pair_work": (C.c_int, [_H, C.POINTER(C.c_longlong), C.POINTER(C.c_longlong)]),
    "sfm_set_partition": (C.c_int, [_H, C.c_int, C.c_int, _I]),
    "sfm_tick_begin": (C.c_int, [_H, C.c_uint32]),
    "sfm_tick_end": (C.c_int, [_H, C.c_uint32]),
}

_lib = None


def load():
    """dlopen libsfm_hip.so once and type every entry point.  Fails loudly."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise SfmLibraryError(
            f"{LIB_PATH} not found: the HIP extension has not been built "
            "(run `make -C carla-social-force-model_amd/csrc` or __graft_entry__.build()). "
            "There is no CPU fallback.")
    try:
        # torch bundles its own libamdhip64 (same SONAME): import it first so that one HIP runtime serves
        # both torch (streams, RCCL) and this library.
        import torch  # noqa: F401
    except Exception:  # pragma: no cover - torch is plumbing only
        pass
    try:
        lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    except OSError as e:
        raise SfmLibraryError(f"cannot load {LIB_PATH}: {e}") from e
    try:
        lib.sfm_abi_version.restype = C.c_int
        v = lib.sfm_abi_version()
    except AttributeError as e:
        raise SfmLibraryError(f"{LIB_PATH} does not export sfm_abi_version (stale build?)") from e
    # the in-tree build must be THE version; another build named by SFM_LIB_PATH (tools/ab_lib.sh) may be older, additions-only ABI
    if v != ABI_VERSION and not (os.environ.get("SFM_LIB_PATH") and 3 <= v < ABI_VERSION):
        raise SfmLibraryError(f"libsfm_hip ABI {v} != expected {ABI_VERSION}: rebuild the extension")
    for name, (res, args) in SYMBOLS.items():
        if SINCE.get(name, 0) > v:
            continue
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise SfmLibraryError(f"{LIB_PATH} does not export {name} (stale build?)") from e
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def fptr(a):
    """float32 C-contiguous ndarray (or None) -> address for a float* parameter (keep ``a`` alive across the call)"""
    if a is None:
        return None
    assert a.dtype == np.float32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data


def u8ptr(a):
    if a is None:
        return None
    assert a.dtype == np.uint8 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data


def iptr(a):
    if a is None:
        return None
    assert a.dtype == np.int32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data


def f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)
