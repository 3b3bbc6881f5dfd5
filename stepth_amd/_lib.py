"""ctypes binding of include/stepth_amd.h (the C-ABI of the HIP engine).

There is no CPU fallback: if libstepth_amd.so is missing or does not load, importing the
engine raises.  Nothing here touches oracle/ (test infrastructure).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
# STEPTH_AMD_LIB: another build of the same library (A/B measurements of tuning builds only)
LIB_PATH = os.environ.get("STEPTH_AMD_LIB") or os.path.join(_HERE, "libstepth_amd.so")
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "stepth_amd.h")

ABI_VERSION = 4

# stepth_status
OK = 0
ERR_BAD_ARG = -1
ERR_SIZE_MISMATCH = -2
ERR_HIP = -3
ERR_MAX_ZERO = -4
ERR_NO_DEVICE = -5
ERR_OUT_OF_MEMORY = -6
ERR_SPLIT_STEP = -7
ERR_SPLIT_EMPTY_ZONE = -8
ERR_SPLIT_DIVERGED = -9

MODE_RING_FIRST_HIT = 0
MODE_SAD_WINDOW = 1
MEM_HOST = 0
MEM_DEVICE = 1
FLAG_ASYNC = 1
FLAG_RAW_ONLY = 2
FLAG_GAUSSIAN_TAIL = 4
FLAG_HEAVY_FIRST = 8


class Group(C.Structure):
    """disage::DiscretePixel{position,size,value} (reference src/depth_image.rs:110-121)."""
    _fields_ = [("x", C.c_uint32), ("y", C.c_uint32), ("w", C.c_uint32), ("h", C.c_uint32),
                ("value", C.c_uint8 * 3), ("pad", C.c_uint8)]


class Image(C.Structure):
    _fields_ = [("data", C.c_void_p), ("width", C.c_uint32), ("height", C.c_uint32),
                ("channels", C.c_uint32), ("memory", C.c_uint32), ("stride", C.c_size_t)]


class Params(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("mode", C.c_uint32),
                ("precision", C.c_uint8 * 3), ("reserved0", C.c_uint8),
                ("max_steps", C.c_uint32), ("tile_w", C.c_uint32), ("tile_h", C.c_uint32),
                ("d_max", C.c_uint32), ("win", C.c_uint32), ("row_begin", C.c_uint32),
                ("row_end", C.c_uint32), ("flags", C.c_uint32), ("row_block", C.c_uint32),
                ("row_mod", C.c_uint32), ("row_rem", C.c_uint32)]


class Stats(C.Structure):
    _fields_ = [("groups", C.c_uint64), ("ring0_hits", C.c_uint64), ("probes", C.c_uint64),
                ("near_steps", C.c_uint64), ("survey_steps", C.c_uint64),
                ("superblock_steps", C.c_uint64), ("blocks_probed", C.c_uint64)]


class Zone(C.Structure):
    """one (Option<u8>, Option<u8>) of depth_split / slice (reference src/depth_image.rs:162)."""
    _fields_ = [("has_from", C.c_uint8), ("from_", C.c_uint8), ("has_to", C.c_uint8),
                ("to", C.c_uint8)]


# every symbol include/stepth_amd.h declares: name -> (restype, argtypes)
_P = C.POINTER
SYMBOLS = {
    "stepth_abi_version": (C.c_uint32, []),
    "stepth_status_string": (C.c_char_p, [C.c_int]),
    "stepth_engine_create": (C.c_int, [C.c_int, _P(C.c_void_p)]),
    "stepth_engine_destroy": (None, [C.c_void_p]),
    "stepth_engine_set_stream": (C.c_int, [C.c_void_p, C.c_void_p]),
    "stepth_engine_reset_stream": (C.c_int, [C.c_void_p]),
    "stepth_engine_sync": (C.c_int, [C.c_void_p]),
    "stepth_last_error": (C.c_char_p, [C.c_void_p]),
    "stepth_depth_from_additional": (C.c_int, [C.c_void_p, _P(Image), _P(Image), _P(Params),
                                               C.c_void_p, C.c_size_t, C.c_uint32, _P(Image),
                                               C.c_void_p]),
    "stepth_adaptive_groups": (C.c_int, [C.c_void_p, _P(Image), C.c_void_p, C.c_uint32, C.c_uint32,
                                         C.c_void_p, C.c_size_t, C.c_uint32, _P(C.c_size_t)]),
    "stepth_depth_from_additional_batch": (C.c_int, [C.c_void_p, C.c_uint32, _P(Image), _P(Image),
                                                     _P(Params), _P(Image), C.c_void_p]),
    "stepth_adaptive_groups_batch": (C.c_int, [C.c_void_p, C.c_uint32, _P(Image), C.c_void_p, C.c_uint32,
                                               C.c_uint32, _P(C.c_void_p), C.c_size_t, _P(C.c_size_t)]),
    "stepth_depth_from_groups_batch": (C.c_int, [C.c_void_p, C.c_uint32, _P(Image), _P(Image),
                                                 _P(Params), _P(C.c_void_p), _P(C.c_size_t), _P(Image),
                                                 C.c_void_p]),
    "stepth_adaptive_depth_batch": (C.c_int, [C.c_void_p, C.c_uint32, _P(Image), _P(Image), _P(Params), C.c_uint32,
                                              C.c_uint32, _P(Image), C.c_void_p, _P(C.c_size_t)]),
    "stepth_host_alloc": (C.c_int, [_P(C.c_void_p), C.c_size_t]),
    "stepth_host_free": (None, [C.c_void_p]),
    "stepth_normalise_rows": (C.c_int, [C.c_void_p, _P(Image), C.c_void_p, C.c_uint32,
                                        C.c_uint32]),
    "stepth_normalise_rows_pattern": (C.c_int, [C.c_void_p, _P(Image), C.c_uint32, C.c_uint32,
                                                C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint32,
                                                C.c_uint32]),
    "stepth_normalise_rows_batch": (C.c_int, [C.c_void_p, C.c_uint32, _P(Image), C.c_uint32,
                                              C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p,
                                              C.c_uint32]),
    "stepth_gaussian_same_size": (C.c_int, [C.c_void_p, _P(Image), _P(Image), C.c_uint32]),
    "stepth_distance_dot_array_batch": (C.c_int, [C.c_void_p, _P(Image), C.c_void_p, C.c_void_p,
                                                  C.c_size_t, C.c_uint32, C.c_void_p,
                                                  C.c_void_p, C.c_void_p, C.c_void_p]),
    "stepth_invert_depth": (C.c_int, [C.c_void_p, _P(Image), C.c_uint32]),
    "stepth_depth_split": (C.c_int, [C.c_void_p, _P(Image), C.c_uint32, _P(Zone), C.c_uint32,
                                     _P(C.c_uint32)]),
    "stepth_depth_histogram": (C.c_int, [C.c_void_p, _P(Image), C.c_void_p, C.c_uint32, C.c_uint32]),
    "stepth_zones_from_histogram": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32,
                                              _P(Zone), C.c_uint32, _P(C.c_uint32)]),
    "stepth_slice": (C.c_int, [C.c_void_p, _P(Image), _P(Zone), _P(Image), C.c_uint32]),
    "stepth_select_foreground": (C.c_int, [C.c_void_p, _P(Image), _P(Image), _P(Zone)]),
    "stepth_select_foreground_batch": (C.c_int, [C.c_void_p, C.c_uint32, _P(Image), _P(Image),
                                                 _P(Zone), _P(C.c_int), C.c_uint32]),
    "stepth_apply_mask": (C.c_int, [C.c_void_p, _P(Image), _P(Image), C.c_uint32]),
    "stepth_stats_enable": (C.c_int, [C.c_void_p, C.c_int]),
    "stepth_get_stats": (C.c_int, [C.c_void_p, _P(Stats)]),
    "stepth_get_stats_raw": (C.c_int, [C.c_void_p, _P(C.c_uint64), C.c_uint32]),
    "stepth_kernel_time_enable": (C.c_int, [C.c_void_p, C.c_int]),
    "stepth_kernel_time_read": (C.c_int, [C.c_void_p, _P(C.c_double), _P(C.c_uint64)]),
    "stepth_selftest_exception_guard": (C.c_int, [C.c_void_p, C.c_int]),
}


def build(force: bool = False) -> str:
    """hipcc --offload-arch=gfx950 -> stepth_amd/libstepth_amd.so (cross-compiles w/o a GPU)."""
    csrc = os.path.join(_HERE, "csrc")
    cmd = ["make", "-C", csrc] + (["-B"] if force else [])
    subprocess.run(cmd, check=True)
    return LIB_PATH


_lib = None


def load():
    """Load the HIP engine library; raises (never falls back) if it is absent or broken."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: the stepth_amd HIP extension is not built. Run "
            "`python -c 'import __graft_entry__ as g; g.build()'` (or make -C stepth_amd/csrc). "
            "There is no CPU fallback.")
    # PyTorch-ROCm wheels bundle their own libamdhip64/libhsa-runtime64.  Two HSA runtimes in
    # one process cannot both own the GPU ("No HIP GPUs are available" for whichever comes
    # second), so when torch is installed it is imported first: the engine library then binds
    # to the runtime torch has already loaded (same SONAME) and both share it.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if the .so is stale
        fn.restype = res
        fn.argtypes = args
    got = lib.stepth_abi_version()
    if got != ABI_VERSION:
        raise ImportError(f"libstepth_amd.so ABI {got} != binding ABI {ABI_VERSION}; rebuild")
    _lib = lib
    return lib
