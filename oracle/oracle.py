"""ctypes wrapper around oracle/libstepth_oracle.so -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module;
the product package (stepth_amd/) must never do so.  See oracle/stepth_oracle.h for what is
restated from which reference lines and for the parity status.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Optional, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libstepth_oracle.so")

SERIAL, REF8, ALLCORES = 0, 1, 2
OK, ERR_BAD_ARG, ERR_MAX_ZERO = 0, -1, -4


class Group(C.Structure):
    _fields_ = [("x", C.c_uint32), ("y", C.c_uint32), ("w", C.c_uint32), ("h", C.c_uint32),
                ("value", C.c_uint8 * 3), ("pad", C.c_uint8)]


GROUP_DTYPE = np.dtype([("x", "<u4"), ("y", "<u4"), ("w", "<u4"), ("h", "<u4"),
                        ("value", "u1", (3,)), ("pad", "u1")])
assert GROUP_DTYPE.itemsize == C.sizeof(Group) == 20


def build(force: bool = False, lib_path: str = _LIB_PATH, march: Optional[str] = None) -> str:
    """Compile the oracle with gcc (plain C, no GPU)."""
    src = os.path.join(_HERE, "stepth_oracle.c")
    if (not force and os.path.exists(lib_path)
            and os.path.getmtime(lib_path) >= max(os.path.getmtime(src),
                                                  os.path.getmtime(src[:-2] + ".h"))):
        return lib_path
    cmd = ["gcc", "-O3", f"-march={march or 'x86-64-v2'}", "-ffp-contract=off", "-fPIC", "-Wall",
           "-Wextra", "-std=c11", "-D_DEFAULT_SOURCE", src, "-o", lib_path, "-shared", "-lm",
           "-lpthread"]
    subprocess.run(cmd, check=True)
    return lib_path


_lib = None


def lib(path: Optional[str] = None):
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or _LIB_PATH
    if not os.path.exists(p):
        build(lib_path=p)
    L = C.CDLL(p)
    u8p, u32p, u64p = C.POINTER(C.c_uint8), C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)
    L.oracle_distance_dot_dot.restype = C.c_uint32
    L.oracle_distance_dot_dot.argtypes = [C.c_uint32] * 4
    L.oracle_distance_dot_array.restype = C.c_int
    L.oracle_distance_dot_array.argtypes = [u8p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_size_t,
                                            C.c_uint32, C.c_uint32, C.c_uint32, u8p, u32p, u32p,
                                            u32p, u64p, u64p]
    L.oracle_uniform_groups.restype = C.c_size_t
    L.oracle_uniform_groups.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_size_t,
                                        C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p]
    L.oracle_adaptive_groups.restype = C.c_size_t
    L.oracle_adaptive_groups.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_size_t,
                                         C.c_uint32, u8p, C.c_uint32, C.c_uint32, C.c_void_p]
    L.oracle_match_groups.restype = C.c_int
    L.oracle_match_groups.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_uint32, C.c_uint32,
                                      C.c_size_t, C.c_uint32, u8p, C.c_int, C.c_void_p,
                                      C.c_void_p, C.c_void_p, C.c_void_p, u64p]
    L.oracle_normalise.restype = C.c_int
    L.oracle_normalise.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, u8p]
    L.oracle_paint.restype = None
    L.oracle_paint.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint32, C.c_uint32,
                               C.c_void_p, C.c_size_t]
    L.oracle_depth_from_additional.restype = C.c_int
    L.oracle_depth_from_additional.argtypes = [
        C.c_void_p, C.c_uint32, C.c_uint32, C.c_size_t, C.c_uint32, C.c_void_p, C.c_uint32,
        C.c_uint32, C.c_size_t, u8p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_size_t,
        C.c_int, C.c_void_p, C.c_void_p, u8p, u64p]
    L.oracle_sad_rows.restype = C.c_int
    L.oracle_sad_rows.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_size_t,
                                  C.c_size_t, C.c_uint32, C.c_uint32, C.c_uint32, u8p,
                                  C.c_uint32, C.c_uint32, C.c_int, C.c_void_p]
    L.oracle_gaussian_same_size.restype = None
    L.oracle_gaussian_same_size.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_size_t,
                                            C.c_void_p, C.c_size_t]
    L.oracle_depth_split.restype = C.c_int
    L.oracle_depth_split.argtypes = [C.c_void_p, C.c_size_t, C.c_uint32, u8p, u8p, u8p,
                                     C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    L.oracle_invert_depth.restype = None
    L.oracle_invert_depth.argtypes = [C.c_void_p, C.c_size_t]
    L.oracle_slice.restype = None
    L.oracle_slice.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_uint8, C.c_int, C.c_uint8,
                               C.c_void_p]
    L.oracle_select_foreground.restype = C.c_int
    L.oracle_select_foreground.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, u8p, u8p]
    L.oracle_apply_mask.restype = None
    L.oracle_apply_mask.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    L.oracle_hardware_threads.restype = C.c_int
    L.oracle_set_threads.restype = None
    L.oracle_set_threads.argtypes = [C.c_int]
    if path is None:
        _lib = L
    return L


def _u8x3(v):
    return (C.c_uint8 * 3)(*[int(x) for x in v])


def _img(a: np.ndarray, channels=(3,)) -> np.ndarray:
    a = np.ascontiguousarray(a, dtype=np.uint8)
    assert a.ndim == 3 and a.shape[2] in channels, a.shape
    return a


def distance_dot_dot(f: Tuple[int, int], s: Tuple[int, int]) -> int:
    return int(lib().oracle_distance_dot_dot(f[0], f[1], s[0], s[1]))


def distance_dot_array(what, array: np.ndarray, from_xy: Tuple[int, int], max_steps: int,
                       precision, count: bool = False):
    """helpers::distance_dot_array -> None or (distance, (x, y)) [, probes, cells]."""
    array = _img(array)
    ah, aw = array.shape[:2]
    d, px, py = C.c_uint32(0), C.c_uint32(0), C.c_uint32(0)
    probes, cells = C.c_uint64(0), C.c_uint64(0)
    found = lib().oracle_distance_dot_array(
        _u8x3(what), array.ctypes.data, aw, ah, array.strides[0], from_xy[0], from_xy[1],
        max_steps, _u8x3(precision), C.byref(d), C.byref(px), C.byref(py), C.byref(probes),
        C.byref(cells))
    res = (int(d.value), (int(px.value), int(py.value))) if found else None
    if count:
        return res, int(probes.value), int(cells.value)
    return res


def uniform_groups(main: np.ndarray, tile_w: int, tile_h: int) -> np.ndarray:
    main = _img(main, (3, 4))
    h, w, ch = main.shape
    n = lib().oracle_uniform_groups(main.ctypes.data, w, h, main.strides[0], ch, tile_w, tile_h,
                                    None)
    groups = np.zeros(n, dtype=GROUP_DTYPE)
    lib().oracle_uniform_groups(main.ctypes.data, w, h, main.strides[0], ch, tile_w, tile_h,
                                groups.ctypes.data)
    return groups


def adaptive_groups(main: np.ndarray, precision, min_splits: int = 16,
                    max_splits: int = 0) -> np.ndarray:
    """Adaptive stand-in for disage::open::rgb_discrete (own design, parity unpinned):
    GROUP_DTYPE array in depth-first order; max_splits = 0 -> ceil(log2(w*h)) like
    src/depth_image.rs:101-103."""
    main = _img(main, (3, 4))
    h, w, ch = main.shape
    p = _u8x3(precision)
    n = lib().oracle_adaptive_groups(main.ctypes.data, w, h, main.strides[0], ch, p, min_splits,
                                     max_splits, None)
    g = np.zeros(n, GROUP_DTYPE)
    lib().oracle_adaptive_groups(main.ctypes.data, w, h, main.strides[0], ch, p, min_splits,
                                 max_splits, g.ctypes.data)
    return g


def match_groups(groups: np.ndarray, add: np.ndarray, max_steps: int, precision,
                 threads: int = ALLCORES):
    """PASS 1 -> dict(raw u8 [n], raw32 u32 [n], pos [n,2], found [n], probes)."""
    add = _img(add)
    groups = np.ascontiguousarray(groups, dtype=GROUP_DTYPE)
    ah, aw = add.shape[:2]
    raw = np.zeros(len(groups), np.uint8)
    raw32 = np.zeros(len(groups), np.uint32)
    pos = np.zeros((len(groups), 2), np.uint32)
    found = np.zeros(len(groups), np.uint8)
    probes = C.c_uint64(0)
    rc = lib().oracle_match_groups(groups.ctypes.data, len(groups), add.ctypes.data, aw, ah,
                                   add.strides[0], max_steps, _u8x3(precision), threads,
                                   raw.ctypes.data, raw32.ctypes.data, pos.ctypes.data,
                                   found.ctypes.data, C.byref(probes))
    assert rc == OK, rc
    return {"raw": raw, "raw32": raw32, "pos": pos, "found": found,
            "probes": int(probes.value)}


def normalise(raw: np.ndarray):
    raw = np.ascontiguousarray(raw, dtype=np.uint8).ravel()
    out = np.zeros_like(raw)
    mx = C.c_uint8(0)
    rc = lib().oracle_normalise(raw.ctypes.data, raw.size, out.ctypes.data, C.byref(mx))
    return rc, out, int(mx.value)


def paint(groups: np.ndarray, values: np.ndarray, w: int, h: int) -> np.ndarray:
    groups = np.ascontiguousarray(groups, dtype=GROUP_DTYPE)
    values = np.ascontiguousarray(values, dtype=np.uint8)
    depth = np.zeros((h, w), np.uint8)
    lib().oracle_paint(groups.ctypes.data, values.ctypes.data, len(groups), w, h,
                       depth.ctypes.data, w)
    return depth


def depth_from_additional(main: np.ndarray, add: np.ndarray, precision, max_steps: int = 255,
                          tile: Tuple[int, int] = (1, 1), groups: Optional[np.ndarray] = None,
                          threads: int = ALLCORES):
    """Whole path (no Gaussian tail) -> dict(status, depth, raw, max, probes)."""
    main = _img(main, (3, 4))
    add = _img(add)
    h, w, ch = main.shape
    ah, aw = add.shape[:2]
    depth = np.zeros((h, w), np.uint8)
    raw = np.zeros((h, w), np.uint8)
    mx, probes = C.c_uint8(0), C.c_uint64(0)
    gptr, gn = None, 0
    if groups is not None:
        groups = np.ascontiguousarray(groups, dtype=GROUP_DTYPE)
        gptr, gn = groups.ctypes.data, len(groups)
    rc = lib().oracle_depth_from_additional(
        main.ctypes.data, w, h, main.strides[0], ch, add.ctypes.data, aw, ah, add.strides[0],
        _u8x3(precision), max_steps, tile[0], tile[1], gptr, gn, threads, depth.ctypes.data,
        raw.ctypes.data, C.byref(mx), C.byref(probes))
    return {"status": rc, "depth": depth, "raw": raw, "max": int(mx.value),
            "probes": int(probes.value)}


def sad_rows(left: np.ndarray, right: np.ndarray, d_max: int, win: int, precision,
             row0: int = 0, row1: Optional[int] = None, threads: int = ALLCORES) -> np.ndarray:
    left = _img(left, (3, 4))
    right = _img(right)
    h, w, ch = left.shape
    assert right.shape[:2] == (h, w)
    row1 = h if row1 is None else row1
    raw = np.zeros((h, w), np.uint8)
    rc = lib().oracle_sad_rows(left.ctypes.data, right.ctypes.data, w, h, left.strides[0],
                               right.strides[0], ch, d_max, win, _u8x3(precision), row0, row1,
                               threads, raw.ctypes.data)
    assert rc == OK, rc
    return raw


def gaussian_same_size(img: np.ndarray) -> np.ndarray:
    """src/depth_image.rs:133-134 (image 0.23.14 sampler restated from memory; unpinned)"""
    img = np.ascontiguousarray(img, dtype=np.uint8)
    assert img.ndim == 2
    out = np.zeros_like(img)
    h, w = img.shape
    lib().oracle_gaussian_same_size(img.ctypes.data, w, h, img.strides[0], out.ctypes.data, w)
    return out


SPLIT_ERR_STEP, SPLIT_ERR_EMPTY, SPLIT_ERR_DIVERGED, SPLIT_ERR_INPUT = -7, -8, -9, -1


def depth_split(depth: np.ndarray, zones: int):
    """src/depth_image.rs:162-219 -> {"status", "zones": [(lo|None, hi|None), ...], "iters"}"""
    depth = np.ascontiguousarray(depth, dtype=np.uint8)
    lo, hi, has = ((C.c_uint8 * 256)() for _ in range(3))
    n, it = C.c_uint32(0), C.c_uint32(0)
    rc = lib().oracle_depth_split(depth.ctypes.data, depth.size, zones, lo, hi, has,
                                  C.byref(n), C.byref(it))
    zs = [((lo[i], hi[i]) if has[i] else (None, None)) for i in range(n.value)] if rc == 0 else []
    return {"status": rc, "zones": zs, "iters": it.value}


def invert_depth(depth: np.ndarray) -> np.ndarray:
    """src/depth_image.rs:226-228 (returns a new array)"""
    out = np.array(depth, dtype=np.uint8, order="C")
    lib().oracle_invert_depth(out.ctypes.data, out.size)
    return out


def slice_mask(depth: np.ndarray, from_: Optional[int], to: Optional[int]) -> np.ndarray:
    """src/depth_image.rs:230-247, the mask plane only"""
    depth = np.ascontiguousarray(depth, dtype=np.uint8)
    mask = np.empty_like(depth)
    lib().oracle_slice(depth.ctypes.data, depth.size, from_ is not None, from_ or 0,
                       to is not None, 255 if to is None else to, mask.ctypes.data)
    return mask


def select_foreground(depth: np.ndarray):
    """src/depth_image.rs:221-224 -> {"status", "mask", "zone"}"""
    depth = np.ascontiguousarray(depth, dtype=np.uint8)
    mask = np.zeros_like(depth)
    f, t = C.c_uint8(0), C.c_uint8(0)
    rc = lib().oracle_select_foreground(depth.ctypes.data, depth.size, mask.ctypes.data,
                                        C.byref(f), C.byref(t))
    return {"status": rc, "mask": mask, "zone": (f.value, t.value) if rc == 0 else None}


def apply_mask(rgba: np.ndarray, mask: np.ndarray) -> np.ndarray:
    """src/mask_image.rs:205-213 (returns a new array)"""
    out = np.array(rgba, dtype=np.uint8, order="C")
    assert out.ndim == 3 and out.shape[2] == 4 and mask.shape == out.shape[:2]
    mask = np.ascontiguousarray(mask, dtype=np.uint8)
    lib().oracle_apply_mask(out.ctypes.data, mask.ctypes.data, mask.size)
    return out


def hardware_threads() -> int:
    return int(lib().oracle_hardware_threads())


def usable_cpus() -> int:
    """CPUs this process may actually use: affinity mask capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period) + 0.5)))
    except Exception:
        pass
    return max(1, n)


def set_threads(n: int) -> None:
    lib().oracle_set_threads(int(n))
