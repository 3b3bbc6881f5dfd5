"""Thin Python handle over the C-ABI engine (include/stepth_amd.h).

Host arrays are numpy uint8, device arrays are torch CUDA(=HIP) uint8 tensors; torch is only
plumbing for device memory and streams.  Every compute call goes through libstepth_amd.so --
there is no Python/numpy implementation of the path in this package.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence, Tuple

import numpy as np

from . import _lib as L


class StepthError(RuntimeError):
    def __init__(self, status: int, message: str):
        super().__init__(f"stepth status {status}: {message}")
        self.status = status


class MaxZeroError(StepthError):
    """Every distance is 0 -- the reference panics here (src/depth_image.rs:127)."""


class SplitError(StepthError):
    """depth_split input on which the reference panics (src/depth_image.rs:203-217)."""


def _is_tensor(a) -> bool:
    return hasattr(a, "data_ptr") and hasattr(a, "is_cuda")


def _describe(a, channels: Sequence[int], name: str) -> Tuple[L.Image, object]:
    """-> (stepth_image, keepalive).  Accepts [H,W,C] (or [H,W] for 1 channel)."""
    if _is_tensor(a):
        import torch
        if a.dtype != torch.uint8:
            raise TypeError(f"{name}: expected uint8 tensor, got {a.dtype}")
        if not a.is_cuda:
            raise TypeError(f"{name}: torch tensors must live on the GPU (use numpy for host)")
        t = a if a.dim() == 3 else a.unsqueeze(-1)
        if t.dim() != 3 or t.shape[2] not in channels:
            raise ValueError(f"{name}: shape {tuple(a.shape)} not [H,W,{list(channels)}]")
        if (t.shape[2] > 1 and t.stride(2) != 1) or (t.shape[1] > 1 and t.stride(1) != t.shape[2]):
            raise ValueError(f"{name}: pixels must be contiguous within a row")
        im = L.Image(t.data_ptr(), t.shape[1], t.shape[0], t.shape[2], L.MEM_DEVICE,
                     t.stride(0) if t.shape[0] > 1 else t.shape[1] * t.shape[2])
        return im, a
    arr = np.asarray(a)
    if arr.dtype != np.uint8:
        raise TypeError(f"{name}: expected uint8 array, got {arr.dtype}")
    v = arr if arr.ndim == 3 else arr[..., None]
    if v.ndim != 3 or v.shape[2] not in channels:
        raise ValueError(f"{name}: shape {arr.shape} not [H,W,{list(channels)}]")
    if ((v.shape[2] > 1 and v.strides[2] != 1) or
            (v.shape[1] > 1 and v.strides[1] != v.shape[2])):
        raise ValueError(f"{name}: pixels must be contiguous within a row")
    stride = v.strides[0] if v.shape[0] > 1 else v.shape[1] * v.shape[2]
    im = L.Image(v.ctypes.data, v.shape[1], v.shape[0], v.shape[2], L.MEM_HOST, stride)
    return im, v


def _u8x3(v) -> "C.Array":
    v = [int(x) for x in v]
    if len(v) != 3 or any(x < 0 or x > 255 for x in v):
        raise ValueError("precision must be three u8 values")
    return (C.c_uint8 * 3)(*v)


GROUP_DTYPE = np.dtype([("x", "<u4"), ("y", "<u4"), ("w", "<u4"), ("h", "<u4"),
                        ("value", "u1", (3,)), ("pad", "u1")])


class _PinnedBlock:
    """owner of one stepth_host_alloc block; numpy keeps it alive as the array's base"""

    def __init__(self, lib, ptr: int, n: int):
        self._lib, self._ptr = lib, ptr
        self.__array_interface__ = {"shape": (n,), "typestr": "|u1", "data": (ptr, False),
                                    "version": 3}

    def __del__(self):
        try:
            self._lib.stepth_host_free(self._ptr)
        except Exception:
            pass


class PlaneList:
    """ctypes descriptions of a list of CUDA tensors (built once, reused by batch calls)."""

    def __init__(self, tensors, channels=(1,), name: str = "plane"):
        self.tensors = list(tensors)
        self.images = (L.Image * len(self.tensors))()
        self._keep = []
        for f, t in enumerate(self.tensors):
            self.images[f], k = _describe(t, channels, name)
            self._keep.append(k)

    def __len__(self):
        return len(self.tensors)

    def __getitem__(self, i):
        return self.tensors[i]


class Engine:
    """One engine per device/stream (stepth_engine_create)."""

    def __init__(self, device: int = -1):
        self._lib = L.load()
        h = C.c_void_p()
        st = self._lib.stepth_engine_create(device, C.byref(h))
        if st != L.OK:
            raise StepthError(st, self._lib.stepth_status_string(st).decode())
        self._h = h

    # -- lifetime -------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None):
            self._lib.stepth_engine_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _check(self, st: int):
        if st == L.OK:
            return
        msg = self._lib.stepth_last_error(self._h).decode() or \
            self._lib.stepth_status_string(st).decode()
        kind = (MaxZeroError if st == L.ERR_MAX_ZERO else
                SplitError if L.ERR_SPLIT_DIVERGED <= st <= L.ERR_SPLIT_STEP else StepthError)
        raise kind(st, msg)

    def set_stream(self, hip_stream: int):
        """Run on this hipStream_t handle (0 = HIP's default stream) -- an explicit choice: it
        holds for every later call, CUDA tensors or not, until reset_stream() or
        use_torch_stream().  (The new stream waits for what the old one holds; the host does not.)"""
        self._explicit = True
        self._bound = hip_stream
        self._check(self._lib.stepth_engine_set_stream(self._h, hip_stream or None))

    def reset_stream(self):
        """Back to the engine's own non-blocking stream, and to following torch's current stream
        for calls with CUDA tensors."""
        self._explicit = False
        self._bound = None
        self._check(self._lib.stepth_engine_reset_stream(self._h))

    def use_torch_stream(self):
        """Bind to torch's current stream now, and keep following it from call to call."""
        import torch
        self._explicit = False
        self._bound = torch.cuda.current_stream().cuda_stream
        self._check(self._lib.stepth_engine_set_stream(self._h, self._bound or None))

    def _follow(self, *args):
        """Unless set_stream() made an explicit choice, calls whose arguments are CUDA tensors
        run on torch's CURRENT stream: whatever produced the tensors (fills, earlier kernels,
        collectives) is ordered before the engine's kernels and whatever consumes the results
        after them.  The engine's own non-blocking stream serves numpy (host-memory) calls,
        which synchronise themselves."""
        if getattr(self, "_explicit", False):
            return
        for a in args:
            if isinstance(a, (list, tuple)):
                a = a[0] if a else None
            if a is not None and _is_tensor(a) and a.is_cuda:
                import torch
                cur = torch.cuda.current_stream(a.device).cuda_stream
                if getattr(self, "_bound", None) != cur:
                    self._bound = cur
                    # (device-side ordering after the stream used before; no host wait)
                    self._check(self._lib.stepth_engine_set_stream(self._h, cur or None))
                return

    def sync(self):
        self._check(self._lib.stepth_engine_sync(self._h))

    # -- the path -------------------------------------------------------------------------
    def depth_from_additional(self, main, additional, precision, *, max_steps: int = 255,
                              tile: Tuple[int, int] = (1, 1), groups=None,
                              mode: int = L.MODE_RING_FIRST_HIT, d_max: int = 0, win: int = 0,
                              rows: Optional[Tuple[int, int]] = None,
                              row_pattern: Optional[Tuple[int, int, int]] = None,
                              raw_only: bool = False, gaussian_tail: bool = False, heavy_first: bool = False, out=None,
                              max_out=None,
                              asynchronous: bool = False):
        """stepth_depth_from_additional.  Returns (depth [H,W] u8, max or None).

        numpy in -> numpy out (synchronous).  torch CUDA tensors in -> tensor out; with
        asynchronous=True nothing is synchronised and `max_out` must be a CUDA int32/uint32
        tensor (or None)."""
        self._follow(main)
        mi, k1 = _describe(main, (3, 4), "main")
        ai, k2 = _describe(additional, (3,), "additional")
        dev = _is_tensor(main)
        if dev != _is_tensor(additional):
            raise TypeError("main and additional must both be numpy or both be CUDA tensors")
        h, w = mi.height, mi.width
        if out is None:
            if dev:
                import torch
                out = torch.zeros((h, w), dtype=torch.uint8, device=main.device)
            else:
                out = np.zeros((h, w), np.uint8)
        di, k3 = _describe(out, (1,), "depth")
        p = L.Params()
        p.struct_size = C.sizeof(L.Params)
        p.mode = mode
        p.precision = _u8x3(precision)
        p.max_steps = max_steps
        p.tile_w, p.tile_h = tile
        p.d_max, p.win = d_max, win
        if rows is not None:
            p.row_begin, p.row_end = rows
            if rows[0] == rows[1]:
                return out, 0  # empty strip; 0,0 would mean "all rows" in the C-ABI
        if row_pattern is not None:  # (row_block, row_mod, row_rem): interleaved row blocks
            p.row_block, p.row_mod, p.row_rem = row_pattern
        p.flags = ((L.FLAG_ASYNC if asynchronous else 0) | (L.FLAG_RAW_ONLY if raw_only else 0) |
                   (L.FLAG_GAUSSIAN_TAIL if gaussian_tail else 0) | (L.FLAG_HEAVY_FIRST if heavy_first else 0))
        gptr, gn, gmem, k4 = None, 0, L.MEM_HOST, None
        if groups is not None:
            if _is_tensor(groups):
                gptr, gn, gmem, k4 = groups.data_ptr(), groups.numel() // 20, L.MEM_DEVICE, groups
            else:
                k4 = np.ascontiguousarray(groups, dtype=GROUP_DTYPE)
                gptr, gn = k4.ctypes.data, len(k4)
        if asynchronous:
            mptr = max_out.data_ptr() if max_out is not None else None
            st = self._lib.stepth_depth_from_additional(self._h, C.byref(mi), C.byref(ai),
                                                        C.byref(p), gptr, gn, gmem,
                                                        C.byref(di), mptr)
            self._check(st)
            return out, None
        mx = C.c_uint32(0)
        st = self._lib.stepth_depth_from_additional(self._h, C.byref(mi), C.byref(ai),
                                                    C.byref(p), gptr, gn, gmem, C.byref(di),
                                                    C.cast(C.byref(mx), C.c_void_p))
        del k1, k2, k3, k4
        self._check(st)
        return out, int(mx.value)

    def adaptive_groups(self, main, precision, min_splits: int = 16, max_splits: int = 0):
        """stepth_adaptive_groups: the stand-in for disage::open::rgb_discrete (own design,
        parity unpinned).  numpy image -> GROUP_DTYPE array; CUDA tensor -> uint8 CUDA tensor
        [n, 20] (pass either as `groups=` to depth_from_additional).  Order is arbitrary."""
        self._follow(main)
        mi, _k = _describe(main, (3, 4), "main")
        prec = _u8x3(precision)
        n = C.c_size_t(0)
        cap = mi.width * mi.height
        if _is_tensor(main):
            import torch
            buf = torch.empty((cap, 20), dtype=torch.uint8, device=main.device)
            self._check(self._lib.stepth_adaptive_groups(self._h, C.byref(mi), prec, min_splits,
                                                         max_splits, buf.data_ptr(), cap,
                                                         L.MEM_DEVICE, C.byref(n)))
            return buf[: n.value]
        out = np.empty(cap, GROUP_DTYPE)
        self._check(self._lib.stepth_adaptive_groups(self._h, C.byref(mi), prec, min_splits,
                                                     max_splits, out.ctypes.data, cap,
                                                     L.MEM_HOST, C.byref(n)))
        return out[: n.value].copy()

    def depth_from_additional_batch(self, mains, additionals, precision, *, max_steps: int = 255,
                                    tile: Tuple[int, int] = (1, 1),
                                    mode: int = L.MODE_RING_FIRST_HIT, d_max: int = 0, win: int = 0,
                                    rows: Optional[Tuple[int, int]] = None,
                                    row_pattern: Optional[Tuple[int, int, int]] = None,
                                    raw_only: bool = False, gaussian_tail: bool = False, heavy_first: bool = False,
                                    outs=None, max_out=None, asynchronous: bool = False):
        """stepth_depth_from_additional_batch: many frames of equal shape in one set of launches,
        ring mode with uniform tiles (`tile`, default one group per pixel), or the windowed-SAD
        mode (`mode`, `d_max`, `win`).  Frames are CUDA uint8 tensors, or numpy arrays (host
        memory: uploaded, computed and downloaded in overlapping chunks; page-locked arrays
        from host_array() copy fastest).  Returns (outs, max): `outs` a list of [H,W] tensors
        / arrays; max a list of ints, or None with asynchronous=True (then `max_out` must be a
        CUDA int32 tensor with one element per frame, or None)."""
        self._follow(mains.tensors if isinstance(mains, PlaneList) else mains)
        n = len(mains)
        if n == 0 or len(additionals) != n:
            raise ValueError("mains and additionals must be equally long, non-empty lists")
        if outs is None:
            if isinstance(mains, PlaneList):
                mains_t = mains.tensors
            else:
                mains_t = mains
            if _is_tensor(mains_t[0]):
                import torch
                outs = [torch.zeros(m.shape[:2], dtype=torch.uint8, device=m.device) for m in mains_t]
            else:
                outs = [np.zeros(m.shape[:2], np.uint8) for m in mains_t]
        # PlaneList arguments carry their descriptions (built once): a step of 64 frames is 192
        # images, and describing them anew on every call costs the host more than the GPU takes
        mi = mains.images if isinstance(mains, PlaneList) else PlaneList(mains, (3, 4), "main").images
        ai = additionals.images if isinstance(additionals, PlaneList) else \
            PlaneList(additionals, (3,), "additional").images
        di = outs.images if isinstance(outs, PlaneList) else PlaneList(outs, (1,), "depth").images
        p = L.Params()
        p.struct_size = C.sizeof(L.Params)
        p.mode = mode
        p.precision = _u8x3(precision)
        p.max_steps = max_steps
        p.tile_w, p.tile_h = tile
        p.d_max, p.win = d_max, win
        if rows is not None:
            p.row_begin, p.row_end = rows
        if row_pattern is not None:
            p.row_block, p.row_mod, p.row_rem = row_pattern
        p.flags = ((L.FLAG_ASYNC if asynchronous else 0) | (L.FLAG_RAW_ONLY if raw_only else 0) |
                   (L.FLAG_GAUSSIAN_TAIL if gaussian_tail else 0) | (L.FLAG_HEAVY_FIRST if heavy_first else 0))
        if asynchronous:
            mptr = max_out.data_ptr() if max_out is not None else None
            self._check(self._lib.stepth_depth_from_additional_batch(self._h, n, mi, ai,
                                                                     C.byref(p), di, mptr))
            return outs, None
        mx = (C.c_uint32 * n)()
        st = self._lib.stepth_depth_from_additional_batch(self._h, n, mi, ai, C.byref(p), di,
                                                          C.cast(mx, C.c_void_p))
        self._check(st)
        return outs, [int(v) for v in mx]

    def adaptive_groups_batch(self, mains, precision, min_splits: int = 16, max_splits: int = 0,
                              buffers=None):
        """stepth_adaptive_groups_batch: adaptive groups of CUDA frames of one shape, one set of
        launches.  -> (buffers, counts): `buffers` a CUDA uint8 tensor [F, H*W, 20] (pass it
        back in to reuse it), counts a list of F ints; frame f's groups = buffers[f, :counts[f]]."""
        import torch
        self._follow(mains)
        n = len(mains)
        mi = (L.Image * n)()
        keep = []
        for f in range(n):
            mi[f], k = _describe(mains[f], (3, 4), "main")
            keep.append(k)
        cap = mi[0].width * mi[0].height
        if buffers is None or tuple(buffers.shape) != (n, cap, 20):
            buffers = torch.empty((n, cap, 20), dtype=torch.uint8, device=mains[0].device)
        ptrs = (C.c_void_p * n)(*[buffers[f].data_ptr() for f in range(n)])
        counts = (C.c_size_t * n)()
        self._check(self._lib.stepth_adaptive_groups_batch(self._h, n, mi, _u8x3(precision), min_splits,
                                                           max_splits, ptrs, cap, counts))
        return buffers, [int(c) for c in counts]

    def depth_from_groups_batch(self, mains, additionals, precision, groups, counts, *,
                                max_steps: int = 255, rows: Optional[Tuple[int, int]] = None,
                                raw_only: bool = False, gaussian_tail: bool = False, outs=None, max_out=None,
                                asynchronous: bool = False):
        """stepth_depth_from_groups_batch: `groups[f]` = CUDA uint8 tensor holding at least
        counts[f] stepth_group records (20 bytes each) of frame f.  Returns (outs, max)."""
        import torch
        self._follow(mains)
        n = len(mains)
        if outs is None:
            outs = [torch.zeros(m.shape[:2], dtype=torch.uint8, device=m.device) for m in mains]
        mi, ai, di = (L.Image * n)(), (L.Image * n)(), (L.Image * n)()
        keep = []
        for f in range(n):
            mi[f], k1 = _describe(mains[f], (3, 4), "main")
            ai[f], k2 = _describe(additionals[f], (3,), "additional")
            di[f], k3 = _describe(outs[f], (1,), "depth")
            keep += [k1, k2, k3]
        p = L.Params()
        p.struct_size = C.sizeof(L.Params)
        p.mode = L.MODE_RING_FIRST_HIT
        p.precision = _u8x3(precision)
        p.max_steps = max_steps
        p.tile_w = p.tile_h = 1
        if rows is not None:
            p.row_begin, p.row_end = rows
        p.flags = (L.FLAG_ASYNC if asynchronous else 0) | (L.FLAG_RAW_ONLY if raw_only else 0) | \
            (L.FLAG_GAUSSIAN_TAIL if gaussian_tail else 0)
        gp = (C.c_void_p * n)(*[groups[f].data_ptr() for f in range(n)])
        gn = (C.c_size_t * n)(*[int(c) for c in counts])
        if asynchronous:
            mptr = max_out.data_ptr() if max_out is not None else None
            self._check(self._lib.stepth_depth_from_groups_batch(self._h, n, mi, ai, C.byref(p), gp, gn,
                                                                 di, mptr))
            return outs, None
        mx = (C.c_uint32 * n)()
        self._check(self._lib.stepth_depth_from_groups_batch(self._h, n, mi, ai, C.byref(p), gp, gn, di,
                                                             C.cast(mx, C.c_void_p)))
        return outs, [int(v) for v in mx]

    def adaptive_depth_batch(self, mains, additionals, precision, *, max_steps: int = 255,
                             min_splits: int = 16, max_splits: int = 0, outs=None, max_out=None,
                             gaussian_tail: bool = False, asynchronous: bool = False, two_calls: bool = False):
        """The path the way the reference runs it (src/depth_image.rs:101-135) for a batch of
        CUDA frames: adaptive groups (the engine's stand-in for disage) -> ring first hit per
        group -> normalise -> paint [-> Gaussian tail]; the groups never leave the device.
        One C call (stepth_adaptive_depth_batch); `two_calls` goes through
        stepth_adaptive_groups_batch + stepth_depth_from_groups_batch instead (same result).
        Returns (outs, max); `last_group_counts` holds the frames' group counts."""
        import torch
        if two_calls:
            self._group_buffers, counts = self.adaptive_groups_batch(
                mains, precision, min_splits, max_splits, getattr(self, "_group_buffers", None))
            gb = self._group_buffers
            self.last_group_counts = counts
            return self.depth_from_groups_batch(mains, additionals, precision, [gb[f] for f in range(len(mains))],
                                                counts, max_steps=max_steps, outs=outs, max_out=max_out,
                                                gaussian_tail=gaussian_tail, asynchronous=asynchronous)
        self._follow(mains.tensors if isinstance(mains, PlaneList) else mains)
        n = len(mains)
        if outs is None:
            outs = [torch.zeros(m.shape[:2], dtype=torch.uint8, device=m.device) for m in mains]
        mi = mains.images if isinstance(mains, PlaneList) else PlaneList(mains, (3, 4), "main").images
        ai = additionals.images if isinstance(additionals, PlaneList) else PlaneList(additionals, (3,), "additional").images
        di = outs.images if isinstance(outs, PlaneList) else PlaneList(outs, (1,), "depth").images
        p = L.Params()
        p.struct_size = C.sizeof(L.Params)
        p.mode = L.MODE_RING_FIRST_HIT
        p.precision = _u8x3(precision)
        p.max_steps = max_steps
        p.tile_w = p.tile_h = 1
        p.flags = (L.FLAG_ASYNC if asynchronous else 0) | (L.FLAG_GAUSSIAN_TAIL if gaussian_tail else 0)
        counts = (C.c_size_t * n)()
        if asynchronous:
            mptr = max_out.data_ptr() if max_out is not None else None
            self._check(self._lib.stepth_adaptive_depth_batch(self._h, n, mi, ai, C.byref(p), min_splits, max_splits,
                                                              di, mptr, counts))
            self.last_group_counts = [int(c) for c in counts]
            return outs, None
        mx = (C.c_uint32 * n)()
        self._check(self._lib.stepth_adaptive_depth_batch(self._h, n, mi, ai, C.byref(p), min_splits, max_splits, di,
                                                          C.cast(mx, C.c_void_p), counts))
        self.last_group_counts = [int(c) for c in counts]
        return outs, [int(v) for v in mx]

    def host_array(self, shape) -> np.ndarray:
        """uint8 numpy array in page-locked memory (stepth_host_alloc), freed with the array."""
        n = int(np.prod(shape))
        ptr = C.c_void_p()
        self._check(self._lib.stepth_host_alloc(C.byref(ptr), max(n, 1)))
        return np.asarray(_PinnedBlock(self._lib, ptr.value, n)).reshape(shape)

    def normalise_rows(self, rows, max_value, asynchronous: bool = False, first_row: int = 0,
                       row_pattern: Optional[Tuple[int, int, int]] = None):
        """stepth_normalise_rows[_pattern], in place.  max_value: int (host) or CUDA tensor
        (device).  With `row_pattern` = (row_block, row_mod, row_rem) only the owned rows of
        the view, whose row 0 is image row `first_row`, are touched."""
        self._follow(rows)
        ri, _k = _describe(rows, (1,), "rows")
        flags = L.FLAG_ASYNC if asynchronous else 0
        rb, rm, rr = row_pattern if row_pattern is not None else (0, 1, 0)
        if _is_tensor(max_value):
            st = self._lib.stepth_normalise_rows_pattern(self._h, C.byref(ri), first_row, rb, rm,
                                                         rr, max_value.data_ptr(), L.MEM_DEVICE,
                                                         flags)
        else:
            mx = C.c_uint32(int(max_value))
            st = self._lib.stepth_normalise_rows_pattern(self._h, C.byref(ri), first_row, rb, rm,
                                                         rr, C.cast(C.byref(mx), C.c_void_p),
                                                         L.MEM_HOST, flags)
        self._check(st)
        return rows

    def normalise_rows_batch(self, rows_list, max_tensor, asynchronous: bool = False,
                             first_row: int = 0,
                             row_pattern: Optional[Tuple[int, int, int]] = None):
        """stepth_normalise_rows_batch: PASS 2 on equally shaped CUDA views in one launch;
        `max_tensor` holds one int32 per view."""
        self._follow(rows_list.tensors if isinstance(rows_list, PlaneList) else rows_list)
        n = len(rows_list)
        ri = rows_list.images if isinstance(rows_list, PlaneList) else PlaneList(rows_list, (1,), "rows").images
        rb, rm, rr = row_pattern if row_pattern is not None else (0, 1, 0)
        self._check(self._lib.stepth_normalise_rows_batch(
            self._h, n, ri, first_row, rb, rm, rr, max_tensor.data_ptr(),
            L.FLAG_ASYNC if asynchronous else 0))

    def gaussian_same_size(self, src, out=None, asynchronous: bool = False):
        """stepth_gaussian_same_size: the reference's `.resize(w, h, Gaussian)` tail."""
        self._follow(src)
        si, _k1 = _describe(src, (1,), "src")
        if out is None:
            if _is_tensor(src):
                import torch
                out = torch.empty_like(src)
            else:
                out = np.zeros((si.height, si.width), np.uint8)
        di, _k2 = _describe(out, (1,), "dst")
        self._check(self._lib.stepth_gaussian_same_size(
            self._h, C.byref(si), C.byref(di), L.FLAG_ASYNC if asynchronous else 0))
        return out

    # ---- consumers of a finished depth map (reference src/depth_image.rs:162-247) ---------
    def invert_depth(self, depth, asynchronous: bool = False):
        """stepth_invert_depth: `depth` (numpy [H,W] or CUDA tensor) becomes 255 - depth in place."""
        self._follow(depth)
        di, _k = _describe(depth, (1,), "depth")
        self._check(self._lib.stepth_invert_depth(self._h, C.byref(di),
                                                  L.FLAG_ASYNC if asynchronous else 0))
        return depth

    def depth_split(self, depth, zones: int):
        """stepth_depth_split -> [(lo, hi), ...] in centroid order; [(None, None)] if zones < 2."""
        self._follow(depth)
        di, _k = _describe(depth, (1,), "depth")
        out = (L.Zone * 256)()
        n = C.c_uint32(0)
        self._check(self._lib.stepth_depth_split(self._h, C.byref(di), int(zones), out, 256,
                                                 C.byref(n)))
        return [(z.from_ if z.has_from else None, z.to if z.has_to else None)
                for z in out[:n.value]]

    def depth_histogram(self, depth, out=None, asynchronous: bool = False):
        """stepth_depth_histogram -> 256 counts (int32 CUDA tensor for a CUDA plane, else a
        numpy uint32 array): the per-rank half of depth_split for row-sharded depth maps."""
        self._follow(depth)
        di, _k = _describe(depth, (1,), "depth")
        if _is_tensor(depth):
            import torch
            if out is None:
                out = torch.empty(256, dtype=torch.int32, device=depth.device)
            self._check(self._lib.stepth_depth_histogram(
                self._h, C.byref(di), out.data_ptr(), L.MEM_DEVICE,
                L.FLAG_ASYNC if asynchronous else 0))
            return out
        if out is None:
            out = np.zeros(256, np.uint32)
        self._check(self._lib.stepth_depth_histogram(self._h, C.byref(di), out.ctypes.data,
                                                     L.MEM_HOST, 0))
        return out

    def zones_from_histogram(self, counts, zones: int):
        """stepth_zones_from_histogram: depth_split's zones from (summed) 256 counts."""
        out = (L.Zone * 256)()
        n = C.c_uint32(0)
        if _is_tensor(counts):
            ptr, mem, keep = counts.data_ptr(), L.MEM_DEVICE, counts
            if counts.numel() != 256 or counts.element_size() != 4 or not counts.is_contiguous():
                raise ValueError("counts: 256 contiguous 32-bit words")
        else:
            keep = np.ascontiguousarray(counts, dtype=np.uint32)
            if keep.size != 256:
                raise ValueError("counts: 256 words")
            ptr, mem = keep.ctypes.data, L.MEM_HOST
        self._check(self._lib.stepth_zones_from_histogram(self._h, ptr, mem, int(zones), out, 256,
                                                          C.byref(n)))
        return [(z.from_ if z.has_from else None, z.to if z.has_to else None)
                for z in out[:n.value]]

    def slice(self, depth, from_=None, to=None, out=None, asynchronous: bool = False):
        """stepth_slice -> mask [H,W] u8 (255 inside [from, to], 0 outside)."""
        self._follow(depth)
        di, _k1 = _describe(depth, (1,), "depth")
        if out is None:
            if _is_tensor(depth):
                import torch
                out = torch.empty((di.height, di.width), dtype=torch.uint8, device=depth.device)
            else:
                out = np.zeros((di.height, di.width), np.uint8)
        mi, _k2 = _describe(out, (1,), "mask")
        z = L.Zone(from_ is not None, from_ or 0, to is not None, 255 if to is None else to)
        self._check(self._lib.stepth_slice(self._h, C.byref(di), C.byref(z), C.byref(mi),
                                           L.FLAG_ASYNC if asynchronous else 0))
        return out

    def select_foreground(self, depth, out=None):
        """stepth_select_foreground -> (mask, (from, to))."""
        self._follow(depth)
        di, _k1 = _describe(depth, (1,), "depth")
        if out is None:
            if _is_tensor(depth):
                import torch
                out = torch.empty((di.height, di.width), dtype=torch.uint8, device=depth.device)
            else:
                out = np.zeros((di.height, di.width), np.uint8)
        mi, _k2 = _describe(out, (1,), "mask")
        z = L.Zone()
        self._check(self._lib.stepth_select_foreground(self._h, C.byref(di), C.byref(mi),
                                                       C.byref(z)))
        return out, (z.from_, z.to)

    @staticmethod
    def plane_list(tensors, channels=(1,), name: str = "plane") -> "PlaneList":
        """Describes a list of CUDA tensors once, for repeated batch calls on the same buffers."""
        return PlaneList(tensors, channels, name)

    def select_foreground_batch(self, depths, outs=None, asynchronous: bool = False,
                                raise_on_error: bool = True):
        """stepth_select_foreground_batch on CUDA tensors of one shape (lists of tensors, or
        PlaneList objects from plane_list() when the same buffers are used call after call).

        -> (masks, zones, statuses); zones[f] is (from, to) or None when frame f's split
        failed (its mask is then left as it was).  Asynchronous calls return (masks, None, None)."""
        self._follow(depths.tensors if isinstance(depths, PlaneList) else depths)
        import torch
        if not isinstance(depths, PlaneList):
            depths = PlaneList(depths, (1,), "depth")
        if outs is None:
            outs = [torch.empty_like(d) for d in depths.tensors]
        if not isinstance(outs, PlaneList):
            outs = PlaneList(outs, (1,), "mask")
        n = len(depths.tensors)
        if len(outs.tensors) != n:
            raise ValueError("one mask per depth map")
        dims, mims, outs = depths.images, outs.images, outs.tensors
        if asynchronous:
            self._check(self._lib.stepth_select_foreground_batch(self._h, n, dims, mims, None, None,
                                                                 L.FLAG_ASYNC))
            return outs, None, None
        zs = (L.Zone * n)()
        st = (C.c_int * n)()
        rc = self._lib.stepth_select_foreground_batch(self._h, n, dims, mims, zs, st, 0)
        if rc != L.OK and (raise_on_error or rc > L.ERR_SPLIT_STEP or rc < L.ERR_SPLIT_DIVERGED):
            self._check(rc)
        zones = [(zs[f].from_, zs[f].to) if st[f] == 0 else None for f in range(n)]
        return outs, zones, list(st)

    def apply_mask(self, rgba, mask, asynchronous: bool = False):
        """stepth_apply_mask: RGBA pixels with mask == 0 become [0,0,0,0], in place."""
        self._follow(rgba)
        ri, _k1 = _describe(rgba, (4,), "rgba")
        mi, _k2 = _describe(mask, (1,), "mask")
        self._check(self._lib.stepth_apply_mask(self._h, C.byref(ri), C.byref(mi),
                                                L.FLAG_ASYNC if asynchronous else 0))
        return rgba

    def distance_dot_array_batch(self, array, what, from_xy, max_steps: int, precision):
        """helpers::distance_dot_array for n queries -> (found u8[n], dist u32[n], pos u32[n,2])."""
        ai, _k = _describe(array, (3,), "array")
        what = np.ascontiguousarray(what, dtype=np.uint8).reshape(-1, 3)
        from_xy = np.ascontiguousarray(from_xy, dtype=np.uint32).reshape(-1, 2)
        n = len(what)
        if len(from_xy) != n:
            raise ValueError("what and from_xy lengths differ")
        found = np.zeros(n, np.uint8)
        dist = np.zeros(n, np.uint32)
        pos = np.zeros((n, 2), np.uint32)
        st = self._lib.stepth_distance_dot_array_batch(
            self._h, C.byref(ai), what.ctypes.data, from_xy.ctypes.data, n, max_steps,
            C.cast(_u8x3(precision), C.c_void_p), found.ctypes.data, dist.ctypes.data,
            pos.ctypes.data)
        self._check(st)
        return found, dist, pos

    # -- instrumentation ------------------------------------------------------------------
    def stats_enable(self, on: bool = True):
        self._check(self._lib.stepth_stats_enable(self._h, int(on)))

    def stats(self) -> dict:
        s = L.Stats()
        self._check(self._lib.stepth_get_stats(self._h, C.byref(s)))
        return {k: int(getattr(s, k)) for k, _t in L.Stats._fields_}

    def stats_raw(self, n: int = 32) -> list:
        """the first n raw 64-bit statistic words (stepth_get_stats_raw): the documented seven,
        then [6] superblock steps without a candidate block and [7] block probes without gain of
        the dense search, then slots only tuning builds fill"""
        raw = (C.c_uint64 * n)()
        self._check(self._lib.stepth_get_stats_raw(self._h, raw, n))
        return [int(v) for v in raw]

    def kernel_time_enable(self, on: bool = True):
        self._check(self._lib.stepth_kernel_time_enable(self._h, int(on)))

    def kernel_time_read(self) -> Tuple[float, int]:
        ms, n = C.c_double(0), C.c_uint64(0)
        self._check(self._lib.stepth_kernel_time_read(self._h, C.byref(ms), C.byref(n)))
        return float(ms.value), int(n.value)
