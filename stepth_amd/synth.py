"""Seeded synthetic stereo pairs: the reference ships no generator, and there
is no network for datasets, so benches and parity tests use these.

Everything is a pure function of (width, height, d_max, seed) built from an integer hash, so
the same bytes come out on every machine and numpy version (no numpy RNG streams involved).

  left  : posterised value-noise luminance + weak per-channel chroma (piecewise smooth,
          natural-image-like -- NOT white noise, which would end every ring search within
          ~3 rings) + +-3 uniform sensor noise
  disp  : a slanted background plane plus 4 fronto-parallel rectangles, all in [0, d_max)
  right : left warped by disp (x_r = x_l - d), far-to-near so nearer surfaces occlude,
          holes filled from the background plane, independent sensor noise
"""
from __future__ import annotations

from typing import Tuple

import numpy as np

_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)
_G = np.uint64(0x9E3779B97F4A7C15)


def _mix(z: np.ndarray) -> np.ndarray:
    """splitmix64 finaliser, vectorised (uint64 wrap-around is intended)."""
    z = (z ^ (z >> np.uint64(30))) * _M1
    z = (z ^ (z >> np.uint64(27))) * _M2
    return z ^ (z >> np.uint64(31))


def _hash01(ix: np.ndarray, iy: np.ndarray, key: int) -> np.ndarray:
    """uniform [0,1) float64 from integer lattice coordinates and a key"""
    with np.errstate(over="ignore"):
        z = (ix.astype(np.uint64) * np.uint64(0x1F123BB5) +
             iy.astype(np.uint64) * np.uint64(0x5F356495) +
             np.uint64(key & 0xFFFFFFFFFFFFFFFF) * _G)
        z = _mix(z + _G)
    return (z >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53))


def _value_noise(w: int, h: int, cell: float, key: int) -> np.ndarray:
    # lattice coordinates and blend weights depend on the column resp. the row only, and the hash
    # on the lattice point only: work on 1-D axes and a small lattice table, then broadcast (the
    # same float64 operations per pixel as the per-pixel formulation, a fraction of the work)
    fx, fy = np.arange(w) / cell, np.arange(h) / cell
    x0, y0 = np.floor(fx).astype(np.int64), np.floor(fy).astype(np.int64)
    tx, ty = fx - x0, fy - y0
    tx = (tx * tx * (3 - 2 * tx))[None, :]
    ty = (ty * ty * (3 - 2 * ty))[:, None]
    ly, lx = np.mgrid[0:int(y0.max()) + 2, 0:int(x0.max()) + 2]
    lat = _hash01(lx, ly, key)
    v00, v10 = lat[np.ix_(y0, x0)], lat[np.ix_(y0, x0 + 1)]
    v01, v11 = lat[np.ix_(y0 + 1, x0)], lat[np.ix_(y0 + 1, x0 + 1)]
    return (v00 * (1 - tx) + v10 * tx) * (1 - ty) + (v01 * (1 - tx) + v11 * tx) * ty


def _sensor_noise(w: int, h: int, key: int) -> np.ndarray:
    ys, xs = np.mgrid[0:h, 0:w]
    out = np.empty((h, w, 3), np.int16)
    for c in range(3):
        out[..., c] = np.floor(_hash01(xs, ys, key * 8 + c) * 7).astype(np.int16) - 3
    return out


def _fbm(w: int, h: int, base: float, key: int, octaves: int = 4) -> np.ndarray:
    acc, amp, tot = np.zeros((h, w)), 1.0, 0.0
    for o in range(octaves):
        acc += amp * _value_noise(w, h, max(base / (2 ** o), 2.0), key * 8 + o)
        tot += amp
        amp *= 0.5
    return acc / tot


def texture(w: int, h: int, seed: int) -> np.ndarray:
    """piecewise-smooth RGB texture, float64 in [0,255]: a shared luminance field, 80 %
    posterised to 5 levels (flat regions with sharp edges, like objects in a scene), plus a
    weak low-frequency chroma field per channel (channels of natural images are correlated).
    Tuned so that the reference algorithm does about as much work per pixel on a 600x400
    pair (about 2.0e3 probes, precision 36) as on the reference's own asset pair (2.1e3)."""
    base = max(w, h) / 6.0
    lum = _fbm(w, h, base, seed * 8)
    lo, hi = lum.min(), lum.max()
    lum = (lum - lo) / max(hi - lo, 1e-9)
    post = np.clip(np.floor(lum * 5.0) / 4.0, 0.0, 1.0)
    lum = 0.8 * post + 0.2 * lum
    img = np.empty((h, w, 3))
    for c in range(3):
        img[..., c] = 0.8 * lum + 0.2 * _fbm(w, h, base * 1.5, seed * 8 + 1 + c, 3)
    lo, hi = img.min(), img.max()
    return (img - lo) / max(hi - lo, 1e-9) * 255.0


def disparity(w: int, h: int, d_max: int, seed: int) -> np.ndarray:
    """integer ground-truth disparity in [0, d_max)"""
    ys, xs = np.mgrid[0:h, 0:w]
    top = max(d_max - 1, 0)
    d = np.floor(0.05 * top + 0.25 * top * ys / max(h - 1, 1)).astype(np.int32)  # slanted floor
    k = np.arange(4)
    r = _hash01(k, k * 7 + 3, seed * 16 + 5)
    r2 = _hash01(k * 3 + 1, k, seed * 16 + 6)
    for i in range(4):  # far to near
        cx, cy = int(r[i] * w), int(r2[i] * h)
        rw, rh = int(w * (0.15 + 0.1 * r2[i])), int(h * (0.15 + 0.1 * r[i]))
        dv = int(top * (0.35 + 0.6 * (i + 1) / 4.0))
        m = (np.abs(xs - cx) <= rw) & (np.abs(ys - cy) <= rh)
        d[m] = np.maximum(d[m], dv)
    return np.clip(d, 0, top)


def stereo_pair(w: int, h: int, d_max: int = 128, seed: int = 0x5EED
                ) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """-> (left RGB8 [h,w,3], right RGB8 [h,w,3], disparity int32 [h,w])"""
    tex = texture(w, h, seed)
    disp = disparity(w, h, d_max, seed)
    ys, xs = np.mgrid[0:h, 0:w]
    bg = disp.min(axis=1, keepdims=True)  # per-row background disparity for hole filling
    src = np.clip(xs + bg, 0, w - 1)
    right = tex[ys, src].copy()
    for dv in np.unique(disp):  # ascending = far to near, nearer overwrites
        m = disp == dv
        xr = xs[m] - dv
        ok = xr >= 0
        right[ys[m][ok], xr[ok]] = tex[ys[m][ok], xs[m][ok]]
    left8 = np.clip(np.rint(tex) + _sensor_noise(w, h, seed * 4 + 1), 0, 255).astype(np.uint8)
    right8 = np.clip(np.rint(right) + _sensor_noise(w, h, seed * 4 + 2), 0, 255).astype(np.uint8)
    return left8, right8, disp
