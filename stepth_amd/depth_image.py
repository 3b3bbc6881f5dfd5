"""Host-side mirror of the reference's `DepthImage` for the open_depth_from_additional path.

Mirrors /root/reference/src/depth_image.rs:6-160 (names, argument meaning, error behaviour):
the container, file/array constructors, `load_depth`'s size check and the two entry points of
the hot path.  All compute of the path runs in the HIP engine through the C-ABI; this module
only decodes files (Pillow) and moves numpy arrays.  Rust's
`io::Error::new(InvalidInput, msg)` becomes `InvalidInput(msg)` with the same messages.

The consumers of a finished depth map that the README's second half uses (Readme.md:18-24,
SURVEY.md 8f N3) are mirrored too and also run in the engine: `invert_depth`, `depth_split`,
`slice`, `select_foreground` (src/depth_image.rs:162-247) and `MaskImage.apply_mask`
(src/mask_image.rs:205-213).  highlight_depth, resize and the rest of MaskImage are outside
the scope.
"""
from __future__ import annotations

import errno
from typing import Optional, Sequence, Tuple

import numpy as np

from . import _lib as L
from .engine import Engine, MaxZeroError

MASK_TRUE, MASK_FALSE = 255, 0  # src/mask_image.rs:3-4

_engine: Optional[Engine] = None


def default_engine() -> Engine:
    """Process-wide engine on the current device (created on first use)."""
    global _engine
    if _engine is None:
        _engine = Engine(-1)
    return _engine


class InvalidInput(OSError):
    """std::io::ErrorKind::InvalidInput with the reference's fixed messages."""

    def __init__(self, message: str):
        super().__init__(errno.EINVAL, message)
        self.message = message


def _decode(path: str):
    try:
        from PIL import Image
        with Image.open(path) as im:
            im.load()
            return im.copy()
    except Exception:
        # src/depth_image.rs:13-17, :67-72, :82-87
        raise InvalidInput("Failed to open image") from None


def _to_rgba8(img) -> np.ndarray:
    """DynamicImage::to_rgba8 (src/depth_image.rs:18, :24)."""
    if isinstance(img, np.ndarray):
        a = img
        if a.dtype != np.uint8 or a.ndim not in (2, 3):
            raise TypeError("expected a uint8 [H,W], [H,W,3] or [H,W,4] array")
        if a.ndim == 2:
            a = np.repeat(a[..., None], 3, axis=2)
        if a.shape[2] == 3:
            a = np.concatenate([a, np.full(a.shape[:2] + (1,), 255, np.uint8)], axis=2)
        return np.ascontiguousarray(a)
    return np.ascontiguousarray(np.asarray(img.convert("RGBA"), dtype=np.uint8))


def _to_rgb8(img) -> np.ndarray:
    """DynamicImage::to_rgb8 (src/depth_image.rs:96)."""
    if isinstance(img, np.ndarray):
        a = img
        if a.dtype != np.uint8 or a.ndim not in (2, 3):
            raise TypeError("expected a uint8 [H,W], [H,W,3] or [H,W,4] array")
        if a.ndim == 2:
            a = np.repeat(a[..., None], 3, axis=2)
        return np.ascontiguousarray(a[..., :3])
    return np.ascontiguousarray(np.asarray(img.convert("RGB"), dtype=np.uint8))


def _to_luma8(img) -> np.ndarray:
    if isinstance(img, np.ndarray):
        if img.dtype != np.uint8 or img.ndim != 2:
            raise TypeError("expected a uint8 [H,W] array")
        return np.ascontiguousarray(img)
    return np.ascontiguousarray(np.asarray(img.convert("L"), dtype=np.uint8))


class DepthImage:
    """`pub struct DepthImage { pub image: RgbaImage, pub depth: GrayImage }`
    (src/depth_image.rs:6-10): `image` is [H,W,4] uint8, `depth` is [H,W] uint8."""

    def __init__(self, image: Optional[np.ndarray] = None, depth: Optional[np.ndarray] = None):
        # #[derive(Default)]: empty buffers
        self.image = image if image is not None else np.zeros((0, 0, 4), np.uint8)
        self.depth = depth if depth is not None else np.zeros(self.image.shape[:2], np.uint8)

    # src/depth_image.rs:12-21
    @classmethod
    def open(cls, image_path: str) -> "DepthImage":
        return cls.from_image(_decode(image_path))

    # src/depth_image.rs:23-27
    @classmethod
    def from_image(cls, img) -> "DepthImage":
        image = _to_rgba8(img)
        return cls(image, np.zeros(image.shape[:2], np.uint8))

    def clone(self) -> "DepthImage":
        return DepthImage(self.image.copy(), self.depth.copy())

    # src/depth_image.rs:138-144, :155-160
    def width(self) -> int:
        return int(self.image.shape[1])

    def height(self) -> int:
        return int(self.image.shape[0])

    def dimensions(self) -> Tuple[int, int]:
        """disage::Dimensions{width,height}"""
        return self.width(), self.height()

    # src/depth_image.rs:37-49
    def load_depth(self, depth: np.ndarray) -> None:
        depth = _to_luma8(depth)
        if depth.shape[1] == self.width() and depth.shape[0] == self.height():
            self.depth = depth
            return
        raise InvalidInput("Sizes don't match")

    # src/depth_image.rs:65-74
    def open_depth(self, depth_path: str) -> None:
        self.load_depth(_to_luma8(_decode(depth_path)))

    # src/depth_image.rs:76-89
    def open_depth_from_additional(self, add_path: str, precision: Sequence[int],
                                   **engine_options) -> None:
        self.load_depth_from_additional(_decode(add_path), precision, **engine_options)

    # src/depth_image.rs:91-136
    def load_depth_from_additional(self, add_image, precision: Sequence[int], *,
                                   groups=None,
                                   tile: Tuple[int, int] = (1, 1), max_steps: int = 255,
                                   gaussian_tail: bool = True,
                                   engine: Optional[Engine] = None) -> None:
        """Runs the whole path on the GPU and stores the result in `self.depth`.

        `precision` is the reference's `[u8;3]`.  Keyword options are what the reference
        hard-codes or delegates to disage: `groups` (a GROUP_DTYPE array standing in for
        disage::open::rgb_discrete's output, :104-110, or the string "adaptive" for the engine's
        own adaptive stand-in with the reference's (16, ceil(log2(W*H))) splits -- parity with
        disage unpinned; default: uniform `tile`s, 1x1 = one group per pixel, the configuration
        the asset pin validates) and `max_steps` (the literal 255 at :119).  `gaussian_tail` applies
        the same-size Gaussian resample of :133-134 like the reference does (image 0.23.14's
        sampler restated -- parity unpinned); pass False for the bare depth map.

        Where the reference panics on a zero maximum (:127) this raises MaxZeroError and
        leaves `self.depth` all zero."""
        add = _to_rgb8(add_image)
        eng = engine or default_engine()
        if isinstance(groups, str):
            if groups != "adaptive":
                raise ValueError("groups: a GROUP_DTYPE array or \"adaptive\"")
            groups = eng.adaptive_groups(self.image, precision, 16, 0)  # :101-103
        try:
            depth, _mx = eng.depth_from_additional(self.image, add, precision,
                                                   max_steps=max_steps, tile=tile,
                                                   groups=groups, mode=L.MODE_RING_FIRST_HIT,
                                                   gaussian_tail=gaussian_tail,
                                                   heavy_first=True)  # photographs: see STEPTH_FLAG_HEAVY_FIRST
        except MaxZeroError:
            self.depth = np.zeros(self.image.shape[:2], np.uint8)
            raise
        self.load_depth(depth)

    # ---- consumers of a finished depth map -------------------------------------------------
    # src/depth_image.rs:226-228
    def invert_depth(self, engine: Optional[Engine] = None) -> None:
        if not self.depth.flags.writeable:  # e.g. a view of a decoded file
            self.depth = self.depth.copy()
        (engine or default_engine()).invert_depth(self.depth)

    # src/depth_image.rs:162-219
    def depth_split(self, zones: int, engine: Optional[Engine] = None):
        """-> [(min, max), ...] per zone in centroid order; [(None, None)] when zones < 2.
        Inputs on which the reference panics raise SplitError."""
        if not 0 <= zones <= 255:
            raise ValueError("zones is a u8")
        return (engine or default_engine()).depth_split(self.depth, zones)

    # src/depth_image.rs:230-247
    def slice(self, from_: Optional[int], to: Optional[int],
              engine: Optional[Engine] = None) -> "MaskImage":
        mask = (engine or default_engine()).slice(self.depth, from_, to)
        return MaskImage(self.image.copy(), mask)

    # src/depth_image.rs:221-224
    def select_foreground(self, engine: Optional[Engine] = None) -> "MaskImage":
        mask, _zone = (engine or default_engine()).select_foreground(self.depth)
        return MaskImage(self.image.copy(), mask)


class MaskImage:
    """`pub struct MaskImage { pub image: RgbaImage, pub mask: GrayImage }`
    (src/mask_image.rs:6-10) -- only what DepthImage::slice returns and the README flow uses."""

    def __init__(self, image: np.ndarray, mask: Optional[np.ndarray] = None):
        self.image = image
        # src/mask_image.rs:17-21: a fresh mask is MASK_TRUE everywhere
        self.mask = mask if mask is not None else np.full(image.shape[:2], MASK_TRUE, np.uint8)

    def width(self) -> int:
        return int(self.image.shape[1])

    def height(self) -> int:
        return int(self.image.shape[0])

    # src/mask_image.rs:205-213
    def apply_mask(self, engine: Optional[Engine] = None) -> None:
        if not self.image.flags.writeable:
            self.image = self.image.copy()
        (engine or default_engine()).apply_mask(self.image, self.mask)

    # src/mask_image.rs:196-198 (PNG/JPEG by extension, through Pillow)
    def save(self, path: str) -> None:
        from PIL import Image
        img = Image.fromarray(self.image, "RGBA")
        if path.lower().endswith((".jpg", ".jpeg")):
            img = img.convert("RGB")
        img.save(path)
