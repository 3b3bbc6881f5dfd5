"""stepth_amd -- MI355X (gfx950) engine for stepth's DepthImage::open_depth_from_additional.

Only what the hot path needs lives here: `csrc/` (HIP kernels + the C-ABI of
include/stepth_amd.h), the ctypes binding, the `DepthImage`/`MaskImage` host mirror, row-strip sharding
and the synthetic-pair generator.  Importing the package does not load the HIP library;
the first Engine() does, and fails loudly if it is missing (there is no CPU fallback).
"""
from ._lib import (FLAG_ASYNC, FLAG_GAUSSIAN_TAIL, FLAG_RAW_ONLY, MEM_DEVICE, MEM_HOST, MODE_RING_FIRST_HIT,
                   MODE_SAD_WINDOW)
from .depth_image import DepthImage, InvalidInput, MaskImage, default_engine
from .engine import GROUP_DTYPE, Engine, MaxZeroError, SplitError, StepthError

__all__ = ["DepthImage", "MaskImage", "SplitError", "InvalidInput", "Engine", "StepthError", "MaxZeroError", "GROUP_DTYPE",
           "default_engine", "MODE_RING_FIRST_HIT", "MODE_SAD_WINDOW", "MEM_HOST", "MEM_DEVICE",
           "FLAG_ASYNC", "FLAG_RAW_ONLY", "FLAG_GAUSSIAN_TAIL"]
