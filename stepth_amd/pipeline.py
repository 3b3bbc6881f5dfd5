"""File-level pipeline for many image pairs: decode on CPU threads while the
GPU works on the previous chunk.

The reference's `open_depth_from_additional(path, precision)` decodes a file and then computes
(src/depth_image.rs:76-89); for a folder of pairs the decode (Pillow / libjpeg-turbo, ~5-10 ms
per 1080p JPEG on one core) dwarfs the GPU time (0.5 ms per pair), so the only way to keep the
GPU busy is to decode many files at once and to overlap decode, upload, kernels and download:

    decode chunk k+1 (thread pool, into page-locked arrays)
        ||  stepth_depth_from_additional_batch on chunk k (host memory: its uploads, kernels and
            downloads already overlap inside the engine)

Decoding stays on the CPU (no hardware JPEG decoder library in the image); everything after
the decoded bytes is the HIP engine.
"""
from __future__ import annotations

from concurrent.futures import ThreadPoolExecutor
from typing import Iterable, Iterator, List, Optional, Sequence, Tuple

import numpy as np

from .depth_image import InvalidInput, _decode, _to_rgb8, default_engine
from .engine import Engine


def _decode_pair(paths: Tuple[str, str], main_buf: Optional[np.ndarray],
                 add_buf: Optional[np.ndarray]):
    """-> (main RGB8, additional RGB8), written into the given page-locked buffers if they fit"""
    main, add = _to_rgb8(_decode(paths[0])), _to_rgb8(_decode(paths[1]))
    if main_buf is not None and main_buf.shape == main.shape:
        main_buf[...] = main
        main = main_buf
    if add_buf is not None and add_buf.shape == add.shape:
        add_buf[...] = add
        add = add_buf
    return main, add


def depth_from_files(pairs: Iterable[Tuple[str, str]], precision: Sequence[int], *,
                     engine: Optional[Engine] = None, max_steps: int = 255,
                     gaussian_tail: bool = True, chunk: int = 8, workers: int = 8
                     ) -> Iterator[Tuple[np.ndarray, int]]:
    """Yields (depth [H,W] u8, max) for every (main path, additional path) pair, in order.

    Within a chunk (one batched engine call) all main images must have one size and all
    additional images one size (the additional size may differ from the main size); files that
    cannot be decoded raise InvalidInput("Failed to open image") like the reference
    (src/depth_image.rs:82-87); a pair whose distances are all zero raises MaxZeroError from
    the chunk it is in."""
    eng = engine or default_engine()
    pairs = list(pairs)
    if chunk < 1 or workers < 1:
        raise ValueError("chunk and workers must be positive")
    slots: List[dict] = [{}, {}]  # page-locked buffers of the two chunks in flight

    def start(k: int, pool: ThreadPoolExecutor):
        part = pairs[k * chunk:(k + 1) * chunk]
        bufs = slots[k & 1]
        futs = []
        for i, pth in enumerate(part):
            futs.append(pool.submit(_decode_pair, pth, bufs.get(("m", i)), bufs.get(("a", i))))
        return futs

    n_chunks = (len(pairs) + chunk - 1) // chunk
    with ThreadPoolExecutor(max_workers=workers) as pool:
        nxt = start(0, pool) if n_chunks else []
        for k in range(n_chunks):
            decoded = [f.result() for f in nxt]
            bufs = slots[k & 1]
            for i, (m, a) in enumerate(decoded):  # first use of a size: move into pinned memory
                if ("m", i) not in bufs or bufs[("m", i)].shape != m.shape:
                    bufs[("m", i)] = eng.host_array(m.shape)
                    bufs[("m", i)][...] = m
                    bufs[("a", i)] = eng.host_array(a.shape)
                    bufs[("a", i)][...] = a
                    decoded[i] = (bufs[("m", i)], bufs[("a", i)])
            if k + 1 < n_chunks:
                nxt = start(k + 1, pool)  # decodes while the GPU works on this chunk
            mains = [m for m, _ in decoded]
            adds = [a for _, a in decoded]
            # one batched engine call per chunk: all main images of one size, all additional
            # images of one size -- the two sizes may differ from each other, like in the
            # reference (src/depth_image.rs:91-100 never compares them)
            if len({m.shape for m in mains}) != 1 or len({a.shape for a in adds}) != 1:
                raise InvalidInput("images of one chunk differ in size")
            outs, mx = eng.depth_from_additional_batch(mains, adds, precision, max_steps=max_steps,
                                                       gaussian_tail=gaussian_tail)
            for d, v in zip(outs, mx):
                yield d, v
