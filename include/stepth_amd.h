/*
 * stepth_amd.h -- C-ABI of the MI355X (gfx950) engine for stepth's
 * DepthImage::open_depth_from_additional hot path.
 *
 * The reference (nikilark/stepth, Rust) has no FFI, plugin or operator interface: the
 * boundary is its own method DepthImage::load_depth_from_additional
 * (/root/reference/src/depth_image.rs:91-136).  The entry points below are what a Rust
 * `extern "C"` block for that method would bind (stub: INTEGRATION.md); each comment cites
 * the reference lines the call replaces.  Plain pointers and sizes only; no C++/torch types.
 *
 * Conventions
 *  - Every call returns a stepth_status (0 = ok, negative = error) and never throws/aborts.
 *    stepth_last_error(engine) gives the message of the last failing call on that engine.
 *  - All image memory is caller-owned and only borrowed: the engine never frees it and does
 *    not keep pointers past the return of a synchronous call, or past stepth_engine_sync()
 *    for calls made with STEPTH_FLAG_ASYNC.
 *  - Images are row-major, 8-bit, `stride` bytes per row (>= width*channels), `memory` says
 *    whether `data` is a host pointer or a device (HIP) pointer on the engine's device.
 *  - An engine is bound to one device and one stream and is NOT thread-safe; different
 *    engines may be used from different threads concurrently (the reference's &mut self
 *    exclusivity, src/depth_image.rs:91-92).
 */
#ifndef STEPTH_AMD_H
#define STEPTH_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define STEPTH_ABI_VERSION 4u

typedef struct stepth_engine stepth_engine; /* opaque */

typedef enum {
    STEPTH_OK = 0,
    STEPTH_ERR_BAD_ARG = -1,
    /* "Sizes don't match", src/depth_image.rs:37-49 (load_depth's check) */
    STEPTH_ERR_SIZE_MISMATCH = -2,
    STEPTH_ERR_HIP = -3,
    /* every group's distance was 0: the reference divides by max and panics
     * (src/depth_image.rs:127); here the output is all zero and this status is returned */
    STEPTH_ERR_MAX_ZERO = -4,
    STEPTH_ERR_NO_DEVICE = -5,
    STEPTH_ERR_OUT_OF_MEMORY = -6,
    /* depth_split inputs on which the reference panics (src/depth_image.rs:162-219):
     * (max - min) / (zones - 1) < 2 makes `step_by` assert or its argument underflow (:217) */
    STEPTH_ERR_SPLIT_STEP = -7,
    /* a zone with no pixels at convergence: `.min().unwrap()` of nothing (:203-206) */
    STEPTH_ERR_SPLIT_EMPTY_ZONE = -8,
    /* the centroid iteration did not settle within 4096 passes (the reference recurses
     * without a bound, :210) */
    STEPTH_ERR_SPLIT_DIVERGED = -9
} stepth_status;

typedef enum {
    /* reference-exact: square-ring first-hit search, src/helpers.rs:9-54 */
    STEPTH_MODE_RING_FIRST_HIT = 0,
    /* horizontal-disparity windowed SAD (BASELINE north_star's wording; not in the
     * reference, checked against the engine's own CPU statement only) */
    STEPTH_MODE_SAD_WINDOW = 1
} stepth_mode;

typedef enum { STEPTH_MEM_HOST = 0, STEPTH_MEM_DEVICE = 1 } stepth_memory;

enum {
    /* enqueue on the engine's stream and return without synchronising; every pointer must
     * then be device memory and STEPTH_ERR_MAX_ZERO is not reported (output is all zero) */
    STEPTH_FLAG_ASYNC = 1u << 0,
    /* stop after PASS 1 (src/depth_image.rs:112-123): write the painted, un-normalised
     * (d as u8) map; used by row-strip sharding, where the max is reduced across ranks
     * before stepth_normalise_rows() */
    STEPTH_FLAG_RAW_ONLY = 1u << 1,
    /* also apply the tail of the reference, src/depth_image.rs:133-134: a same-size resample
     * with image's Gaussian filter (full-frame calls only; parity unpinned) */
    STEPTH_FLAG_GAUSSIAN_TAIL = 1u << 2,
    /* ring mode, one group per pixel, for natural images (most pixels match nearby, a few
     * percent search far): the search's workgroups start heaviest first (a cheap estimate per
     * 16x16-pixel tile, three small extra launches; frame sets in which most tiles search far
     * keep image order and the plain search, decided on the device) and probe candidate blocks as
     * pooled tasks, 64 at a time by whichever lanes.  Results are the same bit for bit.  The
     * reference's own pair at 1080p: 8 frames 1.24 -> 0.91 ms per call, one frame 0.52 -> 0.30 ms;
     * synthetic frames with large occlusions: + 5 % (the three launches), which is why it is OFF
     * by default in the C-ABI; the DepthImage mirrors of the reference API (photographs) set it. */
    STEPTH_FLAG_HEAVY_FIRST = 1u << 3
};

/* disage::DiscretePixel{position{x,y}, size{width,height}, value:[u8;3]} as consumed at
 * src/depth_image.rs:110-121.  20 bytes, little-endian, no padding surprises. */
typedef struct {
    uint32_t x, y, w, h;
    uint8_t value[3];
    uint8_t pad;
} stepth_group;

typedef struct {
    const void *data;
    uint32_t width, height;
    uint32_t channels; /* main: 3 (RGB) or 4 (RGBA as stored in DepthImage.image,
                          src/depth_image.rs:8; alpha ignored like to_rgb8 at :105);
                          additional: 3; depth output: 1 */
    uint32_t memory;   /* stepth_memory */
    size_t stride;     /* bytes per row */
} stepth_image;

typedef struct {
    uint32_t struct_size;  /* sizeof(stepth_params), for forward compatibility */
    uint32_t mode;         /* stepth_mode */
    uint8_t precision[3];  /* src/depth_image.rs:79 */
    uint8_t reserved0;
    uint32_t max_steps;    /* ring mode: `max` of helpers.rs:13; the reference passes 255
                              (src/depth_image.rs:119) */
    uint32_t tile_w, tile_h; /* ring mode, groups == NULL: uniform tiles, value = floor(mean);
                                1x1 = one group per pixel */
    uint32_t d_max, win;   /* SAD mode: disparities [0,d_max), odd window side */
    uint32_t row_begin, row_end; /* main-image rows to compute; 0,0 = all.  A group belongs
                                    to the strip when its top row y is inside it */
    uint32_t flags;        /* STEPTH_FLAG_* */
    /* interleaved row blocks (ring mode, uniform tiles): inside [row_begin,row_end) the call
     * owns only rows y with (y / row_block) % row_mod == row_rem; row_block == 0 = all rows.
     * row_block must be even and a multiple of tile_h.  The work per row is very uneven
     * (data dependent), so N ranks balance best with row_mod = N, row_rem = rank. */
    uint32_t row_block, row_mod, row_rem;
} stepth_params;

/* counters of the last ring-mode call, for "mean probes/pixel" reporting */
typedef struct {
    uint64_t groups;       /* groups (queries) processed */
    uint64_t ring0_hits;   /* groups that matched at their centre cell */
    uint64_t probes;       /* predicate evaluations issued by the kernels (lane-granular) */
    /* wave-steps of the search kernel by kind (each = 64 lane-probes) */
    uint64_t near_steps;       /* block tests of block-rings 0..3 (or spiral steps)          */
    uint64_t survey_steps;     /* superblock surveys                                          */
    uint64_t superblock_steps; /* block tests inside a surviving superblock                   */
    uint64_t blocks_probed;    /* 8x8 blocks whose 64 cells were read                         */
} stepth_stats;

uint32_t stepth_abi_version(void);
const char *stepth_status_string(int status);

/* One engine per device/stream.  device_id < 0 = current device. */
int stepth_engine_create(int device_id, stepth_engine **out);
void stepth_engine_destroy(stepth_engine *engine);
/* Run on the caller's HIP stream (a hipStream_t; NULL is HIP's default stream, which is what
 * torch.cuda.current_stream() usually is) instead of the engine's own non-blocking stream;
 * stepth_engine_reset_stream() goes back to the engine's own.  The new stream is ordered after
 * whatever the old one still holds (an event it waits for); the host is not blocked. */
int stepth_engine_set_stream(stepth_engine *engine, void *hip_stream);
int stepth_engine_reset_stream(stepth_engine *engine);
int stepth_engine_sync(stepth_engine *engine);
const char *stepth_last_error(const stepth_engine *engine);

/*
 * The whole path of DepthImage::load_depth_from_additional up to (not including) the
 * Gaussian resample: src/depth_image.rs:96-132.
 *   main        self.image (RGBA8 or RGB8)                       :104-109
 *   additional  add_image.to_rgb8()                              :96-100
 *   groups      what disage::open::rgb_discrete would return (:104-110); NULL = uniform
 *               tiles from params.  `groups_memory` is a stepth_memory.
 *   depth       [H][W] u8, same width/height as main, else STEPTH_ERR_SIZE_MISMATCH (:37-49)
 *   max_out     optional; the global max of (d as u8) (:124).  Host pointer for synchronous
 *               calls, device pointer (uint32_t) with STEPTH_FLAG_ASYNC.
 */
int stepth_depth_from_additional(stepth_engine *engine, const stepth_image *main,
                                 const stepth_image *additional, const stepth_params *params,
                                 const stepth_group *groups, size_t n_groups,
                                 uint32_t groups_memory, const stepth_image *depth,
                                 uint32_t *max_out);

/*
 * Adaptive pixel groups for `groups` above: the engine's stand-in for
 * disage::open::rgb_discrete(img, MeanBrightnessHasher, BrightnessChecker{precision},
 * (min_splits, max_splits)), src/depth_image.rs:101-110.  disage's source is absent from the
 * reference tree, so the rule is this engine's own and its parity with disage is unpinned:
 * a rectangle at depth d is a group if it is a single pixel, or d >= max_splits, or
 * d >= min_splits and max - min < precision[c] on every channel; otherwise it is cut across
 * its longer side into floor and ceiling halves.  value = per-channel floor(mean).
 *   min_splits / max_splits   the reference passes 16 and ceil(log2(W*H)) (:101-103);
 *                             max_splits = 0 means that formula
 *   groups, capacity          receive the groups (any order), host or device memory
 *                             (`groups_memory`); W*H entries always suffice.  groups = NULL
 *                             only counts.
 *   n_out                     number of groups
 * Pass the result (device memory avoids the round trip) as `groups` to
 * stepth_depth_from_additional.
 */
int stepth_adaptive_groups(stepth_engine *engine, const stepth_image *main,
                           const uint8_t precision[3], uint32_t min_splits, uint32_t max_splits,
                           stepth_group *groups, size_t capacity, uint32_t groups_memory,
                           size_t *n_out);

/*
 * The same path for a batch of frames in ONE set of kernel launches (the per-frame maxima stay
 * separate: each frame is normalised by its own max, src/depth_image.rs:124-129).  A single
 * small frame or row strip cannot fill the GPU -- the cost of a query varies by two orders of
 * magnitude and one heavy query is a ~100 us critical path -- so callers with several frames
 * in flight (video, row-strip sharding of many frames) should batch them.
 *   - ring mode with uniform tiles (1x1 = one group per pixel, the dense search; larger tiles
 *     take their floor mean colour as value, device memory only), or the windowed-SAD mode
 *     (contiguous row strips only); every frame the same sizes; params (rows, row pattern,
 *     flags) apply to every frame;
 *   - every image in device memory (same strides for every frame), or every image in host
 *     memory (what a caller holding decoded images in RAM has, src/depth_image.rs:8-9, :96).
 *     Host batches are synchronous and pipelined: chunks of frames are uploaded on a copy
 *     stream while the previous chunk's kernels run and the one before downloads, so the PCIe
 *     transfers hide behind the kernels; page-locked memory (stepth_host_alloc) makes the
 *     copies fully asynchronous;
 *   - STEPTH_FLAG_GAUSSIAN_TAIL (ring mode, whole frames, not RAW_ONLY) applies the same-size
 *     Gaussian resample of src/depth_image.rs:133-134 to every frame, two launches per chunk;
 *   - n_frames is not limited (the engine works through it in chunks of 64);
 *   - max_out: n_frames words, device memory with STEPTH_FLAG_ASYNC, host memory otherwise
 *     (synchronous calls return STEPTH_ERR_MAX_ZERO if ANY frame's max is 0).
 */
int stepth_depth_from_additional_batch(stepth_engine *engine, uint32_t n_frames,
                                       const stepth_image *mains, const stepth_image *additionals,
                                       const stepth_params *params, const stepth_image *depths,
                                       uint32_t *max_out);

/*
 * The reference's real flow for many frames (src/depth_image.rs:104-132): grouping -> match ->
 * normalise -> paint, each stage ONE set of launches for all frames.
 *
 * stepth_adaptive_groups_batch: stepth_adaptive_groups for n_frames device images of one shape;
 *   groups_out[f] = device memory for `capacity` >= width*height groups of frame f, n_out[f]
 *   (host) = the number written.  One host synchronisation at the end (the counts).
 * stepth_depth_from_groups_batch: stepth_depth_from_additional with an explicit group list per
 *   frame: groups[f] = n_groups[f] stepth_group in DEVICE memory (e.g. straight from
 *   stepth_adaptive_groups_batch: the groups never leave the device), images in device memory
 *   of one shape, ring mode, contiguous row strips (params.row_begin/row_end), flags ASYNC /
 *   RAW_ONLY / GAUSSIAN_TAIL (full frames); max_out as in stepth_depth_from_additional_batch.
 */
/*
 * stepth_adaptive_depth_batch: the two calls above in one -- what DepthImage::
 * load_depth_from_additional does for a user of the crate (src/depth_image.rs:101-135), for
 * n_frames device frames: adaptive groups (kept in the engine's own buffers) -> ring first hit per
 * group -> normalise -> paint [-> Gaussian tail].  Whole frames; flags ASYNC / RAW_ONLY /
 * GAUSSIAN_TAIL.  The group counts still come back to the host once (they size the launches), but
 * only after the pack kernels have been enqueued: the device works on while the host waits.
 * n_groups_out: optional, n_frames counts.  max_out as in stepth_depth_from_additional_batch.
 */
int stepth_adaptive_depth_batch(stepth_engine *engine, uint32_t n_frames, const stepth_image *mains,
                                const stepth_image *additionals, const stepth_params *params,
                                uint32_t min_splits, uint32_t max_splits, const stepth_image *depths,
                                uint32_t *max_out, size_t *n_groups_out);
int stepth_adaptive_groups_batch(stepth_engine *engine, uint32_t n_frames, const stepth_image *mains,
                                 const uint8_t precision[3], uint32_t min_splits,
                                 uint32_t max_splits, stepth_group *const *groups_out,
                                 size_t capacity, size_t *n_out);
int stepth_depth_from_groups_batch(stepth_engine *engine, uint32_t n_frames,
                                   const stepth_image *mains, const stepth_image *additionals,
                                   const stepth_params *params, const stepth_group *const *groups,
                                   const size_t *n_groups, const stepth_image *depths,
                                   uint32_t *max_out);

/*
 * Page-locked host memory for image buffers handed to the engine as STEPTH_MEM_HOST (decode
 * straight into it): uploads and downloads of such memory run at full PCIe rate and overlap
 * the kernels.  Free with stepth_host_free.  Ordinary (pageable) memory works everywhere too.
 */
int stepth_host_alloc(void **out, size_t bytes);
void stepth_host_free(void *ptr);

/*
 * PASS 2 on a strip, src/depth_image.rs:125-129: data[r][c] = data[r][c]*255/max (u64
 * floor), in place.  `max` is read from `max_ptr` (`max_memory` = host or device) -- after
 * a cross-rank max reduction in the sharded flow.  max == 0 -> rows are zeroed and
 * STEPTH_ERR_MAX_ZERO is returned (synchronous calls only).
 */
int stepth_normalise_rows(stepth_engine *engine, const stepth_image *rows,
                          const uint32_t *max_ptr, uint32_t max_memory, uint32_t flags);
/* The same restricted to the rows of an interleaved pattern (see stepth_params.row_block):
 * `rows` starts at absolute image row `first_row`. */
int stepth_normalise_rows_pattern(stepth_engine *engine, const stepth_image *rows,
                                  uint32_t first_row, uint32_t row_block, uint32_t row_mod,
                                  uint32_t row_rem, const uint32_t *max_ptr,
                                  uint32_t max_memory, uint32_t flags);

/* PASS 2 for a batch of equally shaped device images in one launch; `max_ptr` holds one word
 * per frame (device memory). */
int stepth_normalise_rows_batch(stepth_engine *engine, uint32_t n_frames,
                                const stepth_image *rows, uint32_t first_row, uint32_t row_block,
                                uint32_t row_mod, uint32_t row_rem, const uint32_t *max_ptr,
                                uint32_t flags);

/*
 * The tail alone, src/depth_image.rs:133-134: DynamicImage::resize(w, h, Gaussian) of a Luma8
 * image to its own size (a 7-tap separable blur).  src and dst have the same size and may be
 * the same buffer.  Used after the gather in the sharded flow.
 */
int stepth_gaussian_same_size(stepth_engine *engine, const stepth_image *src,
                              const stepth_image *dst, uint32_t flags);

/*
 * helpers::distance_dot_array for a batch of independent queries, src/helpers.rs:9-54:
 * query i looks for the first cell of `array` (additional image, RGB8) whose per-channel
 * |cell - what[i]| < precision, scanning square rings around from[i] for max steps.
 *   what   n x 3 bytes, from_xy n x 2 uint32 (x, y)           (host pointers)
 *   found  n bytes (1 = Some), dist n uint32 (src/helpers.rs:3-7), pos_xy n x 2 uint32
 * Exists so that the matcher can be tested in isolation against the oracle.
 */
int stepth_distance_dot_array_batch(stepth_engine *engine, const stepth_image *array,
                                    const uint8_t *what, const uint32_t *from_xy, size_t n,
                                    uint32_t max, const uint8_t precision[3], uint8_t *found,
                                    uint32_t *dist, uint32_t *pos_xy);

/* ---- consumers of a finished depth map (README's second half, Readme.md:18-24) --------- */

/* one `(Option<u8>, Option<u8>)` of depth_split's result / slice's arguments */
typedef struct stepth_zone {
    uint8_t has_from, from; /* has_from = 0 is None -> 0   (src/depth_image.rs:231) */
    uint8_t has_to, to;     /* has_to   = 0 is None -> 255 (src/depth_image.rs:232) */
} stepth_zone;

/*
 * DepthImage::invert_depth, src/depth_image.rs:226-228: every byte of `depth` becomes
 * 255 - byte, in place ([H][W] u8, host or device).
 */
int stepth_invert_depth(stepth_engine *engine, const stepth_image *depth, uint32_t flags);

/*
 * DepthImage::depth_split(zones), src/depth_image.rs:162-219: 1-D k-means over the depth
 * bytes; writes the zones as (Some(min), Some(max)) in centroid order to out[0..*n_out).
 * zones < 2 gives the single zone (None, None) (:163-165).  `capacity` entries are available
 * at `out` (255 always suffice); more zones than that is STEPTH_ERR_BAD_ARG.  Inputs on which
 * the reference panics return STEPTH_ERR_SPLIT_*.  Synchronous.
 */
int stepth_depth_split(stepth_engine *engine, const stepth_image *depth, uint32_t zones,
                       stepth_zone *out, uint32_t capacity, uint32_t *n_out);

/*
 * DepthImage::slice(from, to), src/depth_image.rs:230-247, the mask plane of the returned
 * MaskImage: mask = 255 (MASK_TRUE, src/mask_image.rs:3) unless depth < from or depth > to,
 * then 0 (MASK_FALSE, :4).  `mask` [H][W] u8 of the depth map's size, else
 * STEPTH_ERR_SIZE_MISMATCH.
 */
int stepth_slice(stepth_engine *engine, const stepth_image *depth, const stepth_zone *range,
                 const stepth_image *mask, uint32_t flags);

/*
 * DepthImage::select_foreground, src/depth_image.rs:221-224: slice(depth_split(2)[0]).
 * zone_out (optional) receives that zone.  The batch form runs n frames of one shape in
 * three launches (histogram, split, slice) and needs device memory; statuses[f] (optional)
 * is STEPTH_OK or the STEPTH_ERR_SPLIT_* of frame f, whose mask is then left unwritten, and
 * the call returns the first non-OK frame status.  With STEPTH_FLAG_ASYNC the batch form only
 * enqueues: zones_out/statuses must be NULL and failed frames simply keep their mask.
 */
int stepth_select_foreground(stepth_engine *engine, const stepth_image *depth,
                             const stepth_image *mask, stepth_zone *zone_out);
int stepth_select_foreground_batch(stepth_engine *engine, uint32_t n_frames,
                                   const stepth_image *depths, const stepth_image *masks,
                                   stepth_zone *zones_out, int *statuses, uint32_t flags);

/*
 * depth_split in two halves, for a depth map whose rows are sharded over several GPUs: every
 * rank takes the 256-bin histogram of its rows (the only pass over pixels, :171-183), the
 * ranks sum their counts (one all_reduce of 1 KB), and every rank derives the same zones from
 * the summed counts.  `counts` = 256 uint32 in host or device memory (`counts_memory`).
 * stepth_depth_split(depth) == stepth_zones_from_histogram(stepth_depth_histogram(depth)).
 */
int stepth_depth_histogram(stepth_engine *engine, const stepth_image *depth, uint32_t *counts,
                           uint32_t counts_memory, uint32_t flags);
int stepth_zones_from_histogram(stepth_engine *engine, const uint32_t *counts,
                                uint32_t counts_memory, uint32_t zones, stepth_zone *out,
                                uint32_t capacity, uint32_t *n_out);

/*
 * MaskImage::apply_mask, src/mask_image.rs:205-213: RGBA pixels whose mask byte is 0 become
 * [0;4], in place.  rgba [H][W][4], mask [H][W] u8 of the same size.
 */
int stepth_apply_mask(stepth_engine *engine, const stepth_image *rgba, const stepth_image *mask,
                      uint32_t flags);

/* Counting costs a little kernel time, so it is off by default: enable it, run a call, read
 * the counters of that call (synchronises the stream). */
int stepth_stats_enable(stepth_engine *engine, int enable);
int stepth_get_stats(stepth_engine *engine, stepth_stats *out);
/* diagnostics: the first n <= 32 raw 64-bit statistic words of the last call: [0..5] the device
 * counters above (ring0_hits .. blocks_probed), [6] superblock visits of the dense search in
 * which no block passed its bound test, [7] its block probes that did not improve a result,
 * then slots that only tuning builds of the library fill */
int stepth_get_stats_raw(stepth_engine *engine, uint64_t *out, uint32_t n);

/*
 * Timing hook for bench.py's roofline leg: brackets the dominant kernel (the ring search or
 * the SAD search) of every subsequent call with HIP events on the engine's stream and
 * accumulates; stepth_kernel_time_read() synchronises and returns the sum in milliseconds
 * and the number of launches, then resets.
 */
int stepth_kernel_time_enable(stepth_engine *engine, int enable);
int stepth_kernel_time_read(stepth_engine *engine, double *total_ms, uint64_t *launches);

/*
 * Self-test of the promise above that no call throws or aborts: raises a C++ exception inside
 * the guard every status-returning entry point runs in, and returns what the guard made of it.
 * kind 0: std::bad_alloc -> STEPTH_ERR_OUT_OF_MEMORY; 1: std::runtime_error, 2: a non-standard
 * exception -> STEPTH_ERR_HIP; anything else: STEPTH_OK.  Needs no device (engine may be NULL).
 */
int stepth_selftest_exception_guard(stepth_engine *engine, int kind);

#ifdef __cplusplus
}
#endif
#endif /* STEPTH_AMD_H */
