/*
 * stepth_oracle.h -- CPU restatement of nikilark/stepth's open_depth_from_additional path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under stepth_amd/ (the product) may include, link or
 * call this.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it,
 * and there only as the checker / the timed CPU baseline ("kind": "port").
 *
 * This is a C restatement of Rust code, NOT the Rust binary: the reference cannot be built in
 * this environment (no rustc/cargo; nightly feature gate at src/lib.rs:1; the `disage`
 * dependency is an empty, un-vendored submodule, .gitmodules:1-3 / Cargo.toml:9).
 *
 * Parity status (see DESIGN.md "Oracle"):
 *   - ring first-hit matcher, middle, u8 wrap, max, normalise: restated line by line from
 *     src/helpers.rs:3-54 and src/depth_image.rs:111-129; pinned by tolerance against the
 *     reference's only shipped result, assets/depth.jpg (tests/test_oracle_asset.py).
 *   - disage::PixelOpps::{substract,lt} (source absent): per-channel |a-b| and ALL channels
 *     strictly `<`, pinned empirically by the same asset.
 *   - disage grouping (rgb_discrete), to_luma8_from_rgb8 and image 0.23.14's Gaussian
 *     resample: source absent, no fixture -> "parity unpinned"; groups are an explicit input
 *     here (uniform tiles or a caller-supplied rectangle list).
 *   - depth_split / select_foreground / slice / invert_depth / apply_mask: restated item by item
 *     from src/depth_image.rs:162-247 and src/mask_image.rs:205-213; pinned by tolerance against
 *     the reference's second shipped result, assets/foreground.jpg (tests/test_oracle_zones.py).
 *   - oracle_adaptive_groups is this project's own stand-in for disage's grouping (parity
 *     unpinned by construction).
 *   - the windowed-SAD mode has no counterpart in the reference at all (self-oracle).
 *
 * All citations are relative to /root/reference.
 */
#ifndef STEPTH_ORACLE_H
#define STEPTH_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Mirrors disage::DiscretePixel{position{x,y}, size{width,height}, value:[u8;3]} as used at
 * src/depth_image.rs:110-121.  Same layout as stepth_group in include/stepth_amd.h. */
typedef struct {
    uint32_t x, y, w, h;
    uint8_t value[3];
    uint8_t pad;
} oracle_group;

enum {
    ORACLE_OK = 0,
    ORACLE_ERR_BAD_ARG = -1,
    ORACLE_ERR_MAX_ZERO = -4 /* src/depth_image.rs:127 divides by max; the reference panics */
};

enum { ORACLE_THREADS_SERIAL = 0, ORACLE_THREADS_REF8 = 1, ORACLE_THREADS_ALLCORES = 2 };

/* src/helpers.rs:3-7 -- (((x1-x2)^2 + (y1-y2)^2) as f64).sqrt() as u32 */
uint32_t oracle_distance_dot_dot(uint32_t fx, uint32_t fy, uint32_t sx, uint32_t sy);

/* src/helpers.rs:9-54.  `array` is the additional image as row-major RGB8, indexed [y][x]
 * (helpers.rs:31-32), `stride` bytes per row.  Returns 1 and fills dist/pos on a hit, 0 for
 * None.  `probes`, when non-NULL, is incremented once per predicate evaluation (in-bounds
 * cell), `cells` once per get2d call. */
int oracle_distance_dot_array(const uint8_t what[3], const uint8_t *array, uint32_t aw,
                              uint32_t ah, size_t stride, uint32_t from_x, uint32_t from_y,
                              uint32_t max, const uint8_t precision[3], uint32_t *dist,
                              uint32_t *pos_x, uint32_t *pos_y, uint64_t *probes,
                              uint64_t *cells);

/* Uniform-tile stand-in for disage::open::rgb_discrete (src/depth_image.rs:104-110):
 * tiles of tile_w x tile_h (edge tiles clipped), row-major order, value = per-channel
 * floor(mean).  tile 1x1 -> value is the pixel itself.  Returns the number of groups
 * written (or needed, when groups == NULL).  `main_px` has `channels` (3 or 4) bytes per
 * pixel; alpha is dropped like DynamicImage::to_rgb8 (src/depth_image.rs:105). */
size_t oracle_uniform_groups(const uint8_t *main_px, uint32_t w, uint32_t h, size_t stride,
                             uint32_t channels, uint32_t tile_w, uint32_t tile_h,
                             oracle_group *groups);

/* Adaptive stand-in for disage::open::rgb_discrete(img, MeanBrightnessHasher,
 * BrightnessChecker{precision}, (min_splits, max_splits)) (src/depth_image.rs:101-110).
 * disage's source is absent from the reference tree, so this is the ENGINE'S OWN DESIGN and
 * its parity with disage is UNPINNED; it keeps what the call site fixes: rectangular groups
 * {position, size, value} that tile the image, at least min_splits levels of binary splitting
 * (>= 2^16 groups on images of >= 2^16 pixels), at most max_splits levels (down to single
 * pixels when max_splits = ceil(log2(w*h)), :103), the value a mean colour, uniformity
 * judged with `precision`:
 *   a rectangle at depth d is a leaf if it is a single pixel, or d >= max_splits, or
 *   d >= min_splits and for every channel max - min < precision[c];
 *   otherwise it is cut across its longer side (width >= height: into widths w/2 and w - w/2,
 *   else heights h/2 and h - h/2), left/top child first;
 *   value = per-channel floor(mean).
 * Groups are written in depth-first order.  Returns the number of groups (also when
 * groups == NULL).  max_splits = 0 means ceil(log2(w*h)) like :103. */
size_t oracle_adaptive_groups(const uint8_t *main_px, uint32_t w, uint32_t h, size_t stride,
                              uint32_t channels, const uint8_t precision[3],
                              uint32_t min_splits, uint32_t max_splits, oracle_group *groups);

/* PASS 1 of src/depth_image.rs:112-123 over groups [0,n): raw[i] = (d as u8), d = 0 on None.
 * Optional outputs: raw_u32 (d before the cast), pos_xy (winning cell, 2 per group), found.
 * threads: ORACLE_THREADS_*; REF8 reproduces par_chunks_mut(len/8) (depth_image.rs:111-112). */
int oracle_match_groups(const oracle_group *groups, size_t n, const uint8_t *add_px,
                        uint32_t aw, uint32_t ah, size_t astride, uint32_t max_steps,
                        const uint8_t precision[3], int threads, uint8_t *raw,
                        uint32_t *raw_u32, uint32_t *pos_xy, uint8_t *found, uint64_t *probes);

/* src/depth_image.rs:124-129: max over raw, then raw*255/max in u64 (floor).
 * Returns ORACLE_ERR_MAX_ZERO (and zero-fills out) when max == 0. */
int oracle_normalise(const uint8_t *raw, size_t n, uint8_t *out, uint8_t *max_out);

/* discr_main.collect() + to_luma8 (src/depth_image.rs:130-132) without the Gaussian tail:
 * paints each group's value over its rectangle into depth[H][W]. */
void oracle_paint(const oracle_group *groups, const uint8_t *values, size_t n, uint32_t w,
                  uint32_t h, uint8_t *depth, size_t dstride);

/* Whole path, src/depth_image.rs:96-132 (no Gaussian resample).  groups==NULL -> uniform
 * tiles.  depth is [H][W] u8.  raw_out (optional) receives the painted pre-normalise map. */
int oracle_depth_from_additional(const uint8_t *main_px, uint32_t w, uint32_t h, size_t mstride,
                                 uint32_t mchannels, const uint8_t *add_px, uint32_t aw,
                                 uint32_t ah, size_t astride, const uint8_t precision[3],
                                 uint32_t max_steps, uint32_t tile_w, uint32_t tile_h,
                                 const oracle_group *groups, size_t n_groups, int threads,
                                 uint8_t *depth, uint8_t *raw_out, uint8_t *max_out,
                                 uint64_t *probes);

/* The tail of load_depth_from_additional, src/depth_image.rs:133-134:
 * DynamicImage::resize(w, h, imageops::Gaussian) to the SAME size.  image 0.23.14 is not in
 * the reference tree, so this follows its published sampler from memory (vertical pass then
 * horizontal pass; kernel gaussian(x, 0.5) with support 3 -> taps i in [o-3, o+3], window
 * clipped at the borders, weights normalised by their sum, f32 accumulation in tap order,
 * round-half-away to u8 after each pass): PARITY UNPINNED -- checked only by tolerance
 * against assets/depth.jpg. */
void oracle_gaussian_same_size(const uint8_t *src, uint32_t w, uint32_t h, size_t sstride,
                               uint8_t *dst, size_t dstride);

/* Windowed-SAD mode (north_star's literal description; NOT in the reference; self-oracle).
 * cost(x,y,d) = sum over win x win window, 3 channels of |L(x+i,y+j) - R(x+i-d,y+j)| with
 * clamp-to-edge coordinates; raw = smallest d in [0,d_max) with cost < win*win*(p0+p1+p2),
 * else the argmin (smallest d on ties).  Rows [row0,row1). */
int oracle_sad_rows(const uint8_t *left, const uint8_t *right, uint32_t w, uint32_t h,
                    size_t lstride, size_t rstride, uint32_t channels, uint32_t d_max,
                    uint32_t win, const uint8_t precision[3], uint32_t row0, uint32_t row1,
                    int threads, uint8_t *raw);

/* ---- consumers of a finished depth map ------------------------------- */

#define ORACLE_SPLIT_OK 0
#define ORACLE_SPLIT_ERR_STEP (-7)     /* src/depth_image.rs:217: step_by(0) / usize underflow  */
#define ORACLE_SPLIT_ERR_EMPTY (-8)    /* :203-206: min()/max() of an empty zone -> unwrap panic */
#define ORACLE_SPLIT_ERR_DIVERGED (-9) /* :210 recursion did not end within the iteration cap   */
#define ORACLE_SPLIT_ERR_INPUT (-1)    /* :214-215: min() of an empty depth map -> unwrap panic  */
#define ORACLE_SPLIT_MAX_ITERS 4096

/* DepthImage::depth_split, src/depth_image.rs:162-219, item by item (no histogram shortcut):
 * 1-D k-means over the depth bytes.  zones < 2 -> one zone (None, None): *n_zones = 1 and
 * has[0] = 0.  Otherwise zone i is (Some(lo[i]), Some(hi[i])), has[i] = 1, zones ordered by
 * centroid.  The reference keys its clusters by centroid VALUE (HashMap<u8, Vec<u8>>, :167),
 * assigns every item to the FIRST nearest centre in list order (min_by, :172-178), takes
 * floor means with an empty cluster giving 0 (:184-187), sorts them (:188) and compares them
 * with the previous list pairwise over the shorter length (:189-195).  Inputs on which the
 * reference panics return one of the error codes above.  lo/hi/has hold up to 256 entries.
 * Returns the number of assignment passes through *iters (may be NULL). */
int oracle_depth_split(const uint8_t *depth, size_t n, uint32_t zones, uint8_t *lo, uint8_t *hi,
                       uint8_t *has, uint32_t *n_zones, uint32_t *iters);

/* DepthImage::invert_depth, src/depth_image.rs:226-228 */
void oracle_invert_depth(uint8_t *depth, size_t n);

/* DepthImage::slice, src/depth_image.rs:230-247: mask = 255 (MASK_TRUE, mask_image.rs:3)
 * unless depth < from or depth > to, then 0 (MASK_FALSE, :4); None -> 0 / 255 */
void oracle_slice(const uint8_t *depth, size_t n, int has_from, uint8_t from, int has_to,
                  uint8_t to, uint8_t *mask);

/* DepthImage::select_foreground, src/depth_image.rs:221-224: slice(depth_split(2)[0]) */
int oracle_select_foreground(const uint8_t *depth, size_t n, uint8_t *mask, uint8_t *from,
                             uint8_t *to);

/* MaskImage::apply_mask, src/mask_image.rs:205-213: RGBA pixel := [0;4] where mask == 0 */
void oracle_apply_mask(uint8_t *rgba, const uint8_t *mask, size_t n);

/* threads used by ORACLE_THREADS_ALLCORES: every online core, or the count set here (a
 * container's CPU share is usually smaller than the host's core count) */
void oracle_set_threads(int n);
int oracle_hardware_threads(void);

#ifdef __cplusplus
}
#endif
#endif
