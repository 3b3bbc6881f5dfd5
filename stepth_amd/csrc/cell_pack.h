// cell_pack.h -- HBM layout of one additional-image cell and the 3-channel box test.
//
// The reference tests `v.substract(what).lt(precision)` per probe
// (/root/reference/src/helpers.rs:35): per-channel |v-what| strictly below precision on all
// three channels, i.e. v_c in [what_c-p_c+1, what_c+p_c-1] -- membership in a 3-D box.
//
// A cell is stored as one dword with the three channels spread into 10-bit fields:
//
//      bits  0..7  R   bits  8..9  zero
//      bits 10..17 G   bits 18..19 zero
//      bits 20..27 B   bits 28..31 zero
//
// The query's box [l_c, h_c] is packed as two ADDENDS,  lo = sum (256 - l_c) << 10c  and
// hi = sum (255 - h_c) << 10c.  No field of cell + lo or cell + hi can carry into its neighbour
// (at most 511), bit 8 of field c of cell + lo is set exactly when cell_c >= l_c, and bit 8 of
// cell + hi exactly when cell_c > h_c.  One probe is therefore
//      A = cell + lo,  B = cell + hi,  match  <=>  (A & ~B & guard) == guard
// -- two adds and one three-input bit operation for all three channels.  Because both tests are
// additions of a per-query constant, two of them fit one 64-bit add (v_lshl_add_u64 issues at
// nearly the rate of a 32-bit add on gfx950, tools/issue_microbench.hip): two cells of a line
// against one addend, or the {max, min} pair of a block's bounds against the {lo, hi} pair.
//
// kNoCell -- 256 in every field, which no pixel has -- fails the "<= h" test of every box: it
// is the "no cell here" value of row padding and masked lanes, which is how the reference's
// `None => continue` (helpers.rs:44) is expressed without a branch.  Bounds of something empty
// are {anything, kNoCell}.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define STEPTH_HD __host__ __device__ __forceinline__
#else
#define STEPTH_HD inline
#endif

namespace stepth {

constexpr uint32_t kGuard = (1u << 8) | (1u << 18) | (1u << 28);
constexpr uint32_t kNoCell = kGuard;

STEPTH_HD uint32_t pack_cell(uint32_t r, uint32_t g, uint32_t b) { return r | (g << 10) | (b << 20); }

// Box of a query colour `v` under `precision` (helpers.rs:35), as the two addends.  precision 0
// on a channel can never be satisfied (nothing is < 0): l = 256 makes the >= test fail for
// every cell.
STEPTH_HD void make_box(const uint8_t v[3], const uint8_t p[3], uint32_t &lo, uint32_t &hi)
{
    lo = 0;
    hi = 0;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        int l, h;
        if (p[c] == 0) {
            l = 256;
            h = 0;
        } else {
            l = (int)v[c] - (int)p[c] + 1;
            h = (int)v[c] + (int)p[c] - 1;
            l = l < 0 ? 0 : l;
            h = h > 255 ? 255 : h;
        }
        lo |= (uint32_t)(256 - l) << (10 * c);
        hi |= (uint32_t)(255 - h) << (10 * c);
    }
}

STEPTH_HD bool cell_in_box(uint32_t cell, uint32_t lo, uint32_t hi)
{
    return ((cell + lo) & ~(cell + hi) & kGuard) == kGuard;
}

// bounds {max, min} (cell format) of a block / superblock / line meet the box on every channel:
// max_c >= l_c and min_c <= h_c.  Empty bounds (min = kNoCell) never do.
STEPTH_HD bool box_meets(uint32_t bmax, uint32_t bmin, uint32_t lo, uint32_t hi)
{
    return ((bmax + lo) & ~(bmin + hi) & kGuard) == kGuard;
}

// floor(sqrt(n)) for the reference's `(.. as f64).sqrt() as u32` (helpers.rs:3-7); exact for
// every n < 2^24 regardless of how the f32 square root rounds.
STEPTH_HD uint32_t isqrt_floor(uint32_t n)
{
    uint32_t r = (uint32_t)__builtin_sqrtf((float)n);
    r = (r * r > n) ? r - 1 : r;
    r = ((r + 1) * (r + 1) <= n) ? r + 1 : r;
    return r;
}

} // namespace stepth
