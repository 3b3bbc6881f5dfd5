// cell_pack.h -- HBM layout of one additional-image cell and the 3-channel box test.
//
// The reference tests `v.substract(what).lt(precision)` per probe
// (/root/reference/src/helpers.rs:35): per-channel |v-what| strictly below precision on all
// three channels, i.e. v_c in [what_c-p_c+1, what_c+p_c-1] -- membership in a 3-D box.
//
// A cell is stored as one dword with the three channels spread into 10-bit fields and a
// guard bit (bit 8 of each field) pre-set:
//
//      bits  0..7  R   bit  8 guard   bit  9 zero
//      bits 10..17 G   bit 18 guard   bit 19 zero
//      bits 20..27 B   bit 28 guard   bits 29..31 zero
//
// With the query's box packed as  lo = sum lo_c << 10c  and  hi = sum (hi_c + 512) << 10c
// no field of (cell - lo) or (hi - cell) can borrow from its neighbour, and bit 8 of field c
// is set exactly when cell_c >= lo_c resp. cell_c <= hi_c.  One probe is therefore
//      v_sub, v_sub, v_and_or, v_cmp           (4 VALU ops for all three channels)
// instead of three byte extractions, three abs-diffs and three compares.
//
// The all-zero dword (guard bits clear) can never satisfy the test: it is the "no cell here"
// value used for row padding and masked lanes, which is how the reference's `None => continue`
// (helpers.rs:44) is expressed without a branch.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define STEPTH_HD __host__ __device__ __forceinline__
#else
#define STEPTH_HD inline
#endif

namespace stepth {

constexpr uint32_t kGuard = (1u << 8) | (1u << 18) | (1u << 28);
constexpr uint32_t kNoCell = 0u;

STEPTH_HD uint32_t pack_cell(uint32_t r, uint32_t g, uint32_t b)
{
    return (r | (g << 10) | (b << 20)) | kGuard;
}

// Box of a query colour `v` under `precision` (helpers.rs:35).  precision 0 on a channel can
// never be satisfied (nothing is < 0): lo = 256 makes the >= test fail for every cell.
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
        lo |= (uint32_t)l << (10 * c);
        hi |= (uint32_t)(h + 512) << (10 * c);
    }
}

STEPTH_HD bool cell_in_box(uint32_t cell, uint32_t lo, uint32_t hi)
{
    const uint32_t t = cell - lo;
    const uint32_t u = hi - cell;
    return ((t & u) | ~kGuard) == 0xFFFFFFFFu;
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
