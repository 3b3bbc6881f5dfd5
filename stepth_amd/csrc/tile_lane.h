// tile_lane.h -- the per-query ("lane = query") half of the dense ring search.
//
// Path replaced: helpers::distance_dot_array, /root/reference/src/helpers.rs:9-54, for one
// query; the wave-level half (which blocks to look at, in which order) lives in
// tile_kernels.h.  Everything here is plain integer code that compiles for the host as well:
// tests/tools/tile_emulate.cpp runs the very same functions on the CPU against the oracle.
//
// The reference walks rings s = 0,1,2,.. around `middle` and inside a ring: row y+s, row y-s,
// column x+s, column x-s, each ascending (helpers.rs:26-31), returning the first cell inside
// the colour box.  With the key  ring | side | position  of that scan the answer is the
// matching cell with the smallest key, so cells can be examined in any order (ring_kernels.h,
// scan_key).  Here one lane owns one query and examines 8x8-cell blocks that a bound test let
// through, a LINE of 8 cells at a time, nearest line first:
//
//   * the cells of a block that lie on scan side 0 ("row y+s", dy = s >= |dx|) form rows of the
//     block; row by row s grows, and inside a row the scan position grows with x -- so the
//     first row with a hit, and its first hit, is the smallest key the block holds on side 0;
//     likewise side 1 (rows, dy = -s), side 2 (columns, dx = s > |dy|), side 3 (dx = -s);
//   * a line whose ring s is >= `limit` (= ring of the best key so far + 1, or the end of the
//     search) cannot improve the result: the scan of that side stops there.
//
// Rows come from T (cells row-major inside the block), columns from TT (column-major), so a
// line is always 32 contiguous bytes.
#pragma once
#include <stdint.h>

#include "cell_pack.h"

namespace stepth {

#ifndef STEPTH_KNOKEY_DEFINED
#define STEPTH_KNOKEY_DEFINED
constexpr uint32_t kTileNoKey = 0xFFFFFFFFu;
#endif

struct LaneQ {
    int qx, qy;      // search centre `middle` (depth_image.rs:114-117), additional-image cells
    uint32_t lo, hi; // colour box (cell_pack.h make_box)
    uint32_t best;   // smallest scan key found so far, kTileNoKey = none
    uint32_t limit;  // rings >= limit are of no use any more; 0 = search over / not searching
};

STEPTH_HD int imax(int a, int b) { return a > b ? a : b; }
STEPTH_HD int imin(int a, int b) { return a < b ? a : b; }

// bounds {max, min} (cell format) of a block / superblock meet the box on every channel.
// The all-zero pair (padding) never does.
STEPTH_HD bool box_meets(uint32_t bmax, uint32_t bmin, uint32_t lo, uint32_t hi)
{
    return ((((bmax - lo) & (hi - bmin)) | ~kGuard)) == 0xFFFFFFFFu;
}

// smallest ring of any cell of a square of `side` cells whose cell (0,0) sits at offset
// (x0, y0) from the centre
STEPTH_HD int square_dmin(int x0, int y0, int side)
{
    const int dx = imax(imax(x0, -(x0 + side - 1)), 0), dy = imax(imax(y0, -(y0 + side - 1)), 0);
    return imax(dx, dy);
}

// 8-bit mask: cell p of the line lies in the box
STEPTH_HD uint32_t line_hits(const uint32_t *c, uint32_t lo, uint32_t hi)
{
#if defined(__HIP_DEVICE_COMPILE__)
    const uint4 a = *reinterpret_cast<const uint4 *>(c), b = *reinterpret_cast<const uint4 *>(c + 4);
    const uint32_t v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#else
    const uint32_t *v = c;
#endif
    uint32_t m = 0;
#pragma unroll
    for (int p = 7; p >= 0; --p)
        m = m + m + (cell_in_box(v[p], lo, hi) ? 1u : 0u);
    return m;
}

// Examines one block for query q.  (x0, y0) = offset of the block's cell (0,0) from the
// centre; rows / cols = its 64 cells row-major (T) and column-major (TT); pb = position bits
// of the key.  Returns the number of lines read (statistics).
STEPTH_HD uint32_t probe_block(LaneQ &q, int x0, int y0, uint32_t pb, const uint32_t *rows,
                               const uint32_t *cols)
{
    uint32_t n_lines = 0;
#pragma unroll 1
    for (int side = 0; side < 4; ++side) {
        const bool is_row = side < 2, neg = side & 1;
        const int a0 = is_row ? y0 : x0; // coordinate across the lines
        const int b0 = is_row ? x0 : y0; // coordinate along a line
        const int strict = is_row ? 0 : 1; // corners belong to the rows (helpers.rs:28-31)
        const int mdb = imax(imax(b0, -(b0 + 7)), 0);
        int s = imax(imax(neg ? -(a0 + 7) : a0, 1), mdb + strict);
        const int last = neg ? -a0 : a0 + 7;
        const uint32_t *tbl = is_row ? rows : cols;
#pragma unroll 1
        for (; s <= last && (uint32_t)s < q.limit; ++s) {
            const int line = (neg ? -s : s) - a0; // 0..7
            const int r = s - strict;             // cells with |b| <= r belong to this side
            const int p0 = imax(0, -r - b0), p1 = imin(7, r - b0);
            uint32_t hits = line_hits(tbl + line * 8, q.lo, q.hi);
            hits &= (0xFFu >> (7 - p1)) & (0xFFu << p0);
            ++n_lines;
            if (hits) {
                const int p = __builtin_ctz(hits);
                const uint32_t key =
                    ((((uint32_t)s << 2) | (uint32_t)side) << pb) | (uint32_t)(b0 + p + s);
                q.best = key < q.best ? key : q.best;
                q.limit = (uint32_t)s + 1u < q.limit ? (uint32_t)s + 1u : q.limit;
                break;
            }
        }
    }
    return n_lines;
}

// Blocks of an 8x8-block window (block (0,0) of the window = block (bx0, by0) of the image)
// that lie in block-rings <= r around the centre block (cbx, cby): bit row*8+col.
STEPTH_HD uint64_t window_le_mask(int bx0, int by0, int cbx, int cby, int r)
{
    const int c_lo = imax(cbx - r - bx0, 0), c_hi = imin(cbx + r - bx0, 7);
    const int r_lo = imax(cby - r - by0, 0), r_hi = imin(cby + r - by0, 7);
    if (c_lo > c_hi || r_lo > r_hi)
        return 0ull;
    const uint64_t colbits = (uint64_t)((0xFFu >> (7 - c_hi)) & (0xFFu << c_lo) & 0xFFu);
    const uint64_t rows = colbits * 0x0101010101010101ull;
    const uint64_t sel = (~0ull >> (8 * (7 - r_hi))) & (~0ull << (8 * r_lo));
    return rows & sel;
}

// block-rings a window spans around the centre block
STEPTH_HD void window_ring_range(int bx0, int by0, int cbx, int cby, int &r_lo, int &r_hi)
{
    const int dx_lo = imax(imax(bx0 - cbx, cbx - (bx0 + 7)), 0);
    const int dy_lo = imax(imax(by0 - cby, cby - (by0 + 7)), 0);
    r_lo = imax(dx_lo, dy_lo);
    const int dx_hi = imax(cbx - bx0, bx0 + 7 - cbx), dy_hi = imax(cby - by0, by0 + 7 - cby);
    r_hi = imax(dx_hi, dy_hi);
}

// key -> (d as u8), helpers.rs:3-7 (floor sqrt) and depth_image.rs:121
STEPTH_HD uint32_t key_distance(uint32_t key, uint32_t pb)
{
    const int s = (int)(key >> (pb + 2u)), side = (int)((key >> pb) & 3u);
    const int off = (int)(key & ((1u << pb) - 1u)) - s;
    const int dx = side < 2 ? off : side == 2 ? s : -s;
    const int dy = side == 0 ? s : side == 1 ? -s : off;
    return isqrt_floor((uint32_t)(dx * dx + dy * dy));
}

} // namespace stepth
