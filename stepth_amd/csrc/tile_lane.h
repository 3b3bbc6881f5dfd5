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

// Tuning builds (-DSTEPTH_TILE_PROFILE=2) count how often each section of the probe runs, per
// lane and per wave (a wave runs a section whenever one of its lanes does); never in the
// shipped library.
#if defined(STEPTH_TILE_PROFILE) && STEPTH_TILE_PROFILE == 2
struct ProbeCounters {
    uint32_t lane[8], wave[8]; // 0 pop, 1 probe, 2 side entered, 3 bounds tested, 4 line, 5 hit path, 6 hit
};
#if defined(__HIP_DEVICE_COMPILE__)
#define STEPTH_CNT(PC, I)                                                                      \
    do {                                                                                       \
        ++(PC)->lane[I];                                                                       \
        const unsigned long long m_ = __builtin_amdgcn_ballot_w64(true);                       \
        if ((uint32_t)__builtin_ctzll(m_) == (threadIdx.x & 63u))                              \
            ++(PC)->wave[I];                                                                   \
    } while (0)
#else
#define STEPTH_CNT(PC, I) ((void)0)
#endif
#else
struct ProbeCounters {};
#define STEPTH_CNT(PC, I) ((void)0)
#endif

struct LaneQ {
    int qx, qy;      // search centre `middle` (depth_image.rs:114-117), additional-image cells
    uint32_t lo, hi; // colour box (cell_pack.h make_box)
    uint32_t best;   // smallest scan key found so far, kTileNoKey = none
    uint32_t limit;  // rings >= limit are of no use any more; 0 = search over / not searching
};

STEPTH_HD int imax(int a, int b) { return a > b ? a : b; }
STEPTH_HD int imin(int a, int b) { return a < b ? a : b; }

// smallest ring of any cell of a square of `side` cells whose cell (0,0) sits at offset
// (x0, y0) from the centre
STEPTH_HD int square_dmin(int x0, int y0, int side)
{
    const int dx = imax(imax(x0, -(x0 + side - 1)), 0), dy = imax(imax(y0, -(y0 + side - 1)), 0);
    return imax(dx, dy);
}

// One bound test shifted into `acc`: bounds = {max, min} of a block or line as one 64-bit value
// (max in the low half), k = the query's {lo, hi} addends likewise.  Four VALU instructions:
// {A, B} = bounds + k (one 64-bit add, no carry between the halves: cell_pack.h), r = A & ~B &
// guard, vcc = (r == guard), acc = acc + acc + vcc.
STEPTH_HD uint32_t bound_bit(uint32_t acc, uint64_t bounds, uint64_t k)
{
    const uint64_t t = bounds + k;
    const uint32_t r = (uint32_t)t & ~(uint32_t)(t >> 32) & kGuard;
#if defined(__HIP_DEVICE_COMPILE__)
    const uint32_t g = kGuard;
    asm("v_cmp_eq_u32 vcc, %2, %1\n v_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(acc) : "v"(r), "s"(g) : "vcc");
    return acc;
#else
    return acc + acc + (r == kGuard ? 1u : 0u);
#endif
}

// The 8 cells of a line against the box.  Per cell r = (cell+lo) & ~(cell+hi) & guard; the cell
// matches iff r == guard, and since every other subset of the three guard bits is numerically
// smaller, "some cell of the line matches" is max(r) == guard.  Two cells share each add (one
// 64-bit add of the addend in both halves): 2 instructions per cell plus 4 for the max.  Only a
// line that holds a match pays for the per-cell bits.
struct LineTest {
    uint32_t r[8];
    STEPTH_HD void load(const uint32_t *c, uint32_t lo, uint32_t hi)
    {
        const uint64_t lo2 = ((uint64_t)lo << 32) | lo, hi2 = ((uint64_t)hi << 32) | hi;
#if defined(__HIP_DEVICE_COMPILE__)
        const ulonglong2 a = *reinterpret_cast<const ulonglong2 *>(c), b = *reinterpret_cast<const ulonglong2 *>(c + 4);
        const uint64_t v[4] = {a.x, a.y, b.x, b.y};
#else
        uint64_t v[4];
        for (int p = 0; p < 4; ++p)
            v[p] = ((uint64_t)c[2 * p + 1] << 32) | c[2 * p];
#endif
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const uint64_t t = v[p] + lo2, u = v[p] + hi2;
            r[2 * p] = (uint32_t)t & ~(uint32_t)u & kGuard;
            r[2 * p + 1] = (uint32_t)(t >> 32) & ~(uint32_t)(u >> 32) & kGuard;
        }
    }
    STEPTH_HD static uint32_t umax(uint32_t a, uint32_t b) { return a > b ? a : b; }
    STEPTH_HD bool any() const
    {
        const uint32_t m0 = umax(umax(r[0], r[1]), r[2]), m1 = umax(umax(r[3], r[4]), r[5]);
        return umax(umax(umax(r[6], r[7]), m0), m1) == kGuard;
    }
    STEPTH_HD uint32_t bits() const // bit p: cell p matches
    {
        uint32_t m = 0;
#if defined(__HIP_DEVICE_COMPILE__)
        const uint32_t g = kGuard; // vcc = (r == guard), m = m + m + vcc: two instructions per cell
#define STEPTH_BIT(R) "v_cmp_eq_u32 vcc, %9, " R "\n v_addc_co_u32 %0, vcc, %0, %0, vcc\n"
        asm(STEPTH_BIT("%8") STEPTH_BIT("%7") STEPTH_BIT("%6") STEPTH_BIT("%5") STEPTH_BIT("%4")
                STEPTH_BIT("%3") STEPTH_BIT("%2") STEPTH_BIT("%1")
            : "+v"(m)
            : "v"(r[0]), "v"(r[1]), "v"(r[2]), "v"(r[3]), "v"(r[4]), "v"(r[5]), "v"(r[6]), "v"(r[7]), "s"(g)
            : "vcc");
#undef STEPTH_BIT
#else
        for (int p = 7; p >= 0; --p)
            m = m + m + (r[p] == kGuard ? 1u : 0u);
#endif
        return m;
    }
};

// One scan side of a block (SIDE 0: row y+s, 1: row y-s, 2: column x+s, 3: column x-s,
// helpers.rs:28-31), nearest line first; stops at the first hit or at ring `limit`.
// a0 / b0 = offset of the block's cell (0,0) across / along the lines, mdb = smallest |b| in
// the block, lb = {max,min} bounds of the 8 lines (rows for sides 0/1, columns for 2/3): only
// a line whose bounds meet the box is read.  Returns the lines read.
template <int SIDE>
STEPTH_HD uint32_t scan_side(LaneQ &q, int a0, int b0, int mdb, uint32_t pb, const uint8_t *tbl_base,
                             uint32_t tbl_off, const uint8_t *lb_base, uint32_t lb_off,
                             [[maybe_unused]] ProbeCounters *pc = nullptr)
{
    STEPTH_CNT(pc, 2);
    constexpr bool neg = SIDE & 1;
    constexpr int strict = SIDE < 2 ? 0 : 1; // corners belong to the rows (helpers.rs:28-31)
    const int s_base = neg ? -(a0 + 7) : a0; // ring of the line with index 0 in scan order
    const int first = imax(imax(s_base, 1), mdb + strict);
    // lines i = 0..7 in scan order (i-th nearest), ring s_base + i; line i = row/column i of the
    // block, or 7 - i on the sides that walk towards the centre's far side
    if (first > s_base + 7 || (uint32_t)first >= q.limit)
        return 0;
    uint32_t cand = 0; // bit i: line i (scan order) may hold a match
    STEPTH_CNT(pc, 3);
    {
#if defined(__HIP_DEVICE_COMPILE__)
        // the 8 {max, min} pairs as 64-bit values, highest scan index first (bound_bit)
        const ulonglong2 *l2 = reinterpret_cast<const ulonglong2 *>(lb_base + lb_off);
        const ulonglong2 w0 = l2[0], w1 = l2[1], w2 = l2[2], w3 = l2[3];
        const uint64_t k = ((uint64_t)q.hi << 32) | q.lo;
        const uint64_t w[8] = {w0.x, w0.y, w1.x, w1.y, w2.x, w2.y, w3.x, w3.y};
#pragma unroll
        for (int i = 7; i >= 0; --i)
            cand = bound_bit(cand, w[neg ? 7 - i : i], k);
#else
        const uint32_t *lb = reinterpret_cast<const uint32_t *>(lb_base + lb_off);
        for (int i = 7; i >= 0; --i) { // highest scan index first: it ends in the highest bit
            const int l = neg ? 7 - i : i;
            cand = cand + cand + (box_meets(lb[2 * l], lb[2 * l + 1], q.lo, q.hi) ? 1u : 0u);
        }
#endif
    }
    cand &= 0xFFu << (first - s_base);
    uint32_t n_lines = 0;
#pragma unroll 1
    while (cand) {
        const int i = __builtin_ctz(cand);
        cand &= cand - 1u;
        const int s = s_base + i;
        if ((uint32_t)s >= q.limit)
            break;
        const int line = neg ? 7 - i : i;
        LineTest t;
        t.load(reinterpret_cast<const uint32_t *>(tbl_base + (tbl_off + (uint32_t)line * 32u)), q.lo, q.hi);
        ++n_lines;
        STEPTH_CNT(pc, 4);
        if (t.any()) {
            STEPTH_CNT(pc, 5);
            const int r = s - strict; // cells with |b| <= r belong to this side
            const int p0 = imax(0, -r - b0), p1 = imin(7, r - b0);
            const uint32_t hits = t.bits() & (0xFFu >> (7 - p1)) & (0xFFu << p0);
            if (hits) {
                STEPTH_CNT(pc, 6);
                const int p = __builtin_ctz(hits);
                const uint32_t key =
                    ((((uint32_t)s << 2) | (uint32_t)SIDE) << pb) | (uint32_t)(b0 + p + s);
                q.best = key < q.best ? key : q.best;
                q.limit = (uint32_t)s + 1u < q.limit ? (uint32_t)s + 1u : q.limit;
                break;
            }
        }
    }
    return n_lines;
}

// Examines one block for query q.  (x0, y0) = offset of the block's cell (0,0) from the
// centre.  The tables arrive as their (wave-uniform) bases and the block's 32-bit BYTE offset
// `off`: its 64 cells row-major at T + off, column-major at TT + off, its 16 line bounds (8 rows,
// then 8 columns; uint2 {max,min} each) at LB + off / 2 -- every load is then scalar base + vector
// offset, no per-lane 64-bit pointer is formed or kept (the whole of T is < 2^32 bytes, engine.hip).
// pb = position bits of the key.  Returns the number of lines read (statistics).
STEPTH_HD uint32_t probe_block(LaneQ &q, int x0, int y0, uint32_t pb, const uint8_t *T,
                               const uint8_t *TT, const uint8_t *LB, uint32_t off,
                               [[maybe_unused]] ProbeCounters *pc = nullptr)
{
    STEPTH_CNT(pc, 1);
    const int mdx = imax(imax(x0, -(x0 + 7)), 0), mdy = imax(imax(y0, -(y0 + 7)), 0);
    const uint32_t lbo = off >> 1;
    uint32_t n = scan_side<0>(q, y0, x0, mdx, pb, T, off, LB, lbo, pc);
    n += scan_side<1>(q, y0, x0, mdx, pb, T, off, LB, lbo, pc);
    n += scan_side<2>(q, x0, y0, mdy, pb, TT, off, LB, lbo + 64u, pc);
    n += scan_side<3>(q, x0, y0, mdy, pb, TT, off, LB, lbo + 64u, pc);
    return n;
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
