// tile_emulate.cpp -- CPU emulation of ring_tile_kernel and ring_list_kernel
// (stepth_amd/csrc/tile_kernels.h), and a self-check of cell_pack.h's arithmetic.
//
// Test infrastructure: runs the kernel's per-lane code (tile_lane.h, cell_pack.h -- the very
// same headers) and mirrors its wave-level control flow with plain loops over 64 "lanes", so
// that the logic of the dense search can be checked against the oracle without a GPU
// (tests/test_tile_emulation.py).  Nothing in the product uses this file.
//
// build: g++ -O2 -shared -fPIC -I stepth_amd/csrc tests/tools/tile_emulate.cpp -o <out>.so
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <vector>

#include "tile_lane.h"

using namespace stepth;

namespace {
constexpr int kPadB = 4;
uint32_t tile_bwp(uint32_t bw) { return ((bw + 7u) & ~7u) + 8u; }
uint32_t tile_bhp(uint32_t bh) { return ((bh + 7u) & ~7u) + 8u; }
struct U2 {
    uint32_t x, y;
};
} // namespace

// the packed additional image (tile_pack_kernel / tile_super_kernel), built on the CPU
struct Packed {
    uint32_t wa, ha, bw, bh, sbw, sbh, bwp, bhp;
    std::vector<uint32_t> T, TT;
    std::vector<U2> BBp, SB, LB;
    Packed(const uint8_t *add, uint32_t wa_, uint32_t ha_)
        : wa(wa_), ha(ha_), bw((wa_ + 7) / 8), bh((ha_ + 7) / 8), sbw((bw + 7) / 8), sbh((bh + 7) / 8),
          bwp(tile_bwp(bw)), bhp(tile_bhp(bh)), T((size_t)bw * bh * 64, kNoCell), TT((size_t)bw * bh * 64, kNoCell),
          BBp((size_t)bwp * bhp, U2{0, kNoCell}), SB((size_t)sbw * sbh, U2{0, kNoCell}),
          LB((size_t)bw * bh * 16, U2{0, kNoCell}) // per block: 8 row bounds, 8 column bounds
    {
    // pack (tile_pack_kernel / tile_super_kernel)
    for (uint32_t by = 0; by < bh; ++by)
        for (uint32_t bx = 0; bx < bw; ++bx) {
            uint32_t mn[3] = {255, 255, 255}, mx[3] = {0, 0, 0};
            for (uint32_t l = 0; l < 64; ++l) {
                const uint32_t x = bx * 8 + (l & 7), y = by * 8 + (l >> 3);
                uint32_t cell = kNoCell;
                if (x < wa && y < ha) {
                    const uint8_t *p = add + ((size_t)y * wa + x) * 3;
                    for (int c = 0; c < 3; ++c) {
                        mn[c] = std::min<uint32_t>(mn[c], p[c]);
                        mx[c] = std::max<uint32_t>(mx[c], p[c]);
                    }
                    cell = pack_cell(p[0], p[1], p[2]);
                }
                const size_t base = ((size_t)by * bw + bx) * 64;
                T[base + l] = cell;
                TT[base + ((l & 7) << 3) + (l >> 3)] = cell;
            }
            BBp[(size_t)(by + kPadB) * bwp + bx + kPadB] =
                U2{pack_cell(mx[0], mx[1], mx[2]), pack_cell(mn[0], mn[1], mn[2])};
            for (uint32_t i = 0; i < 16; ++i) { // line bounds: rows 0..7, then columns 0..7
                uint32_t ln[3] = {255, 255, 255}, lx[3] = {0, 0, 0};
                bool any = false;
                for (uint32_t p = 0; p < 8; ++p) {
                    const uint32_t x = bx * 8 + (i < 8 ? p : i - 8), y = by * 8 + (i < 8 ? i : p);
                    if (x >= wa || y >= ha)
                        continue;
                    any = true;
                    const uint8_t *px = add + ((size_t)y * wa + x) * 3;
                    for (int c = 0; c < 3; ++c) {
                        ln[c] = std::min<uint32_t>(ln[c], px[c]);
                        lx[c] = std::max<uint32_t>(lx[c], px[c]);
                    }
                }
                if (any)
                    LB[((size_t)by * bw + bx) * 16 + i] =
                        U2{pack_cell(lx[0], lx[1], lx[2]), pack_cell(ln[0], ln[1], ln[2])};
            }
        }
    for (uint32_t sy = 0; sy < sbh; ++sy)
        for (uint32_t sx = 0; sx < sbw; ++sx) {
            uint32_t mn[3] = {255, 255, 255}, mx[3] = {0, 0, 0};
            for (uint32_t l = 0; l < 64; ++l) {
                const uint32_t bx = sx * 8 + (l & 7), by = sy * 8 + (l >> 3);
                if (bx >= bw || by >= bh)
                    continue;
                const U2 bb = BBp[(size_t)(by + kPadB) * bwp + bx + kPadB];
                for (int c = 0; c < 3; ++c) {
                    mx[c] = std::max(mx[c], (bb.x >> (10 * c)) & 0xffu);
                    mn[c] = std::min(mn[c], (bb.y >> (10 * c)) & 0xffu);
                }
            }
            SB[(size_t)sy * sbw + sx] = U2{pack_cell(mx[0], mx[1], mx[2]), pack_cell(mn[0], mn[1], mn[2])};
        }

    }
};

// Probe policy of the emulated kernel: g_flush_min == 0 -> ring_tile_kernel<.., POOL = false>,
// every lane probes its own candidate blocks as it finds them; > 0 -> the POOL variant:
// candidate (query, block) pairs go into a wave-level task pool that is executed 64 tasks at a
// time against a snapshot of the queries, results merged by minimum, once it holds g_flush_min
// tasks, before it could overflow (g_pool_cap), at the end of step 0 and of the search.
static int g_flush_min = 0, g_pool_cap = 128; // 0: the plain variant; 16 / 128 = kPoolFlushMin / kPoolCap of the POOL variant
extern "C" void tile_emulate_policy(int flush_min, int pool_cap)
{
    g_flush_min = flush_min;
    g_pool_cap = pool_cap;
}

extern "C" int tile_emulate(const uint8_t *main_px, uint32_t W, uint32_t H, uint32_t mch,
                            const uint8_t *add, uint32_t wa, uint32_t ha, const uint8_t *prec,
                            uint32_t max_steps_in, uint32_t row0, uint32_t row1, uint8_t *raw,
                            uint32_t *max_out,
                            uint64_t *stats /* [6]: probes, lines, steps, wave-level probe iterations, push iterations, far waves */)
{
    const Packed P(add, wa, ha);
    const uint32_t bw = P.bw, bh = P.bh, sbw = P.sbw, bwp = P.bwp;
    const std::vector<uint32_t> &T = P.T, &TT = P.TT;
    const std::vector<U2> &BBp = P.BBp, &SB = P.SB, &LB = P.LB;
    const uint32_t max_steps = std::min(max_steps_in, std::max(wa, ha));
    const uint32_t top = std::max(max_steps, 2u) - 1u;
    uint32_t pb = 1;
    while ((2u * top) >> pb)
        ++pb;
    const uint32_t ring_shift = pb + 2;
    const uint32_t cbx_n = ((W >> 1) + 8u) >> 3, cby0 = ((row0 + 1u) >> 1) >> 3;
    const uint32_t cby_n = ((row1 >> 1) >> 3) - cby0 + 1u;
    uint32_t mxd = 0;
    uint64_t n_probes = 0, n_lines = 0, n_steps = 0, n_iters = 0, n_push = 0, n_far = 0;

    for (uint32_t cbyi = 0; cbyi < cby_n; ++cbyi)
        for (uint32_t cbxi = 0; cbxi < cbx_n; ++cbxi)
            for (uint32_t wv = 0; wv < 4; ++wv) {
                const int CBX = (int)cbxi, CBY = (int)(cby0 + cbyi);
                LaneQ q[64];
                uint32_t s_end[64], px[64], py[64];
                bool valid[64];
                bool any = false;
                for (uint32_t lane = 0; lane < 64; ++lane) {
                    const uint32_t c = lane >> 2, m = lane & 3;
                    const uint32_t cx = CBX * 8u + (wv & 1) * 4 + (c & 3), cy = CBY * 8u + (wv >> 1) * 4 + (c >> 2);
                    const uint32_t x = 2 * cx - 1 + (m & 1), y = 2 * cy - 1 + (m >> 1);
                    valid[lane] = x < W && y >= row0 && y < row1;
                    px[lane] = x;
                    py[lane] = y;
                    LaneQ &l = q[lane];
                    l.qx = (int)cx;
                    l.qy = (int)cy;
                    l.lo = l.hi = 0;
                    l.best = kTileNoKey;
                    l.limit = 0;
                    s_end[lane] = 0;
                    if (!valid[lane])
                        continue;
                    const uint8_t *p = main_px + ((size_t)y * W + x) * mch;
                    const uint8_t v[3] = {p[0], p[1], p[2]};
                    make_box(v, prec, l.lo, l.hi);
                    if (cx < wa && cy < ha && max_steps > 0) {
                        const uint32_t cell = T[((size_t)CBY * bw + CBX) * 64 + ((cy & 7) << 3) + (cx & 7)];
                        const uint32_t smax = std::max(std::max(cx, wa - 1 - cx), std::max(cy, ha - 1 - cy));
                        s_end[lane] = std::min(max_steps, smax + 1);
                        if (cell_in_box(cell, l.lo, l.hi))
                            l.best = 0;
                        else if (s_end[lane] > 1)
                            l.limit = s_end[lane];
                    }
                    any |= l.limit != 0;
                }
                if (any) {
                    const int R = (int)max_steps - 1;
                    const int wx0 = CBX * 8 + (int)(wv & 1) * 4, wy0 = CBY * 8 + (int)(wv >> 1) * 4;
                    const int sx0 = std::max(wx0 - R, 0) >> 6, sy0 = std::max(wy0 - R, 0) >> 6;
                    const int sx1 = std::min(wx0 + 3 + R, (int)wa - 1) >> 6;
                    const int sy1 = std::min(wy0 + 3 + R, (int)ha - 1) >> 6;
                    const int nsx = sx1 - sx0 + 1, nsy = sy1 - sy0 + 1;
                    if (nsx > 16 || nsy > 12)
                        return -1; // the kernel's slots would not fit
                    std::vector<uint32_t> skeys; // wave-level slots
                    struct Task {
                        int lane, bx, by;
                    };
                    std::vector<Task> pool;
                    int fault = 0;
                    // executes the pooled tasks, 64 at a time: the tasks of one go see the same
                    // snapshot of their queries, results merge by minimum
                    auto flush = [&]() {
                        for (size_t b = 0; b < pool.size(); b += 64) {
                            ++n_iters;
                            const size_t n = std::min<size_t>(64, pool.size() - b);
                            LaneQ res[64];
                            for (size_t i = 0; i < n; ++i) {
                                const Task &t = pool[b + i];
                                LaneQ tq = q[t.lane];
                                const int x0 = t.bx * 8 - tq.qx, y0 = t.by * 8 - tq.qy;
                                if ((uint32_t)square_dmin(x0, y0, 8) < tq.limit) {
                                    if (t.bx < 0 || t.by < 0 || t.bx >= (int)bw || t.by >= (int)bh) {
                                        fault = -2;
                                        return;
                                    }
                                    const size_t boff = ((size_t)t.by * bw + t.bx) * 64;
                                    n_lines += probe_block(tq, x0, y0, pb, reinterpret_cast<const uint8_t *>(T.data()), reinterpret_cast<const uint8_t *>(TT.data()),
                                                           reinterpret_cast<const uint8_t *>(LB.data()), (uint32_t)(boff * 4));
                                    ++n_probes;
                                }
                                res[i] = tq;
                            }
                            for (size_t i = 0; i < n; ++i) {
                                LaneQ &o = q[pool[b + i].lane];
                                o.best = std::min(o.best, res[i].best);
                                o.limit = std::min(o.limit, res[i].limit);
                            }
                        }
                        pool.clear();
                    };
                    for (uint32_t it = 0;; ++it) {
                        int bx0, by0, r_lo, r_hi;
                        uint64_t allow;
                        bool want[64];
                        if (it == 0) {
                            bx0 = CBX - 3;
                            by0 = CBY - 3;
                            allow = 0x007F7F7F7F7F7F7Full;
                            r_lo = 0;
                            r_hi = 3;
                            for (int l = 0; l < 64; ++l)
                                want[l] = q[l].limit != 0;
                        } else {
                            if (it == 1) {
                                flush(); // the settle test needs step 0's results
                                if (fault)
                                    return fault;
                                bool left = false;
                                for (int l = 0; l < 64; ++l) {
                                    if (q[l].limit != 0 &&
                                        (s_end[l] <= 25 || (q[l].best != kTileNoKey && (q[l].best >> ring_shift) <= 24)))
                                        q[l].limit = 0;
                                    left |= q[l].limit != 0;
                                }
                                if (!left)
                                    break;
                                ++n_far;
                                for (int ty = 0; ty < nsy; ++ty)
                                    for (int tx = 0; tx < nsx; ++tx) {
                                        const int sx = sx0 + tx, sy = sy0 + ty;
                                        const U2 sb = SB[(size_t)sy * sbw + sx];
                                        bool anyw = false;
                                        for (int l = 0; l < 64; ++l) {
                                            const uint32_t dS = (uint32_t)square_dmin((sx << 6) - q[l].qx, (sy << 6) - q[l].qy, 64);
                                            anyw |= dS < q[l].limit && box_meets(sb.x, sb.y, q[l].lo, q[l].hi);
                                        }
                                        if (anyw) {
                                            const uint32_t dref = (uint32_t)square_dmin((sx << 6) - (wx0 + 2), (sy << 6) - (wy0 + 2), 64);
                                            skeys.push_back((dref << 8) | (uint32_t)(ty * 16 + tx));
                                        }
                                    }
                            }
                            if (skeys.empty())
                                break;
                            const auto itmin = std::min_element(skeys.begin(), skeys.end());
                            const uint32_t pick = *itmin;
                            // the kernel stops at the first slot beyond the largest limit + 2 (slots
                            // are keyed by the distance from the middle of the wave's 4x4 centres,
                            // which is within 2 rings of every lane's own distance)
                            uint32_t maxlimit = 0;
                            for (int l = 0; l < 64; ++l)
                                maxlimit = std::max(maxlimit, q[l].limit);
                            if ((pick >> 8) >= maxlimit + 2u)
                                break;
                            skeys.erase(itmin);
                            const int sx = sx0 + (int)(pick & 15u), sy = sy0 + (int)((pick >> 4) & 15u);
                            const U2 sb = SB[(size_t)sy * sbw + sx];
                            bool anyw = false;
                            for (int l = 0; l < 64; ++l) {
                                const uint32_t dS = (uint32_t)square_dmin((sx << 6) - q[l].qx, (sy << 6) - q[l].qy, 64);
                                want[l] = dS < q[l].limit && box_meets(sb.x, sb.y, q[l].lo, q[l].hi);
                                anyw |= want[l];
                            }
                            if (!anyw)
                                continue;
                            bx0 = sx * 8;
                            by0 = sy * 8;
                            allow = ~window_le_mask(bx0, by0, CBX, CBY, 3);
                            window_ring_range(bx0, by0, CBX, CBY, r_lo, r_hi);
                            r_lo = std::max(r_lo, 4);
                        }
                        ++n_steps;
                        uint64_t cand[64];
                        for (int l = 0; l < 64; ++l) {
                            uint64_t mm = 0;
                            for (int row = 0; row < 8; ++row)
                                for (int col = 0; col < 8; ++col) {
                                    const U2 bb = BBp[(size_t)(by0 + kPadB + row) * bwp + (size_t)(bx0 + kPadB + col)];
                                    if (box_meets(bb.x, bb.y, q[l].lo, q[l].hi))
                                        mm |= 1ull << (row * 8 + col);
                                }
                            cand[l] = want[l] ? mm & allow : 0ull;
                        }
                        uint64_t le_prev = r_lo > 0 ? window_le_mask(bx0, by0, CBX, CBY, r_lo - 1) : 0ull;
                        for (int r = r_lo; r <= r_hi; ++r) {
                            const uint64_t le = window_le_mask(bx0, by0, CBX, CBY, r);
                            if (g_flush_min > 0) {
                                uint64_t curs[64];
                                bool more = false;
                                for (int l = 0; l < 64; ++l) {
                                    curs[l] = cand[l] & le & ~le_prev;
                                    if (8 * (r - 1) + 1 >= (int)q[l].limit)
                                        curs[l] = 0;
                                    more |= curs[l] != 0;
                                }
                                while (more) { // one push per lane and iteration
                                    ++n_push;
                                    if ((int)pool.size() + 64 > g_pool_cap)
                                        flush();
                                    more = false;
                                    for (int l = 0; l < 64; ++l) {
                                        if (!curs[l])
                                            continue;
                                        const int k = __builtin_ctzll(curs[l]);
                                        curs[l] &= curs[l] - 1;
                                        more |= curs[l] != 0;
                                        const int bx = bx0 + (k & 7), by = by0 + (k >> 3);
                                        if ((uint32_t)square_dmin(bx * 8 - q[l].qx, by * 8 - q[l].qy, 8) < q[l].limit)
                                            pool.push_back(Task{l, bx, by});
                                    }
                                }
                                if ((int)pool.size() >= g_flush_min)
                                    flush();
                                if (fault)
                                    return fault;
                                le_prev = le;
                                continue;
                            }
                            uint64_t probed_at = 0; // bit j: some lane probed a block in its j-th pop of the round
                            for (int l = 0; l < 64; ++l) {
                                uint64_t cur = cand[l] & le & ~le_prev;
                                if (8 * (r - 1) + 1 >= (int)q[l].limit)
                                    cur = 0;
                                for (int j = 0; cur; ++j) {
                                    if ((uint32_t)square_dmin((bx0 + (__builtin_ctzll(cur) & 7)) * 8 - q[l].qx,
                                                              (by0 + (__builtin_ctzll(cur) >> 3)) * 8 - q[l].qy, 8) < q[l].limit)
                                        probed_at |= 1ull << j;
                                    const int k = __builtin_ctzll(cur);
                                    cur &= cur - 1;
                                    const int bx = bx0 + (k & 7), by = by0 + (k >> 3);
                                    const int x0 = bx * 8 - q[l].qx, y0 = by * 8 - q[l].qy;
                                    if ((uint32_t)square_dmin(x0, y0, 8) < q[l].limit) {
                                        if (bx < 0 || by < 0 || bx >= (int)bw || by >= (int)bh)
                                            return -2; // a padded block got through: memory fault on the GPU
                                        const size_t boff = ((size_t)by * bw + bx) * 64;
                                        n_lines += probe_block(q[l], x0, y0, pb, reinterpret_cast<const uint8_t *>(T.data()), reinterpret_cast<const uint8_t *>(TT.data()),
                                                           reinterpret_cast<const uint8_t *>(LB.data()), (uint32_t)(boff * 4));
                                        ++n_probes;
                                    }
                                }
                            }
                            n_iters += (uint64_t)__builtin_popcountll(probed_at);
                            le_prev = le;
                        }
                    }
                    flush(); // whatever is still pooled when the search ends
                    if (fault)
                        return fault;
                }
                for (int l = 0; l < 64; ++l) {
                    uint32_t d8 = 0;
                    if (q[l].best != kTileNoKey && q[l].best != 0)
                        d8 = key_distance(q[l].best, pb) & 0xff;
                    if (valid[l]) {
                        raw[(size_t)(py[l] - row0) * W + px[l]] = (uint8_t)d8;
                        mxd = std::max(mxd, d8);
                    }
                }
            }
    if (max_out)
        *max_out = mxd;
    if (stats) {
        stats[0] = n_probes;
        stats[1] = n_lines;
        stats[2] = n_steps;
        stats[3] = n_iters;
        stats[4] = n_push;
        stats[5] = n_far;
    }
    return 0;
}

// Exhaustive check of cell_pack.h's box arithmetic against the reference predicate
// `|v - what| < precision` on every channel (/root/reference/src/helpers.rs:35), of
// bound_bit's 64-bit form and of the empty / no-cell values.  Returns 0, or the number of the
// first check that failed.
extern "C" int cell_pack_selfcheck(void)
{
    const uint8_t precs[] = {0, 1, 2, 3, 36, 37, 128, 254, 255};
    // one channel at a time (the others held where they match), every value pair
    for (int ch = 0; ch < 3; ++ch)
        for (uint8_t p : precs)
            for (int what = 0; what < 256; ++what) {
                uint8_t w3[3] = {77, 77, 77}, p3[3] = {5, 5, 5};
                w3[ch] = (uint8_t)what;
                p3[ch] = p;
                uint32_t lo, hi;
                make_box(w3, p3, lo, hi);
                for (int v = 0; v < 256; ++v) {
                    uint32_t c3[3] = {77, 77, 77};
                    c3[ch] = (uint32_t)v;
                    const bool want = std::abs(v - what) < (int)p;
                    if (cell_in_box(pack_cell(c3[0], c3[1], c3[2]), lo, hi) != want)
                        return 1;
                    // a single-cell block: bounds {v, v}; and a block [v, v+9] against the box
                    const uint32_t pc = pack_cell(c3[0], c3[1], c3[2]);
                    if (box_meets(pc, pc, lo, hi) != want)
                        return 2;
                    const int v1 = std::min(v + 9, 255);
                    uint32_t m3[3] = {77, 77, 77};
                    m3[ch] = (uint32_t)v1;
                    bool any = false;
                    for (int u = v; u <= v1; ++u)
                        any |= std::abs(u - what) < (int)p;
                    const uint32_t bmax = pack_cell(m3[0], m3[1], m3[2]);
                    if (box_meets(bmax, pc, lo, hi) != any)
                        return 3;
                    const uint64_t k = ((uint64_t)hi << 32) | lo, b = ((uint64_t)pc << 32) | bmax;
                    if ((bound_bit(2u, b, k) == 5u) != any)
                        return 4;
                }
                if (cell_in_box(kNoCell, lo, hi) || box_meets(0u, kNoCell, lo, hi) ||
                    box_meets(kNoCell, kNoCell, lo, hi))
                    return 5; // padding must never match
            }
    // three channels together, pseudo-random, incl. LineTest on 8 cells
    uint32_t s = 12345u;
    auto rnd = [&]() { return (s = s * 1664525u + 1013904223u) >> 24; };
    for (int it = 0; it < 200000; ++it) {
        uint8_t w3[3] = {(uint8_t)rnd(), (uint8_t)rnd(), (uint8_t)rnd()};
        uint8_t p3[3] = {(uint8_t)(rnd() % 80), (uint8_t)(rnd() % 80), (uint8_t)(rnd() % 80)};
        uint32_t lo, hi, cells[8], want_bits = 0;
        make_box(w3, p3, lo, hi);
        for (int i = 0; i < 8; ++i) {
            uint32_t c3[3];
            bool in = true;
            for (int c = 0; c < 3; ++c) {
                c3[c] = (rnd() & 1u) ? rnd() : (uint32_t)std::min(255, std::max(0, (int)w3[c] + (int)(rnd() % 9) - 4));
                in = in && std::abs((int)c3[c] - (int)w3[c]) < (int)p3[c];
            }
            cells[i] = (it % 17 == 0 && i == 3) ? kNoCell : pack_cell(c3[0], c3[1], c3[2]);
            if (cells[i] != kNoCell && in)
                want_bits |= 1u << i;
        }
        LineTest t;
        t.load(cells, lo, hi);
        if (t.any() != (want_bits != 0) || t.bits() != want_bits)
            return 6;
    }
    return 0;
}

// ---------------------------------------------------------------------------------------
// CPU emulation of ring_list_kernel (prepared queries: group lists, tiles > 1x1,
// stepth_distance_dot_array_batch): the same per-lane headers, the kernel's wave-level control
// flow mirrored with loops over the `lanes` (16, 32 or 64) queries a wave takes -- mode 0 the
// lane's own 7x7 window block-ring by block-ring, then either mode 1, the wave-shared window of
// superblocks (visited nearest first from the middle of the wave's centres, with the kernel's
// tolerance and early exit), or mode 2, the lane's own ring walk for a spread-out wave, and
// mode 3, the drain.  Candidate blocks become tasks in a wave-level pool (capacity 128) that is
// executed 64 tasks at a time against a snapshot of the queries, results merged by minimum, once
// it holds 16 tasks, before it could overflow, before the own-window results are judged and at
// the end -- like the kernel (kListPoolCap, kListFlushMin).  Query k: centre (cx[k], cy[k]),
// colour what[3k..].  Outputs per query: d (distance), found, pos (x, y).  Returns 0, or < 0
// where the GPU would fault.
// ---------------------------------------------------------------------------------------
extern "C" int list_emulate(const uint32_t *cx, const uint32_t *cy, const uint8_t *what, uint32_t n,
                            const uint8_t *add, uint32_t wa, uint32_t ha, const uint8_t *prec,
                            uint32_t max_steps_in, uint32_t lanes, uint32_t *d_out, uint8_t *found_out,
                            uint32_t *pos_out, uint64_t *stats /* [4]: shared far waves, per-lane far waves, probes, pool goes */)
{
    static const uint64_t kNearLe[4] = {0x0000000008000000ull, 0x0000001C1C1C0000ull, 0x00003E3E3E3E3E00ull,
                                        0x007F7F7F7F7F7F7Full};
    constexpr size_t kCap = 128, kMin = 16;
    if (lanes != 16 && lanes != 32 && lanes != 64)
        return -3;
    const Packed P(add, wa, ha);
    const uint32_t max_steps = std::min(max_steps_in, std::max(wa, ha));
    const uint32_t top = std::max(max_steps, 2u) - 1u;
    uint32_t pb = 1;
    while ((2u * top) >> pb)
        ++pb;
    const uint32_t ring_shift = pb + 2;
    uint64_t n_shared = 0, n_own = 0, n_probes = 0, n_goes = 0;

    for (uint32_t w0 = 0; w0 < n; w0 += lanes) {
        const uint32_t nl = std::min(lanes, n - w0);
        LaneQ q[64];
        uint32_t s_end[64];
        int cbx[64], cby[64];
        bool any = false;
        for (uint32_t l = 0; l < nl; ++l) {
            const uint32_t k = w0 + l;
            q[l].qx = (int)cx[k];
            q[l].qy = (int)cy[k];
            make_box(what + 3 * k, prec, q[l].lo, q[l].hi);
            q[l].best = kTileNoKey;
            q[l].limit = 0;
            s_end[l] = 0;
            cbx[l] = (int)(cx[k] >> 3);
            cby[l] = (int)(cy[k] >> 3);
            if (cx[k] < wa && cy[k] < ha && max_steps > 0) { // else: the search ends at once (helpers.rs:49-51)
                const uint32_t cell = P.T[((size_t)cby[l] * P.bw + cbx[l]) * 64 + ((cy[k] & 7) << 3) + (cx[k] & 7)];
                const uint32_t smax = std::max(std::max(cx[k], wa - 1 - cx[k]), std::max(cy[k], ha - 1 - cy[k]));
                s_end[l] = std::min(max_steps, smax + 1);
                if (cell_in_box(cell, q[l].lo, q[l].hi))
                    q[l].best = 0;
                else if (s_end[l] > 1)
                    q[l].limit = s_end[l];
            }
            any |= q[l].limit != 0;
        }
        auto bounds_mask = [&](int l, int bx0, int by0) { // the 64 bound tests of a window
            uint64_t mm = 0;
            for (int row = 0; row < 8; ++row)
                for (int col = 0; col < 8; ++col) {
                    const U2 bb = P.BBp[(size_t)(by0 + kPadB + row) * P.bwp + (size_t)(bx0 + kPadB + col)];
                    if (box_meets(bb.x, bb.y, q[l].lo, q[l].hi))
                        mm |= 1ull << (row * 8 + col);
                }
            return mm;
        };
        struct Task {
            int lane, bx, by;
        };
        std::vector<Task> pool;
        int fault = 0;
        auto flush = [&]() {
            for (size_t b = 0; b < pool.size(); b += 64) {
                ++n_goes;
                const size_t cnt = std::min<size_t>(64, pool.size() - b);
                LaneQ res[64];
                for (size_t i = 0; i < cnt; ++i) {
                    const Task &t = pool[b + i];
                    LaneQ tq = q[t.lane];
                    const int x0 = t.bx * 8 - tq.qx, y0 = t.by * 8 - tq.qy;
                    if ((uint32_t)square_dmin(x0, y0, 8) < tq.limit) {
                        if (t.bx < 0 || t.by < 0 || t.bx >= (int)P.bw || t.by >= (int)P.bh) {
                            fault = -2; // a padded block got through
                            return;
                        }
                        const size_t boff = ((size_t)t.by * P.bw + t.bx) * 64;
                        probe_block(tq, x0, y0, pb, reinterpret_cast<const uint8_t *>(P.T.data()), reinterpret_cast<const uint8_t *>(P.TT.data()),
                                    reinterpret_cast<const uint8_t *>(P.LB.data()), (uint32_t)(boff * 4));
                        ++n_probes;
                    }
                    res[i] = tq;
                }
                for (size_t i = 0; i < cnt; ++i) {
                    LaneQ &o = q[pool[b + i].lane];
                    o.best = std::min(o.best, res[i].best);
                    o.limit = std::min(o.limit, res[i].limit);
                }
            }
            pool.clear();
        };
        if (any) {
            const int R = (int)max_steps - 1;
            int sx0 = 0, sy0 = 0, mxc = 0, myc = 0, ring = 0, rt = -1;
            uint32_t tol = 0;
            std::vector<uint32_t> skeys;
            for (int mode = 0;;) {
                uint64_t cand[64] = {0};
                int ox[64] = {0}, oy[64] = {0};
                int rounds = 1;
                if (mode == 0) {
                    for (uint32_t l = 0; l < nl; ++l) {
                        if (q[l].limit == 0)
                            continue;
                        if (cby[l] - 3 + kPadB < 0 || cbx[l] - 3 + kPadB < 0 || cby[l] + 4 + kPadB >= (int)P.bhp + 1 ||
                            cbx[l] + 4 + kPadB >= (int)P.bwp + 1)
                            return -4; // the window would leave the padded bounds
                        cand[l] = bounds_mask((int)l, cbx[l] - 3, cby[l] - 3) & kNearLe[3];
                        ox[l] = cbx[l] - 3;
                        oy[l] = cby[l] - 3;
                    }
                    rounds = 4;
                } else if (mode == 1) {
                    if (skeys.empty()) {
                        mode = 3;
                        continue;
                    }
                    const auto itmin = std::min_element(skeys.begin(), skeys.end());
                    const uint32_t pick = *itmin;
                    uint32_t maxlimit = 0;
                    for (uint32_t l = 0; l < nl; ++l)
                        maxlimit = std::max(maxlimit, q[l].limit);
                    if ((pick >> 8) >= maxlimit + tol) {
                        mode = 3;
                        continue;
                    }
                    skeys.erase(itmin);
                    const int sx = sx0 + (int)(pick & 15u), sy = sy0 + (int)((pick >> 4) & 15u);
                    const U2 sb = P.SB[(size_t)sy * P.sbw + sx];
                    bool anyw = false;
                    for (uint32_t l = 0; l < nl; ++l) {
                        const uint32_t dS = (uint32_t)square_dmin((sx << 6) - q[l].qx, (sy << 6) - q[l].qy, 64);
                        if (!(dS < q[l].limit && box_meets(sb.x, sb.y, q[l].lo, q[l].hi)))
                            continue;
                        anyw = true;
                        const int bx0 = sx * 8, by0 = sy * 8;
                        cand[l] = bounds_mask((int)l, bx0, by0) & ~window_le_mask(bx0, by0, cbx[l], cby[l], 3);
                        ox[l] = bx0;
                        oy[l] = by0;
                    }
                    if (!anyw)
                        continue;
                } else if (mode == 2) {
                    ++rt;
                    const int n_in_ring = ring == 0 ? 1 : 8 * ring;
                    if (rt >= n_in_ring) {
                        rt = 0;
                        ++ring;
                    }
                    const int lb = ring == 0 ? 0 : 64 * (ring - 1) + 1;
                    bool go = false;
                    for (uint32_t l = 0; l < nl; ++l)
                        go |= q[l].limit != 0 && lb < (int)q[l].limit;
                    if (!go) {
                        mode = 3;
                        continue;
                    }
                    int dx = 0, dy = 0;
                    if (ring > 0) {
                        const int side = rt / (2 * ring), o = rt - side * 2 * ring;
                        dx = side == 0 ? -ring + o : side == 1 ? ring : side == 2 ? ring - o : -ring;
                        dy = side == 0 ? -ring : side == 1 ? -ring + o : side == 2 ? ring : ring - o;
                    }
                    for (uint32_t l = 0; l < nl; ++l) {
                        const int lx0 = std::max(q[l].qx - R, 0) >> 6, ly0 = std::max(q[l].qy - R, 0) >> 6;
                        const int lx1 = std::min(q[l].qx + R, (int)wa - 1) >> 6, ly1 = std::min(q[l].qy + R, (int)ha - 1) >> 6;
                        const int sx = (q[l].qx >> 6) + dx, sy = (q[l].qy >> 6) + dy;
                        if (!(q[l].limit != 0 && sx >= lx0 && sx <= lx1 && sy >= ly0 && sy <= ly1))
                            continue;
                        const U2 sb = P.SB[(size_t)sy * P.sbw + sx];
                        const uint32_t dS = (uint32_t)square_dmin((sx << 6) - q[l].qx, (sy << 6) - q[l].qy, 64);
                        if (!(dS < q[l].limit && box_meets(sb.x, sb.y, q[l].lo, q[l].hi)))
                            continue;
                        const int bx0 = sx * 8, by0 = sy * 8;
                        cand[l] = bounds_mask((int)l, bx0, by0) & ~window_le_mask(bx0, by0, cbx[l], cby[l], 3);
                        ox[l] = bx0;
                        oy[l] = by0;
                    }
                }
                for (int ps = 0; ps < rounds; ++ps) {
                    uint64_t cur[64];
                    for (uint32_t l = 0; l < nl; ++l) {
                        cur[l] = cand[l];
                        if (mode == 0) {
                            cur[l] = cand[l] & kNearLe[ps] & ~(ps ? kNearLe[ps - 1] : 0ull);
                            if (8 * (ps - 1) + 1 >= (int)q[l].limit)
                                cur[l] = 0;
                        }
                    }
                    const bool drain = mode == 3 || (mode == 0 && ps == 3);
                    for (;;) {
                        bool more = false;
                        for (uint32_t l = 0; l < nl; ++l)
                            more |= cur[l] != 0;
                        if (more && pool.size() <= kCap - 64) {
                            for (uint32_t l = 0; l < nl; ++l) {
                                if (!cur[l])
                                    continue;
                                const int k = __builtin_ctzll(cur[l]);
                                cur[l] &= cur[l] - 1;
                                const int bx = ox[l] + (k & 7), by = oy[l] + (k >> 3);
                                if ((uint32_t)square_dmin(bx * 8 - q[l].qx, by * 8 - q[l].qy, 8) < q[l].limit)
                                    pool.push_back(Task{(int)l, bx, by});
                            }
                            continue;
                        }
                        if (pool.size() > kCap - 64 || (!more && (pool.size() >= kMin || (drain && !pool.empty())))) {
                            flush();
                            if (fault)
                                return fault;
                        }
                        if (!more)
                            break;
                    }
                }
                if (mode == 3)
                    break;
                if (mode == 0) {
                    bool any_far = false;
                    int wx_lo = 0x7fffffff, wy_lo = 0x7fffffff, wx_hi = 0, wy_hi = 0;
                    for (uint32_t l = 0; l < nl; ++l) {
                        if (q[l].limit != 0 &&
                            (s_end[l] <= 25 || (q[l].best != kTileNoKey && (q[l].best >> ring_shift) <= 24)))
                            q[l].limit = 0;
                        if (q[l].limit != 0) {
                            any_far = true;
                            wx_lo = std::min(wx_lo, q[l].qx), wy_lo = std::min(wy_lo, q[l].qy);
                            wx_hi = std::max(wx_hi, q[l].qx), wy_hi = std::max(wy_hi, q[l].qy);
                        }
                    }
                    if (!any_far)
                        break;
                    sx0 = std::max(wx_lo - R, 0) >> 6;
                    sy0 = std::max(wy_lo - R, 0) >> 6;
                    const int sx1 = std::min(wx_hi + R, (int)wa - 1) >> 6, sy1 = std::min(wy_hi + R, (int)ha - 1) >> 6;
                    const int nsx = sx1 - sx0 + 1, nsy = sy1 - sy0 + 1;
                    if (nsx <= 10 && nsy <= 12) {
                        ++n_shared;
                        mode = 1;
                        mxc = (wx_lo + wx_hi) >> 1;
                        myc = (wy_lo + wy_hi) >> 1;
                        tol = (uint32_t)std::max(wx_hi - mxc, wy_hi - myc) + 1u;
                        for (int ty = 0; ty < nsy; ++ty)
                            for (int tx = 0; tx < nsx; ++tx) {
                                const int sx = sx0 + tx, sy = sy0 + ty;
                                const U2 sb = P.SB[(size_t)sy * P.sbw + sx];
                                bool anyw = false;
                                for (uint32_t l = 0; l < nl; ++l) {
                                    const uint32_t dS = (uint32_t)square_dmin((sx << 6) - q[l].qx, (sy << 6) - q[l].qy, 64);
                                    anyw |= dS < q[l].limit && box_meets(sb.x, sb.y, q[l].lo, q[l].hi);
                                }
                                if (anyw)
                                    skeys.push_back(((uint32_t)square_dmin((sx << 6) - mxc, (sy << 6) - myc, 64) << 8) |
                                                    (uint32_t)(ty * 16 + tx));
                            }
                    } else {
                        ++n_own;
                        mode = 2;
                    }
                }
            }
        }
        for (uint32_t l = 0; l < nl; ++l) {
            const uint32_t k = w0 + l;
            const bool fnd = q[l].best != kTileNoKey;
            int ddx = 0, ddy = 0;
            uint32_t d = 0;
            if (fnd && q[l].best != 0) {
                const int s = (int)(q[l].best >> (pb + 2)), side = (int)((q[l].best >> pb) & 3);
                const int off = (int)(q[l].best & ((1u << pb) - 1)) - s;
                ddx = side < 2 ? off : side == 2 ? s : -s;
                ddy = side == 0 ? s : side == 1 ? -s : off;
                d = isqrt_floor((uint32_t)(ddx * ddx + ddy * ddy));
            }
            d_out[k] = d;
            found_out[k] = fnd ? 1 : 0;
            pos_out[2 * k] = fnd ? (uint32_t)(q[l].qx + ddx) : 0;
            pos_out[2 * k + 1] = fnd ? (uint32_t)(q[l].qy + ddy) : 0;
        }
    }
    if (stats) {
        stats[0] = n_shared;
        stats[1] = n_own;
        stats[2] = n_probes;
        stats[3] = n_goes;
    }
    return 0;
}
