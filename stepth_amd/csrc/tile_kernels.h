// tile_kernels.h -- gfx950 dense ring search, "lane = query" (round 2).
//
// Path replaced: PASS 1 of DepthImage::load_depth_from_additional with one group per pixel,
// /root/reference/src/depth_image.rs:112-123, and helpers::distance_dot_array,
// /root/reference/src/helpers.rs:9-54.  One launch does ring 0, the search, the distance and
// the per-frame max: no pending lists, no second pass over the queries.
//
// Why this shape.  The round-1 kernel gave one WAVE to the <= 4 queries that share a search
// centre and spread the 64 blocks of a step over the lanes; it retired 123 VALU + 85 SALU
// wave-instructions per pixel and was VALU-issue bound (a wave64 integer VALU instruction
// holds its SIMD for ~3.9 cycles whatever the occupancy: tools/issue_microbench.hip,
// profiles/r2_issue_microbench.txt).  Here one LANE owns one query:
//
//   workgroup = one 8x8 block of search centres of one frame = 16x16 main-image pixels
//               (the four pixels 2cx-1..2cx x 2cy-1..2cy share the centre (cx,cy),
//               depth_image.rs:114-117 with a 1x1 group); wave = 4x4 centres x 4 pixels (a
//               square patch: its lanes look for similar colours in similar places; 8x2
//               centres measured 3 % slower on the bench frames).
//   All 64 queries of a wave have their centre in the same 8x8-cell block of the additional
//   image, so the wave walks ONE sequence of blocks: their bounds arrive by scalar loads
//   (s_load_dwordx16 = a row of 8 blocks) and each lane tests them against its own colour box
//   -- 4 VALU instructions per block for 64 queries, against ~70 per 64 blocks for <= 4
//   queries before.  Blocks that pass are probed by the lane itself, a line of 8 cells at a
//   time, nearest line first (tile_lane.h), with an early exit at the first hit.
//
//   step 0   the 7x7 blocks around the centre block (block-rings 0..3 = every cell of rings
//            <= 24), probed block-ring by block-ring.  A query whose best ring is <= 24 is
//            settled: every nearer cell was seen or excluded by bounds.
//   survey   the others test the bounds of the 64x64-cell superblocks overlapping their
//            window; the superblocks some lane wants go into lane slots keyed by distance.
//   step k   nearest superblock first: its 64 block bounds (minus the 7x7 already done), the
//            passing blocks probed block-ring by block-ring.  Every probe tightens the lane's
//            `limit` (ring of its best key + 1); blocks and superblocks whose nearest cell is
//            at ring >= limit are skipped -- the result is the exact first hit of the
//            reference's scan order (tile_lane.h), bit for bit.
//
// Bounds are stored with a border of empty (min = kNoCell: never matching) entries, so the block
// windows need no edge cases.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ring_kernels.h"
#include "tile_lane.h"

namespace stepth {

constexpr int kPadB = 4; // empty blocks left of / above the image in BBp
// padded bounds layouts (host and device agree through these)
STEPTH_HD uint32_t tile_bwp(uint32_t bw) { return ((bw + 7u) & ~7u) + 8u; }
STEPTH_HD uint32_t tile_bhp(uint32_t bh) { return ((bh + 7u) & ~7u) + 8u; }
STEPTH_HD uint32_t tile_sbwp(uint32_t sbw) { return sbw + 16u; } // rows are read 10 entries at a time

struct TileArgs {
    const uint32_t *T, *TT; // cells, 8x8 blocks: row-major / column-major inside a block
    const uint2 *LB;        // per block: {max,min} of its 8 rows, then of its 8 columns
    const uint2 *BBp;       // block bounds {max,min}, [bhp][bwp], image block (0,0) at (kPadB,kPadB)
    const uint2 *SBp;       // superblock bounds, [sbh][sbwp]
    uint32_t wa, ha, bw, bh, bwp, sbwp;
    uint32_t t_stride, bb_stride, sb_stride; // per frame: dwords of T/TT, uint2 of BBp / SBp
    const uint8_t *main_px[kMaxBatch];
    size_t mstride;
    uint32_t mch, w, row0, row1;
    RowPattern rp;
    uint32_t cbx_n, cby0, cby_n, n_rows; // centre blocks of the strip; rows of them over all frames
    uint32_t max_steps, pos_bits;
    uint8_t prec[3];
    uint8_t *raw; // (d as u8) per pixel of the strip, depth_image.rs:121
    uint32_t raw_stride;
    uint32_t *max_word; // per frame, depth_image.rs:124
    unsigned long long *stats;
    // optional (heavy-first launch order, tile_cost_kernel below): workgroup blockIdx.x does the
    // work of workgroup order[blockIdx.x]
    const uint32_t *order;
};

__device__ __forceinline__ uint32_t rfl(uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
}

// Half a row of block bounds (4 x {max, min} = 32 bytes) into scalar registers.  Written out
// as the instruction: the compiler selects s_load for a plain load only while it can prove the
// table unmodified since kernel entry, and loses that proof to a store, a fence or a second
// code path elsewhere in the kernel.  The wait is part of it -- the compiler's
// own counter bookkeeping does not see the load.
typedef uint64_t u64x4 __attribute__((ext_vector_type(4)));
template <int BYTE_OFFSET>
__device__ __forceinline__ u64x4 sload_half_row(const void *p)
{
    u64x4 v;
    asm("s_load_dwordx8 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=&s"(v) : "s"(p), "n"(BYTE_OFFSET));
    return v;
}

// Four block-bound tests (one half of a row of 8 blocks), highest column first, shifted into
// `acc` (bit = column): four VALU instructions per block (tile_lane.h bound_bit), the bounds
// straight from the scalar registers the s_load left them in.  b0 = the lowest column.
__device__ __forceinline__ uint32_t test_half_row(uint32_t acc, uint64_t b0, uint64_t b1, uint64_t b2,
                                                  uint64_t b3, uint64_t k)
{
    acc = bound_bit(acc, b3, k);
    acc = bound_bit(acc, b2, k);
    acc = bound_bit(acc, b1, k);
    return bound_bit(acc, b0, k);
}

// wave-wide unsigned max in 4 DPP steps + 4 readlanes (see wave_min_dpp)
__device__ __forceinline__ uint32_t wave_max_dpp(uint32_t v)
{
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0xB1, 0xF, 0xF, false));
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x4E, 0xF, 0xF, false));
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x124, 0xF, 0xF, false));
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x128, 0xF, 0xF, false));
    return max(max(bcast(v, 0), bcast(v, 16)), max(bcast(v, 32), bcast(v, 48)));
}

// STEPTH_TILE_PROFILE (a tuning build, never the shipped one): the STATS variant reports
// s_memtime ticks per phase in place of the step counts -- [2] bound tests, [3] survey,
// [4] probes, [5] whole wave.
#ifdef STEPTH_TILE_PROFILE
#define STEPTH_TICK() (STATS ? __builtin_amdgcn_s_memtime() : 0ull)
#else
#define STEPTH_TICK() 0ull
#endif

// POOL variant (round 3; natural images, STEPTH_FLAG_HEAVY_FIRST).  The bound tests of a window
// tell every lane which blocks ITS query must look into; in the plain variant each lane then
// probes its own blocks, and the wave sits through a ~300-instruction probe whenever ANY lane has
// one.  On the synthetic frames 46 of 64 lanes have, on natural images 1-2 -- the slowest waves
// ran ~190 probes back to back, a 0.6 ms chain of dependent loads in a 1.2 ms kernel.  With POOL a
// candidate (query, block) pair is a TASK in a wave-level pool in LDS; tasks are executed 64 at a
// time by whichever lanes, against the query's box and limit kept in LDS, and merge into its best
// key by minimum.  The pool is emptied once it holds kPoolFlushMin tasks (waiting longer lets
// limits go stale and queues probes a hit has made useless), before it could overflow, before
// step 0's results are judged and at the end of the search.  Results are unchanged: a block is
// skipped only if its nearest cell lies at or beyond the query's limit of that moment.  Probe
// goes per wave: 2.9 -> 1.5 on the reference's pair, 4.1 -> 3.5 on the synthetic frames -- where
// queueing (~25 instructions per round) costs more than the goes it saves: + 10 % time, hence a
// variant and not a replacement.
constexpr uint32_t kPoolCap = 128, kPoolFlushMin = 16;

// <= 80 SGPRs keep 8 workgroups of 256 threads resident per CU (MI355X_MICROARCH.md,
// "Residency": 800 / (sgprs + 16)); the kernel wants 106, the spills to vector lanes cost less
// than the eighth wave per SIMD brings (1.28 -> 1.22 ms; 96: 1.23, 64: 1.24)
template <bool STATS, bool POOL>
__device__ __forceinline__ void ring_tile_body(const TileArgs &a, const uint32_t *order)
{
    // POOL: per wave the queries as the probe tasks see them, and the task pool
    __shared__ uint2 s_box[1][POOL ? 64 : 1];           // colour box {lo, hi}
    __shared__ uint32_t s_best[1][POOL ? 64 : 1];       // smallest key so far (merged by minimum)
    __shared__ uint32_t s_lim[1][POOL ? 64 : 1];        // its limit
    __shared__ uint32_t s_task[1][POOL ? kPoolCap : 1]; // query lane | block x << 6 | block y << 19
    [[maybe_unused]] unsigned long long pc_tests = 0, pc_survey = 0, pc_probe = 0;
    [[maybe_unused]] ProbeCounters pcnt{};
    [[maybe_unused]] const unsigned long long pc_t0 = STEPTH_TICK();
    // One wave per workgroup (the four waves of a tile share nothing but its position): block
    // x = xcd + 8 * (wave + 4 * tile-within-XCD), so a tile's waves stay on one XCD and follow each
    // other in the dispatch order.
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wv = (blockIdx.x >> 3) & 3u;
    constexpr uint32_t lw = 0; // the wave's row of the LDS arrays
    // Workgroup ids go round-robin over the 8 XCDs.  Rows of centre blocks (the frames of a batch
    // stacked) are dealt to the XCDs in turn: every XCD gets an eighth of every frame and of
    // every part of it -- frames and image regions differ a lot in work, and an XCD that owned
    // one frame (or one band of a single frame) finished well before or after the others
    // (1.16 -> 1.02 ms on the bench frames).  The workgroups an XCD runs at one time are still
    // neighbours in a row and read the same cells through its L2.
    const uint32_t tb = (blockIdx.x & 7u) + 8u * (blockIdx.x >> 5); // the tile's place in the launch order
    const uint32_t bid = order ? order[tb] : tb;
    const uint32_t xcd = bid & 7u, xi = bid >> 3;
    const uint32_t g = xcd + 8u * (xi / a.cbx_n); // row of centre blocks, all frames stacked
    if (g >= a.n_rows)
        return;
    const uint32_t f = g / a.cby_n;
    const int CBY = (int)(a.cby0 + g % a.cby_n), CBX = (int)(xi % a.cbx_n);
    const uint32_t tf = f * a.t_stride;
    const uint2 *BBf = a.BBp + (size_t)f * a.bb_stride;
    const uint2 *SBf = a.SBp + (size_t)f * a.sb_stride;
    const uint32_t pb = a.pos_bits, ring_shift = pb + 2u;

    // ---- the lane's query ----
    const uint32_t c = lane >> 2, m = lane & 3u;
    const uint32_t cx = (uint32_t)CBX * 8u + (wv & 1u) * 4u + (c & 3u), cy = (uint32_t)CBY * 8u + (wv >> 1) * 4u + (c >> 2);
    const uint32_t x = 2u * cx - 1u + (m & 1u), y = 2u * cy - 1u + (m >> 1); // may wrap: invalid
    const bool valid = x < a.w && y >= a.row0 && y < a.row1 && a.rp.owns(y);
    LaneQ q;
    q.qx = (int)cx;
    q.qy = (int)cy;
    q.lo = q.hi = 0u;
    q.best = kTileNoKey;
    q.limit = 0u;
    uint32_t s_end = 0;
    unsigned long long n_ring0 = 0, n_tests = 0, n_probed = 0, n_near = 0, n_survey = 0, n_super = 0;
    unsigned long long n_super_empty = 0, n_probe_void = 0; // steps with no candidate block / probes without gain
    if (valid) {
        const uint8_t *p = a.main_px[f] + (size_t)y * a.mstride + (size_t)x * a.mch;
        const uint8_t v[3] = {p[0], p[1], p[2]};
        make_box(v, a.prec, q.lo, q.hi);
        if (cx < a.wa && cy < a.ha && a.max_steps > 0) {
            // ring 0: the centre cell (helpers.rs:26-35 with s = 0)
            const uint32_t cell = a.T[tf + ((uint32_t)CBY * a.bw + (uint32_t)CBX) * 64u +
                                      ((cy & 7u) << 3) + (cx & 7u)];
            const uint32_t smax = max(max(cx, a.wa - 1u - cx), max(cy, a.ha - 1u - cy));
            s_end = min(a.max_steps, smax + 1u); // rings 0..s_end-1 hold in-bounds cells
            if (cell_in_box(cell, q.lo, q.hi))
                q.best = 0u; // ring 0, distance 0
            else if (s_end > 1u)
                q.limit = s_end;
            if constexpr (STATS)
                n_ring0 = q.best == 0u;
        }
    }

#if defined(STEPTH_ABLATE) && STEPTH_ABLATE == 3 // tuning build: ring 0 only
    q.limit = 0u;
#endif
    [[maybe_unused]] uint32_t n_pool = 0; // POOL: tasks in the pool (wave-uniform)
    // POOL: executes the pooled tasks, 64 at a time.  Lane i of a go takes task i: it reads the
    // query (box, best key, limit) from LDS, probes the block like the query's own lane would,
    // and merges an improvement back by minimum -- the tasks of one go see the same snapshot of
    // their queries.  The lane's OWN query lives in LDS meanwhile (the registers of the probe are
    // the kernel's peak; this keeps 8 waves per SIMD) and comes back merged afterwards.  No fence
    // or wave barrier: the LDS operations of a wave complete in program order, and either
    // intrinsic would turn the scalar loads of the bound rows into vector loads (the compiler
    // can no longer prove the tables unmodified -- the same reason keeps every global store at
    // the kernel's end).
    [[maybe_unused]] auto flush = [&]() __attribute__((always_inline)) {
        if constexpr (POOL) {
            s_best[lw][lane] = q.best; // what the lane found out about its own query meanwhile
            s_lim[lw][lane] = q.limit;
            for (uint32_t base = 0; base < n_pool; base += 64u) {
                const uint32_t ti = base + lane;
                if (ti < n_pool) {
                    const uint32_t t = s_task[lw][ti], src = t & 63u;
                    const int bx = (int)((t >> 6) & 0x1FFFu), by = (int)(t >> 19);
                    LaneQ tq;
                    const uint2 box = s_box[lw][src];
                    tq.lo = box.x;
                    tq.hi = box.y;
                    tq.best = s_best[lw][src];
                    tq.limit = s_lim[lw][src];
                    tq.qx = CBX * 8 + (int)((wv & 1u) * 4u + ((src >> 2) & 3u));
                    tq.qy = CBY * 8 + (int)((wv >> 1) * 4u + (src >> 4));
                    const int x0 = bx * 8 - tq.qx, y0 = by * 8 - tq.qy;
                    // (the dmin test also keeps stale bounds of unpacked rows from being probed)
                    if ((uint32_t)square_dmin(x0, y0, 8) < tq.limit) {
                        const uint32_t boff = tf + ((uint32_t)by * a.bw + (uint32_t)bx) * 64u;
                        const uint32_t best_before = tq.best;
                        const uint32_t nl = probe_block(tq, x0, y0, pb, reinterpret_cast<const uint8_t *>(a.T),
                                                        reinterpret_cast<const uint8_t *>(a.TT),
                                                        reinterpret_cast<const uint8_t *>(a.LB), boff << 2, &pcnt);
                        if (tq.best < best_before) {
                            atomicMin(&s_best[lw][src], tq.best);
                            atomicMin(&s_lim[lw][src], tq.limit);
                        }
                        if constexpr (STATS) {
                            ++n_probed;
                            n_tests += 8u * nl;
                            n_probe_void += tq.best == best_before;
                        }
                    }
                }
            }
            n_pool = 0u;
            uint32_t l2 = lane;
            asm("" : "+v"(l2)); // (opaque: keeps the compiler from carrying the values across instead)
            const uint2 box = s_box[lw][l2];
            q.lo = box.x;
            q.hi = box.y;
            q.best = s_best[lw][l2];
            q.limit = s_lim[lw][l2];
            q.qx = CBX * 8 + (int)((wv & 1u) * 4u + ((l2 >> 2) & 3u));
            q.qy = CBY * 8 + (int)((wv >> 1) * 4u + (l2 >> 4));
        }
    };
    if (__builtin_amdgcn_ballot_w64(q.limit != 0u)) {
        if constexpr (POOL)
            s_box[lw][lane] = make_uint2(q.lo, q.hi);
        // superblock window of the wave (far phase): every lane's window of radius
        // max_steps-1 lies inside it
        const int R = (int)a.max_steps - 1;
        const int wx0 = CBX * 8 + (int)(wv & 1u) * 4, wy0 = CBY * 8 + (int)(wv >> 1) * 4;
        const int sx0 = max(wx0 - R, 0) >> 6, sy0 = max(wy0 - R, 0) >> 6;
        const int sx1 = min(wx0 + 3 + R, (int)a.wa - 1) >> 6, sy1 = min(wy0 + 3 + R, (int)a.ha - 1) >> 6;
        const int nsx = sx1 - sx0 + 1, nsy = sy1 - sy0 + 1; // <= 10 each for max_steps <= 258
        uint32_t skey0 = kTileNoKey, skey1 = kTileNoKey, skey2 = kTileNoKey; // slot ty*16+tx

        [[maybe_unused]] bool done = false; // POOL: the search is over but for the tasks still pooled
        for (uint32_t it = 0;; ++it) {
            if constexpr (POOL) {
                // what is still pooled: before step 0's results are judged, and when the search ends
                if (n_pool != 0u && (it == 1u || done))
                    flush();
                if (done)
                    break;
            }
            int bx0, by0, r_lo, r_hi;
            uint64_t allow;
            bool want;
            if (it == 0) {
                // step 0: block-rings 0..3 around the centre block
                bx0 = CBX - 3;
                by0 = CBY - 3;
                allow = 0x007F7F7F7F7F7F7Full; // 7 rows x 7 columns
                r_lo = 0;
                r_hi = 3;
                want = q.limit != 0u;
                if constexpr (STATS)
                    ++n_near;
            } else {
#if defined(STEPTH_ABLATE) && STEPTH_ABLATE == 1 // tuning build: step 0 only
                break;
#endif
                if (it == 1) {
                    // settled: best ring <= 24, or nothing beyond ring 24 to search
                    if (q.limit != 0u &&
                        (s_end <= 25u || (q.best != kTileNoKey && (q.best >> ring_shift) <= 24u)))
                        q.limit = 0u;
                    if (!__builtin_amdgcn_ballot_w64(q.limit != 0u))
                        break;
                    // survey: superblocks whose bounds meet some lane's box within its reach
                    if constexpr (STATS)
                        ++n_survey;
                    [[maybe_unused]] const unsigned long long pc_s0 = STEPTH_TICK();
                    for (int ty = 0; ty < nsy; ++ty) {
                        const int sy = sy0 + ty;
                        // a row of the window in one go: up to 10 superblocks = 20 dwords
                        // a row of the window in one go: lane i holds dword i of its <= 10 {max, min}
                        // pairs, the loop reads them back lane by lane (an indexed array of scalar
                        // registers would be copied whole for every use)
                        const uint32_t row_dw = reinterpret_cast<const uint32_t *>(SBf + (size_t)sy * a.sbwp + sx0)[lane & 31u];
                        const int ay = (sy << 6) - q.qy;
                        const int dsy = max(max(ay, -63 - ay), 0);
                        for (int tx = 0; tx < nsx; ++tx) {
                            const uint32_t bmax = (uint32_t)__builtin_amdgcn_readlane((int)row_dw, 2 * tx);
                            const uint32_t bmin = (uint32_t)__builtin_amdgcn_readlane((int)row_dw, 2 * tx + 1);
                            const int ax = ((sx0 + tx) << 6) - q.qx;
                            const uint32_t dS = (uint32_t)max(max(max(ax, -63 - ax), 0), dsy);
                            const bool w = dS < q.limit && box_meets(bmax, bmin, q.lo, q.hi);
                            if constexpr (STATS)
                                n_tests += q.limit != 0u;
                            if (__builtin_amdgcn_ballot_w64(w)) {
                                // distance from the middle of the wave's centres orders the visits
                                const int rx = ((sx0 + tx) << 6) - (wx0 + 2), ry = (sy << 6) - (wy0 + 2);
                                const uint32_t dref = (uint32_t)max(max(max(rx, -63 - rx), max(ry, -63 - ry)), 0);
                                const uint32_t t = (uint32_t)(ty * 16 + tx);
                                const uint32_t key = (dref << 8) | t;
                                if (lane == (t & 63u)) {
                                    if (t < 64u)
                                        skey0 = key;
                                    else if (t < 128u)
                                        skey1 = key;
                                    else
                                        skey2 = key;
                                }
                            }
                        }
                    }
                    pc_survey += STEPTH_TICK() - pc_s0;
                }
                // nearest superblock some lane may still want.  Slots come in the order of their
                // distance from the middle of the wave's centres, which is within 2 rings of
                // every lane's own distance: past the largest limit + 2 nothing is left.
                const uint32_t pick = wave_min_dpp(min(min(skey0, skey1), skey2));
                if (pick == kTileNoKey || (pick >> 8) >= wave_max_dpp(q.limit) + 2u) {
                    if constexpr (POOL) {
                        done = true; // (limits only shrink when the pooled tasks run: the search stays over)
                        continue;
                    }
                    break;
                }
                skey0 = skey0 == pick ? kTileNoKey : skey0;
                skey1 = skey1 == pick ? kTileNoKey : skey1;
                skey2 = skey2 == pick ? kTileNoKey : skey2;
                const int sx = sx0 + (int)(pick & 15u), sy = sy0 + (int)((pick >> 4) & 15u);
                const uint2 sb = SBf[(size_t)sy * a.sbwp + sx];
                const int ax = (sx << 6) - q.qx, ay = (sy << 6) - q.qy;
                const uint32_t dS = (uint32_t)max(max(max(ax, -63 - ax), max(ay, -63 - ay)), 0);
                want = dS < q.limit && box_meets(sb.x, sb.y, q.lo, q.hi);
                if (!__builtin_amdgcn_ballot_w64(want))
                    continue;
                bx0 = sx * 8;
                by0 = sy * 8;
                allow = ~window_le_mask(bx0, by0, CBX, CBY, 3); // step 0 did block-rings 0..3
                window_ring_range(bx0, by0, CBX, CBY, r_lo, r_hi);
                r_lo = max(r_lo, 4);
                if constexpr (STATS)
                    ++n_super;
            }

            // ---- bounds of the window's 8x8 blocks against the lane's box ----
            // Rows and half rows of blocks that lie beyond every lane's limit are skipped
            // (wave-uniform: distances from the wave's 4x4 centres).
            uint32_t m_lo = 0, m_hi = 0; // bit row*8+col
            [[maybe_unused]] const unsigned long long pc_b0 = STEPTH_TICK();
            const uint32_t wlim = wave_max_dpp(want ? q.limit : 0u);
            {
                const int cx_l = bx0 * 8 - (wx0 + 3), cx_r = wx0 - (bx0 * 8 + 31); // cols 0..3
                const bool need_l = max(max(cx_l, cx_r), 0) < (int)wlim;
                const bool need_h = max(max(cx_l + 32, cx_r - 32), 0) < (int)wlim;   // cols 4..7
                const uint2 *win = BBf + (size_t)(by0 + kPadB) * a.bwp + (size_t)(bx0 + kPadB);
                const uint64_t kq = ((uint64_t)q.hi << 32) | q.lo;
                auto do_row = [&](uint32_t acc, int row) __attribute__((always_inline)) -> uint32_t {
                    const int ry0 = (by0 + row) * 8 - (wy0 + 3), ry1 = wy0 - ((by0 + row) * 8 + 7);
                    if (max(max(ry0, ry1), 0) >= (int)wlim || (it == 0 && row == 7))
                        return acc << 8;
                    const ulonglong4 *rp4 = reinterpret_cast<const ulonglong4 *>(win + (size_t)row * a.bwp);
                    if (need_h) {
                        const u64x4 v1 = sload_half_row<32>(rp4);
                        acc = test_half_row(acc, v1.x, v1.y, v1.z, v1.w, kq);
                    } else {
                        acc <<= 4;
                    }
                    if (need_l) {
                        const u64x4 v0 = sload_half_row<0>(rp4);
                        acc = test_half_row(acc, v0.x, v0.y, v0.z, v0.w, kq);
                    } else {
                        acc <<= 4;
                    }
                    return acc;
                };
#pragma unroll
                for (int row = 7; row >= 4; --row)
                    m_hi = do_row(m_hi, row);
#pragma unroll
                for (int row = 3; row >= 0; --row)
                    m_lo = do_row(m_lo, row);
            }
            if constexpr (STATS)
                n_tests += want ? 64u : 0u;
            uint64_t cand = want ? (((uint64_t)m_hi << 32) | m_lo) & allow : 0ull;
            if constexpr (STATS)
                n_super_empty += it != 0 && !__builtin_amdgcn_ballot_w64(cand != 0ull);
            [[maybe_unused]] const unsigned long long pc_p0 = STEPTH_TICK();
            pc_tests += pc_p0 - pc_b0;

            // ---- probe the passing blocks, block-ring by block-ring (nearest first) ----
            uint64_t le_prev = r_lo > 0 ? window_le_mask(bx0, by0, CBX, CBY, r_lo - 1) : 0ull;
            for (int r = r_lo; r <= r_hi; ++r) {
                const uint64_t le = window_le_mask(bx0, by0, CBX, CBY, r);
                uint64_t cur = cand & le & ~le_prev;
                le_prev = le;
                // a block of block-ring r has no cell nearer than ring 8(r-1)+1
                if (8 * (r - 1) + 1 >= (int)wlim)
                    break; // (wlim: the lanes' largest limit when the step began)
                if (8 * (r - 1) + 1 >= (int)q.limit)
                    cand = 0ull; // this lane is through with the window
                if (!__builtin_amdgcn_ballot_w64((cand & ~(le & ~cur)) != 0ull))
                    break; // no lane has a candidate in this or a later block-ring
                if (8 * (r - 1) + 1 >= (int)q.limit)
                    cur = 0ull;
                if (!__builtin_amdgcn_ballot_w64(cur != 0ull))
                    continue;
                if constexpr (POOL) {
                    // the block-ring's candidates become tasks; the pool runs when it holds enough
                    for (;;) {
                        const bool more = __builtin_amdgcn_ballot_w64(cur != 0ull) != 0ull;
                        if (n_pool > kPoolCap - 64u || (!more && n_pool >= kPoolFlushMin))
                            flush();
                        if (!more)
                            break;
                        bool ok = false; // every lane with a candidate left queues its next one
                        uint32_t task = 0;
                        if (cur != 0ull) {
                            const int k = __builtin_ctzll(cur);
                            cur &= cur - 1ull;
                            const int bx = bx0 + (k & 7), by = by0 + (k >> 3);
                            ok = (uint32_t)square_dmin(bx * 8 - q.qx, by * 8 - q.qy, 8) < q.limit;
                            task = lane | ((uint32_t)bx << 6) | ((uint32_t)by << 19);
                        }
                        const unsigned long long okm = __builtin_amdgcn_ballot_w64(ok);
                        if (ok)
                            s_task[lw][n_pool + __builtin_amdgcn_mbcnt_hi((uint32_t)(okm >> 32),
                                                                           __builtin_amdgcn_mbcnt_lo((uint32_t)okm, 0u))] = task;
                        n_pool += (uint32_t)__builtin_popcountll(okm);
                    }
                    continue;
                }
                while (cur != 0ull) {
                    STEPTH_CNT(&pcnt, 0);
                    const int k = __builtin_ctzll(cur);
                    cur &= cur - 1ull;
                    const int bx = bx0 + (k & 7), by = by0 + (k >> 3);
                    const int x0 = bx * 8 - q.qx, y0 = by * 8 - q.qy;
                    // the dmin test also keeps stale bounds of rows this call did not pack
                    // (they are out of every query's reach) from ever being probed
#if defined(STEPTH_ABLATE) && STEPTH_ABLATE == 2 // tuning build: no probes
                    if (false) {
#else
                    if ((uint32_t)square_dmin(x0, y0, 8) < q.limit) {
#endif
                        const uint32_t boff = tf + ((uint32_t)by * a.bw + (uint32_t)bx) * 64u;
                        [[maybe_unused]] const uint32_t best_before = q.best;
                        const uint32_t nl =
                            probe_block(q, x0, y0, pb, reinterpret_cast<const uint8_t *>(a.T),
                                        reinterpret_cast<const uint8_t *>(a.TT),
                                        reinterpret_cast<const uint8_t *>(a.LB), boff << 2, &pcnt);
                        if constexpr (STATS) {
                            ++n_probed;
                            n_tests += 8u * nl;
                            n_probe_void += q.best == best_before;
                        }
                    }
                }
            }
            pc_probe += STEPTH_TICK() - pc_p0;
        }
    }

    // ---- distance, store, max (depth_image.rs:118-124) ----
    uint32_t d8 = 0;
    if (q.best != kTileNoKey && q.best != 0u)
        d8 = key_distance(q.best, pb) & 0xffu; // `d as u8` wraps (depth_image.rs:121)
    if constexpr (POOL) {
        // the lane's pixel, worked out again (not carried through the probes in registers)
        uint32_t l2 = lane;
        asm("" : "+v"(l2));
        const uint32_t c2 = l2 >> 2, m2 = l2 & 3u;
        const uint32_t cx2 = (uint32_t)CBX * 8u + (wv & 1u) * 4u + (c2 & 3u), cy2 = (uint32_t)CBY * 8u + (wv >> 1) * 4u + (c2 >> 2);
        const uint32_t x2 = 2u * cx2 - 1u + (m2 & 1u), y2 = 2u * cy2 - 1u + (m2 >> 1);
        if (x2 < a.w && y2 >= a.row0 && y2 < a.row1 && a.rp.owns(y2))
            a.raw[(size_t)f * a.raw_stride + (size_t)(y2 - a.row0) * a.w + x2] = (uint8_t)d8;
    } else if (valid) {
        a.raw[(size_t)f * a.raw_stride + (size_t)(y - a.row0) * a.w + x] = (uint8_t)d8;
    }
    // Most waves find nothing beyond what is already there: a plain read first keeps the
    // same-address atomics (~11 ns each, chip-wide) off the common path.  No workgroup
    // barrier: a wave that is done leaves, whatever its neighbours still have to search.
    const uint32_t wmax = wave_max_dpp(d8);
    if (lane == 0 && wmax > __hip_atomic_load(a.max_word + f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
        atomicMax(a.max_word + f, wmax);
    if constexpr (STATS) {
        // lane-level counts: ring-0 hits, bound tests + cells examined, blocks probed
        unsigned long long v0 = n_ring0, v1 = n_tests, v5 = n_probed, v7 = n_probe_void;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            v0 += __shfl_xor(v0, o, 64);
            v1 += __shfl_xor(v1, o, 64);
            v5 += __shfl_xor(v5, o, 64);
            v7 += __shfl_xor(v7, o, 64);
        }
        if (lane == 0) {
            atomicAdd(&stats_replica(a.stats)[0], v0);
            atomicAdd(&stats_replica(a.stats)[1], v1);
#if defined(STEPTH_TILE_PROFILE) && STEPTH_TILE_PROFILE == 2
        }
        for (int i = 0; i < 7; ++i) { // raw slots 8.. : lane counts, 16.. : wave counts
            unsigned long long lv = pcnt.lane[i], wv2 = pcnt.wave[i];
            for (int o = 32; o > 0; o >>= 1) {
                lv += __shfl_xor(lv, o, 64);
                wv2 += __shfl_xor(wv2, o, 64);
            }
            if (lane == 0) {
                atomicAdd(&stats_replica(a.stats)[8 + i], lv);
                atomicAdd(&stats_replica(a.stats)[16 + i], wv2);
            }
        }
        if (lane == 0) {
            atomicAdd(&stats_replica(a.stats)[2], n_near);
            atomicAdd(&stats_replica(a.stats)[3], n_survey);
            atomicAdd(&stats_replica(a.stats)[4], n_super);
            atomicAdd(&stats_replica(a.stats)[5], v5);
#elif defined(STEPTH_TILE_PROFILE)
            atomicAdd(&stats_replica(a.stats)[2], pc_tests);
            atomicAdd(&stats_replica(a.stats)[3], pc_survey);
            atomicAdd(&stats_replica(a.stats)[4], pc_probe);
            atomicAdd(&stats_replica(a.stats)[5], STEPTH_TICK() - pc_t0);
#else
            atomicAdd(&stats_replica(a.stats)[2], n_near);
            atomicAdd(&stats_replica(a.stats)[3], n_survey);
            atomicAdd(&stats_replica(a.stats)[4], n_super);
            atomicAdd(&stats_replica(a.stats)[5], v5);
            atomicAdd(&stats_replica(a.stats)[6], n_super_empty); // raw slots (stepth_get_stats_raw)
            atomicAdd(&stats_replica(a.stats)[7], v7);
#endif
        }
    }
}

template <bool STATS, bool POOL = false>
__global__ __launch_bounds__(64) __attribute__((amdgpu_num_sgpr(80))) void ring_tile_kernel(TileArgs a)
{
    ring_tile_body<STATS, POOL>(a, a.order);
}

// STEPTH_FLAG_HEAVY_FIRST: both variants in one kernel, chosen by what tile_scan_kernel found out
// about the frames (`verdict` = its word 16: 1 = no light majority, the order is the identity):
// natural images run the pooled variant in the heavy-first order, frames like the synthetic ones
// -- issue bound, where probe tasks cost 10 % -- the plain variant in image order.  The host
// cannot know which without waiting for the device; a wave-uniform branch costs nothing.
__global__ __launch_bounds__(64) __attribute__((amdgpu_num_sgpr(80))) void ring_tile_auto_kernel(TileArgs a, const uint32_t *verdict)
{
    if (verdict[0])
        ring_tile_body<false, false>(a, nullptr);
    else
        ring_tile_body<false, true>(a, a.order);
}

// ---------------------------------------------------------------------------------------
// Heavy-first launch order for ring_tile_kernel (round 3; STEPTH_FLAG_HEAVY_FIRST).
//
// On natural images most workgroups are done after ring 0 / step 0 while a few percent search far
// (the reference's own pair at 1080p: median workgroup 36 k clock ticks, p99 1.2 M, max 3.6 M):
// dispatched in image order, the kernel ends in a long tail of a few slow workgroups -- VALU 58 %
// busy, 3.6 of 8 waves resident per SIMD.  Started heaviest first, the same workgroups take 0.81
// instead of 1.22 ms (order from measured costs); with the estimate below 0.95-0.97 ms.
//
// tile_cost_kernel   per workgroup of ring_tile_kernel ("tile": 8x8 centres), 16 sample queries
//                    (every other centre): a sample whose box meets none of the 7x7 block bounds
//                    around the centre block cannot be settled by step 0; it weighs in with the
//                    distance of the nearest superblock whose bounds meet its box (in units of
//                    16 rings; nothing in reach = the full reach).  key = sum of the weights
//                    (0..240), class = min(floor(sqrt(key)), 15).  Per group of 64 tiles: the
//                    tiles' classes, their ranks within the class (image order) and the 16 class
//                    sizes -- no atomics, the resulting order is deterministic.
// tile_scan_kernel   exclusive prefix of the class sizes over the groups; tiles of the heavier classes.
//                    If fewer than half of the tiles are of class 0 the frame set has no light
//                    majority to hide behind (synthetic frames: the search is issue bound, not
//                    tail bound, and image order serves the caches better) -> identity order.
// tile_scatter_kernel  order[position] = tile.
// ---------------------------------------------------------------------------------------
struct TileOrderBufs {
    uint8_t *cls, *rank; // per tile
    uint8_t *cnt;        // [groups][16] class sizes
    uint32_t *off;       // [groups][16] position of the group's first tile of the class
    uint32_t *order;     // [n_wg]
    uint32_t n_wg, n_groups;
};

__global__ __launch_bounds__(1024) void tile_cost_kernel(TileArgs a, TileOrderBufs o)
{
    __shared__ uint2 s_win[64][49]; // per tile of the group: the 7x7 block bounds around its centre block
    __shared__ int4 s_tile[64];     // per tile: frame, CBX, CBY, 1 = there is one (the divisions once per tile)
    __shared__ uint8_t s_cls[64];
    const uint32_t s = threadIdx.x & 15u, tl = threadIdx.x >> 4;
    if (threadIdx.x < 64u) {
        // tile t = the tile's place in ring_tile_kernel's launch order
        const uint32_t t = blockIdx.x * 64u + threadIdx.x, xcd = t & 7u, xi = t >> 3, g = xcd + 8u * (xi / a.cbx_n);
        int4 v = make_int4(0, 0, 0, 0);
        if (t < o.n_wg && g < a.n_rows)
            v = make_int4((int)(g / a.cby_n), (int)(xi % a.cbx_n), (int)(a.cby0 + g % a.cby_n), 1);
        s_tile[threadIdx.x] = v;
    }
    __syncthreads();
    // ---- the groups' windows -> LDS (the 16 samples of a tile test the same 49 blocks) ----
    for (uint32_t i = threadIdx.x; i < 64u * 49u; i += 1024u) {
        const uint32_t wt = i / 49u, e = i - wt * 49u, dy = e / 7u, dx = e - dy * 7u;
        const int4 ti = s_tile[wt];
        uint2 b = make_uint2(0u, kNoCell);
        if (ti.w)
            b = a.BBp[(size_t)ti.x * a.bb_stride + (size_t)(ti.z - 3 + (int)dy + kPadB) * a.bwp + (size_t)(ti.y - 3 + (int)dx + kPadB)];
        s_win[wt][e] = b;
    }
    __syncthreads();
    uint32_t w = 0;
    const int4 ti = s_tile[tl];
    if (ti.w) {
        const uint32_t f = (uint32_t)ti.x;
        const int CBX = ti.y, CBY = ti.z;
        const uint32_t cx = (uint32_t)CBX * 8u + 2u * (s & 3u), cy = (uint32_t)CBY * 8u + 2u * (s >> 2);
        const uint32_t x = 2u * cx, y = 2u * cy;
        if (x < a.w && y >= a.row0 && y < a.row1 && a.rp.owns(y) && cx < a.wa && cy < a.ha && a.max_steps > 1u) {
            const uint8_t *p = a.main_px[f] + (size_t)y * a.mstride + (size_t)x * a.mch;
            const uint8_t v[3] = {p[0], p[1], p[2]};
            uint32_t lo, hi;
            make_box(v, a.prec, lo, hi);
            // a full match is the numerically largest value r = (max+lo) & ~(min+hi) & guard can take
            // (tile_lane.h LineTest::any): "some block's bounds meet the box" = max(r) == guard --
            // one 64-bit add, one v_bitop3 and half a v_max3 per block
            const uint64_t kq = ((uint64_t)hi << 32) | lo;
            uint32_t rmax = 0;
#pragma unroll 7
            for (int e = 0; e < 49; ++e) {
                const uint64_t t = (((uint64_t)s_win[tl][e].y << 32) | s_win[tl][e].x) + kq;
                rmax = max(rmax, (uint32_t)t & ~(uint32_t)(t >> 32) & kGuard);
            }
            const bool near = rmax == kGuard;
            if (!near && a.max_steps <= 160u) {
                w = 4u; // a short reach: how far matters little next to how many (measured)
            } else if (!near) {
                const uint2 *SBf = a.SBp + (size_t)f * a.sb_stride;
                // (superblocks up to 159 rings away: whatever meets nothing nearer counts as the
                // full reach -- a 6x6 survey instead of up to 9x9)
                const int R = min((int)a.max_steps - 1, 159), qx = (int)cx, qy = (int)cy;
                const int sx0 = max(qx - R, 0) >> 6, sy0 = max(qy - R, 0) >> 6;
                const int sx1 = min(qx + R, (int)a.wa - 1) >> 6, sy1 = min(qy + R, (int)a.ha - 1) >> 6;
                uint32_t dmin = 255u;
                for (int sy = sy0; sy <= sy1; ++sy)
                    for (int sx = sx0; sx <= sx1; ++sx) {
                        const uint2 sb = SBf[(size_t)sy * a.sbwp + sx];
                        const uint32_t dS = (uint32_t)square_dmin((sx << 6) - qx, (sy << 6) - qy, 64);
                        if (dS <= (uint32_t)R && box_meets(sb.x, sb.y, lo, hi))
                            dmin = min(dmin, dS);
                    }
                w = min(max(dmin, 25u), 255u) >> 4; // 1..15
            }
        }
    }
    // key of the tile: the 16 samples are 16 consecutive lanes
#pragma unroll
    for (int sh = 1; sh < 16; sh <<= 1)
        w += (uint32_t)__shfl_xor((int)w, sh, 64);
    if (s == 0)
        s_cls[tl] = (uint8_t)min(isqrt_floor(w), 15u);
    __syncthreads();
    if (threadIdx.x < 64u) {
        const uint32_t c = s_cls[threadIdx.x], tt = blockIdx.x * 64u + threadIdx.x;
        uint32_t rank = 0, mine = 0;
#pragma unroll
        for (uint32_t k = 0; k < 16u; ++k) {
            const unsigned long long m = __builtin_amdgcn_ballot_w64(c == k && tt < o.n_wg);
            if (c == k)
                rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
            if (threadIdx.x == k)
                mine = (uint32_t)__builtin_popcountll(m);
        }
        if (tt < o.n_wg) {
            o.cls[tt] = (uint8_t)c;
            o.rank[tt] = (uint8_t)rank;
        }
        if (threadIdx.x < 16u)
            o.cnt[(size_t)blockIdx.x * 16u + threadIdx.x] = (uint8_t)mine;
    }
}

// inclusive prefix sum over the 64 lanes of a wave: four steps inside the rows of 16 lanes, two
// row broadcasts (DPP; lanes without a source add 0)
__device__ __forceinline__ uint32_t wave_scan_incl(uint32_t v)
{
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, true); // row_shr:1
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, true); // row_shr:2
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, true); // row_shr:4
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, true); // row_shr:8
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false); // row_bcast:15 -> rows 1, 3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false); // row_bcast:31 -> rows 2, 3
    return v;
}

// One workgroup of 16 waves, wave c = class c: an exclusive prefix of the class's sizes over the
// groups (1024 groups at a time through LDS), then base[c] = tiles of the heavier classes
// (o.off[n_groups * 16 ..]; word 16 = 1: identity order, class 0 is not the majority).
__global__ __launch_bounds__(1024) void tile_scan_kernel(TileOrderBufs o)
{
    __shared__ uint4 s_cnt[1024];        // 16 class sizes (bytes) per group
    __shared__ uint32_t s_off[1024][17]; // the turn's offsets (17: no bank conflicts between the waves' columns)
    __shared__ uint32_t s_total[16];
    const uint32_t lane = threadIdx.x & 63u, c = threadIdx.x >> 6;
    uint32_t run = 0;
    for (uint32_t g0 = 0; g0 < o.n_groups; g0 += 1024u) {
        const uint32_t gl = g0 + threadIdx.x;
        s_cnt[threadIdx.x] = gl < o.n_groups ? reinterpret_cast<const uint4 *>(o.cnt)[gl] : make_uint4(0u, 0u, 0u, 0u);
        __syncthreads();
        for (uint32_t j = 0; j < 16u; ++j) {
            const uint32_t g = j * 64u + lane;
            const uint32_t n = reinterpret_cast<const uint8_t *>(&s_cnt[g])[c];
            const uint32_t inc = wave_scan_incl(n);
            s_off[g][c] = run + inc - n;
            run += bcast(inc, 63);
        }
        __syncthreads();
        if (gl < o.n_groups) { // the group's 16 offsets in one 64-byte piece
            uint4 *dst = reinterpret_cast<uint4 *>(o.off + (size_t)gl * 16u);
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4)
                dst[q4] = make_uint4(s_off[threadIdx.x][4 * q4], s_off[threadIdx.x][4 * q4 + 1], s_off[threadIdx.x][4 * q4 + 2],
                                     s_off[threadIdx.x][4 * q4 + 3]);
        }
        __syncthreads();
    }
    if (lane == 0)
        s_total[c] = run;
    __syncthreads();
    if (threadIdx.x < 17u) {
        uint32_t v = 0;
        if (threadIdx.x < 16u)
            for (uint32_t k = threadIdx.x + 1u; k < 16u; ++k)
                v += s_total[k];
        else
            v = 2u * s_total[0] < o.n_wg ? 1u : 0u;
        o.off[(size_t)o.n_groups * 16u + threadIdx.x] = v;
    }
}

__global__ __launch_bounds__(256) void tile_scatter_kernel(TileOrderBufs o, uint32_t identity_too)
{
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    if (t >= o.n_wg)
        return;
    const uint32_t *base = o.off + (size_t)o.n_groups * 16u;
    if (base[16]) { // identity order: ring_tile_auto_kernel does not read it, the list search does
        if (identity_too)
            o.order[t] = t;
        return;
    }
    const uint32_t c = o.cls[t];
    o.order[base[c] + o.off[(size_t)(t >> 6) * 16u + c] + o.rank[t]] = t;
}

// Heavy-first order for ring_list_kernel: the same estimate per WAVE of 64 sorted queries (the
// waves are what the kernel's tail is made of: 2.2 of 4 wave slots filled on average).  A query
// whose box meets none of the 7x7 block bounds around its centre block weighs 4 (reach <= 160)
// or the distance of the nearest meeting superblock within 159 rings / 16; the wave's key is the
// sum, the classes, scan and scatter are those of the dense search (groups of 64 waves).
struct ListCostArgs {
    const uint2 *BBp, *SBp;
    uint32_t wa, ha, bwp, sbwp, bb_stride, sb_stride;
    const Query *queries;
    uint32_t q_stride, wpf, max_steps;
    uint32_t nq[kMaxBatch];
};

__global__ __launch_bounds__(256) void list_cost_kernel(ListCostArgs a, TileOrderBufs o)
{
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    const uint32_t wid = blockIdx.x * 4u + wv; // one workgroup = 4 waves of the search
    uint32_t w = 0;
    if (wid < o.n_wg) {
        const uint32_t f = wid / a.wpf, i = (wid - f * a.wpf) * 64u + lane;
        if (i < a.nq[f]) {
            const Query q = a.queries[(size_t)f * a.q_stride + i];
            if (q.cx < a.wa && q.cy < a.ha && a.max_steps > 1u) {
                const int cbx = (int)(q.cx >> 3), cby = (int)(q.cy >> 3);
                const uint2 *BBf = a.BBp + (size_t)f * a.bb_stride;
                bool near = false;
                for (int dy = -3; dy <= 3; ++dy) {
                    const uint2 *row = BBf + (size_t)(cby + dy + kPadB) * a.bwp + (size_t)(cbx - 3 + kPadB);
#pragma unroll
                    for (int dx = 0; dx < 7; ++dx)
                        near |= box_meets(row[dx].x, row[dx].y, q.lo, q.hi);
                }
                if (!near && a.max_steps <= 160u) {
                    w = 4u;
                } else if (!near) {
                    const uint2 *SBf = a.SBp + (size_t)f * a.sb_stride;
                    const int R = min((int)a.max_steps - 1, 159), qx = (int)q.cx, qy = (int)q.cy;
                    const int sx0 = max(qx - R, 0) >> 6, sy0 = max(qy - R, 0) >> 6;
                    const int sx1 = min(qx + R, (int)a.wa - 1) >> 6, sy1 = min(qy + R, (int)a.ha - 1) >> 6;
                    uint32_t dmin = 255u;
                    for (int sy = sy0; sy <= sy1; ++sy)
                        for (int sx = sx0; sx <= sx1; ++sx) {
                            const uint2 sb = SBf[(size_t)sy * a.sbwp + sx];
                            const uint32_t dS = (uint32_t)square_dmin((sx << 6) - qx, (sy << 6) - qy, 64);
                            if (dS <= (uint32_t)R && box_meets(sb.x, sb.y, q.lo, q.hi))
                                dmin = min(dmin, dS);
                        }
                    w = min(max(dmin, 25u), 255u) >> 4;
                }
            }
        }
    }
#pragma unroll
    for (int sh = 1; sh < 64; sh <<= 1)
        w += (uint32_t)__shfl_xor((int)w, sh, 64);
    if (wid < o.n_wg && lane == 0)
        o.cls[wid] = (uint8_t)min(isqrt_floor(w >> 2), 15u); // (64 queries per wave against 16 samples per tile)
}

// ranks and class sizes per group of 64 waves, from the classes list_cost_kernel wrote
__global__ __launch_bounds__(64) void list_rank_kernel(TileOrderBufs o)
{
    const uint32_t wid = blockIdx.x * 64u + threadIdx.x;
    const uint32_t c = wid < o.n_wg ? o.cls[wid] : 16u;
    uint32_t rank = 0, mine = 0;
#pragma unroll
    for (uint32_t k = 0; k < 16u; ++k) {
        const unsigned long long m = __builtin_amdgcn_ballot_w64(c == k);
        if (c == k)
            rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
        if (threadIdx.x == k)
            mine = (uint32_t)__builtin_popcountll(m);
    }
    if (wid < o.n_wg)
        o.rank[wid] = (uint8_t)rank;
    if (threadIdx.x < 16u)
        o.cnt[(size_t)blockIdx.x * 16u + threadIdx.x] = (uint8_t)mine;
}

// ---------------------------------------------------------------------------------------
// PASS 1 for PREPARED QUERIES (explicit group lists, uniform tiles larger than one pixel,
// stepth_distance_dot_array_batch): src/depth_image.rs:112-123 for arbitrary `middle`s.
//
// Same lane = query scheme and the same per-lane probes (tile_lane.h) as ring_tile_kernel, but
// the queries of a wave have nothing in common, so everything is per lane: each lane walks the
// windows around ITS centre -- first the 7x7 blocks around its centre block, then the
// superblocks of its window ring by ring around its own superblock -- and reads their bounds
// with its own (vector) loads.  The wave only shares the loop counters; a lane that does not
// want the window of an iteration sits it out.  No reach limit: one group per pixel beyond the
// dense search's 263 rings runs here as 1x1 tiles.
// ---------------------------------------------------------------------------------------
struct ListArgs {
    const uint32_t *T, *TT;
    const uint2 *LB, *BBp, *SBp;
    uint32_t wa, ha, bw, bh, bwp, sbwp;
    uint32_t t_stride, bb_stride, sb_stride;
    const Query *queries; // frame f's at queries + f * q_stride, nq[f] of them
    const uint32_t *ids;  // optional: results of query k go to slot ids[k] (queries sorted by bin)
    uint32_t q_stride;
    uint32_t nq[kMaxBatch];
    uint32_t lanes; // queries per wave: 64, or fewer for small launches
    uint32_t max_steps, pos_bits;
    uint8_t *raw; // (d as u8) per query, depth_image.rs:121
    uint32_t raw_stride;
    uint32_t *raw32, *pos; // optional (single frame): d, winning cell (x, y)
    uint8_t *found;        // optional: Some / None
    uint32_t *max_word;
    unsigned long long *stats;
    // optional (group lists): the search paints its result itself -- the raw value into the
    // group's rectangle when that holds at most kPaintSmall pixels (the reference's groups hold a
    // few dozen, src/depth_image.rs:102), else the group is queued for paint_big_kernel; PASS 2
    // then runs in place over the rows (lut_rows_kernel).  Saves the separate paint launch, whose
    // byte stores took as long as a fifth of the search.
    const void *paint_groups[kMaxBatch]; // GroupRec lists, frame f's group = the query's slot
    uint8_t *paint_out[kMaxBatch];       // nullptr: no painting here
    size_t paint_stride;
    uint32_t paint_w, paint_h, paint_row0, paint_row1;
    uint32_t *big_count, *big_list; // per frame: count, list of `big_cap` slots
    uint32_t big_cap;
    // waves of all frames in one launch order: wave w of frame f = wid f * wpf + w; optional
    // heavy-first order (list_cost_kernel below): workgroup blockIdx.x takes wave order[blockIdx.x]
    uint32_t wpf;
    const uint32_t *order;
};

// eight bound tests of one row of blocks held in vector registers, highest column first
__device__ __forceinline__ uint32_t test_row_v(uint32_t acc, const ulonglong2 &v0, const ulonglong2 &v1,
                                               const ulonglong2 &v2, const ulonglong2 &v3, uint64_t k)
{
    acc = bound_bit(acc, v3.y, k);
    acc = bound_bit(acc, v3.x, k);
    acc = bound_bit(acc, v2.y, k);
    acc = bound_bit(acc, v2.x, k);
    acc = bound_bit(acc, v1.y, k);
    acc = bound_bit(acc, v1.x, k);
    acc = bound_bit(acc, v0.y, k);
    return bound_bit(acc, v0.x, k);
}

// blocks of the 7x7 window (origin = centre block - 3) in block-rings <= r: bit row*8+col
constexpr uint64_t kNearLe[4] = {0x0000000008000000ull, 0x0000001C1C1C0000ull,
                                 0x00003E3E3E3E3E00ull, 0x007F7F7F7F7F7F7Full};

// Probe tasks (round 3).  The windows' bound tests tell every lane which blocks ITS query must
// look into.  Until round 3 each lane then probed its own blocks one after the other, and the wave
// sat through a ~300-instruction probe whenever any lane had one: 23 of 64 lanes worked per
// instruction, the kernel was bound by its slowest waves (2.5 of 8 wave slots filled, VALU 45 %
// busy).  Now a candidate (query, block) pair is a TASK in a wave-level pool in LDS; tasks are
// executed 64 at a time by whichever lanes -- query box, centre, best key and limit come from LDS,
// an improvement merges back by minimum -- once the pool holds kListFlushMin of them (waiting
// longer lets limits go stale and queues probes a hit has made useless), before it could
// overflow, before the own-window results are judged and when the search ends.  The result is
// the same: a block is skipped only if its nearest cell lies at or beyond the query's limit of
// that moment.
constexpr uint32_t kListPoolCap = 128, kListFlushMin = 16;

template <bool STATS>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 8))) void ring_list_kernel(ListArgs a)
{
    // One wave per workgroup: the waves share nothing, and their run times differ by an order of
    // magnitude -- in a workgroup of four the slots of the early ones stayed taken until the last
    // was done.
    __shared__ uint2 s_box[1][64];            // the queries as the tasks see them: box {lo, hi},
    __shared__ uint2 s_qxy[1][64];            // centre,
    __shared__ uint32_t s_best[1][64];        // smallest key so far (merged by minimum),
    __shared__ uint32_t s_lim[1][64];         // its limit;
    __shared__ uint32_t s_task[1][kListPoolCap]; // the pool: query lane | block x << 6 | block y << 19
    const uint32_t lane = threadIdx.x & 63u;
    constexpr uint32_t wv = 0;
#ifdef STEPTH_LIST_PROFILE // tuning build: how long the waves run, and what the long ones do
    [[maybe_unused]] const unsigned long long lp_t0 = STATS ? __builtin_amdgcn_s_memtime() : 0ull;
    [[maybe_unused]] unsigned long long lp_goes = 0, lp_visits = 0, lp_go_ticks = 0, lp_m0_ticks = 0, lp_visit_ticks = 0, lp_survey_ticks = 0, lp_queue_ticks = 0;
#endif
    const uint32_t wid = a.order ? a.order[blockIdx.x] : blockIdx.x;
    const uint32_t f = wid / a.wpf;
    // a wave takes `lanes` (16, 32 or 64) consecutive queries: a small launch is bound by the
    // latency of its slowest wave, not by issue slots, and a wave of fewer queries has fewer
    // windows to sit through
    const uint32_t i = (wid - f * a.wpf) * a.lanes + lane;
    const bool valid = lane < a.lanes && i < a.nq[f];
    const uint32_t tf = f * a.t_stride;
    const uint2 *BBf = a.BBp + (size_t)f * a.bb_stride;
    const uint2 *SBf = a.SBp + (size_t)f * a.sb_stride;
    const uint32_t pb = a.pos_bits, ring_shift = pb + 2u;
    Query qq = Query{~0u, ~0u, 0u, 0u};
    if (valid)
        qq = a.queries[(size_t)f * a.q_stride + i];
    LaneQ q;
    q.qx = (int)qq.cx;
    q.qy = (int)qq.cy;
    q.lo = qq.lo;
    q.hi = qq.hi;
    const uint64_t kq = ((uint64_t)q.hi << 32) | q.lo; // the addends as a pair (tile_lane.h bound_bit)
    q.best = kTileNoKey;
    q.limit = 0u;
    uint32_t s_end = 0;
    unsigned long long n_ring0 = 0, n_tests = 0, n_probed = 0, n_near = 0, n_super = 0;
    // a centre outside the array ends the search at once (helpers.rs:49-51), cx = ~0: not in the strip
    const bool inb = valid && qq.cx < a.wa && qq.cy < a.ha && a.max_steps > 0;
    const int cbx = (int)(qq.cx >> 3), cby = (int)(qq.cy >> 3); // only used when inb
    if (inb) {
        const uint32_t cell = a.T[tf + ((uint32_t)cby * a.bw + (uint32_t)cbx) * 64u + ((qq.cy & 7u) << 3) + (qq.cx & 7u)];
        const uint32_t smax = max(max(qq.cx, a.wa - 1u - qq.cx), max(qq.cy, a.ha - 1u - qq.cy));
        s_end = min(a.max_steps, smax + 1u);
        if (cell_in_box(cell, q.lo, q.hi))
            q.best = 0u;
        else if (s_end > 1u)
            q.limit = s_end;
        if constexpr (STATS)
            n_ring0 = q.best == 0u;
    }

    if (__builtin_amdgcn_ballot_w64(q.limit != 0u)) {
        s_box[wv][lane] = make_uint2(q.lo, q.hi);
        s_qxy[wv][lane] = make_uint2((uint32_t)q.qx, (uint32_t)q.qy);
        uint32_t n_pool = 0; // tasks in the pool (wave-uniform)
        // shared far phase: ONE window of superblocks for the wave (see mode 1 below)
        int sx0 = 0, sy0 = 0, nsx = 0, nsy = 0, wx_lo = 0, wy_lo = 0, wx_hi = 0, wy_hi = 0, mx = 0, my = 0;
        uint32_t tol = 0, skey0 = kTileNoKey, skey1 = kTileNoKey, skey2 = kTileNoKey; // slot ty*16+tx
        // spread-out far phase: every lane walks its own window ring by ring (mode 2)
        int lx0 = 0, ly0 = 0, lx1 = 0, ly1 = 0, scx = 0, scy = 0, ring = 0, rt = -1;
        const int R = (int)a.max_steps - 1;

        // 0: the 7x7 blocks around the lane's own centre block, 1: the wave's window of
        // superblocks nearest first, 2: per-lane windows, 3: only the pool is left
        for (int mode = 0;;) {
            uint64_t cand = 0ull; // blocks of the window the lane's query must look into
            int ox = 0, oy = 0;   // the window's block (0,0), per lane
            int rounds = 1;
            if (mode == 0) {
#ifdef STEPTH_LIST_PROFILE
                const unsigned long long lp_w0 = STATS ? __builtin_amdgcn_s_memtime() : 0ull;
#endif
                // ---- the 7x7 blocks around the lane's own centre block (block-rings 0..3) ----
                if (q.limit != 0u) {
                    if constexpr (STATS)
                        ++n_near;
                    uint32_t m_lo = 0, m_hi = 0; // bit row*8+col
                    const uint2 *win = BBf + (size_t)(cby - 3 + kPadB) * a.bwp + (size_t)(cbx - 3 + kPadB);
#pragma unroll
                    for (int row = 6; row >= 4; --row) {
                        const ulonglong2 *rp4 = reinterpret_cast<const ulonglong2 *>(win + (size_t)row * a.bwp);
                        m_hi = test_row_v(m_hi, rp4[0], rp4[1], rp4[2], rp4[3], kq);
                    }
#pragma unroll
                    for (int row = 3; row >= 0; --row) {
                        const ulonglong2 *rp4 = reinterpret_cast<const ulonglong2 *>(win + (size_t)row * a.bwp);
                        m_lo = test_row_v(m_lo, rp4[0], rp4[1], rp4[2], rp4[3], kq);
                    }
                    if constexpr (STATS)
                        n_tests += 56u;
                    cand = (((uint64_t)m_hi << 32) | m_lo) & kNearLe[3];
                    ox = cbx - 3;
                    oy = cby - 3;
                }
#ifdef STEPTH_LIST_PROFILE
                if constexpr (STATS)
                    lp_m0_ticks += __builtin_amdgcn_s_memtime() - lp_w0 + (cand & 1ull) * 0ull;
#endif
                rounds = 4;
            } else if (mode == 1) {
#ifdef STEPTH_LIST_PROFILE
                const unsigned long long lp_v0 = STATS ? __builtin_amdgcn_s_memtime() : 0ull;
#define STEPTH_LP_VISIT_END() do { if constexpr (STATS) lp_visit_ticks += __builtin_amdgcn_s_memtime() - lp_v0; } while (0)
#else
#define STEPTH_LP_VISIT_END() ((void)0)
#endif
                // ---- the lanes still searching share what they can: the bounding box of their
                // centres gives ONE window of superblocks for the wave, surveyed once and visited
                // nearest first (bounds by scalar loads, every lane against its own box and its
                // own distance) ----
                const uint32_t pick = wave_min_dpp(min(min(skey0, skey1), skey2));
                if (pick == kTileNoKey || (pick >> 8) >= wave_max_dpp(q.limit) + tol) {
                    mode = 3; // (limits only shrink when the pooled tasks run: the search stays over)
                    STEPTH_LP_VISIT_END();
                    continue;
                }
                skey0 = skey0 == pick ? kTileNoKey : skey0;
                skey1 = skey1 == pick ? kTileNoKey : skey1;
                skey2 = skey2 == pick ? kTileNoKey : skey2;
                const int sx = sx0 + (int)(pick & 15u), sy = sy0 + (int)((pick >> 4) & 15u);
                const uint2 sb = SBf[(size_t)sy * a.sbwp + sx];
                const int ax = (sx << 6) - q.qx, ay = (sy << 6) - q.qy;
                const uint32_t dS = (uint32_t)max(max(max(ax, -63 - ax), max(ay, -63 - ay)), 0);
                const bool want = dS < q.limit && box_meets(sb.x, sb.y, q.lo, q.hi);
                if (!__builtin_amdgcn_ballot_w64(want)) {
                    STEPTH_LP_VISIT_END();
                    continue;
                }
                if constexpr (STATS)
                    n_super += want;
#ifdef STEPTH_LIST_PROFILE
                ++lp_visits;
#endif
                const int bx0 = sx * 8, by0 = sy * 8;
                // bounds of the superblock's 64 blocks: rows and half rows beyond every lane's
                // limit (distances from the box of the wave's centres) are skipped
                const uint32_t wlim = wave_max_dpp(want ? q.limit : 0u);
                uint32_t m_lo = 0, m_hi = 0;
                const int cx_l = bx0 * 8 - wx_hi, cx_r = wx_lo - (bx0 * 8 + 31); // cols 0..3
                const bool need_l = max(max(cx_l, cx_r), 0) < (int)wlim;
                const bool need_h = max(max(cx_l + 32, cx_r - 32), 0) < (int)wlim; // cols 4..7
                const uint2 *win = BBf + (size_t)(by0 + kPadB) * a.bwp + (size_t)(bx0 + kPadB);
                auto do_row = [&](uint32_t acc, int row) __attribute__((always_inline)) -> uint32_t {
                    const int ry0 = (by0 + row) * 8 - wy_hi, ry1 = wy_lo - ((by0 + row) * 8 + 7);
                    if (max(max(ry0, ry1), 0) >= (int)wlim)
                        return acc << 8;
                    const ulonglong4 *rp4 = reinterpret_cast<const ulonglong4 *>(win + (size_t)row * a.bwp);
                    if (need_h) {
                        const u64x4 v1 = sload_half_row<32>(rp4);
                        acc = test_half_row(acc, v1.x, v1.y, v1.z, v1.w, kq);
                    } else {
                        acc <<= 4;
                    }
                    if (need_l) {
                        const u64x4 v0 = sload_half_row<0>(rp4);
                        acc = test_half_row(acc, v0.x, v0.y, v0.z, v0.w, kq);
                    } else {
                        acc <<= 4;
                    }
                    return acc;
                };
#pragma unroll
                for (int row = 7; row >= 4; --row)
                    m_hi = do_row(m_hi, row);
#pragma unroll
                for (int row = 3; row >= 0; --row)
                    m_lo = do_row(m_lo, row);
                if constexpr (STATS)
                    n_tests += want ? 64u : 0u;
                if (want) // the lane's own 7x7 window did block-rings 0..3 around ITS centre block
                    cand = (((uint64_t)m_hi << 32) | m_lo) & ~window_le_mask(bx0, by0, cbx, cby, 3);
                ox = bx0;
                oy = by0;
                STEPTH_LP_VISIT_END();
            } else if (mode == 2) {
                // ---- spread-out wave: every lane walks the superblocks of its own window, ring
                // by ring around its own superblock (the wave shares only the loop counters) ----
                ++rt;
                const int n_in_ring = ring == 0 ? 1 : 8 * ring;
                if (rt >= n_in_ring) {
                    rt = 0;
                    ++ring;
                }
                // a superblock of ring r lies at least 64 (r - 1) + 1 rings away
                const int lb = ring == 0 ? 0 : 64 * (ring - 1) + 1;
                if (!__builtin_amdgcn_ballot_w64(q.limit != 0u && lb < (int)q.limit)) {
                    mode = 3;
                    continue;
                }
                int dx = 0, dy = 0;
                if (ring > 0) {
                    const int side = rt / (2 * ring), o = rt - side * 2 * ring;
                    dx = side == 0 ? -ring + o : side == 1 ? ring : side == 2 ? ring - o : -ring;
                    dy = side == 0 ? -ring : side == 1 ? -ring + o : side == 2 ? ring : ring - o;
                }
                const int sx = scx + dx, sy = scy + dy;
                const bool inwin = q.limit != 0u && sx >= lx0 && sx <= lx1 && sy >= ly0 && sy <= ly1;
                uint2 sb = make_uint2(0u, 0u);
                if (inwin)
                    sb = SBf[(size_t)sy * a.sbwp + sx];
                const int ax = (sx << 6) - q.qx, ay = (sy << 6) - q.qy;
                const uint32_t dS = (uint32_t)max(max(max(ax, -63 - ax), max(ay, -63 - ay)), 0);
                const bool want = inwin && dS < q.limit && box_meets(sb.x, sb.y, q.lo, q.hi);
                if constexpr (STATS)
                    n_tests += inwin;
                if (want) {
                    if constexpr (STATS)
                        ++n_super;
                    const int bx0 = sx * 8, by0 = sy * 8;
                    uint32_t m_lo = 0, m_hi = 0;
                    const uint2 *win = BBf + (size_t)(by0 + kPadB) * a.bwp + (size_t)(bx0 + kPadB);
#pragma unroll 1
                    for (int row = 7; row >= 4; --row) {
                        const ulonglong2 *rp4 = reinterpret_cast<const ulonglong2 *>(win + (size_t)row * a.bwp);
                        m_hi = test_row_v(m_hi, rp4[0], rp4[1], rp4[2], rp4[3], kq);
                    }
#pragma unroll 1
                    for (int row = 3; row >= 0; --row) {
                        const ulonglong2 *rp4 = reinterpret_cast<const ulonglong2 *>(win + (size_t)row * a.bwp);
                        m_lo = test_row_v(m_lo, rp4[0], rp4[1], rp4[2], rp4[3], kq);
                    }
                    if constexpr (STATS)
                        n_tests += 64u;
                    cand = (((uint64_t)m_hi << 32) | m_lo) & ~window_le_mask(bx0, by0, cbx, cby, 3);
                    ox = bx0;
                    oy = by0;
                }
            }
            // ---- the window's candidates become tasks (the own window block-ring by block-ring,
            // nearest first); the pool is executed whenever it holds enough of them ----
#ifdef STEPTH_LIST_PROFILE
            const unsigned long long lp_q0 = STATS ? __builtin_amdgcn_s_memtime() : 0ull, lp_go_before = lp_go_ticks;
#endif
            for (int ps = 0; ps < rounds; ++ps) {
                uint64_t cur = cand;
                if (mode == 0) {
                    cur = cand & kNearLe[ps] & ~(ps ? kNearLe[ps - 1] : 0ull);
                    if (8 * (ps - 1) + 1 >= (int)q.limit)
                        cur = 0ull;
                }
                // before the own-window results are judged, and at the end, nothing may stay pooled
                const bool drain = mode == 3 || (mode == 0 && ps == 3);
                for (;;) {
                    const bool more = __builtin_amdgcn_ballot_w64(cur != 0ull) != 0ull;
                    if (more && n_pool <= kListPoolCap - 64u) {
                        // every lane with a candidate left queues its next one
                        bool ok = false;
                        uint32_t task = 0;
                        if (cur != 0ull) {
                            const int k = __builtin_ctzll(cur);
                            cur &= cur - 1ull;
                            const int bx = ox + (k & 7), by = oy + (k >> 3);
                            // (the dmin test also keeps stale bounds of unpacked rows from being probed)
                            ok = (uint32_t)square_dmin(bx * 8 - q.qx, by * 8 - q.qy, 8) < q.limit;
                            task = lane | ((uint32_t)bx << 6) | ((uint32_t)by << 19);
                        }
                        const unsigned long long okm = __builtin_amdgcn_ballot_w64(ok);
                        if (ok)
                            s_task[wv][n_pool + __builtin_amdgcn_mbcnt_hi((uint32_t)(okm >> 32),
                                                                           __builtin_amdgcn_mbcnt_lo((uint32_t)okm, 0u))] = task;
                        n_pool += (uint32_t)__builtin_popcountll(okm);
                        continue;
                    }
                    if (n_pool > kListPoolCap - 64u || (!more && (n_pool >= kListFlushMin || (drain && n_pool != 0u)))) {
                        // ---- the pooled tasks, 64 at a time.  Lane i of a go takes task i: query
                        // box, centre, best key and limit from LDS, the probe like the query's own
                        // lane would run it, an improvement merged back by minimum -- the tasks of
                        // one go see the same snapshot of their queries.  (No fence or wave barrier:
                        // the LDS operations of a wave complete in program order, and either
                        // intrinsic would turn the scalar loads of the bound rows into vector loads.)
                        s_best[wv][lane] = q.best; // what the lane knows of its own query
                        s_lim[wv][lane] = q.limit;
#ifdef STEPTH_LIST_PROFILE
                        const unsigned long long lp_g0 = STATS ? __builtin_amdgcn_s_memtime() : 0ull;
#endif
                        for (uint32_t base = 0; base < n_pool; base += 64u) {
#ifdef STEPTH_LIST_PROFILE
                            ++lp_goes;
#endif
                            const uint32_t ti = base + lane;
                            if (ti < n_pool) {
                                const uint32_t t = s_task[wv][ti], src = t & 63u;
                                const int bx = (int)((t >> 6) & 0x1FFFu), by = (int)(t >> 19);
                                LaneQ tq;
                                const uint2 box = s_box[wv][src], xy = s_qxy[wv][src];
                                tq.lo = box.x;
                                tq.hi = box.y;
                                tq.qx = (int)xy.x;
                                tq.qy = (int)xy.y;
                                tq.best = s_best[wv][src];
                                tq.limit = s_lim[wv][src];
                                const int x0 = bx * 8 - tq.qx, y0 = by * 8 - tq.qy;
                                if ((uint32_t)square_dmin(x0, y0, 8) < tq.limit) {
                                    const uint32_t boff = tf + ((uint32_t)by * a.bw + (uint32_t)bx) * 64u;
                                    const uint32_t best_before = tq.best;
                                    const uint32_t nl = probe_block(tq, x0, y0, pb, reinterpret_cast<const uint8_t *>(a.T),
                                                                    reinterpret_cast<const uint8_t *>(a.TT),
                                                                    reinterpret_cast<const uint8_t *>(a.LB), boff << 2);
                                    if (tq.best < best_before) {
                                        atomicMin(&s_best[wv][src], tq.best);
                                        atomicMin(&s_lim[wv][src], tq.limit);
                                    }
                                    if constexpr (STATS) {
                                        ++n_probed;
                                        n_tests += 8u * nl;
                                    }
                                }
                            }
                        }
                        n_pool = 0u;
                        if (q.limit != 0u) {
                            q.best = s_best[wv][lane];
                            q.limit = s_lim[wv][lane];
                        }
#ifdef STEPTH_LIST_PROFILE
                        if constexpr (STATS)
                            lp_go_ticks += __builtin_amdgcn_s_memtime() - lp_g0;
#endif
                    }
                    if (!more)
                        break;
                }
            }
#ifdef STEPTH_LIST_PROFILE
            if constexpr (STATS) // the queue-and-go stage without its goes
                lp_queue_ticks += __builtin_amdgcn_s_memtime() - lp_q0 - (lp_go_ticks - lp_go_before);
#endif
            if (mode == 3)
                break;
            if (mode == 0) {
                // settled: best ring <= 24, or nothing beyond ring 24 to search
                if (q.limit != 0u && (s_end <= 25u || (q.best != kTileNoKey && (q.best >> ring_shift) <= 24u)))
                    q.limit = 0u;
                if (!__builtin_amdgcn_ballot_w64(q.limit != 0u))
                    break;
                // ---- far phase: one window for the wave if the lanes still searching lie close
                // enough together for the 10 x 12 slots, else per-lane windows ----
                const bool far = q.limit != 0u;
#ifdef STEPTH_LIST_PROFILE
                const unsigned long long lp_s0 = STATS ? __builtin_amdgcn_s_memtime() : 0ull;
#endif
                wx_lo = (int)wave_min_dpp(far ? (uint32_t)q.qx : 0x7fffffffu);
                wy_lo = (int)wave_min_dpp(far ? (uint32_t)q.qy : 0x7fffffffu);
                wx_hi = (int)wave_max_dpp(far ? (uint32_t)q.qx : 0u);
                wy_hi = (int)wave_max_dpp(far ? (uint32_t)q.qy : 0u);
                sx0 = max(wx_lo - R, 0) >> 6;
                sy0 = max(wy_lo - R, 0) >> 6;
                const int sx1 = min(wx_hi + R, (int)a.wa - 1) >> 6, sy1 = min(wy_hi + R, (int)a.ha - 1) >> 6;
                nsx = sx1 - sx0 + 1;
                nsy = sy1 - sy0 + 1;
                if (nsx <= 10 && nsy <= 12) {
                    mode = 1;
                    // middle of the box orders the visits; a lane's own distance differs by <= tol
                    mx = (wx_lo + wx_hi) >> 1;
                    my = (wy_lo + wy_hi) >> 1;
                    tol = (uint32_t)max(wx_hi - mx, wy_hi - my) + 1u;
                    for (int ty = 0; ty < nsy; ++ty) {
                        const int sy = sy0 + ty;
                        const uint32_t row_dw = reinterpret_cast<const uint32_t *>(SBf + (size_t)sy * a.sbwp + sx0)[lane & 31u];
                        const int ay = (sy << 6) - q.qy;
                        const int dsy = max(max(ay, -63 - ay), 0);
                        for (int tx = 0; tx < nsx; ++tx) { // (lane i holds dword i of the row, see ring_tile_kernel)
                            const uint32_t bmax = (uint32_t)__builtin_amdgcn_readlane((int)row_dw, 2 * tx);
                            const uint32_t bmin = (uint32_t)__builtin_amdgcn_readlane((int)row_dw, 2 * tx + 1);
                            const int ax = ((sx0 + tx) << 6) - q.qx;
                            const uint32_t dS = (uint32_t)max(max(max(ax, -63 - ax), 0), dsy);
                            const bool w = dS < q.limit && box_meets(bmax, bmin, q.lo, q.hi);
                            if constexpr (STATS)
                                n_tests += q.limit != 0u;
                            if (__builtin_amdgcn_ballot_w64(w)) {
                                const int rx = ((sx0 + tx) << 6) - mx, ry = (sy << 6) - my;
                                const uint32_t dref = (uint32_t)max(max(max(rx, -63 - rx), max(ry, -63 - ry)), 0);
                                const uint32_t t = (uint32_t)(ty * 16 + tx);
                                const uint32_t key = (dref << 8) | t;
                                if (lane == (t & 63u)) {
                                    if (t < 64u)
                                        skey0 = key;
                                    else if (t < 128u)
                                        skey1 = key;
                                    else
                                        skey2 = key;
                                }
                            }
                        }
                    }
                } else {
                    mode = 2;
                    lx0 = max(q.qx - R, 0) >> 6;
                    ly0 = max(q.qy - R, 0) >> 6;
                    lx1 = min(q.qx + R, (int)a.wa - 1) >> 6;
                    ly1 = min(q.qy + R, (int)a.ha - 1) >> 6;
                    scx = q.qx >> 6;
                    scy = q.qy >> 6;
                }
#ifdef STEPTH_LIST_PROFILE
                if constexpr (STATS)
                    lp_survey_ticks += __builtin_amdgcn_s_memtime() - lp_s0;
#endif
            }
        }
    }
#ifdef STEPTH_LIST_PROFILE
    [[maybe_unused]] const unsigned long long lp_e0 = STATS ? __builtin_amdgcn_s_memtime() : 0ull;
#endif

    // ---- distance, store, max (depth_image.rs:118-124) ----
    uint32_t d = 0;
    int ddx = 0, ddy = 0;
    const bool fnd = q.best != kTileNoKey;
    if (fnd && q.best != 0u) {
        const int s = (int)(q.best >> (pb + 2u)), side = (int)((q.best >> pb) & 3u);
        const int off = (int)(q.best & ((1u << pb) - 1u)) - s;
        ddx = side < 2 ? off : side == 2 ? s : -s;
        ddy = side == 0 ? s : side == 1 ? -s : off;
        d = isqrt_floor((uint32_t)(ddx * ddx + ddy * ddy));
    }
    if (valid) {
        const uint32_t o = a.ids ? a.ids[(size_t)f * a.q_stride + i] : i;
        a.raw[(size_t)f * a.raw_stride + o] = (uint8_t)(d & 0xffu); // `d as u8` wraps
        if (a.raw32)
            a.raw32[o] = d;
        if (a.pos) {
            a.pos[2 * (size_t)o] = fnd ? (uint32_t)(q.qx + ddx) : 0u;
            a.pos[2 * (size_t)o + 1] = fnd ? (uint32_t)(q.qy + ddy) : 0u;
        }
        if (a.found)
            a.found[o] = fnd ? 1 : 0;
        if (a.paint_out[0]) { // depth_image.rs:130-132, raw value (PASS 2 follows in place)
            const GroupRec r = reinterpret_cast<const GroupRec *>(a.paint_groups[f])[o];
            if (r.y >= a.paint_row0 && r.y < a.paint_row1 && r.x < a.paint_w && r.w != 0u && r.h != 0u) {
                const uint32_t w = min(r.w, a.paint_w - r.x), h = min(r.h, a.paint_h - r.y);
                if ((uint64_t)w * h <= kPaintSmall) {
                    uint8_t *out = a.paint_out[f] + (size_t)r.y * a.paint_stride + r.x;
                    for (uint32_t yy = 0; yy < h; ++yy)
                        for (uint32_t xx = 0; xx < w; ++xx)
                            out[(size_t)yy * a.paint_stride + xx] = (uint8_t)(d & 0xffu);
                } else {
                    const uint32_t at = atomicAdd(a.big_count + f, 1u);
                    if (at < a.big_cap)
                        a.big_list[(size_t)f * a.big_cap + at] = o;
                }
            }
        }
    }
    const uint32_t wmax = wave_max_dpp(d & 0xffu);
    if (lane == 0 && wmax > __hip_atomic_load(a.max_word + f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
        atomicMax(a.max_word + f, wmax);
    if constexpr (STATS) {
        unsigned long long v0 = n_ring0, v1 = n_tests, v2 = n_near, v4 = n_super, v5 = n_probed;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            v0 += __shfl_xor(v0, o, 64);
            v1 += __shfl_xor(v1, o, 64);
            v2 += __shfl_xor(v2, o, 64);
            v4 += __shfl_xor(v4, o, 64);
            v5 += __shfl_xor(v5, o, 64);
        }
        if (lane == 0) {
            atomicAdd(&stats_replica(a.stats)[0], v0);
            atomicAdd(&stats_replica(a.stats)[1], v1);
            atomicAdd(&stats_replica(a.stats)[2], v2);
            atomicAdd(&stats_replica(a.stats)[4], v4);
            atomicAdd(&stats_replica(a.stats)[5], v5);
#ifdef STEPTH_LIST_PROFILE
            // waves by run time (s_memtime ticks of 10 ns): bucket b = log2(ticks) - 13 in 0..7 -- [8+b]
            // waves, [16+b] their shared-window visits, [24+b] their goes
            const unsigned long long ticks = __builtin_amdgcn_s_memtime() - lp_t0;
            const int b = min(max(63 - __builtin_clzll(ticks | 1ull) - 13, 0), 7);
            atomicAdd(&stats_replica(a.stats)[8 + b], 1ull);
            if (b < 4)
                atomicAdd(&stats_replica(a.stats)[16 + b], lp_visits);
            atomicAdd(&stats_replica(a.stats)[24 + b], lp_goes);
            atomicAdd(&stats_replica(a.stats)[6], ticks);
            atomicAdd(&stats_replica(a.stats)[7], lp_go_ticks);
            atomicAdd(&stats_replica(a.stats)[3], lp_m0_ticks);
            // (this build only: the words 20..23 replace the per-bucket visit sums of buckets 4..7)
            atomicAdd(&stats_replica(a.stats)[21], lp_visit_ticks);
            atomicAdd(&stats_replica(a.stats)[22], lp_survey_ticks);
            atomicAdd(&stats_replica(a.stats)[23], lp_e0 - lp_t0);
            atomicAdd(&stats_replica(a.stats)[20], lp_queue_ticks);
#endif
        }
    }
}

// ---------------------------------------------------------------------------------------
// pack for the dense search: RGB8 [Ha][Wa] -> T, TT (cells, row- and column-major inside
// the 8x8 block) and the padded block bounds BBp.  One wave per block.  Replaces
// disage::converters::raw_rgb + pixels_to_array (src/depth_image.rs:97-100).
// ---------------------------------------------------------------------------------------
struct TilePackArgs {
    const uint8_t *src[kMaxBatch];
    size_t stride;
    uint32_t wa, ha, bw, bh, bwp;
    uint32_t *T, *TT;
    uint2 *LB, *BBp;
    uint32_t by0, nby; // block rows to pack
    uint32_t *zero;    // words cleared for the kernels that follow
    uint32_t n_zero;
    uint32_t t_stride, bb_stride;
};

// One workgroup packs 32 blocks of one block row (256 x 8 pixels): the pixel bytes go through
// LDS (dword loads when the rows are 4-byte aligned), then thread (block, row) builds the 8
// cells of its row -- T, row bounds, and by a 3-step exchange the block bounds -- and, after a
// barrier, thread (block, column) builds its column from the cells left in LDS -- TT, column
// bounds.  No per-cell shuffles: ~30 instructions per block instead of ~150.
constexpr uint32_t kPackBlocks = 32;                  // blocks per workgroup
constexpr uint32_t kPackRowBytes = kPackBlocks * 24u; // 768 pixel bytes per row of the strip

__device__ __forceinline__ void fold_bounds(uint32_t cell, uint32_t (&mn)[3], uint32_t (&mx)[3])
{
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const uint32_t v = (cell >> (10 * c)) & 0xffu;
        mn[c] = min(mn[c], v);
        mx[c] = max(mx[c], v);
    }
}

__global__ __launch_bounds__(256) void tile_pack_kernel(TilePackArgs a)
{
    __shared__ uint32_t s_px[8][kPackRowBytes / 4 + 1]; // RGB bytes of the strip, row by row
    __shared__ uint32_t s_cell[8][kPackBlocks * 8 + 1]; // its cells
    const uint32_t t = threadIdx.x, f = blockIdx.z;
    if (blockIdx.x == 0 && blockIdx.y == 0 && f == 0)
        for (uint32_t i = t; i < a.n_zero; i += 256u)
            a.zero[i] = 0u;
    const uint32_t bx0 = blockIdx.x * kPackBlocks, by = a.by0 + blockIdx.y;
    const uint32_t x0 = bx0 * 8u, y0 = by * 8u;
    const uint32_t n_px = min(256u, a.wa - x0), n_bytes = n_px * 3u; // pixels / bytes per row here
    const uint8_t *src = a.src[f] + (size_t)y0 * a.stride + (size_t)x0 * 3u;
    // ---- pixel bytes -> LDS ----
    if (((reinterpret_cast<uintptr_t>(src) | a.stride) & 3u) == 0) {
        for (uint32_t i = t; i < 8u * (kPackRowBytes / 4u); i += 256u) {
            const uint32_t r = i / (kPackRowBytes / 4u), d = i - r * (kPackRowBytes / 4u);
            if (y0 + r < a.ha && d * 4u < n_bytes) {
                const uint8_t *q = src + (size_t)r * a.stride + d * 4u;
                uint32_t v;
                if (d * 4u + 4u <= n_bytes) { // whole dword inside the row
                    v = *reinterpret_cast<const uint32_t *>(q);
                } else { // the row's last bytes: never read past the image
                    v = q[0];
                    if (d * 4u + 1u < n_bytes)
                        v |= (uint32_t)q[1] << 8;
                    if (d * 4u + 2u < n_bytes)
                        v |= (uint32_t)q[2] << 16;
                }
                s_px[r][d] = v;
            }
        }
    } else {
        uint8_t *pb = reinterpret_cast<uint8_t *>(&s_px[0][0]);
        for (uint32_t i = t; i < 8u * kPackRowBytes; i += 256u) {
            const uint32_t r = i / kPackRowBytes, b = i - r * kPackRowBytes;
            if (y0 + r < a.ha && b < n_bytes)
                pb[r * sizeof(s_px[0]) + b] = src[(size_t)r * a.stride + b];
        }
    }
    __syncthreads();
    // ---- thread = (block, row): cells of the row, T, row bounds, block bounds ----
    const uint32_t b = t >> 3, l = t & 7u, bx = bx0 + b;
    const bool in_blocks = bx < a.bw;
    const size_t base = (size_t)f * a.t_stride + ((size_t)by * a.bw + bx) * 64u;
    uint2 *lb = a.LB + base / 4u; // 16 line bounds per block
    {
        uint32_t cell[8], mn[3] = {255u, 255u, 255u}, mx[3] = {0u, 0u, 0u};
        const uint8_t *row = reinterpret_cast<const uint8_t *>(&s_px[l][0]) + b * 24u;
#pragma unroll
        for (uint32_t p = 0; p < 8; ++p) {
            cell[p] = kNoCell;
            if (x0 + b * 8u + p < a.wa && y0 + l < a.ha) {
                cell[p] = pack_cell(row[3 * p], row[3 * p + 1], row[3 * p + 2]);
                fold_bounds(cell[p], mn, mx);
            }
            s_cell[l][b * 8u + p] = cell[p];
        }
        if (in_blocks) {
            uint4 *dst = reinterpret_cast<uint4 *>(a.T + base + l * 8u);
            dst[0] = make_uint4(cell[0], cell[1], cell[2], cell[3]);
            dst[1] = make_uint4(cell[4], cell[5], cell[6], cell[7]);
            // a line with no cell inside the image gets empty bounds: never matches
            lb[l] = mx[0] < mn[0] ? make_uint2(0u, kNoCell)
                                  : make_uint2(pack_cell(mx[0], mx[1], mx[2]), pack_cell(mn[0], mn[1], mn[2]));
        }
        // the block's bounds: fold the 8 rows (8 consecutive lanes)
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int o = 1; o < 8; o <<= 1) {
                mn[c] = min(mn[c], (uint32_t)__shfl_xor((int)mn[c], o, 64));
                mx[c] = max(mx[c], (uint32_t)__shfl_xor((int)mx[c], o, 64));
            }
        if (in_blocks && l == 0)
            a.BBp[(size_t)f * a.bb_stride + (size_t)(by + kPadB) * a.bwp + bx + kPadB] =
                make_uint2(pack_cell(mx[0], mx[1], mx[2]), pack_cell(mn[0], mn[1], mn[2]));
    }
    __syncthreads();
    // ---- thread = (block, column): TT and the column bounds, from the cells in LDS ----
    if (in_blocks) {
        uint32_t cell[8], mn[3] = {255u, 255u, 255u}, mx[3] = {0u, 0u, 0u};
#pragma unroll
        for (uint32_t p = 0; p < 8; ++p) {
            cell[p] = s_cell[p][b * 8u + l];
            if (cell[p] != kNoCell)
                fold_bounds(cell[p], mn, mx);
        }
        uint4 *dst = reinterpret_cast<uint4 *>(a.TT + base + l * 8u);
        dst[0] = make_uint4(cell[0], cell[1], cell[2], cell[3]);
        dst[1] = make_uint4(cell[4], cell[5], cell[6], cell[7]);
        lb[8u + l] = mx[0] < mn[0] ? make_uint2(0u, kNoCell)
                                   : make_uint2(pack_cell(mx[0], mx[1], mx[2]), pack_cell(mn[0], mn[1], mn[2]));
    }
}

// superblock bounds from the padded block bounds: one wave folds 8x8 blocks
struct TileSuperArgs {
    const uint2 *BBp;
    uint32_t bw, bh, bwp, sbw, sbwp;
    uint2 *SBp;
    uint32_t sby0, nsby; // superblock rows to fold
    uint32_t by0, by1;   // block rows that are valid (others hold stale bounds)
    uint32_t bb_stride, sb_stride;
};

__global__ __launch_bounds__(256) void tile_super_kernel(TileSuperArgs a)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t idx = blockIdx.x * 4u + (threadIdx.x >> 6);
    if (idx >= a.sbw * a.nsby)
        return;
    const uint32_t sx = idx % a.sbw, sy = a.sby0 + idx / a.sbw;
    const uint32_t bx = sx * 8u + (lane & 7u), by = sy * 8u + (lane >> 3);
    uint32_t mn[3] = {255u, 255u, 255u}, mx[3] = {0u, 0u, 0u};
    bool any = false;
    if (bx < a.bw && by >= a.by0 && by < a.by1) {
        const uint2 bb = a.BBp[(size_t)blockIdx.y * a.bb_stride + (size_t)(by + kPadB) * a.bwp + bx + kPadB];
        any = true;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            mx[c] = (bb.x >> (10 * c)) & 0xffu;
            mn[c] = (bb.y >> (10 * c)) & 0xffu;
        }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        mn[c] = wave_min_dpp(mn[c]);
        mx[c] = wave_max_dpp(mx[c]);
    }
    const bool some = __builtin_amdgcn_ballot_w64(any) != 0ull;
    if (lane == 0)
        a.SBp[(size_t)blockIdx.y * a.sb_stride + (size_t)sy * a.sbwp + sx] =
            some ? make_uint2(pack_cell(mx[0], mx[1], mx[2]), pack_cell(mn[0], mn[1], mn[2]))
                 : make_uint2(0u, kNoCell); // nothing valid here: never matches
}

} // namespace stepth
