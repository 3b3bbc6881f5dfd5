// ring_kernels.h -- gfx950 kernels of the reference-exact mode (square-ring first hit).
//
// Path replaced: PASS 1 / max / PASS 2 / paint of DepthImage::load_depth_from_additional,
// /root/reference/src/depth_image.rs:96-132, and helpers::distance_dot_array,
// /root/reference/src/helpers.rs:9-54.  Wave = 64 lanes everywhere.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "cell_pack.h"

namespace stepth {

// ---------------------------------------------------------------------------------------
// pack_additional: RGB8 [Ha][Wa] -> the search structure in HBM.  Replaces
// disage::converters::raw_rgb + pixels_to_array (src/depth_image.rs:97-100): instead of a
// Vec<Vec<[u8;3]>> the additional image becomes
//
//   T  [bh][bw][8][8] dwords  cells (cell_pack.h) tiled in 8x8 blocks, one block = 256
//                             contiguous bytes = one fully coalesced wave read; cells of
//                             edge blocks that fall outside the image are kNoCell
//   BB [bh][bw] uint2         per-block channel bounds {max, min} in the same spread format
//
// with bw = ceil(Wa/8), bh = ceil(Ha/8).  One wave packs one block.
// ---------------------------------------------------------------------------------------
constexpr int kMaxBatch = 64; // frames one set of launches can carry (blockIdx.y / record bits)

// Statistic counters are kept in kStatsReplicas copies of kStatsWords words, a workgroup adds to
// copy blockIdx.x % kStatsReplicas and the host sums the copies: every wave of a launch adds
// several counters, and atomics on one 128-byte line serialise chip-wide (~11 ns each -- the
// counting variant of the dense search took 15-50 ms instead of 2.4 while they shared one copy).
constexpr uint32_t kStatsWords = 32, kStatsReplicas = 64;
__device__ __forceinline__ unsigned long long *stats_replica(unsigned long long *base)
{
    return base + (blockIdx.x % kStatsReplicas) * kStatsWords;
}
constexpr int kBlk = 8;       // block side
constexpr int kBlkShift = 3;

struct PackArgs {
    const uint8_t *src[kMaxBatch]; // one additional image per frame (blockIdx.y)
    size_t stride;
    uint32_t wa, ha;
    uint32_t bw, bh;
    uint32_t *T;
    uint2 *BB;
    uint32_t by0, nby; // block rows to pack (a row strip only reaches part of the image)
    uint32_t *zero;    // words to clear for the kernels that follow (max, counters) ...
    uint32_t n_zero;   // ... instead of a separate memset launch
    uint32_t t_stride, bb_stride; // per-frame strides of T (dwords) and BB (uint2)
};

__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
        v = min(v, (uint32_t)__shfl_xor((int)v, o, 64));
    return v;
}

__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
        v = max(v, (uint32_t)__shfl_xor((int)v, o, 64));
    return v;
}

__global__ __launch_bounds__(256) void pack_additional_kernel(PackArgs a)
{
    const uint32_t lane = threadIdx.x & 63u, f = blockIdx.y;
    if (blockIdx.x == 0 && f == 0)
        for (uint32_t i = threadIdx.x; i < a.n_zero; i += blockDim.x)
            a.zero[i] = 0u;
    const uint32_t idx = blockIdx.x * 4u + (threadIdx.x >> 6);
    if (idx >= a.bw * a.nby)
        return;
    const uint32_t bx = idx % a.bw, by = a.by0 + idx / a.bw;
    const uint32_t blk = by * a.bw + bx;
    const uint32_t x = bx * kBlk + (lane & 7u), y = by * kBlk + (lane >> 3);
    uint32_t cell = kNoCell;
    uint32_t mn[3] = {255u, 255u, 255u}, mx[3] = {0u, 0u, 0u};
    if (x < a.wa && y < a.ha) {
        const uint8_t *p = a.src[f] + (size_t)y * a.stride + (size_t)x * 3;
        mn[0] = mx[0] = p[0];
        mn[1] = mx[1] = p[1];
        mn[2] = mx[2] = p[2];
        cell = pack_cell(p[0], p[1], p[2]);
    }
    a.T[(size_t)f * a.t_stride + (size_t)blk * 64u + lane] = cell;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        mn[c] = wave_min_u32(mn[c]);
        mx[c] = wave_max_u32(mx[c]);
    }
    if (lane == 0)
        a.BB[(size_t)f * a.bb_stride + blk] =
            make_uint2(pack_cell(mx[0], mx[1], mx[2]), pack_cell(mn[0], mn[1], mn[2]));
}

// Superblock bounds: one wave folds the bounds of 8x8 blocks (64x64 cells) into SB.
struct SuperArgs {
    const uint2 *BB;
    uint32_t bw, bh, sbw, sbh;
    uint2 *SB;
    uint32_t sby0, nsby; // superblock rows to fold, and the block rows that are valid: blocks
    uint32_t by0, by1;   // outside [by0,by1) hold stale bounds and are left out
    uint32_t bb_stride, sb_stride; // per-frame strides (blockIdx.y = frame)
};

__global__ __launch_bounds__(256) void pack_super_kernel(SuperArgs a)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t idx = blockIdx.x * 4u + (threadIdx.x >> 6);
    if (idx >= a.sbw * a.nsby)
        return;
    const uint32_t sb = (a.sby0 + idx / a.sbw) * a.sbw + idx % a.sbw;
    const uint32_t bx = (sb % a.sbw) * 8u + (lane & 7u), by = (sb / a.sbw) * 8u + (lane >> 3);
    uint32_t mn[3] = {255u, 255u, 255u}, mx[3] = {0u, 0u, 0u};
    if (bx < a.bw && by >= a.by0 && by < a.by1) {
        const uint2 bb = a.BB[(size_t)blockIdx.y * a.bb_stride + (size_t)by * a.bw + bx];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            mx[c] = (bb.x >> (10 * c)) & 0xffu;
            mn[c] = (bb.y >> (10 * c)) & 0xffu;
        }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        mn[c] = wave_min_u32(mn[c]);
        mx[c] = wave_max_u32(mx[c]);
    }
    if (lane == 0)
        a.SB[(size_t)blockIdx.y * a.sb_stride + sb] =
            make_uint2(pack_cell(mx[0], mx[1], mx[2]), pack_cell(mn[0], mn[1], mn[2]));
}

// Rows a call owns inside [row_begin,row_end): all of them (block == 0), or every mod-th block
// of `block` rows -- interleaved row blocks balance the data-dependent work between ranks.
struct RowPattern {
    uint32_t block, mod, rem;
    __device__ __forceinline__ bool owns(uint32_t y) const
    {
        return block == 0u || (y / block) % mod == rem;
    }
};

// ---------------------------------------------------------------------------------------
// Queries.  One query = one disage group (src/depth_image.rs:113-121): the search centre
// `middle` and the colour box of its value.
// ---------------------------------------------------------------------------------------
struct Query {
    uint32_t cx, cy; // middle = ((x+w)/2, (y+h)/2), depth_image.rs:114-117; ~0u = not in strip
    uint32_t lo, hi; // make_box(value, precision)
};

struct PrepArgs {
    // uniform tiles from the main image (one image per frame, blockIdx.y)
    const uint8_t *main_px[kMaxBatch];
    size_t mstride;
    uint32_t mch, w, h;
    uint32_t tile_w, tile_h, gx, gy;
    // or an explicit list per frame
    const void *groups[kMaxBatch]; // stepth_group[n[f]]
    uint32_t n[kMaxBatch];         // queries of frame f (tiles: the same for every frame)
    uint32_t q_stride;             // queries between the frames' arrays in `out`
    uint32_t row_begin, row_end;
    RowPattern rp;
    uint8_t prec[3];
    Query *out;
    // optional (group lists): per frame, how many queries fall into each 32x32-cell bin of
    // search centres (last bin: queries that search nothing) -- the first pass of a counting sort
    uint32_t *bin_counts;
    uint32_t bins_x, bins_y, n_bins; // n_bins = bins_x * bins_y + 1
};

// Group lists arrive in any order, but the per-lane search kernel shares a window of
// superblocks among the 64 queries of a wave: queries are sorted by the 32x32-cell bin of their
// search centre (counting sort: count in prep_list_kernel, scan, scatter).
// bins[key] += 1 for every active lane, one atomic per DISTINCT key of the wave (consecutive
// groups mostly share a bin, and same-address atomics serialise); returns the lane's slot
__device__ __forceinline__ uint32_t bin_add(uint32_t *bins, uint32_t key, bool active)
{
    const uint32_t lane = threadIdx.x & 63u;
    unsigned long long todo = __builtin_amdgcn_ballot_w64(active);
    uint32_t slot = 0;
    while (todo) {
        const int leader = __ffsll(todo) - 1;
        const uint32_t lk = (uint32_t)__builtin_amdgcn_readlane((int)key, leader);
        const unsigned long long same = __builtin_amdgcn_ballot_w64(active && key == lk);
        uint32_t base = 0;
        if ((int)lane == leader)
            base = atomicAdd(bins + lk, (uint32_t)__popcll(same));
        base = (uint32_t)__builtin_amdgcn_readlane((int)base, leader);
        if (active && key == lk)
            slot = base + (uint32_t)__popcll(same & ((1ull << lane) - 1ull));
        todo &= ~same;
    }
    return slot;
}

__device__ __forceinline__ uint32_t query_bin(const Query &q, uint32_t bins_x, uint32_t bins_y)
{
    if (q.cx == ~0u)
        return bins_x * bins_y;
    return min(q.cy >> 5, bins_y - 1u) * bins_x + min(q.cx >> 5, bins_x - 1u);
}

struct GroupRec { // == stepth_group
    uint32_t x, y, w, h;
    uint8_t value[3];
    uint8_t pad;
};

// uniform tiles larger than 1x1: value = per-channel floor(mean) (the oracle's stand-in for
// disage's MeanBrightnessHasher grouping, src/depth_image.rs:104-109; parity unpinned)
__global__ __launch_bounds__(256) void prep_tiles_kernel(PrepArgs a)
{
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x, f = blockIdx.y;
    if (g >= a.n[f])
        return;
    const uint32_t gy0 = a.row_begin / a.tile_h; // strips are tile-aligned (checked on host)
    const uint32_t tyi = gy0 + g / a.gx, txi = g % a.gx;
    const uint32_t x = txi * a.tile_w, y = tyi * a.tile_h;
    const uint32_t w = min(a.tile_w, a.w - x), h = min(a.tile_h, a.h - y);
    uint32_t sum[3] = {0, 0, 0};
    for (uint32_t yy = y; yy < y + h; ++yy) {
        const uint8_t *row = a.main_px[f] + (size_t)yy * a.mstride;
        for (uint32_t xx = x; xx < x + w; ++xx) {
            const uint8_t *p = row + (size_t)xx * a.mch;
            sum[0] += p[0];
            sum[1] += p[1];
            sum[2] += p[2];
        }
    }
    const uint32_t cnt = w * h;
    const uint8_t v[3] = {(uint8_t)(sum[0] / cnt), (uint8_t)(sum[1] / cnt),
                          (uint8_t)(sum[2] / cnt)};
    Query q;
    q.cx = (x + w) / 2u; // depth_image.rs:115
    q.cy = (y + h) / 2u; // depth_image.rs:116
    if (!a.rp.owns(y))
        q.cx = q.cy = ~0u; // another rank's tile row
    make_box(v, a.prec, q.lo, q.hi);
    a.out[(size_t)f * a.q_stride + g] = q;
}

__global__ __launch_bounds__(256) void prep_list_kernel(PrepArgs a)
{
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x, f = blockIdx.y;
    if (g >= a.n[f])
        return; // (bin_add below works on whatever lanes are left)
    const GroupRec r = reinterpret_cast<const GroupRec *>(a.groups[f])[g];
    Query q;
    if (r.y >= a.row_begin && r.y < a.row_end) {
        q.cx = (r.x + r.w) / 2u; // u32 arithmetic like the reference (wraps on overflow)
        q.cy = (r.y + r.h) / 2u;
    } else {
        q.cx = q.cy = ~0u;
    }
    make_box(r.value, a.prec, q.lo, q.hi);
    a.out[(size_t)f * a.q_stride + g] = q;
    if (a.bin_counts)
        (void)bin_add(a.bin_counts + (size_t)f * a.n_bins, query_bin(q, a.bins_x, a.bins_y), true);
}

struct BinArgs {
    const Query *in; // unsorted queries, frame f's at in + f * q_stride
    Query *out;      // sorted by bin
    uint32_t *ids;   // ids[f * q_stride + k] = index of sorted query k in `in` (= its group)
    uint32_t *bins;  // per frame n_bins counts -> start offsets (bin_scan) -> fill cursors (scatter)
    uint32_t q_stride, bins_x, bins_y, n_bins;
    uint32_t n[kMaxBatch];
};

// exclusive prefix sum over the bins of one frame (one workgroup per frame)
__global__ __launch_bounds__(256) void bin_scan_kernel(BinArgs a)
{
    __shared__ uint32_t s_part[256];
    uint32_t *bins = a.bins + (size_t)blockIdx.x * a.n_bins;
    const uint32_t per = (a.n_bins + 255u) / 256u, lo = threadIdx.x * per, hi = min(lo + per, a.n_bins);
    uint32_t sum = 0;
    for (uint32_t i = lo; i < hi; ++i)
        sum += bins[i];
    s_part[threadIdx.x] = sum;
    __syncthreads();
    uint32_t base = 0;
    for (uint32_t t = 0; t < threadIdx.x; ++t)
        base += s_part[t];
    for (uint32_t i = lo; i < hi; ++i) {
        const uint32_t c = bins[i];
        bins[i] = base;
        base += c;
    }
}

__global__ __launch_bounds__(256) void bin_scatter_kernel(BinArgs a)
{
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x, f = blockIdx.y;
    const bool active = g < a.n[f];
    Query q = Query{~0u, ~0u, 0u, 0u};
    if (active)
        q = a.in[(size_t)f * a.q_stride + g];
    const uint32_t at = bin_add(a.bins + (size_t)f * a.n_bins, query_bin(q, a.bins_x, a.bins_y), active);
    if (active) {
        a.out[(size_t)f * a.q_stride + at] = q;
        a.ids[(size_t)f * a.q_stride + at] = g;
    }
}

// ---------------------------------------------------------------------------------------
// PASS 1 (src/depth_image.rs:112-123) with helpers::distance_dot_array inlined: two kernels.
//
// The reference walks rings s = 0,1,2,.. around `middle` and, inside a ring, row y+s, row
// y-s, column x+s, column x-s, each ascending (helpers.rs:26-31), and returns the first cell
// inside the colour box.  Give every cell the key ring|side|position of that scan: the answer
// is the matching cell with the smallest key, so cells may be examined in ANY order and most
// of them need not be examined at all.
//
//   ring0_kernel   dense, one lane per query (blockIdx.y = frame of a batch): tests the centre cell (ring 0) -- on natural
//                  images a large share of the queries ends here -- writes the result 0 for
//                  every query and compacts the survivors into `pending` (one wave-aggregated
//                  atomic per wave).
//   ring_search_kernel   persistent waves, ONE query per wave at a time, queries dealt
//                  round-robin (wave w takes pending[w], pending[w+NW], ..): the work per
//                  query is wildly uneven (0.2 .. 30 steps), interleaving gives every wave
//                  the same mix and keeps all concurrently running waves inside one narrow
//                  band of the image (L2 locality).  The 8x8 blocks around the centre block
//                  are visited along a square spiral, 64 blocks per step (one lane each,
//                  offsets from an LDS-staged table, bounds prefetched one step ahead); a
//                  block whose channel bounds BB miss the colour box cannot hold a match and
//                  is dropped after one 8-byte read -- 1/64 of the reference's probes.  Every
//                  surviving block costs one step: 64 lanes read its 64 cells (256 contiguous
//                  bytes, two blocks in flight), test them and, only if some cell matched,
//                  compute keys and keep the smallest.
//                  All cells of rings s <= 8 M lie in block-rings <= M, so once block-rings
//                  0..M are done and the best key's ring is <= 8 M no unvisited cell can beat
//                  it: stop.  Blocks whose nearest cell lies beyond the best ring are skipped.
//
// A ring search ends at the first ring with no in-bounds cell (helpers.rs:49-51): ring
// max(cx, Wa-1-cx, cy, Ha-1-cy)+1 for an in-bounds centre, and immediately (None) for a
// centre outside the array -- both folded into s_end.
// ---------------------------------------------------------------------------------------
struct RingArgs {
    const uint32_t *T;
    const uint2 *BB;
    const uint2 *SB; // superblock (8x8 blocks) bounds, [sbh][sbw]
    uint32_t wa, ha, bw, bh, sbw, sbh;
    // QMODE 0: one query per main pixel (uniform 1x1 tiles), read straight from the image;
    // one image per frame (blockIdx.y)
    const uint8_t *main_px[kMaxBatch];
    size_t mstride;
    uint32_t mch, w, row0, row1;
    uint32_t ntx, ty0, n_slots; // 2x2 tiles of the strip: columns, first tile row, slots
    RowPattern rp;              // with a pattern: slots walk the owned row blocks only
    // QMODE 1: prepared queries, frame f's at queries + f * q_stride, nq[f] of them
    const Query *queries;
    uint32_t q_stride;
    uint32_t nq[kMaxBatch];
    uint32_t n; // queries per frame (QMODE 1: the largest nq[f])
    const uint32_t *query_ids; // optional (ring_list_kernel): queries are sorted, results go to these slots
    uint32_t max_steps;
    uint32_t id_bits; // pending ids: query id in the low id_bits, frame above (32: one frame)
    uint8_t prec[3];
    uint8_t *raw;      // n: (d as u8), depth_image.rs:121
    uint32_t *raw32;   // optional: d
    uint32_t *pos;     // optional: winning cell (x,y)
    uint8_t *found;    // optional: Some/None
    uint32_t *max_word; // atomicMax of raw, depth_image.rs:124
    unsigned long long *stats; // optional [ring0_hits, probes]
    const uint32_t *spiral;      // block-spiral slots, spiral_chunks*64 entries (host-built)
    const uint32_t *spiral_done; // per chunk k: complete block-rings after slots [0,64(k+1))
    uint32_t pos_bits;           // key layout, see scan_key
    uint32_t *pending;           // n: ids of the queries that missed at ring 0 (compacted by
                                 // ring0_kernel), and their records {cx,cy,lo,hi}
    Query *pending_rec;
    uint32_t *pending_count;     // 1 word, zeroed before ring0_kernel
    uint32_t *pending_next;      // kParts counters kPartStride words apart, zeroed: runs handed out
    uint32_t t_stride, raw_stride; // per-frame strides of T (dwords) and raw (bytes)
    bool heavy_first;              // host side: STEPTH_FLAG_HEAVY_FIRST (launch_tile)
};

constexpr uint32_t kNoKey = 0xFFFFFFFFu;
#ifndef STEPTH_RUN
#define STEPTH_RUN 64
#endif
// <= 80 SGPRs keeps 8 workgroups of 256 threads resident per CU (MI355X_MICROARCH.md,
// "Residency"); the kernel wants ~100 without the cap
#ifndef STEPTH_SEARCH_ATTR
#define STEPTH_SEARCH_ATTR __attribute__((amdgpu_num_sgpr(80)))
#endif
constexpr uint32_t kRun = STEPTH_RUN; // consecutive pending queries a wave takes at a time
#ifndef STEPTH_PARTS
#define STEPTH_PARTS 64
#endif
constexpr uint32_t kParts = STEPTH_PARTS; // run counters
constexpr uint32_t kPartStride = 32; // words between counters (128-byte lines)

__device__ __forceinline__ uint32_t bcast(uint32_t v, int src_lane)
{
    return (uint32_t)__builtin_amdgcn_readlane((int)v, src_lane);
}

// wave-wide unsigned min in 4 DPP steps + 4 readlanes (no LDS crossbar):
// quad xor 1, quad xor 2, row rotate 4, row rotate 8 leave every 16-lane row reduced.
__device__ __forceinline__ uint32_t wave_min_dpp(uint32_t v)
{
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0xB1, 0xF, 0xF, false));
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x4E, 0xF, 0xF, false));
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x124, 0xF, 0xF, false));
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x128, 0xF, 0xF, false));
    return min(min(bcast(v, 0), bcast(v, 16)), min(bcast(v, 32), bcast(v, 48)));
}

// Order-preserving key of a cell in the reference's scan (helpers.rs:26-31): ring
// s = max(|dx|,|dy|) first, then the side (row y+s, row y-s, column x+s, column x-s), then
// the position along that side (ascending).  A corner belongs to the row that reaches it
// first.  Layout: position in bits [pb-1:0] (<= 2 (s_cap-1)), side in [pb+1:pb], ring above;
// pb is chosen on the host from the largest ring the launch can reach (pb = 9 for the
// reference's max = 255).
__device__ __forceinline__ uint32_t scan_key(int dx, int dy, uint32_t pb, uint32_t &ring)
{
    const int ndx = -dx, ndy = -dy;
    const int s = max(max(dx, ndx), max(dy, ndy));
    const bool top = dy == s, bot = ndy == s;
    const uint32_t side = top ? 0u : bot ? 1u : dx == s ? 2u : 3u;
    const int pos = ((top || bot) ? dx : dy) + s;
    ring = (uint32_t)s;
    return ((((uint32_t)s << 2) | side) << pb) | (uint32_t)pos;
}

__device__ __forceinline__ void key_to_offset(uint32_t key, uint32_t pb, int &dx, int &dy)
{
    const int s = (int)(key >> (pb + 2u)), side = (int)((key >> pb) & 3u);
    const int off = (int)(key & ((1u << pb) - 1u)) - s;
    dx = side < 2 ? off : side == 2 ? s : -s;
    dy = side == 0 ? s : side == 1 ? -s : off;
}

// Block spiral: slot u -> block offset (bi, bj) from the centre block.  Block-ring m holds
// slots [(2m-1)^2, (2m+1)^2), walked as a pinwheel of four sides of 2m blocks; slot 0 is the
// centre block.  The table is query independent and host-built (entry = (bi & 0xffff) |
// (bj << 16), then one word per 64-slot chunk: complete block-rings after that chunk); its
// first kSpiralChunks chunks -- enough for max_steps <= 258, the reference uses 255 -- are
// staged in LDS, later chunks are read from global memory.
constexpr uint32_t kSpiralRings = 33;                                   // block-rings 0..33
constexpr uint32_t kSpiralSlots = (2 * kSpiralRings + 1) * (2 * kSpiralRings + 1); // 4489
constexpr uint32_t kSpiralChunks = (kSpiralSlots + 63) / 64;            // 71

STEPTH_HD void block_slot(uint32_t u, int &i, int &j)
{
    if (u == 0) {
        i = j = 0;
        return;
    }
    const int m = (int)((isqrt_floor(u) + 1u) >> 1);
    const int t = (int)u - (2 * m - 1) * (2 * m - 1);
    const int side = t / (2 * m), o = t - side * 2 * m;
    i = side == 0 ? -m + o : side == 1 ? m : side == 2 ? -m + 1 + o : -m;
    j = side == 0 ? -m : side == 1 ? -m + o : side == 2 ? m : -m + 1 + o;
}

// number of complete block-rings once slots [0, done) are processed: largest m with
// (2m+1)^2 <= done
STEPTH_HD uint32_t rings_done(uint32_t done) { return (isqrt_floor(done) - 1u) >> 1; }

// Query `slot` of a launch -> its id q (index into raw), centre and colour box.
// QMODE 0 (one query per main pixel): slots walk the image in 2x2 tiles anchored at odd
// coordinates -- the four pixels (2tx-1..2tx, 2ty-1..2ty) all have middle = ((x+1)/2,
// (y+1)/2) = (tx, ty) (depth_image.rs:114-117 with a 1x1 group) -- so queries that share a
// search centre sit next to each other in the pending list and ring_search_kernel can serve
// them together.  Returns false for slots outside the image / strip.
template <int QMODE>
__device__ __forceinline__ bool load_query(const RingArgs &a, uint32_t f, uint32_t slot, uint32_t &q,
                                           uint32_t &cx, uint32_t &cy, uint32_t &lo,
                                           uint32_t &hi)
{
    if constexpr (QMODE == 0) {
        const uint32_t tile = slot >> 2, m = slot & 3u;
        const uint32_t tx = tile % a.ntx;
        uint32_t ty = a.ty0 + tile / a.ntx, y_lo = a.row0, y_hi = a.row1;
        if (a.rp.block) {
            // k-th owned block of rows = global block rem + k*mod; it touches block/2 + 1
            // tile rows (its first and last one are shared with the neighbouring blocks, which
            // emit the rows on their side)
            const uint32_t per = a.rp.block / 2u + 1u, tr = tile / a.ntx;
            const uint32_t y0 = (a.rp.rem + (tr / per) * a.rp.mod) * a.rp.block;
            ty = y0 / 2u + tr % per;
            y_lo = max(y_lo, y0);
            y_hi = min(y_hi, y0 + a.rp.block);
        }
        const uint32_t x = 2u * tx - 1u + (m & 1u), y = 2u * ty - 1u + (m >> 1); // may wrap
        if (slot >= a.n_slots || x >= a.w || y < y_lo || y >= y_hi)
            return false;
        q = (y - a.row0) * a.w + x;
        const uint8_t *p = a.main_px[f] + (size_t)y * a.mstride + (size_t)x * a.mch;
        const uint8_t v[3] = {p[0], p[1], p[2]};
        cx = tx;
        cy = ty;
        make_box(v, a.prec, lo, hi);
        return true;
    } else {
        if (slot >= a.nq[f])
            return false;
        q = slot;
        const Query qq = a.queries[(size_t)f * a.q_stride + slot];
        cx = qq.cx;
        cy = qq.cy;
        lo = qq.lo;
        hi = qq.hi;
        return true;
    }
}

// One returning atomic on a single address costs ~11 ns chip-wide (MI355X_MICROARCH.md,
// "dequeue"), so the compaction counter is bumped once per WORKGROUP (kRing0Iters * 256
// queries), not once per wave: 2048 queries per workgroup for big launches (ITERS = 8), 512
// for small ones, which would otherwise leave most CUs without a workgroup (ITERS = 2).
constexpr uint32_t kRing0Iters = 8, kRing0ItersSmall = 2;

template <int QMODE, bool STATS, uint32_t ITERS>
__global__ __launch_bounds__(256) void ring0_kernel(RingArgs a)
{
    __shared__ uint32_t s_wave_total[4];
    __shared__ uint32_t s_base;
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    const uint32_t f = blockIdx.y; // frame
    Query rec[ITERS];
    uint32_t qid[ITERS];
    uint32_t pend_bits = 0, n_mine = 0;
    unsigned long long n_hit = 0, n_inb = 0;
#pragma unroll
    for (uint32_t it = 0; it < ITERS; ++it) {
        const uint32_t slot = (blockIdx.x * ITERS + it) * 256u + threadIdx.x;
        uint32_t q = 0, cx = ~0u, cy = ~0u, lo = 0, hi = 0;
        const bool valid = load_query<QMODE>(a, f, slot, q, cx, cy, lo, hi);
        qid[it] = q;
        const bool inb = valid && cx < a.wa && cy < a.ha && a.max_steps > 0;
        bool fnd = false, pending = false;
        if (inb) {
            const uint32_t blk = (cy >> kBlkShift) * a.bw + (cx >> kBlkShift);
            const uint32_t c = a.T[(size_t)f * a.t_stride + blk * 64u + ((cy & 7u) << 3) + (cx & 7u)];
            fnd = cell_in_box(c, lo, hi);
            pending = !fnd && a.max_steps > 1;
        }
        if (valid) { // a ring-0 hit has distance 0; None is 0 too (depth_image.rs:118-121)
            a.raw[(size_t)f * a.raw_stride + q] = 0;
            if (a.raw32)
                a.raw32[q] = 0;
            if (a.pos) {
                a.pos[2 * (size_t)q] = fnd ? cx : 0u;
                a.pos[2 * (size_t)q + 1] = fnd ? cy : 0u;
            }
            if (a.found)
                a.found[q] = fnd ? 1 : 0;
        }
        rec[it] = Query{cx, cy | (f << 16), lo, hi}; // pending records carry their frame
        const unsigned long long pm = __ballot(pending);
        if (pending)
            pend_bits |= 1u << it;
        n_mine += (uint32_t)__popcll(pm); // wave-uniform: pending queries of this wave
        if constexpr (STATS) {
            n_hit += __popcll(__ballot(fnd));
            n_inb += __popcll(__ballot(inb));
        }
    }
    if (lane == 0)
        s_wave_total[wv] = n_mine;
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t tot = s_wave_total[0] + s_wave_total[1] + s_wave_total[2] + s_wave_total[3];
        s_base = tot ? atomicAdd(a.pending_count, tot) : 0u;
    }
    __syncthreads();
    uint32_t slot = s_base;
    for (uint32_t w = 0; w < wv; ++w)
        slot += s_wave_total[w];
#pragma unroll
    for (uint32_t it = 0; it < ITERS; ++it) {
        const bool pending = (pend_bits >> it) & 1u;
        const unsigned long long pm = __ballot(pending);
        if (pending) {
            const uint32_t at = slot + (uint32_t)__popcll(pm & ((1ull << lane) - 1ull));
            // the id carries the frame in its top bits (batches): ring_finish_kernel then needs
            // 8 bytes per pending query, not the 16-byte record as well
            a.pending[at] = a.id_bits < 32u ? qid[it] | (f << a.id_bits) : qid[it];
            a.pending_rec[at] = rec[it];
        }
        slot += (uint32_t)__popcll(pm);
    }
    if constexpr (STATS) {
        if (lane == 0) {
            atomicAdd(&stats_replica(a.stats)[0], n_hit);
            atomicAdd(&stats_replica(a.stats)[1], n_inb);
        }
    }
}

// The search kernel gets its own small argument block (few live scalars: it runs at 8
// workgroups per CU only below ~80 SGPRs) and writes nothing but the winning key of every
// pending query; ring_finish_kernel turns keys into distances.
struct SearchArgs {
    const uint32_t *T;
    const uint2 *BB;
    const uint2 *SB;
    const Query *rec;             // pending records, compacted by ring0_kernel
    uint32_t *key;                // out: best key per pending record (kNoKey = None)
    const uint32_t *count;        // number of pending records
    uint32_t *next;               // kParts run counters
    const uint32_t *spiral;       // block-spiral table (FAR only)
    const uint32_t *spiral_done;
    unsigned long long *stats;
    uint32_t wa, ha, bw, bh, sbw, max_steps, pos_bits;
    uint32_t t_stride, bb_stride, sb_stride; // per-frame strides; a record's frame = cy >> 16
};

__device__ __forceinline__ unsigned long long ballot(bool p)
{
    return __builtin_amdgcn_ballot_w64(p);
}

// FAR = false: max_steps <= 258 (the reference uses 255): near step + superblock-guided far
// phase.  FAR = true: longer searches walk the whole block spiral from a global table.
template <bool STATS, bool FAR>
__global__ __launch_bounds__(256) STEPTH_SEARCH_ATTR void ring_search_kernel(SearchArgs a)
{
    const uint32_t lane = threadIdx.x & 63u;
    const int lx = (int)(lane & 7u), ly = (int)(lane >> 3);
    const uint32_t lane_t = ((lane & 7u) << 3) | (lane >> 3); // transposed cell index
    const uint32_t ring_shift = a.pos_bits + 2u;
    const uint32_t count = *a.count;
    unsigned long long n_probe = 0, n_near = 0, n_survey = 0, n_super = 0, n_blocks = 0;

    // first step of the block spiral (slots 0..63 = block-rings 0..3 and a bit of ring 4):
    // the same 64 blocks for every query, one per lane, kept in registers
    int nbi, nbj;
    block_slot(lane, nbi, nbj);

    // Runs of kRun consecutive pending records are dealt from kParts interleaved partitions
    // (run r belongs to partition r % kParts), each with its own counter on its own cache
    // line: a single counter would cap the whole chip at ~88 grabs/us.  A wave starts at
    // partition wave % kParts and moves on when it is empty.
    // long runs amortise the grab, short runs balance small launches (a row strip of 1/8
    // frame has only ~24 records per wave): aim at >= 8 runs per resident wave, runs of
    // 4..kRun records (multiples of 4: the 2x2 tiles put up to 4 records on one centre)
    const uint32_t run = min(kRun, max(4u, (count / (8192u * 8u)) & ~3u));
    const uint32_t n_runs = (count + run - 1u) / run;
    const uint32_t wave_id =
        (uint32_t)__builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4u + (threadIdx.x >> 6)));
    static_assert(kParts <= 64, "one lane per partition in the refill scan");
    uint32_t part = wave_id % kParts;
    for (;;) {
        const uint32_t part_runs = (n_runs + kParts - 1u - part) / kParts;
        uint32_t idx = 0;
        if (lane == 0)
            idx = atomicAdd(a.next + part * kPartStride, 1u);
        idx = (uint32_t)__builtin_amdgcn_readfirstlane((int)idx);
        if (idx >= part_runs) {
            // this partition is exhausted: look at all of them at once (lane p reads counter
            // p) and move to the next one that still has runs; none left -> done
            bool has = false;
            if (lane < kParts) {
                const uint32_t taken = __hip_atomic_load(a.next + lane * kPartStride,
                                                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                has = taken < (n_runs + kParts - 1u - lane) / kParts;
            }
            const unsigned long long live = ballot(has);
            if (!live)
                break;
            // first live partition after `part`, cyclically
            const unsigned long long after = part + 1u < 64u ? live >> (part + 1u) : 0ull;
            part = after ? part + 1u + (uint32_t)(__ffsll(after) - 1) : (uint32_t)(__ffsll(live) - 1);
            continue;
        }
        const uint32_t run0 = (idx * kParts + part) * run;
        const uint32_t run_len = min(run, count - run0);
        // one coalesced read brings the whole run: lane i holds record run0 + i
        Query rec = Query{0u, 0u, 0u, 0u};
        if (lane < run_len)
            rec = a.rec[run0 + lane];
        uint32_t result = kNoKey; // lane i: best key of record run0 + i

        // Records with the same search centre (the 2x2 pixel tiles of the 1x1 mode put up to
        // four of them side by side) are served together: one set of block tests and one read
        // of every probed block for all their colour boxes.
        unsigned long long starts;
        {
            const uint32_t pcx = (uint32_t)__shfl_up((int)rec.cx, 1, 64);
            const uint32_t pcy = (uint32_t)__shfl_up((int)rec.cy, 1, 64);
            starts = ballot(lane < run_len && (FAR || lane == 0 || rec.cx != pcx || rec.cy != pcy));
        }
        for (uint32_t r = 0; r < run_len;) {
            const unsigned long long later = (starts >> r) >> 1; // group starts after r
            const uint32_t till = later ? (uint32_t)__ffsll(later) : run_len - r;
            const uint32_t M = min(min(till, run_len - r), 4u); // members r .. r+M-1
            const uint32_t qx = bcast(rec.cx, (int)r), qyf = bcast(rec.cy, (int)r);
            const uint32_t qy = qyf & 0xffffu, frame = qyf >> 16;
            // frame offsets stay 32-bit indices (the whole batch holds < 2^30 cells)
            const uint32_t tf = frame * a.t_stride, bf = frame * a.bb_stride;
            const uint32_t sf = frame * a.sb_stride;
            const uint32_t smax = max(max(qx, a.wa - 1u - qx), max(qy, a.ha - 1u - qy));
            const uint32_t s_end = min(a.max_steps, smax + 1u); // rings 0..s_end-1 are searched
            const int cbx = (int)(qx >> kBlkShift), cby = (int)(qy >> kBlkShift);
            const int nox = -(int)(qx & 7u), noy = -(int)(qy & 7u);

            // per member: colour box, best key so far, and `limit`: a block whose nearest cell
            // is at ring >= limit is useless to it (0 = member settled / absent)
            uint32_t qlo[4], qhi[4], best[4], limit[4];
#pragma unroll
            for (uint32_t m = 0; m < 4; ++m) {
                const int src = (int)min(r + m, run_len - 1u);
                qlo[m] = bcast(rec.lo, src);
                qhi[m] = bcast(rec.hi, src);
                best[m] = kNoKey;
                limit[m] = m < M ? s_end : 0u;
            }
            uint32_t glimit = s_end; // max of the members' limits

            // box [lo,hi] meets [bmin,bmax] on every channel
            auto meets = [&](uint2 bb, uint32_t lo, uint32_t hi) {
                return box_meets(bb.x, bb.y, lo, hi);
            };
            // some member that can still use a block at distance `d` has a box meeting `bb`
            auto wanted = [&](uint2 bb, uint32_t d) {
                bool w = false;
#pragma unroll
                for (uint32_t m = 0; m < 4; ++m)
                    w |= d < limit[m] && meets(bb, qlo[m], qhi[m]);
                return w;
            };
            // smallest ring of any cell of block (bi,bj): max over axes of max(8b-o, o-8b-7, 0)
            auto block_dmin = [&](int bi, int bj) {
                const int ax = (bi << 3) + nox, ay = (bj << 3) + noy;
                return (uint32_t)max(max(max(ax, -7 - ax), max(ay, -7 - ay)), 0);
            };
            // Most blocks lie entirely inside ONE side region of the reference's scan (all
            // their cells on "row y+s" sides, or all on "row y-s", "column x+s", "column x-s"
            // sides).  There the scan order inside the block is: nearest line first, then
            // ascending along the line -- so if the lanes read the cells in that order (rows
            // or columns, forwards or backwards: `cell_of`), a member's smallest key sits in
            // the lowest hit lane and no wave-min is needed.  Blocks on the diagonals (type 4)
            // keep the natural order and pay for the reduction.
            // info = bi (14 bits) | bj (14 bits) | type << 28, per lane (lane = block)
            auto block_info = [&](int bi, int bj) {
                const int x0 = (bi << 3) + nox, y0 = (bj << 3) + noy; // offsets of cell (0,0)
                const int x1 = x0 + 7, y1 = y0 + 7;
                const int mx = max(-x0, x1), my = max(-y0, y1);       // max |dx|, max |dy|
                const uint32_t type = (y0 > 0 && y0 >= mx)    ? 0u   // rows below:    side 0
                                      : (y1 < 0 && -y1 >= mx) ? 1u   // rows above:    side 1
                                      : (x0 > 0 && x0 > my)   ? 2u   // columns right: side 2
                                      : (x1 < 0 && -x1 > my)  ? 3u   // columns left:  side 3
                                                              : 4u;
                return ((uint32_t)bi & 0x3fffu) | (((uint32_t)bj & 0x3fffu) << 14) | (type << 28);
            };
            // byte offset of block (bx,by) of this frame in T, with the block's type in bits 0..2
            auto block_off = [&](int bx, int by, uint32_t info) {
                return ((tf + ((uint32_t)by * a.bw + (uint32_t)bx) * 64u) << 2) | (info >> 28);
            };
            // cell (row-major index in the block) read by this lane for a block of `type`
            auto cell_of = [&](uint32_t type) {
                const uint32_t base = (type & 2u) ? lane_t : lane; // 2, 3: columns
                // 1: rows bottom-up (^56), 3: columns right-to-left (^7), else forwards
                return base ^ (uint32_t)((0x07003800ull >> (8u * type)) & 0xffu);
            };
            // all lanes hold cell `cell` of block (sbi,sbj) (type, nearest ring d): test it for
            // every member the block can still help; keys only on a hit
            auto resolve = [&](int sbi, int sbj, uint32_t type, uint32_t d, uint32_t cell,
                               uint32_t c) -> bool {
                // hit masks per member (wave-uniform 64-bit values)
                unsigned long long h[4];
#pragma unroll
                for (uint32_t m = 0; m < 4; ++m)
                    h[m] = d < limit[m] ? ballot(cell_in_box(c, qlo[m], qhi[m])) : 0ull;
                if (!(h[0] | h[1] | h[2] | h[3]))
                    return false;
                uint32_t ring;
                const uint32_t key = scan_key(sbi * 8 + (int)(cell & 7u) + nox,
                                              sbj * 8 + (int)(cell >> 3) + noy, a.pos_bits, ring);
                const unsigned long long in_reach = ballot(ring < s_end);
                bool changed = false;
#pragma unroll
                for (uint32_t m = 0; m < 4; ++m) {
                    const unsigned long long hm = h[m] & in_reach;
                    if (hm) {
                        const uint32_t k = type < 4u
                                               ? bcast(key, __ffsll(hm) - 1)
                                               : wave_min_dpp(((hm >> lane) & 1ull) ? key : kNoKey);
                        best[m] = min(best[m], k);
                        limit[m] = min(limit[m], (best[m] >> ring_shift) + 1u);
                        changed = true;
                    }
                }
                return changed;
            };
            // One step of cell probes per surviving block (two in flight): 64 lanes read its
            // 64 cells (256 contiguous bytes).  `info,dmin` are per lane (lane = block).
            // `boff` = byte offset of the lane's block in T (a multiple of 256) | its type.
            auto probe_blocks = [&](unsigned long long pm, uint32_t info, uint32_t dmin,
                                    uint32_t boff) {
                while (pm) {
                    const int b0 = __ffsll(pm) - 1;
                    pm &= pm - 1;
                    const bool two = pm != 0;
                    const int b1 = two ? __ffsll(pm) - 1 : b0;
                    pm &= pm - 1; // no-op when pm == 0
                    const uint32_t o0 = bcast(boff, b0), o1 = bcast(boff, b1);
                    const uint32_t t0 = o0 & 7u, t1 = o1 & 7u;
                    const uint32_t d0 = bcast(dmin, b0), d1 = bcast(dmin, b1);
                    const uint32_t cell0 = cell_of(t0), cell1 = cell_of(t1);
                    const char *tb = reinterpret_cast<const char *>(a.T);
                    const uint32_t c0 = *reinterpret_cast<const uint32_t *>(
                        tb + (size_t)(o0 & ~255u) + (cell0 << 2));
                    const uint32_t c1 = *reinterpret_cast<const uint32_t *>(
                        tb + (size_t)(o1 & ~255u) + (cell1 << 2));
                    // block coordinates are only needed for the keys of a hit
                    const uint32_t n0 = bcast(info, b0), n1 = bcast(info, b1);
                    const int i0 = (int)(n0 << 18) >> 18, j0 = (int)(n0 << 4) >> 18;
                    const int i1 = (int)(n1 << 18) >> 18, j1 = (int)(n1 << 4) >> 18;
                    bool changed = resolve(i0, j0, t0, d0, cell0, c0);
                    if (two)
                        changed |= resolve(i1, j1, t1, d1, cell1, c1);
                    if constexpr (STATS) {
                        n_probe += two ? 128ull : 64ull;
                        n_blocks += two ? 2ull : 1ull;
                    }
                    if (changed) {
                        // everything farther than every member's best ring is now useless,
                        // including the rest of this step's survivors
                        glimit = max(max(limit[0], limit[1]), max(limit[2], limit[3]));
                        pm &= ballot(dmin < glimit);
                    }
                }
            };

            if constexpr (FAR) {
                const uint32_t m_end = (s_end + 6u) >> 3; // last block-ring with a ring < s_end
                const uint32_t u_end = (2u * m_end + 1u) * (2u * m_end + 1u);
                for (uint32_t u0 = 0, k = 0; u0 < u_end; u0 += 64u, ++k) {
                    const uint32_t e = a.spiral[u0 + lane];
                    const int bi = (int)(short)(e & 0xffffu), bj = (int)e >> 16;
                    const int bx = cbx + bi, by = cby + bj;
                    const uint32_t dmin = block_dmin(bi, bj);
                    bool cand = u0 + lane < u_end && (uint32_t)bx < a.bw &&
                                (uint32_t)by < a.bh && dmin < glimit;
                    if (cand)
                        cand = wanted(a.BB[bf + (uint32_t)by * a.bw + (uint32_t)bx], dmin);
                    if constexpr (STATS) {
                        n_probe += 64ull;
                        ++n_near;
                    }
                    const uint32_t info = block_info(bi, bj);
                    probe_blocks(ballot(cand), info, dmin, block_off(bx, by, info));
                    // block-rings 0..m_done are complete: every cell of rings <= 8 m_done has
                    // been seen or excluded
                    if (best[0] != kNoKey && (best[0] >> ring_shift) <= 8u * a.spiral_done[k])
                        break;
                }
            } else {
                // ---- near: block-rings 0..3 (all cells of rings <= 24) ----
                {
                    const int bx = cbx + nbi, by = cby + nbj;
                    const uint32_t dmin = block_dmin(nbi, nbj);
                    bool cand = (uint32_t)bx < a.bw && (uint32_t)by < a.bh && dmin < glimit;
                    if (cand)
                        cand = wanted(a.BB[bf + (uint32_t)by * a.bw + (uint32_t)bx], dmin);
                    if constexpr (STATS) {
                        n_probe += 64ull;
                        ++n_near;
                    }
                    const uint32_t info = block_info(nbi, nbj);
                    probe_blocks(ballot(cand), info, dmin, block_off(bx, by, info));
                }
                // a member whose best ring is <= 24 is settled; the others need the far phase
#pragma unroll
                for (uint32_t m = 0; m < 4; ++m)
                    if (s_end <= 25u || (best[m] != kNoKey && (best[m] >> ring_shift) <= 24u))
                        limit[m] = 0u;
                glimit = max(max(limit[0], limit[1]), max(limit[2], limit[3]));
                if (glimit) {
                    // ---- far: survey the 64x64-cell superblocks overlapping the window of
                    // radius glimit-1 -- only those whose bounds meet a box can hold a match
                    // -- then one step per surviving superblock, nearest first: its 64 blocks.
                    const int rad = (int)glimit - 1;
                    const int sx0 = max((int)qx - rad, 0) >> 6, sy0 = max((int)qy - rad, 0) >> 6;
                    const int sx1 = min((int)qx + rad, (int)a.wa - 1) >> 6;
                    const int sy1 = min((int)qy + rad, (int)a.ha - 1) >> 6;
                    const int nsx = sx1 - sx0 + 1, nsy = sy1 - sy0 + 1; // <= 10 each
                    // lanes tile the window: 8 x 8 superblocks in one round when it fits (any
                    // window of radius <= 223), else 16 x 4 per round, 3 rounds cover 16 x 12
                    const uint32_t gsh = (nsx <= 8 && nsy <= 8) ? 3u : 4u;
                    const int grows = 64 >> gsh;
                    uint32_t skey[3] = {kNoKey, kNoKey, kNoKey};
#pragma unroll
                    for (int h = 0; h < 3; ++h) {
                        const int tx = (int)(lane & ((1u << gsh) - 1u)), ty = (int)(lane >> gsh) + grows * h;
                        if (tx < nsx && ty < nsy) {
                            const int sx = sx0 + tx, sy = sy0 + ty;
                            const int ax = (sx << 6) - (int)qx, ay = (sy << 6) - (int)qy;
                            const uint32_t dS =
                                (uint32_t)max(max(max(ax, -63 - ax), max(ay, -63 - ay)), 0);
                            if (dS < glimit &&
                                wanted(a.SB[sf + (uint32_t)sy * a.sbw + (uint32_t)sx], dS))
                                skey[h] = (dS << 8) | (uint32_t)(ty * 16 + tx);
                        }
                        if (grows * (h + 1) >= nsy)
                            break;
                    }
                    if constexpr (STATS) {
                        n_probe += 64ull;
                        ++n_survey;
                    }
                    for (;;) {
                        // nearest surviving superblock that can still improve some best key
                        uint32_t k = min(min(skey[0], skey[1]), skey[2]);
                        k = (k >> 8) < glimit ? k : kNoKey;
                        if (ballot(k != kNoKey) == 0)
                            break;
                        const uint32_t pick = wave_min_dpp(k);
                        const int sx = sx0 + (int)(pick & 15u), sy = sy0 + (int)((pick >> 4) & 15u);
#pragma unroll
                        for (int h = 0; h < 3; ++h)
                            skey[h] = skey[h] == pick ? kNoKey : skey[h];
                        // its 64 blocks, one lane each (block-rings 0..3 were done above)
                        const int bx = sx * 8 + lx, by = sy * 8 + ly;
                        const int bi = bx - cbx, bj = by - cby;
                        const uint32_t dmin = block_dmin(bi, bj);
                        bool cand = (uint32_t)bx < a.bw && (uint32_t)by < a.bh && dmin < glimit &&
                                    max(abs(bi), abs(bj)) > 3;
                        if (cand)
                            cand = wanted(a.BB[bf + (uint32_t)by * a.bw + (uint32_t)bx], dmin);
                        if constexpr (STATS) {
                            n_probe += 64ull;
                            ++n_super;
                        }
                        // nearest blocks first, in shells of 8 rings from the superblock's
                        // nearest cell: a hit in an early shell prunes the later ones (lane
                        // order inside a superblock says nothing about distance)
                        unsigned long long pm = ballot(cand);
                        if (!pm)
                            continue;
                        const uint32_t info = block_info(bi, bj), shell = (dmin - (pick >> 8)) >> 3;
                        const uint32_t boff = block_off(bx, by, info);
                        for (uint32_t sh = 0; pm; ++sh) {
                            const unsigned long long now = pm & ballot(shell <= sh);
                            if (now) {
                                probe_blocks(now, info, dmin, boff);
                                pm &= ~now & ballot(dmin < glimit);
                            }
                        }
                    }
                }
            }
#pragma unroll
            for (uint32_t m = 0; m < 4; ++m)
                if (m < M && lane == r + m)
                    result = best[m];
            r += M;
        }
        if (lane < run_len)
            a.key[run0 + lane] = result;
    }
    if constexpr (STATS) {
        if (lane == 0) {
            atomicAdd(&stats_replica(a.stats)[1], n_probe);
            atomicAdd(&stats_replica(a.stats)[2], n_near);
            atomicAdd(&stats_replica(a.stats)[3], n_survey);
            atomicAdd(&stats_replica(a.stats)[4], n_super);
            atomicAdd(&stats_replica(a.stats)[5], n_blocks);
        }
    }
}

// keys -> distances: helpers.rs:3-7 (floor sqrt), depth_image.rs:121 (`d as u8` wraps),
// :124 (max).  Dense over the pending records; None keeps the 0 written by ring0_kernel.
struct FinishArgs {
    const uint32_t *pending; // query ids
    const Query *rec;
    const uint32_t *key;
    const uint32_t *count;
    uint32_t pos_bits;
    uint32_t id_bits; // pending ids: query id in the low id_bits, frame above (32: no frame)
    uint8_t *raw;
    uint32_t *raw32, *pos;
    uint8_t *found;
    uint32_t *max_word;  // one word per frame
    uint32_t raw_stride; // bytes between the frames' raw maps
};

__global__ __launch_bounds__(256) void ring_finish_kernel(FinishArgs a)
{
    // The frames of a batch interleave in the pending list, and same-address global atomics
    // cost ~11 ns each chip-wide: the maxima are folded per frame in LDS and every workgroup
    // flushes its non-zero entries once at the end.
    __shared__ uint32_t s_fmax[kMaxBatch];
    if (threadIdx.x < kMaxBatch)
        s_fmax[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t count = *a.count;
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < count; i += gridDim.x * 256u) {
        const uint32_t key = a.key[i];
        if (key == kNoKey)
            continue;
        const uint32_t idw = a.pending[i];
        const uint32_t frame = a.id_bits < 32u ? idw >> a.id_bits : 0u;
        const uint32_t q = a.id_bits < 32u ? idw & ((1u << a.id_bits) - 1u) : idw;
        int dx, dy;
        key_to_offset(key, a.pos_bits, dx, dy);
        const uint32_t d = isqrt_floor((uint32_t)(dx * dx + dy * dy));
        a.raw[(size_t)frame * a.raw_stride + q] = (uint8_t)(d & 0xffu);
        if (d & 0xffu)
            atomicMax(&s_fmax[frame], d & 0xffu);
        if (a.raw32)
            a.raw32[q] = d;
        if (a.pos) {
            const Query r = a.rec[i];
            a.pos[2 * (size_t)q] = (uint32_t)((int)r.cx + dx);
            a.pos[2 * (size_t)q + 1] = (uint32_t)((int)(r.cy & 0xffffu) + dy);
        }
        if (a.found)
            a.found[q] = 1;
    }
    __syncthreads();
    if (threadIdx.x < kMaxBatch && s_fmax[threadIdx.x])
        atomicMax(&a.max_word[threadIdx.x], s_fmax[threadIdx.x]);
}

// ---------------------------------------------------------------------------------------
// PASS 2 + paint (src/depth_image.rs:125-132).  v*255/max in u64 floor == 32-bit floor here
// (v,max <= 255); a 256-entry table per workgroup replaces the per-pixel division.
// max == 0: the reference panics (:127); the table is all zero.  identity != 0: RAW_ONLY.
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ void build_lut(uint8_t *lut, const uint32_t *max_word, int identity,
                                          uint32_t *max_out = nullptr)
{
    if (max_out && blockIdx.x == 0 && threadIdx.x == 0)
        *max_out = *max_word;
    const uint32_t mx = identity ? 255u : *max_word;
    for (uint32_t v = threadIdx.x; v < 256u; v += blockDim.x)
        lut[v] = (mx == 0) ? 0 : (uint8_t)(v * 255u / mx);
    __syncthreads();
}

struct PaintArgs {
    const uint8_t *raw; // per group
    const uint32_t *max_word;
    uint32_t *max_out; // optional: the caller's copy of the max (device memory)
    int identity;
    uint8_t *out[kMaxBatch]; // [H][W] per frame (blockIdx.y), row 0 of the image
    size_t ostride;
    uint32_t w, h;
    uint32_t row_begin, row_end;
    RowPattern rp;
    uint32_t tile_w, tile_h, gx; // uniform tiles
    const void *groups_list[kMaxBatch]; // or one list per frame ...
    uint32_t n_list[kMaxBatch];         // ... of this many groups
    uint32_t n;
    uint32_t raw_stride; // bytes between the frames' raw maps
};

// uniform tiles (incl. 1x1): one thread per output pixel of the strip
__global__ __launch_bounds__(256) void paint_tiles_kernel(PaintArgs a)
{
    __shared__ uint8_t lut[256];
    const uint32_t f = blockIdx.y;
    build_lut(lut, a.max_word + f, a.identity, a.max_out ? a.max_out + f : nullptr);
    const uint8_t *raw = a.raw + (size_t)f * a.raw_stride;
    uint8_t *out = a.out[f];
    const uint32_t rows = a.row_end - a.row_begin;
    const size_t total = (size_t)rows * a.w;
    const uint32_t gy0 = a.row_begin / a.tile_h;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (size_t)gridDim.x * blockDim.x) {
        const uint32_t y = a.row_begin + (uint32_t)(i / a.w), x = (uint32_t)(i % a.w);
        if (!a.rp.owns(y))
            continue;
        const size_t g = (size_t)(y / a.tile_h - gy0) * a.gx + x / a.tile_w;
        out[(size_t)y * a.ostride + x] = lut[raw[g]];
    }
}

// 1x1 tiles, everything 4-byte aligned: 4 pixels per thread
__global__ __launch_bounds__(256) void paint_1x1_vec4_kernel(PaintArgs a)
{
    __shared__ uint8_t lut[256];
    const uint32_t f = blockIdx.y;
    build_lut(lut, a.max_word + f, a.identity, a.max_out ? a.max_out + f : nullptr);
    const uint8_t *raw = a.raw + (size_t)f * a.raw_stride;
    uint8_t *out = a.out[f];
    const uint32_t rows = a.row_end - a.row_begin, w4 = a.w >> 2;
    const size_t total = (size_t)rows * w4;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (size_t)gridDim.x * blockDim.x) {
        const uint32_t r = (uint32_t)(i / w4), c = (uint32_t)(i % w4);
        if (!a.rp.owns(a.row_begin + r))
            continue;
        const uint32_t v = reinterpret_cast<const uint32_t *>(raw)[i];
        const uint32_t o = (uint32_t)lut[v & 0xff] | ((uint32_t)lut[(v >> 8) & 0xff] << 8) |
                           ((uint32_t)lut[(v >> 16) & 0xff] << 16) |
                           ((uint32_t)lut[v >> 24] << 24);
        *reinterpret_cast<uint32_t *>(out + (size_t)(a.row_begin + r) * a.ostride + 4u * c) = o;
    }
}

// explicit group list: one wave per group, lanes stride over the rectangle.  Rectangles are
// expected to partition the image like disage's; where they overlap, which one wins is
// unspecified (the reference's serial collect() lets the later group win).
__global__ __launch_bounds__(256) void paint_list_kernel(PaintArgs a)
{
    // 16 lanes per group: the reference's groups hold a few dozen pixels at most once the image
    // has >= 2^16 of them (src/depth_image.rs:102), a whole wave per group would mostly idle
    __shared__ uint8_t lut[256];
    const uint32_t f = blockIdx.y;
    build_lut(lut, a.max_word + f, a.identity, a.max_out ? a.max_out + f : nullptr);
    const uint32_t sub = threadIdx.x & 15u;
    const uint32_t g = blockIdx.x * 16u + (threadIdx.x >> 4);
    if (g >= a.n_list[f])
        return;
    const GroupRec r = reinterpret_cast<const GroupRec *>(a.groups_list[f])[g];
    // (an empty rectangle paints nothing -- like the reference's collect(); the walk below
    // divides by the width)
    if (r.y < a.row_begin || r.y >= a.row_end || r.x >= a.w || r.w == 0u || r.h == 0u)
        return;
    const uint8_t v = lut[a.raw[(size_t)f * a.raw_stride + g]];
    const uint32_t w = min(r.w, a.w - r.x), h = min(r.h, a.h - r.y);
    uint8_t *out = a.out[f] + (size_t)r.y * a.ostride + r.x;
    // cells sub, sub + 16, ... of the rectangle in row-major order: one division, then steps
    uint32_t yy = sub / w, xx = sub - yy * w;
    while (yy < h) {
        out[(size_t)yy * a.ostride + xx] = v;
        xx += 16u;
        while (xx >= w) {
            xx -= w;
            ++yy;
        }
    }
}

// PASS 2 in place on rows the search kernel painted with raw values (group lists on
// ring_list_kernel): v -> v*255/max through the LUT, 16 bytes per lane where everything is
// aligned.  src/depth_image.rs:125-129.
__global__ __launch_bounds__(256) void lut_rows_kernel(PaintArgs a)
{
    __shared__ uint8_t lut[256];
    const uint32_t f = blockIdx.y;
    build_lut(lut, a.max_word + f, a.identity, a.max_out ? a.max_out + f : nullptr);
    uint8_t *out = a.out[f];
    const uint32_t rows = a.row_end - a.row_begin;
    auto map4 = [&](uint32_t v) {
        return (uint32_t)lut[v & 0xff] | ((uint32_t)lut[(v >> 8) & 0xff] << 8) | ((uint32_t)lut[(v >> 16) & 0xff] << 16) |
               ((uint32_t)lut[v >> 24] << 24);
    };
    if (((reinterpret_cast<uintptr_t>(out) | a.ostride | a.w) & 15u) == 0) {
        const uint32_t w16 = a.w >> 4;
        const size_t total = (size_t)rows * w16;
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
            const uint32_t r = (uint32_t)(i / w16), c = (uint32_t)(i - (size_t)r * w16);
            uint4 *p = reinterpret_cast<uint4 *>(out + (size_t)(a.row_begin + r) * a.ostride + 16u * c);
            uint4 v = *p;
            v.x = map4(v.x);
            v.y = map4(v.y);
            v.z = map4(v.z);
            v.w = map4(v.w);
            *p = v;
        }
        return;
    }
    const size_t total = (size_t)rows * a.w;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        uint8_t *p = out + (size_t)(a.row_begin + (uint32_t)(i / a.w)) * a.ostride + (i % a.w);
        *p = lut[*p];
    }
}

// The rectangles too large for the search kernel's own lanes (more than kPaintSmall pixels; the
// search queued them): raw value, one wave per rectangle.
constexpr uint32_t kPaintSmall = 64;
__global__ __launch_bounds__(256) void paint_big_kernel(PaintArgs a, const uint32_t *big_count, const uint32_t *big_list,
                                                        uint32_t cap)
{
    const uint32_t f = blockIdx.y, lane = threadIdx.x & 63u;
    const uint32_t n = min(big_count[f], cap);
    for (uint32_t k = blockIdx.x * 4u + (threadIdx.x >> 6); k < n; k += gridDim.x * 4u) {
        const uint32_t g = big_list[(size_t)f * cap + k];
        const GroupRec r = reinterpret_cast<const GroupRec *>(a.groups_list[f])[g];
        const uint8_t v = a.raw[(size_t)f * a.raw_stride + g];
        const uint32_t w = min(r.w, a.w - r.x), h = min(r.h, a.h - r.y);
        uint8_t *out = a.out[f] + (size_t)r.y * a.ostride + r.x;
        for (uint32_t yy = 0; yy < h; ++yy)
            for (uint32_t xx = lane; xx < w; xx += 64u)
                out[(size_t)yy * a.ostride + xx] = v;
    }
}

// in-place PASS 2 on already painted rows (row-strip sharding, after the cross-rank max)
struct NormArgs {
    uint8_t *data[kMaxBatch]; // one image (view) per frame, blockIdx.y
    size_t stride;
    uint32_t w, rows;
    const uint32_t *max_word;
    uint32_t first_row; // absolute row of data's row 0, for the pattern
    RowPattern rp;
};

__global__ __launch_bounds__(256) void normalise_rows_kernel(NormArgs a)
{
    __shared__ uint8_t lut[256];
    build_lut(lut, a.max_word + blockIdx.y, 0);
    uint8_t *data = a.data[blockIdx.y];
    const size_t total = (size_t)a.rows * a.w;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (size_t)gridDim.x * blockDim.x) {
        if (!a.rp.owns(a.first_row + (uint32_t)(i / a.w)))
            continue;
        uint8_t *p = data + (size_t)(i / a.w) * a.stride + (i % a.w);
        *p = lut[*p];
    }
}

// ---------------------------------------------------------------------------------------
// Gaussian same-size resample, the tail of load_depth_from_additional
// (src/depth_image.rs:133-134: DynamicImage::resize(w, h, imageops::Gaussian)).  image 0.23.14
// is not part of the reference tree; this follows its sampler as published (vertical pass,
// then horizontal pass; gaussian(x, 0.5), support 3 -> taps o-3..o+3 clipped at the borders
// and renormalised; f32 accumulation in tap order; round-half-away to u8 after each pass).
// PARITY UNPINNED: checked against the oracle bit for bit and against assets/depth.jpg only
// by tolerance.  Weights come from the host (16 border classes x 7 taps) so that CPU and GPU
// use the very same floats.
// ---------------------------------------------------------------------------------------
struct GaussArgs {
    const uint8_t *src[kMaxBatch]; // one plane per frame (blockIdx.y)
    size_t s_elem, s_line; // strides along the filtered axis / between lines
    uint32_t n, lines;
    uint8_t *dst[kMaxBatch];
    size_t d_elem, d_line;
    float w[4][4][7]; // [clipped taps before][clipped taps after][offset + 3]
};

__global__ __launch_bounds__(256) void gauss_pass_kernel(GaussArgs a)
{
// the reference accumulates t += v * w in f32 with separate roundings; HIP's __fmul_rn /
// __fadd_rn are plain operators, so contraction into FMA has to be switched off here
#pragma clang fp contract(off)
    const uint32_t total = a.n * a.lines; // sides are <= 65535: fits 32 bits
    const uint8_t *src = a.src[blockIdx.y];
    uint8_t *dst = a.dst[blockIdx.y];
    const uint32_t inner = a.d_elem == 1 ? a.n : a.lines; // extent of the contiguous direction
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        // consecutive threads run along the contiguous direction of the destination
        const uint32_t q = i / inner, rem = i - q * inner;
        const uint32_t o = a.d_elem == 1 ? rem : q;
        const uint32_t l = a.d_elem == 1 ? q : rem;
        const int cl = max(0, 3 - (int)o), cr = max(0, (int)o + 3 - (int)(a.n - 1));
        const float *w = a.w[min(cl, 3)][min(cr, 3)];
        float t = 0.0f;
        for (int k = cl - 3; k <= 3 - cr; ++k) {
            const float v = (float)src[(size_t)l * a.s_line + (size_t)((int)o + k) * a.s_elem];
            t = t + v * w[k + 3];
        }
        t = fminf(fmaxf(t, 0.0f), 255.0f);
        dst[(size_t)l * a.d_line + (size_t)o * a.d_elem] = (uint8_t)roundf(t);
    }
}

// The same two passes, four outputs per thread with dword loads and stores (width a multiple of
// 4, pointers and strides 4-byte aligned -- the host checks).  Every output accumulates its taps
// in the same order with the same weights as gauss_pass_kernel, so the bytes are identical.
struct Gauss4Args {
    const uint8_t *src[kMaxBatch];
    uint8_t *dst[kMaxBatch];
    size_t sstride, dstride; // bytes between rows
    uint32_t w, h;
    float wt[4][4][7];
};

__device__ __forceinline__ uint32_t gauss_round(float t)
{
    t = fminf(fmaxf(t, 0.0f), 255.0f);
    return (uint32_t)(uint8_t)roundf(t);
}

// vertical pass: out(x, y) = sum_k in(x, y + k) * w[k]; thread = 4 adjacent columns of one row
__global__ __launch_bounds__(256) void gauss_vertical4_kernel(Gauss4Args a)
{
#pragma clang fp contract(off)
    const uint32_t wq = a.w >> 2, total = wq * a.h;
    const uint8_t *src = a.src[blockIdx.y];
    uint8_t *dst = a.dst[blockIdx.y];
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const uint32_t y = i / wq, x = (i - y * wq) << 2;
        const int cl = max(0, 3 - (int)y), cr = max(0, (int)y + 3 - (int)(a.h - 1));
        float t0 = 0.0f, t1 = 0.0f, t2 = 0.0f, t3 = 0.0f;
        if ((cl | cr) == 0) {
            // interior rows: the one unclipped weight set, read once per wave (an index that
            // depends on the row makes every tap a per-lane load from the argument block)
#pragma unroll
            for (int k = -3; k <= 3; ++k) {
                const uint32_t v = *reinterpret_cast<const uint32_t *>(src + (size_t)((int)y + k) * a.sstride + x);
                const float wk = a.wt[0][0][k + 3];
                t0 = t0 + (float)(v & 255u) * wk;
                t1 = t1 + (float)((v >> 8) & 255u) * wk;
                t2 = t2 + (float)((v >> 16) & 255u) * wk;
                t3 = t3 + (float)(v >> 24) * wk;
            }
            *reinterpret_cast<uint32_t *>(dst + (size_t)y * a.dstride + x) =
                gauss_round(t0) | (gauss_round(t1) << 8) | (gauss_round(t2) << 16) | (gauss_round(t3) << 24);
            continue;
        }
        const float *w = a.wt[min(cl, 3)][min(cr, 3)];
        for (int k = cl - 3; k <= 3 - cr; ++k) {
            const uint32_t v = *reinterpret_cast<const uint32_t *>(src + (size_t)((int)y + k) * a.sstride + x);
            const float wk = w[k + 3];
            t0 = t0 + (float)(v & 255u) * wk;
            t1 = t1 + (float)((v >> 8) & 255u) * wk;
            t2 = t2 + (float)((v >> 16) & 255u) * wk;
            t3 = t3 + (float)(v >> 24) * wk;
        }
        *reinterpret_cast<uint32_t *>(dst + (size_t)y * a.dstride + x) =
            gauss_round(t0) | (gauss_round(t1) << 8) | (gauss_round(t2) << 16) | (gauss_round(t3) << 24);
    }
}

// horizontal pass: out(x, y) = sum_k in(x + k, y) * w[k]; thread = outputs x .. x+3 of one row,
// from the three aligned dwords around them
__global__ __launch_bounds__(256) void gauss_horizontal4_kernel(Gauss4Args a)
{
#pragma clang fp contract(off)
    const uint32_t wq = a.w >> 2, total = wq * a.h;
    const uint8_t *src = a.src[blockIdx.y];
    uint8_t *dst = a.dst[blockIdx.y];
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const uint32_t y = i / wq, q = i - y * wq, x = q << 2;
        const uint32_t *row = reinterpret_cast<const uint32_t *>(src + (size_t)y * a.sstride);
        const uint32_t d0 = q > 0 ? row[q - 1] : 0u, d1 = row[q], d2 = q + 1 < wq ? row[q + 1] : 0u;
        float b[12]; // b[j] = in(x - 4 + j, y)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            b[j] = (float)((d0 >> (8 * j)) & 255u);
            b[4 + j] = (float)((d1 >> (8 * j)) & 255u);
            b[8 + j] = (float)((d2 >> (8 * j)) & 255u);
        }
        uint32_t out = 0;
        if (x >= 3u && x + 6u < a.w) { // all four outputs unclipped: the one interior weight set
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float t = 0.0f;
#pragma unroll
                for (int k = -3; k <= 3; ++k)
                    t = t + b[4 + j + k] * a.wt[0][0][k + 3];
                out |= gauss_round(t) << (8 * j);
            }
            *reinterpret_cast<uint32_t *>(dst + (size_t)y * a.dstride + x) = out;
            continue;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int o = (int)x + j;
            const int cl = max(0, 3 - o), cr = max(0, o + 3 - (int)(a.w - 1));
            const float *w = a.wt[min(cl, 3)][min(cr, 3)];
            float t = 0.0f;
#pragma unroll
            for (int k = -3; k <= 3; ++k)
                if (k >= cl - 3 && k <= 3 - cr)
                    t = t + b[4 + j + k] * w[k + 3];
            out |= gauss_round(t) << (8 * j);
        }
        *reinterpret_cast<uint32_t *>(dst + (size_t)y * a.dstride + x) = out;
    }
}

// clears rows [row_begin,row_end) of every frame's depth map (group lists: pixels no group of
// the strip covers must read 0, like ImageBuffer::from_pixel(.., Luma([0])), depth_image.rs:20)
__global__ __launch_bounds__(256) void zero_rows_kernel(PaintArgs a)
{
    uint8_t *out = a.out[blockIdx.y];
    const uint32_t rows = a.row_end - a.row_begin;
    if (((reinterpret_cast<uintptr_t>(out) | a.ostride | a.w) & 15u) == 0) { // 16 bytes per lane
        const uint32_t w16 = a.w >> 4;
        const size_t total = (size_t)rows * w16;
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
            const uint32_t r = (uint32_t)(i / w16), c = (uint32_t)(i - (size_t)r * w16);
            *reinterpret_cast<uint4 *>(out + (size_t)(a.row_begin + r) * a.ostride + 16u * c) = make_uint4(0u, 0u, 0u, 0u);
        }
        return;
    }
    const size_t total = (size_t)rows * a.w;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (size_t)gridDim.x * blockDim.x)
        out[(size_t)(a.row_begin + (uint32_t)(i / a.w)) * a.ostride + (i % a.w)] = 0;
}

__global__ void zero_words_kernel(uint32_t *p, int n)
{
    if ((int)threadIdx.x < n)
        p[threadIdx.x] = 0;
}

} // namespace stepth
