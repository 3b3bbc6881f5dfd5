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
// What the host hands to the searches of tile_kernels.h (launch_tile: one group per pixel;
// launch_list: prepared queries).  The reference walks rings s = 0,1,2,.. around `middle` and,
// inside a ring, row y+s, row y-s, column x+s, column x-s, each ascending (helpers.rs:26-31),
// and returns the first cell inside the colour box; with the key ring|side|position of that
// scan the answer is the matching cell with the smallest key (tile_lane.h).  A ring search ends
// at the first ring with no in-bounds cell (helpers.rs:49-51): ring max(cx, Wa-1-cx, cy,
// Ha-1-cy)+1 for an in-bounds centre, and immediately (None) for a centre outside the array.
// (The round-1 kernels -- ring0 / ring_search / ring_finish, one wave per <= 4 queries walking a
// block spiral -- lived here until round 3; every reach now runs on the lane = query kernels.)
// ---------------------------------------------------------------------------------------
struct RingArgs {
    const uint32_t *T;
    uint32_t wa, ha, bw, bh, sbw, sbh;
    // one query per main pixel (launch_tile), read straight from the image; one image per frame
    const uint8_t *main_px[kMaxBatch];
    size_t mstride;
    uint32_t mch, w, row0, row1;
    RowPattern rp;
    // prepared queries (launch_list), frame f's at queries + f * q_stride, nq[f] of them
    const Query *queries;
    uint32_t q_stride;
    uint32_t nq[kMaxBatch];
    uint32_t n; // queries per frame (prepared: the largest nq[f])
    const uint32_t *query_ids; // optional: queries are sorted, results go to these slots
    uint32_t max_steps;
    uint8_t prec[3];
    uint8_t *raw;      // n: (d as u8), depth_image.rs:121
    uint32_t *raw32;   // optional: d
    uint32_t *pos;     // optional: winning cell (x,y)
    uint8_t *found;    // optional: Some/None
    uint32_t *max_word; // atomicMax of raw, depth_image.rs:124
    unsigned long long *stats;
    uint32_t t_stride, raw_stride; // per-frame strides of T (dwords) and raw (bytes)
    bool heavy_first;              // host side: STEPTH_FLAG_HEAVY_FIRST (launch_tile)
};

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
    // a workgroup takes (row, run of 256 dwords) items: the row of an item is worked out once per
    // workgroup, not per thread (a 64-bit division per dword used to be most of this kernel)
    const uint32_t rows = a.row_end - a.row_begin, w4 = a.w >> 2, runs = (w4 + 255u) >> 8;
    for (uint32_t item = blockIdx.x; item < rows * runs; item += gridDim.x) {
        const uint32_t r = item / runs, c = (item - r * runs) * 256u + threadIdx.x; // uniform r
        if (c >= w4 || !a.rp.owns(a.row_begin + r))
            continue;
        const uint32_t v = reinterpret_cast<const uint32_t *>(raw)[(size_t)r * w4 + c];
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
