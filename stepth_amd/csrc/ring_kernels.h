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
// pack_additional: RGB8 [Ha][Wa] -> P [Ha][Wp] dwords (row image) and PT [Wa][Hp] dwords
// (column image = transpose), both in the cell format of cell_pack.h, pitch-padded with
// kNoCell.  Replaces disage::converters::raw_rgb + pixels_to_array
// (src/depth_image.rs:97-100): the Vec<Vec<[u8;3]>> becomes two flat dword planes so that
// BOTH the row sides and the column sides of a ring (helpers.rs:28-31) are unit-stride reads.
// ---------------------------------------------------------------------------------------
struct PackArgs {
    const uint8_t *src;
    size_t stride;
    uint32_t wa, ha;
    uint32_t wp, hp; // pitches in dwords
    uint32_t *P, *PT;
};

__global__ __launch_bounds__(256) void pack_additional_kernel(PackArgs a)
{
    __shared__ uint32_t tile[32][33];
    const uint32_t tx = threadIdx.x & 31, ty = threadIdx.x >> 5; // 32 x 8
    const uint32_t x0 = blockIdx.x * 32, y0 = blockIdx.y * 32;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const uint32_t y = y0 + ty + 8 * r, x = x0 + tx;
        uint32_t cell = kNoCell;
        if (y < a.ha && x < a.wa) {
            const uint8_t *p = a.src + (size_t)y * a.stride + (size_t)x * 3;
            cell = pack_cell(p[0], p[1], p[2]);
        }
        if (y < a.ha && x < a.wp)
            a.P[(size_t)y * a.wp + x] = cell;
        tile[ty + 8 * r][tx] = cell;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const uint32_t x = x0 + ty + 8 * r, y = y0 + tx; // transposed write, unit stride in y
        if (x < a.wa && y < a.hp)
            a.PT[(size_t)x * a.hp + y] = tile[tx][ty + 8 * r];
    }
}

// ---------------------------------------------------------------------------------------
// Queries.  One query = one disage group (src/depth_image.rs:113-121): the search centre
// `middle` and the colour box of its value.
// ---------------------------------------------------------------------------------------
struct Query {
    uint32_t cx, cy; // middle = ((x+w)/2, (y+h)/2), depth_image.rs:114-117; ~0u = not in strip
    uint32_t lo, hi; // make_box(value, precision)
};

struct PrepArgs {
    // uniform tiles from the main image
    const uint8_t *main_px;
    size_t mstride;
    uint32_t mch, w, h;
    uint32_t tile_w, tile_h, gx, gy;
    // or an explicit list
    const void *groups; // stepth_group[n]
    uint32_t n;
    uint32_t row_begin, row_end;
    uint8_t prec[3];
    Query *out;
};

struct GroupRec { // == stepth_group
    uint32_t x, y, w, h;
    uint8_t value[3];
    uint8_t pad;
};

// uniform tiles larger than 1x1: value = per-channel floor(mean) (the oracle's stand-in for
// disage's MeanBrightnessHasher grouping, src/depth_image.rs:104-109; parity unpinned)
__global__ __launch_bounds__(256) void prep_tiles_kernel(PrepArgs a)
{
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= a.n)
        return;
    const uint32_t gy0 = a.row_begin / a.tile_h; // strips are tile-aligned (checked on host)
    const uint32_t tyi = gy0 + g / a.gx, txi = g % a.gx;
    const uint32_t x = txi * a.tile_w, y = tyi * a.tile_h;
    const uint32_t w = min(a.tile_w, a.w - x), h = min(a.tile_h, a.h - y);
    uint32_t sum[3] = {0, 0, 0};
    for (uint32_t yy = y; yy < y + h; ++yy) {
        const uint8_t *row = a.main_px + (size_t)yy * a.mstride;
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
    make_box(v, a.prec, q.lo, q.hi);
    a.out[g] = q;
}

__global__ __launch_bounds__(256) void prep_list_kernel(PrepArgs a)
{
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= a.n)
        return;
    const GroupRec r = reinterpret_cast<const GroupRec *>(a.groups)[g];
    Query q;
    if (r.y >= a.row_begin && r.y < a.row_end) {
        q.cx = (r.x + r.w) / 2u; // u32 arithmetic like the reference (wraps on overflow)
        q.cy = (r.y + r.h) / 2u;
    } else {
        q.cx = q.cy = ~0u;
    }
    make_box(r.value, a.prec, q.lo, q.hi);
    a.out[g] = q;
}

// ---------------------------------------------------------------------------------------
// ring_match: PASS 1 (src/depth_image.rs:112-123) with helpers::distance_dot_array inlined.
//
// Work decomposition
//   * one wave owns 64 consecutive queries; phase A tests every lane's ring 0 (its centre
//     cell) -- on natural images most queries end here;
//   * phase B serves the survivors one at a time with the WHOLE wave: the 64 lanes walk the
//     ring sequence in the reference's scan order (row y+s, row y-s, column x+s, column x-s,
//     each ascending -- helpers.rs:28-31), 64..128 consecutive cells per step, unit-stride
//     from P (rows) or PT (columns).  `__ballot` of the box test + lowest set bit is the
//     sequential loop's first hit, bit for bit; later cells of the step are simply ignored.
//   * rings 1..7 (252 probes) are walked with a per-lane table of (dx,dy) offsets in scan
//     order, so that small rings also fill all 64 lanes.
// A ring search ends at the first ring with no in-bounds cell (helpers.rs:49-51), which for an
// in-bounds centre is ring max(cx, Wa-1-cx, cy, Ha-1-cy)+1, and immediately (None) for a
// centre outside the array.
// ---------------------------------------------------------------------------------------
struct RingArgs {
    const uint32_t *P, *PT;
    uint32_t wa, ha, wp, hp;
    // QMODE 0: one query per main pixel (uniform 1x1 tiles), read straight from the image
    const uint8_t *main_px;
    size_t mstride;
    uint32_t mch, w, row0;
    // QMODE 1: prepared queries
    const Query *queries;
    uint32_t n;
    uint32_t max_steps;
    uint8_t prec[3];
    uint8_t *raw;      // n: (d as u8), depth_image.rs:121
    uint32_t *raw32;   // optional: d
    uint32_t *pos;     // optional: winning cell (x,y)
    uint8_t *found;    // optional: Some/None
    uint32_t *max_word; // atomicMax of raw, depth_image.rs:124
    unsigned long long *stats; // optional [ring0_hits, probes]
};

constexpr int kLinRings = 8; // rings 1..kLinRings-1 use the offset table (4*8*8-4 = 252 probes)

__device__ __forceinline__ uint32_t bcast(uint32_t v, int src_lane)
{
    return (uint32_t)__builtin_amdgcn_readlane((int)v, src_lane);
}

// offset table entry for linear probe index t (ring s starts at t = 4 s^2, 8s+4 probes)
__device__ __forceinline__ void probe_offset(uint32_t t, int &dx, int &dy, int &s)
{
    s = (int)isqrt_floor(t >> 2);
    const int r = (int)t - 4 * s * s, len = 2 * s + 1;
    const int side = r / len, off = r - side * len - s;
    dx = side == 0 ? off : side == 1 ? off : side == 2 ? s : -s;
    dy = side == 0 ? s : side == 1 ? -s : off;
}

template <int QMODE, bool STATS>
__global__ __launch_bounds__(256) void ring_match_kernel(RingArgs a)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = blockIdx.x * 4u + (threadIdx.x >> 6);
    const uint32_t q = wave * 64u + lane;
    const bool valid = q < a.n;

    uint32_t cx = ~0u, cy = ~0u, lo = 0, hi = 0;
    if (valid) {
        if constexpr (QMODE == 0) {
            const uint32_t x = q % a.w, y = a.row0 + q / a.w;
            const uint8_t *p = a.main_px + (size_t)y * a.mstride + (size_t)x * a.mch;
            const uint8_t v[3] = {p[0], p[1], p[2]};
            cx = (x + 1u) >> 1; // (x + 1)/2, (y + 1)/2 for a 1x1 group
            cy = (y + 1u) >> 1;
            make_box(v, a.prec, lo, hi);
        } else {
            const Query qq = a.queries[q];
            cx = qq.cx;
            cy = qq.cy;
            lo = qq.lo;
            hi = qq.hi;
        }
    }

    // ---- phase A: ring 0, one lane per query ------------------------------------------
    const bool inb = valid && cx < a.wa && cy < a.ha && a.max_steps > 0;
    bool fnd = false;
    uint32_t fx = 0, fy = 0;
    bool pending = false;
    if (inb) {
        const uint32_t c = a.P[(size_t)cy * a.wp + cx];
        if (cell_in_box(c, lo, hi)) {
            fnd = true;
            fx = cx;
            fy = cy;
        } else {
            pending = a.max_steps > 1;
        }
    }
    unsigned long long n_probe = 0;
    if constexpr (STATS)
        n_probe = __popcll(__ballot(inb));
    const unsigned long long ring0 = STATS ? __popcll(__ballot(fnd)) : 0;

    // ---- phase B: survivors, whole wave per query ---------------------------------------
    unsigned long long pend = __ballot(pending);
    if (pend) {
        // scan-order offsets of probes 4..259 (rings 1..7), 4 per lane
        int odx[4], ody[4], os[4];
#pragma unroll
        for (int k = 0; k < 4; ++k)
            probe_offset(4u + 64u * k + lane, odx[k], ody[k], os[k]);

        while (pend) {
            const int src = __ffsll(pend) - 1;
            pend &= pend - 1;
            const uint32_t qx = bcast(cx, src), qy = bcast(cy, src);
            const uint32_t qlo = bcast(lo, src), qhi = bcast(hi, src);
            const uint32_t smax = max(max(qx, a.wa - 1u - qx), max(qy, a.ha - 1u - qy));
            const uint32_t s_end = min(a.max_steps, smax + 1u);
            bool hit = false;
            uint32_t hx = 0, hy = 0;

            // rings 1..7 through the offset table
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (hit || (uint32_t)(k == 0 ? 1 : k == 1 ? 4 : k == 2 ? 5 : 7) >= s_end)
                    break; // first ring of this step: probes 4,68,132,196 -> rings 1,4,5,7
                const uint32_t px = qx + (uint32_t)odx[k], py = qy + (uint32_t)ody[k];
                // probes 256..259 of the table belong to ring 8, which the side loop covers
                const bool act = (uint32_t)os[k] < min(s_end, (uint32_t)kLinRings) &&
                                 px < a.wa && py < a.ha;
                const uint32_t c = act ? a.P[(size_t)py * a.wp + px] : kNoCell;
                const unsigned long long m = __ballot(cell_in_box(c, qlo, qhi));
                if constexpr (STATS)
                    n_probe += __popcll(__ballot(act));
                if (m) {
                    const int w = __ffsll(m) - 1;
                    hx = bcast(px, w);
                    hy = bcast(py, w);
                    hit = true;
                }
            }

            // rings >= 8: four unit-stride sides per ring
            for (uint32_t s = kLinRings; s < s_end && !hit; ++s) {
#pragma unroll
                for (int side = 0; side < 4 && !hit; ++side) {
                    const bool rowside = side < 2;
                    const uint32_t line = rowside ? (side == 0 ? qy + s : qy - s)
                                                  : (side == 2 ? qx + s : qx - s);
                    const uint32_t nline = rowside ? a.ha : a.wa;
                    if (line >= nline) // also catches the unsigned wrap of qy - s < 0
                        continue;
                    const uint32_t *base = rowside ? a.P + (size_t)line * a.wp
                                                   : a.PT + (size_t)line * a.hp;
                    const uint32_t cen = rowside ? qx : qy;
                    const uint32_t lim = rowside ? a.wa : a.ha;
                    const uint32_t j0 = cen >= s ? cen - s : 0u;
                    const uint32_t j1 = min(cen + s, lim - 1u); // inclusive
                    for (uint32_t jb = j0; jb <= j1; jb += 128u) {
                        const uint32_t ja = jb + lane, jc = ja + 64u;
                        const uint32_t ca = ja <= j1 ? base[ja] : kNoCell;
                        const uint32_t cb = jc <= j1 ? base[jc] : kNoCell;
                        if constexpr (STATS)
                            n_probe += min(j1 - jb + 1u, 128u);
                        unsigned long long m = __ballot(cell_in_box(ca, qlo, qhi));
                        uint32_t jh = jb;
                        if (!m) {
                            m = __ballot(cell_in_box(cb, qlo, qhi));
                            jh = jb + 64u;
                        }
                        if (m) {
                            jh += (uint32_t)(__ffsll(m) - 1);
                            hx = rowside ? jh : line;
                            hy = rowside ? line : jh;
                            hit = true;
                            break;
                        }
                    }
                }
            }

            if ((int)lane == src && hit) {
                fnd = true;
                fx = hx;
                fy = hy;
            }
        }
    }

    // ---- epilogue: distance (helpers.rs:3-7), u8 wrap (depth_image.rs:121), max (:124) ----
    uint32_t d = 0;
    if (fnd) {
        const int dx = (int)fx - (int)cx, dy = (int)fy - (int)cy;
        d = isqrt_floor((uint32_t)(dx * dx + dy * dy));
    }
    const uint32_t d8 = d & 0xffu;
    if (valid) {
        a.raw[q] = (uint8_t)d8;
        if (a.raw32)
            a.raw32[q] = d;
        if (a.pos) {
            a.pos[2 * (size_t)q] = fnd ? fx : 0u;
            a.pos[2 * (size_t)q + 1] = fnd ? fy : 0u;
        }
        if (a.found)
            a.found[q] = fnd ? 1 : 0;
    }
    uint32_t m = valid ? d8 : 0u;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
        m = max(m, (uint32_t)__shfl_xor((int)m, o, 64));
    if (lane == 0) {
        if (m)
            atomicMax(a.max_word, m);
        if constexpr (STATS) {
            atomicAdd(&a.stats[0], ring0);
            atomicAdd(&a.stats[1], n_probe);
        }
    }
}

// ---------------------------------------------------------------------------------------
// PASS 2 + paint (src/depth_image.rs:125-132).  v*255/max in u64 floor == 32-bit floor here
// (v,max <= 255); a 256-entry table per workgroup replaces the per-pixel division.
// max == 0: the reference panics (:127); the table is all zero.  identity != 0: RAW_ONLY.
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ void build_lut(uint8_t *lut, const uint32_t *max_word, int identity)
{
    const uint32_t mx = identity ? 255u : *max_word;
    for (uint32_t v = threadIdx.x; v < 256u; v += blockDim.x)
        lut[v] = (mx == 0) ? 0 : (uint8_t)(v * 255u / mx);
    __syncthreads();
}

struct PaintArgs {
    const uint8_t *raw; // per group
    const uint32_t *max_word;
    int identity;
    uint8_t *out; // [H][W], points at row 0 of the image
    size_t ostride;
    uint32_t w, h;
    uint32_t row_begin, row_end;
    uint32_t tile_w, tile_h, gx; // uniform tiles
    const void *groups;          // or list
    uint32_t n;
};

// uniform tiles (incl. 1x1): one thread per output pixel of the strip
__global__ __launch_bounds__(256) void paint_tiles_kernel(PaintArgs a)
{
    __shared__ uint8_t lut[256];
    build_lut(lut, a.max_word, a.identity);
    const uint32_t rows = a.row_end - a.row_begin;
    const size_t total = (size_t)rows * a.w;
    const uint32_t gy0 = a.row_begin / a.tile_h;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (size_t)gridDim.x * blockDim.x) {
        const uint32_t y = a.row_begin + (uint32_t)(i / a.w), x = (uint32_t)(i % a.w);
        const size_t g = (size_t)(y / a.tile_h - gy0) * a.gx + x / a.tile_w;
        a.out[(size_t)y * a.ostride + x] = lut[a.raw[g]];
    }
}

// 1x1 tiles, everything 4-byte aligned: 4 pixels per thread
__global__ __launch_bounds__(256) void paint_1x1_vec4_kernel(PaintArgs a)
{
    __shared__ uint8_t lut[256];
    build_lut(lut, a.max_word, a.identity);
    const uint32_t rows = a.row_end - a.row_begin, w4 = a.w >> 2;
    const size_t total = (size_t)rows * w4;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (size_t)gridDim.x * blockDim.x) {
        const uint32_t r = (uint32_t)(i / w4), c = (uint32_t)(i % w4);
        const uint32_t v = reinterpret_cast<const uint32_t *>(a.raw)[i];
        const uint32_t o = (uint32_t)lut[v & 0xff] | ((uint32_t)lut[(v >> 8) & 0xff] << 8) |
                           ((uint32_t)lut[(v >> 16) & 0xff] << 16) |
                           ((uint32_t)lut[v >> 24] << 24);
        *reinterpret_cast<uint32_t *>(a.out + (size_t)(a.row_begin + r) * a.ostride + 4u * c) = o;
    }
}

// explicit group list: one wave per group, lanes stride over the rectangle.  Rectangles are
// expected to partition the image like disage's; where they overlap, which one wins is
// unspecified (the reference's serial collect() lets the later group win).
__global__ __launch_bounds__(256) void paint_list_kernel(PaintArgs a)
{
    __shared__ uint8_t lut[256];
    build_lut(lut, a.max_word, a.identity);
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t g = blockIdx.x * 4u + (threadIdx.x >> 6);
    if (g >= a.n)
        return;
    const GroupRec r = reinterpret_cast<const GroupRec *>(a.groups)[g];
    if (r.y < a.row_begin || r.y >= a.row_end || r.x >= a.w)
        return;
    const uint8_t v = lut[a.raw[g]];
    const uint32_t w = min(r.w, a.w - r.x), h = min(r.h, a.h - r.y);
    const size_t cells = (size_t)w * h;
    for (size_t i = lane; i < cells; i += 64u) {
        const uint32_t yy = r.y + (uint32_t)(i / w), xx = r.x + (uint32_t)(i % w);
        a.out[(size_t)yy * a.ostride + xx] = v;
    }
}

// in-place PASS 2 on already painted rows (row-strip sharding, after the cross-rank max)
struct NormArgs {
    uint8_t *data;
    size_t stride;
    uint32_t w, rows;
    const uint32_t *max_word;
};

__global__ __launch_bounds__(256) void normalise_rows_kernel(NormArgs a)
{
    __shared__ uint8_t lut[256];
    build_lut(lut, a.max_word, 0);
    const size_t total = (size_t)a.rows * a.w;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (size_t)gridDim.x * blockDim.x) {
        uint8_t *p = a.data + (size_t)(i / a.w) * a.stride + (i % a.w);
        *p = lut[*p];
    }
}

__global__ void zero_words_kernel(uint32_t *p, int n)
{
    if ((int)threadIdx.x < n)
        p[threadIdx.x] = 0;
}

} // namespace stepth
