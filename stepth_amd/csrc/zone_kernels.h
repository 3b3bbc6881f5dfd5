// zone_kernels.h -- consumers of a finished depth map:
// DepthImage::depth_split / select_foreground / slice / invert_depth
// (/root/reference/src/depth_image.rs:162-247) and MaskImage::apply_mask
// (/root/reference/src/mask_image.rs:205-213).
//
// Every pass over pixels is one HBM-bound streaming kernel (16 bytes per lane per trip):
//   depth_histogram_kernel  1 B/px read             -> 256 counts per frame
//   depth_split_kernel      works on the 256 counts only (one workgroup per frame)
//   slice_kernel            1 B/px read + 1 B/px written, range read from the split result
//   invert_kernel           1 B/px read + 1 B/px written
//   apply_mask_kernel       1 B/px mask read, 16 B written per 4 masked-out pixels
// The k-means of depth_split depends on the pixels only through their histogram (the item
// -> centre assignment is a function of the value), so the reference's per-item loop
// (:171-183) becomes one histogram pass plus an iteration over 256 bins; sums stay exact in
// 64-bit integers.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ring_kernels.h" // kMaxBatch

namespace stepth {

constexpr uint32_t kSplitMaxIters = 4096; // the reference recurses without a bound (:210)
constexpr int kSplitErrStep = -7, kSplitErrEmpty = -8, kSplitErrDiverged = -9;

// planes of a batch: same width/height/stride for every frame
struct PlaneArgs {
    const uint8_t *src[kMaxBatch];
    uint8_t *dst[kMaxBatch];
    size_t src_stride, dst_stride;
    uint32_t W, H; // a contiguous plane is passed as W = w*h, H = 1 (keeps 16-byte alignment)
};

// depth_split result of one frame, written by depth_split_kernel
struct ZoneTable {
    int32_t status; // 0 or kSplitErr*
    uint32_t n;     // zones found
    uint32_t iters; // assignment passes
    uint32_t pad;
    uint8_t lo[256], hi[256];
};

__device__ __forceinline__ uint32_t segs_per_row(uint32_t W) { return (W + 15u) >> 4; }

// 16-byte segment `s` of a plane: row, first column, byte count, aligned?
struct Seg {
    uint32_t row, x, n;
};
__device__ __forceinline__ Seg seg_of(uint32_t s, uint32_t W, uint32_t spr)
{
    Seg g;
    g.row = spr == 0 ? 0 : s / spr;
    g.x = (s - g.row * spr) << 4;
    g.n = min(16u, W - g.x);
    return g;
}

constexpr uint32_t kHistReplicas = 32; // one per LDS bank of a 32-bit access
constexpr uint32_t kHistUnroll = 4;    // 16-byte loads in flight per lane

// counts[frame][256] += histogram of the frame's bytes.  LDS holds 32 interleaved copies of
// the table, word (v * 32 + lane % 32): a 32-bit LDS access is banked by (word % 32) within
// each half-wave, so every lane of a half owns its bank whatever the data looks like (depth
// maps are mostly flat: 71 % zeros on the asset) and ds_add_u32 never conflicts.
__device__ __forceinline__ void hist_word(uint32_t *mine, uint32_t w)
{
    atomicAdd(&mine[(w & 255u) * kHistReplicas], 1u);
    atomicAdd(&mine[((w >> 8) & 255u) * kHistReplicas], 1u);
    atomicAdd(&mine[((w >> 16) & 255u) * kHistReplicas], 1u);
    atomicAdd(&mine[(w >> 24) * kHistReplicas], 1u);
}

__global__ __launch_bounds__(256) void depth_histogram_kernel(PlaneArgs a, uint32_t *counts)
{
    __shared__ uint32_t s_h[kHistReplicas * 256];
    for (uint32_t i = threadIdx.x; i < kHistReplicas * 256; i += 256)
        s_h[i] = 0;
    __syncthreads();
    const uint32_t f = blockIdx.y;
    const uint8_t *src = a.src[f];
    uint32_t *mine = s_h + (threadIdx.x & (kHistReplicas - 1));
    const uint32_t spr = segs_per_row(a.W), total = spr * a.H;
    const uint32_t step = gridDim.x * 256;
    for (uint32_t s0 = blockIdx.x * 256 + threadIdx.x; s0 < total; s0 += step * kHistUnroll) {
        uint4 v[kHistUnroll];
        uint32_t fast = 0; // bit u: segment u was loaded as one aligned 16-byte word
#pragma unroll
        for (uint32_t u = 0; u < kHistUnroll; ++u) {
            const uint32_t s = s0 + u * step;
            if (s < total) {
                const Seg g = seg_of(s, a.W, spr);
                const uint8_t *p = src + (size_t)g.row * a.src_stride + g.x;
                if (g.n == 16 && ((uintptr_t)p & 15u) == 0) {
                    v[u] = *reinterpret_cast<const uint4 *>(p);
                    fast |= 1u << u;
                } else {
                    for (uint32_t k = 0; k < g.n; ++k)
                        atomicAdd(&mine[(uint32_t)p[k] * kHistReplicas], 1u);
                }
            }
        }
#pragma unroll
        for (uint32_t u = 0; u < kHistUnroll; ++u)
            if (fast & (1u << u)) {
                hist_word(mine, v[u].x);
                hist_word(mine, v[u].y);
                hist_word(mine, v[u].z);
                hist_word(mine, v[u].w);
            }
    }
    __syncthreads();
    uint32_t sum = 0; // bin threadIdx.x; start at a different copy per lane: no bank conflicts
#pragma unroll 8
    for (uint32_t r = 0; r < kHistReplicas; ++r)
        sum += s_h[threadIdx.x * kHistReplicas + ((r + threadIdx.x) & (kHistReplicas - 1))];
    if (sum)
        atomicAdd(&counts[f * 256 + threadIdx.x], sum);
}

// DepthImage::depth_split on the 256 counts of frame blockIdx.x (src/depth_image.rs:162-219).
// Thread v owns depth value v.  Clusters are keyed by centre VALUE like the reference's
// HashMap<u8, Vec<u8>> (:167-170), so duplicate centres share one cluster.
__global__ __launch_bounds__(256) void depth_split_kernel(const uint32_t *counts, uint32_t zones,
                                                          ZoneTable *tables)
{
    __shared__ uint8_t s_prev[256], s_next[256];
    __shared__ uint32_t s_nprev, s_nnext, s_min, s_max, s_differ, s_empty, s_wave_total[4];
    __shared__ unsigned long long s_sum[256], s_cnt[256];
    __shared__ uint32_t s_key[256], s_cmin[256], s_cmax[256], s_mean_count[256];
    const uint32_t v = threadIdx.x;
    const uint32_t mine = counts[blockIdx.x * 256 + v];
    ZoneTable *out = tables + blockIdx.x;
    if (v == 0) {
        s_min = 255u;
        s_max = 0u;
    }
    __syncthreads();
    if (mine) { // :214-215
        atomicMin(&s_min, v);
        atomicMax(&s_max, v);
    }
    __syncthreads();
    {
        // :217  ((img_max - img_min) / (zones - 1)) as usize - 1, then step_by(step): a
        // quotient of 0 underflows and of 1 gives step_by(0); both panic in the reference
        const uint32_t lo = s_min, hi = s_max;
        const uint32_t q = hi >= lo ? (hi - lo) / (zones - 1u) : 0u;
        if (q < 2u) {
            if (v == 0) {
                out->status = kSplitErrStep;
                out->n = 0;
                out->iters = 0;
            }
            return;
        }
        const uint32_t step = q - 1u, n = (hi - lo + step - 1u) / step; // :216-218, hi exclusive
        if (v < n)
            s_prev[v] = (uint8_t)(lo + v * step);
        if (v == 0)
            s_nprev = n;
    }
    for (uint32_t it = 1;; ++it) {
        s_sum[v] = 0;
        s_cnt[v] = 0;
        s_key[v] = 0;
        s_cmin[v] = 255u;
        s_cmax[v] = 0u;
        s_mean_count[v] = 0;
        __syncthreads();
        if (v == 0) { // read by everyone at the end of the previous trip, before that barrier
            s_differ = 0u;
            s_empty = 0u;
        }
        const uint32_t np = s_nprev;
        if (v < np)
            s_key[s_prev[v]] = 1u; // :167-170
        if (mine) {                // :171-183 for every item of value v at once
            uint32_t best = s_prev[0];
            int best_d = abs((int)best - (int)v);
            for (uint32_t k = 1; k < np; ++k) { // min_by keeps the FIRST minimum
                const int d = abs((int)s_prev[k] - (int)v);
                if (d < best_d) {
                    best_d = d;
                    best = s_prev[k];
                }
            }
            atomicAdd(&s_sum[best], (unsigned long long)mine * v);
            atomicAdd(&s_cnt[best], (unsigned long long)mine);
            atomicMin(&s_cmin[best], v);
            atomicMax(&s_cmax[best], v);
        }
        __syncthreads();
        if (s_key[v]) { // :184-187 floor mean, 0 for an empty cluster
            const unsigned long long c = s_cnt[v];
            atomicAdd(&s_mean_count[(uint32_t)(s_sum[v] / (c ? c : 1ull))], 1u);
        }
        __syncthreads();
        {
            // :188 sort: mean value v goes to positions [before, before + count) of the list;
            // `before` = exclusive prefix sum of the counts (wave scan + 4 wave totals)
            const uint32_t count = s_mean_count[v], lane = v & 63u;
            uint32_t incl = count;
#pragma unroll
            for (uint32_t off = 1; off < 64; off <<= 1) {
                const uint32_t up = __shfl_up(incl, off);
                if (lane >= off)
                    incl += up;
            }
            if (lane == 63u)
                s_wave_total[v >> 6] = incl;
            __syncthreads();
            uint32_t before = incl - count;
            for (uint32_t w = 0; w < (v >> 6); ++w)
                before += s_wave_total[w];
            for (uint32_t r = 0; r < count; ++r)
                s_next[before + r] = (uint8_t)v;
            if (v == 255u)
                s_nnext = before + count;
        }
        __syncthreads();
        const uint32_t n = s_nnext;
        if (v < n && v < np && s_prev[v] != s_next[v]) // :189-195, zip stops at the shorter list
            s_differ = 1u;
        __syncthreads();
        if (!s_differ) { // :196-208
            if (v < n) {
                const uint32_t c = s_next[v];
                if (!s_key[c] || s_cnt[c] == 0) // unwrap() of None panics
                    s_empty = 1u;
                out->lo[v] = (uint8_t)s_cmin[c];
                out->hi[v] = (uint8_t)s_cmax[c];
            }
            __syncthreads();
            if (v == 0) {
                out->status = s_empty ? kSplitErrEmpty : 0;
                out->n = s_empty ? 0u : n;
                out->iters = it;
            }
            return;
        }
        if (it >= kSplitMaxIters) {
            if (v == 0) {
                out->status = kSplitErrDiverged;
                out->n = 0;
                out->iters = it;
            }
            return;
        }
        if (v < n) // :210
            s_prev[v] = s_next[v];
        if (v == 0)
            s_nprev = n;
        // the barrier after the clears at the top of the loop orders these writes
    }
}

// per-byte (b < from || b > to) ? 0 : 255 on four packed bytes
__device__ __forceinline__ uint32_t slice_word(uint32_t w, uint32_t from, uint32_t to)
{
    uint32_t r = 0;
#pragma unroll
    for (uint32_t k = 0; k < 4; ++k) {
        const uint32_t b = (w >> (8 * k)) & 255u;
        r |= (b < from || b > to) ? 0u : (255u << (8 * k));
    }
    return r;
}

// DepthImage::slice, src/depth_image.rs:230-247 (mask plane).  The range is either the
// immediate [from, to] or, when `tables` is given, zone `zone` of the frame's split result
// (select_foreground, :221-224, is zone 0 of depth_split(2)); a failed split leaves the mask
// unwritten.
__global__ __launch_bounds__(256) void slice_kernel(PlaneArgs a, uint32_t from, uint32_t to,
                                                    const ZoneTable *tables, uint32_t zone)
{
    const uint32_t f = blockIdx.y;
    if (tables) {
        const ZoneTable *t = tables + f;
        if (t->status != 0 || zone >= t->n)
            return;
        from = t->lo[zone];
        to = t->hi[zone];
    }
    const uint8_t *src = a.src[f];
    uint8_t *dst = a.dst[f];
    const uint32_t spr = segs_per_row(a.W), total = spr * a.H;
    for (uint32_t s = blockIdx.x * 256 + threadIdx.x; s < total; s += gridDim.x * 256) {
        const Seg g = seg_of(s, a.W, spr);
        const uint8_t *p = src + (size_t)g.row * a.src_stride + g.x;
        uint8_t *q = dst + (size_t)g.row * a.dst_stride + g.x;
        if (g.n == 16 && (((uintptr_t)p | (uintptr_t)q) & 15u) == 0) {
            uint4 v = *reinterpret_cast<const uint4 *>(p);
            v.x = slice_word(v.x, from, to);
            v.y = slice_word(v.y, from, to);
            v.z = slice_word(v.z, from, to);
            v.w = slice_word(v.w, from, to);
            *reinterpret_cast<uint4 *>(q) = v;
        } else {
            for (uint32_t k = 0; k < g.n; ++k)
                q[k] = (p[k] < from || p[k] > to) ? 0 : 255;
        }
    }
}

// DepthImage::invert_depth, src/depth_image.rs:226-228: p = 255 - p (src may equal dst)
__global__ __launch_bounds__(256) void invert_kernel(PlaneArgs a)
{
    const uint32_t f = blockIdx.y;
    const uint8_t *src = a.src[f];
    uint8_t *dst = a.dst[f];
    const uint32_t spr = segs_per_row(a.W), total = spr * a.H;
    for (uint32_t s = blockIdx.x * 256 + threadIdx.x; s < total; s += gridDim.x * 256) {
        const Seg g = seg_of(s, a.W, spr);
        const uint8_t *p = src + (size_t)g.row * a.src_stride + g.x;
        uint8_t *q = dst + (size_t)g.row * a.dst_stride + g.x;
        if (g.n == 16 && (((uintptr_t)p | (uintptr_t)q) & 15u) == 0) {
            uint4 v = *reinterpret_cast<const uint4 *>(p);
            v.x = ~v.x; // 255 - b == ~b per byte
            v.y = ~v.y;
            v.z = ~v.z;
            v.w = ~v.w;
            *reinterpret_cast<uint4 *>(q) = v;
        } else {
            for (uint32_t k = 0; k < g.n; ++k)
                q[k] = (uint8_t)(255u - p[k]);
        }
    }
}

// MaskImage::apply_mask, src/mask_image.rs:205-213: RGBA pixel := [0;4] where mask == 0.
// One lane per 4 pixels: 4 mask bytes in, one 16-byte store only where something is cleared.
struct ApplyArgs {
    uint8_t *rgba[kMaxBatch];
    const uint8_t *mask[kMaxBatch];
    size_t rgba_stride, mask_stride;
    uint32_t W, H;
};
__global__ __launch_bounds__(256) void apply_mask_kernel(ApplyArgs a)
{
    const uint32_t f = blockIdx.y;
    const uint32_t qpr = (a.W + 3u) >> 2, total = qpr * a.H;
    for (uint32_t s = blockIdx.x * 256 + threadIdx.x; s < total; s += gridDim.x * 256) {
        const uint32_t row = s / qpr, x = (s - row * qpr) << 2, n = min(4u, a.W - x);
        const uint8_t *m = a.mask[f] + (size_t)row * a.mask_stride + x;
        uint8_t *p = a.rgba[f] + (size_t)row * a.rgba_stride + 4u * (size_t)x;
        if (n == 4 && ((uintptr_t)m & 3u) == 0 && ((uintptr_t)p & 15u) == 0) {
            const uint32_t mw = *reinterpret_cast<const uint32_t *>(m);
            const bool z0 = !(mw & 0xffu), z1 = !(mw & 0xff00u), z2 = !(mw & 0xff0000u),
                       z3 = !(mw & 0xff000000u);
            if (z0 & z1 & z2 & z3) {
                *reinterpret_cast<uint4 *>(p) = make_uint4(0, 0, 0, 0);
            } else if (z0 | z1 | z2 | z3) {
                uint32_t *pw = reinterpret_cast<uint32_t *>(p);
                if (z0)
                    pw[0] = 0;
                if (z1)
                    pw[1] = 0;
                if (z2)
                    pw[2] = 0;
                if (z3)
                    pw[3] = 0;
            }
        } else {
            for (uint32_t k = 0; k < n; ++k)
                if (m[k] == 0)
                    for (uint32_t c = 0; c < 4; ++c)
                        p[4 * k + c] = 0;
        }
    }
}

} // namespace stepth
