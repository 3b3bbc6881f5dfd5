// group_kernels.h -- adaptive pixel groups: the engine's stand-in for
// disage::open::rgb_discrete(img, MeanBrightnessHasher, BrightnessChecker{precision},
// (min_splits, max_splits)), /root/reference/src/depth_image.rs:101-110.
//
// disage's source is absent from the reference tree (empty submodule), so the rule below is
// the engine's OWN DESIGN and its parity with disage is unpinned; it keeps what the call
// site fixes (rectangles {position, size, value} tiling the image; >= min_splits levels of
// binary splitting; <= max_splits levels; a mean colour as the value; uniformity judged with
// `precision`):
//   a rectangle at depth d is a leaf if it is a single pixel, or d >= max_splits, or
//   d >= min_splits and max - min < precision[c] on every channel; otherwise it is cut across
//   its longer side into floor and ceiling halves.  value = per-channel floor(mean).
//
// The tree above depth min_splits does not depend on the pixels, so it is never built: one
// thread per depth-min_splits node walks its path from the root (adaptive_seed_kernel), then
// one launch per level tests the surviving nodes and appends leaves / children with
// wave-aggregated atomics (adaptive_level_kernel).  Group order is arbitrary (the reference's
// results do not depend on it: every group is matched independently).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ring_kernels.h" // GroupRec

namespace stepth {

struct GroupNode {
    uint32_t x, y, w, h;
};

constexpr uint32_t kGroupCtrPad = 32;

struct AdaptiveArgs {
    const uint8_t *px[kMaxBatch]; // main images (one per frame, blockIdx.y), `ch` bytes per pixel
    size_t stride;
    uint32_t ch, W, H;
    uint32_t prec[3];
    uint32_t min_splits, max_splits, seed_depth; // seed_depth = min(min_splits, max_splits)
    GroupRec *groups[kMaxBatch];                 // capacity W*H each
    GroupNode *list[3];                          // node lists of levels d, d+1, d+2 (rotating) ...
    size_t list_stride;                          // ... frame f's at list[k] + f * list_stride
    // per frame 4 counters, each on its own 128-byte line (kGroupCtrPad words apart: atomics on
    // one line serialise chip-wide, and every wave of every frame appends): counter c of frame f
    // at counters[(4 f + c) * kGroupCtrPad]; c = 0 groups, 1 + level % 3 list lengths
    uint32_t *counters;
};

// appends `v` for the lanes with `yes` to an array with length counter `count`: one atomic
// per wave (atomics on one address serialise chip-wide)
template <typename T> __device__ __forceinline__ void wave_append(bool yes, T *arr, uint32_t *count, const T &v)
{
    const unsigned long long m = __ballot(yes);
    if (!m)
        return;
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t base = 0;
    if (lane == (uint32_t)(__ffsll(m) - 1))
        base = atomicAdd(count, (uint32_t)__popcll(m));
    base = __shfl(base, __ffsll(m) - 1);
    if (yes)
        arr[base + __popcll(m & ((1ull << lane) - 1ull))] = v;
}

// statistics of a rectangle -> leaf? ; writes the group or the two children
__device__ __forceinline__ void adaptive_visit(const AdaptiveArgs &a, uint32_t f, bool live, GroupNode n,
                                               uint32_t depth, GroupNode *next, uint32_t *next_count)
{
    bool leaf = false;
    uint32_t sum[3] = {0, 0, 0};
    GroupRec g{};
    if (live) {
        leaf = (uint64_t)n.w * n.h == 1 || depth >= a.max_splits;
        if (leaf || depth >= a.min_splits) {
            uint32_t mn[3] = {255, 255, 255}, mx[3] = {0, 0, 0};
            unsigned long long s64[3] = {0, 0, 0};
            // RGB8 rows that start on a 4-byte boundary are read a dword at a time (a byte load
            // moves one byte per lane and instruction; the tree's rectangles are 7..8 x 4 pixels
            // at the seed depth, 96 byte loads a thread).  Twelve bytes = four pixels = three
            // dwords repeat the byte -> channel pattern, so the slots t[r], r = byte offset from
            // the first dword mod 3, are static in the unrolled body; they are turned into
            // channels once at the end (channel = (r + first dword's offset) mod 3).
            const bool by_dwords = a.ch == 3u && ((reinterpret_cast<uintptr_t>(a.px[f]) | a.stride) & 3u) == 0u;
            if (by_dwords) {
                const uint32_t b0 = n.x * 3u, b1 = (n.x + n.w) * 3u, a0 = b0 & ~3u; // bytes of the row
                const uint32_t row_bytes = a.W * 3u;
                uint32_t tmn[3] = {255, 255, 255}, tmx[3] = {0, 0, 0}, tsum[3] = {0, 0, 0};
                for (uint32_t yy = n.y; yy < n.y + n.h; ++yy) {
                    const uint8_t *row = a.px[f] + (size_t)yy * a.stride;
                    for (uint32_t o = a0; o < b1; o += 12u) {
                        uint32_t dw[3]; // the three loads go out together
#pragma unroll
                        for (uint32_t i = 0; i < 3; ++i) {
                            const uint32_t od = o + 4u * i;
                            dw[i] = 0u;
                            if (od + 4u <= row_bytes) { // whole dword inside the row
                                if (od < b1)
                                    dw[i] = *reinterpret_cast<const uint32_t *>(row + od);
                            } else if (od < b1) { // the row's last bytes: never read past the image
                                dw[i] = row[od];
                                if (od + 1u < row_bytes)
                                    dw[i] |= (uint32_t)row[od + 1u] << 8;
                                if (od + 2u < row_bytes)
                                    dw[i] |= (uint32_t)row[od + 2u] << 16;
                            }
                        }
#pragma unroll
                        for (uint32_t i = 0; i < 3; ++i)
#pragma unroll
                            for (uint32_t k = 0; k < 4; ++k) {
                                const uint32_t od = o + 4u * i;
                                const uint32_t r = (4u * i + k) % 3u, v = (dw[i] >> (8u * k)) & 0xffu;
                                const bool in = od + k >= b0 && od + k < b1;
                                tsum[r] += in ? v : 0u;
                                tmn[r] = min(tmn[r], in ? v : 255u);
                                tmx[r] = max(tmx[r], in ? v : 0u);
                            }
                    }
                    if ((yy & 127u) == 127u) { // 128 rows x 65535 px x 255 still fit 32 bits
                        const uint32_t ph = a0 % 3u;
#pragma unroll
                        for (int k = 0; k < 3; ++k) { // channel k sits in slot (k - ph) mod 3
                            const uint32_t t = ph == 0u ? tsum[k] : ph == 1u ? tsum[(k + 2) % 3] : tsum[(k + 1) % 3];
                            s64[k] += t;
                        }
                        tsum[0] = tsum[1] = tsum[2] = 0;
                    }
                }
                const uint32_t ph = a0 % 3u;
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    sum[k] = ph == 0u ? tsum[k] : ph == 1u ? tsum[(k + 2) % 3] : tsum[(k + 1) % 3];
                    mn[k] = ph == 0u ? tmn[k] : ph == 1u ? tmn[(k + 2) % 3] : tmn[(k + 1) % 3];
                    mx[k] = ph == 0u ? tmx[k] : ph == 1u ? tmx[(k + 2) % 3] : tmx[(k + 1) % 3];
                }
            } else {
                for (uint32_t yy = n.y; yy < n.y + n.h; ++yy) {
                    const uint8_t *p = a.px[f] + (size_t)yy * a.stride + (size_t)n.x * a.ch;
                    for (uint32_t xx = 0; xx < n.w; ++xx, p += a.ch) {
#pragma unroll
                        for (int k = 0; k < 3; ++k) {
                            const uint32_t v = p[k];
                            sum[k] += v;
                            mn[k] = min(mn[k], v);
                            mx[k] = max(mx[k], v);
                        }
                    }
                    if ((yy & 127u) == 127u) { // 128 rows x 65535 px x 255 still fit 32 bits
#pragma unroll
                        for (int k = 0; k < 3; ++k) {
                            s64[k] += sum[k];
                            sum[k] = 0;
                        }
                    }
                }
            }
            if (!leaf)
                leaf = mx[0] - mn[0] < a.prec[0] && mx[1] - mn[1] < a.prec[1] && mx[2] - mn[2] < a.prec[2];
            if (leaf) {
                const unsigned long long area = (unsigned long long)n.w * n.h;
                g.x = n.x;
                g.y = n.y;
                g.w = n.w;
                g.h = n.h;
#pragma unroll
                for (int k = 0; k < 3; ++k)
                    g.value[k] = (uint8_t)((s64[k] + sum[k]) / area);
                g.pad = 0;
            }
        }
    }
    wave_append(live && leaf, a.groups[f], &a.counters[4u * f * kGroupCtrPad], g);
    const bool split = live && !leaf;
    GroupNode c0 = n, c1 = n;
    if (n.w >= n.h) {
        c0.w = n.w / 2;
        c1.x = n.x + c0.w;
        c1.w = n.w - c0.w;
    } else {
        c0.h = n.h / 2;
        c1.y = n.y + c0.h;
        c1.h = n.h - c0.h;
    }
    // both children of a node end up adjacent
    const unsigned long long m = __ballot(split);
    if (m) {
        const uint32_t lane = threadIdx.x & 63u;
        uint32_t base = 0;
        if (lane == (uint32_t)(__ffsll(m) - 1))
            base = atomicAdd(next_count, 2u * (uint32_t)__popcll(m));
        base = __shfl(base, __ffsll(m) - 1);
        if (split) {
            const uint32_t at = base + 2u * (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
            next[at] = c0;
            next[at + 1] = c1;
        }
    }
}

// one thread per node index of depth `seed_depth`: bit (seed_depth-1-d) of the index picks the
// child at depth d.  A path that reaches a single pixel early belongs to the index whose
// remaining bits are all zero.
__global__ __launch_bounds__(256) void adaptive_seed_kernel(AdaptiveArgs a)
{
    const uint32_t f = blockIdx.y;
    const uint64_t total = 1ull << a.seed_depth;
    const uint64_t rounds = (total + (uint64_t)gridDim.x * 256 - 1) / ((uint64_t)gridDim.x * 256);
    for (uint64_t r = 0; r < rounds; ++r) { // whole waves stay together for the ballots
        const uint64_t i = (r * gridDim.x + blockIdx.x) * 256 + threadIdx.x;
        bool live = i < total;
        GroupNode n{0, 0, a.W, a.H};
        uint32_t depth = 0;
        for (; live && depth < a.seed_depth; ++depth) {
            if ((uint64_t)n.w * n.h == 1) {
                live = (i & ((1ull << (a.seed_depth - depth)) - 1ull)) == 0; // the all-zero suffix owns it
                break;
            }
            const bool second = (i >> (a.seed_depth - 1 - depth)) & 1ull;
            if (n.w >= n.h) {
                const uint32_t w0 = n.w / 2;
                n.x += second ? w0 : 0;
                n.w = second ? n.w - w0 : w0;
            } else {
                const uint32_t h0 = n.h / 2;
                n.y += second ? h0 : 0;
                n.h = second ? n.h - h0 : h0;
            }
        }
        adaptive_visit(a, f, live, n, depth, a.list[(a.seed_depth + 1) % 3] + f * a.list_stride,
                       &a.counters[(4u * f + 1 + (a.seed_depth + 1) % 3) * kGroupCtrPad]);
    }
}

// nodes of depth `depth` (list[depth % 3]) -> groups and nodes of depth + 1 (list[(depth+1) % 3]).
// The third list is the one level depth - 1 has consumed: its counter is cleared here for
// level depth + 1 to append to (no launch in between needs it).
__global__ __launch_bounds__(256) void adaptive_level_kernel(AdaptiveArgs a, uint32_t depth)
{
    const uint32_t f = blockIdx.y;
    uint32_t *counters = a.counters + 4u * f * kGroupCtrPad;
    const uint32_t count = counters[(1 + depth % 3) * kGroupCtrPad];
    if (blockIdx.x == 0 && threadIdx.x == 0)
        counters[(1 + (depth + 2) % 3) * kGroupCtrPad] = 0;
    const uint32_t rounds = (count + gridDim.x * 256 - 1) / (gridDim.x * 256);
    const GroupNode *in = a.list[depth % 3] + f * a.list_stride;
    for (uint32_t r = 0; r < rounds; ++r) {
        const uint32_t i = (r * gridDim.x + blockIdx.x) * 256 + threadIdx.x;
        const bool live = i < count;
        const GroupNode n = live ? in[i] : GroupNode{0, 0, 1, 1};
        adaptive_visit(a, f, live, n, depth, a.list[(depth + 1) % 3] + f * a.list_stride,
                       &counters[(1 + (depth + 1) % 3) * kGroupCtrPad]);
    }
}

} // namespace stepth
