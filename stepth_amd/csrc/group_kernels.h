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

// min / max / sum per channel over the rows y + sub, y + sub + nsub, ... of a rectangle (one lane's
// share; nsub = 1: the whole rectangle)
__device__ __forceinline__ void node_stats(const AdaptiveArgs &a, uint32_t f, const GroupNode &n, uint32_t sub,
                                           uint32_t nsub, uint32_t (&mn)[3], uint32_t (&mx)[3],
                                           unsigned long long (&total)[3])
{
    uint32_t sum[3] = {0, 0, 0};
    unsigned long long s64[3] = {0, 0, 0};
    mn[0] = mn[1] = mn[2] = 255u;
    mx[0] = mx[1] = mx[2] = 0u;
    uint32_t rows_done = 0;
    // RGB8 rows that start on a 4-byte boundary are read a dword at a time (a byte load
    // moves one byte per lane and instruction).  Twelve bytes = four pixels = three
    // dwords repeat the byte -> channel pattern, so the slots t[r], r = byte offset from
    // the first dword mod 3, are static in the unrolled body; they are turned into
    // channels once at the end (channel = (r + first dword's offset) mod 3).
    const bool by_dwords = a.ch == 3u && ((reinterpret_cast<uintptr_t>(a.px[f]) | a.stride) & 3u) == 0u;
    if (by_dwords) {
        const uint32_t b0 = n.x * 3u, b1 = (n.x + n.w) * 3u, a0 = b0 & ~3u; // bytes of the row
        const uint32_t row_bytes = a.W * 3u;
        const uint32_t ph = a0 % 3u;
        uint32_t tmn[3] = {255, 255, 255}, tmx[3] = {0, 0, 0}, tsum[3] = {0, 0, 0};
        for (uint32_t yy = n.y + sub; yy < n.y + n.h; yy += nsub) {
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
            if ((++rows_done & 127u) == 0u) { // 128 rows x 65535 px x 255 still fit 32 bits
#pragma unroll
                for (int k = 0; k < 3; ++k) { // channel k sits in slot (k - ph) mod 3
                    const uint32_t t = ph == 0u ? tsum[k] : ph == 1u ? tsum[(k + 2) % 3] : tsum[(k + 1) % 3];
                    s64[k] += t;
                }
                tsum[0] = tsum[1] = tsum[2] = 0;
            }
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            sum[k] = ph == 0u ? tsum[k] : ph == 1u ? tsum[(k + 2) % 3] : tsum[(k + 1) % 3];
            mn[k] = ph == 0u ? tmn[k] : ph == 1u ? tmn[(k + 2) % 3] : tmn[(k + 1) % 3];
            mx[k] = ph == 0u ? tmx[k] : ph == 1u ? tmx[(k + 2) % 3] : tmx[(k + 1) % 3];
        }
    } else {
        for (uint32_t yy = n.y + sub; yy < n.y + n.h; yy += nsub) {
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
            if ((++rows_done & 127u) == 0u) { // 128 rows x 65535 px x 255 still fit 32 bits
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    s64[k] += sum[k];
                    sum[k] = 0;
                }
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 3; ++k)
        total[k] = s64[k] + sum[k];
}

__device__ __forceinline__ GroupRec leaf_group(const GroupNode &n, const unsigned long long (&total)[3])
{
    GroupRec g{};
    const unsigned long long area = (unsigned long long)n.w * n.h;
    g.x = n.x;
    g.y = n.y;
    g.w = n.w;
    g.h = n.h;
#pragma unroll
    for (int k = 0; k < 3; ++k)
        g.value[k] = (uint8_t)(total[k] / area);
    g.pad = 0;
    return g;
}

// the two halves of a node: cut across its longer side into floor and ceiling halves
__device__ __forceinline__ void split_node(const GroupNode &n, GroupNode &c0, GroupNode &c1)
{
    c0 = n;
    c1 = n;
    if (n.w >= n.h) {
        c0.w = n.w / 2;
        c1.x = n.x + c0.w;
        c1.w = n.w - c0.w;
    } else {
        c0.h = n.h / 2;
        c1.y = n.y + c0.h;
        c1.h = n.h - c0.h;
    }
}

// statistics of a rectangle -> leaf? ; writes the group or the two children
__device__ __forceinline__ void adaptive_visit(const AdaptiveArgs &a, uint32_t f, bool live, GroupNode n,
                                               uint32_t depth, GroupNode *next, uint32_t *next_count)
{
    bool leaf = false;
    GroupRec g{};
    if (live) {
        leaf = (uint64_t)n.w * n.h == 1 || depth >= a.max_splits;
        if (leaf || depth >= a.min_splits) {
            uint32_t mn[3], mx[3];
            unsigned long long total[3];
            node_stats(a, f, n, 0u, 1u, mn, mx, total);
            if (!leaf)
                leaf = mx[0] - mn[0] < a.prec[0] && mx[1] - mn[1] < a.prec[1] && mx[2] - mn[2] < a.prec[2];
            if (leaf)
                g = leaf_group(n, total);
        }
    }
    wave_append(live && leaf, a.groups[f], &a.counters[4u * f * kGroupCtrPad], g);
    const bool split = live && !leaf;
    GroupNode c0, c1;
    split_node(n, c0, c1);
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

// the path from the root to node `i` of depth `seed_depth`: bit (seed_depth-1-d) of the index
// picks the child at depth d.  A path that reaches a single pixel early belongs to the index
// whose remaining bits are all zero.  Returns whether the node exists.
__device__ __forceinline__ bool seed_node(const AdaptiveArgs &a, uint64_t i, GroupNode &n, uint32_t &depth)
{
    bool live = i < (1ull << a.seed_depth);
    n = GroupNode{0, 0, a.W, a.H};
    depth = 0;
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
    return live;
}

// The whole tree below the seed depth in ONE launch (round 3; the seed + level launches below
// remain for seeds whose subtrees are deeper than kTreeLevels and images wider than 65535).
// A workgroup takes 256 seed nodes, one per thread, and keeps the nodes below them in LDS, one
// array per level: a go pops up to 256 nodes of the DEEPEST level that has any (a thread per
// node, as in the level launches: every lane busy), appends the leaves to the frame's groups
// (one global atomic per workgroup and go) and pushes the halves of the others one level down.
// Deepest first keeps every level at <= 512 nodes: a level is only fed (<= 2 x 256) by a pop of
// the level above, which happens only when it and everything deeper is empty.
// Tried first and slower than the six launches: four lanes per node with a stack per seed walked
// depth first (a wave sits through its unluckiest seed's subtree: 5 x), and with one stack per
// wave (3 x: a quad per node leaves most lanes idle on the small nodes deep in the tree).
constexpr uint32_t kTreeLevels = 8, kTreeLevelCap = 512;

__global__ __launch_bounds__(256) void adaptive_tree_kernel(AdaptiveArgs a, uint32_t levels)
{
    extern __shared__ uint2 s_level[]; // [levels][kTreeLevelCap] {x | y << 16, w | h << 16}; level l = depth seed + 1 + l
    __shared__ uint32_t s_cnt[kTreeLevels];
    __shared__ uint32_t s_leaves[2], s_base[2]; // by the parity of the go
    const uint32_t f = blockIdx.y, t = threadIdx.x, lane = t & 63u;
    if (t < kTreeLevels)
        s_cnt[t] = 0u;
    if (t < 2u)
        s_leaves[t] = 0u;
    const uint64_t total_seeds = 1ull << a.seed_depth;
    const uint64_t rounds = (total_seeds + (uint64_t)gridDim.x * 256 - 1) / ((uint64_t)gridDim.x * 256);
    uint32_t par = 0;
    for (uint64_t r = 0; r < rounds; ++r) {
        bool seeds = true; // the round's first go takes the seed nodes
        // (every go takes at least one node out of the <= 256 (2^(levels+1) - 1) of a round: the
        // guard is never reached, it only keeps a logic error from hanging the GPU)
        for (uint32_t guard = 0; guard < (512u << levels); ++guard) {
            __syncthreads(); // the pushes of the last go
            bool have;
            GroupNode n{0, 0, 1, 1};
            uint32_t depth, to = 0;
            if (seeds) {
                have = seed_node(a, (r * gridDim.x + blockIdx.x) * 256 + t, n, depth);
                seeds = false;
            } else {
                int l = (int)levels - 1;
                while (l >= 0 && s_cnt[l] == 0u)
                    --l;
                if (l < 0)
                    break;
                const uint32_t cnt = s_cnt[l], take = min(cnt, 256u);
                have = t < take;
                if (have) {
                    const uint2 pn = s_level[(uint32_t)l * kTreeLevelCap + cnt - take + t];
                    n = GroupNode{pn.x & 0xffffu, pn.x >> 16, pn.y & 0xffffu, pn.y >> 16};
                }
                __syncthreads();
                if (t == 0)
                    s_cnt[l] = cnt - take;
                depth = a.seed_depth + 1u + (uint32_t)l;
                // (level l + 1 exists whenever a node of level l can split: the host sized `levels`)
                to = min((uint32_t)l + 1u, levels - 1u);
            }
            // ---- the go: the thread's node -> a leaf into the groups or two halves into level `to`
            bool leaf = false;
            GroupRec g{};
            if (have) {
                leaf = n.w * n.h == 1u || depth >= a.max_splits;
                if (leaf || depth >= a.min_splits) {
                    uint32_t mn[3], mx[3];
                    unsigned long long total[3];
                    node_stats(a, f, n, 0u, 1u, mn, mx, total);
                    if (!leaf)
                        leaf = mx[0] - mn[0] < a.prec[0] && mx[1] - mn[1] < a.prec[1] && mx[2] - mn[2] < a.prec[2];
                    if (leaf)
                        g = leaf_group(n, total);
                }
            }
            const bool is_leaf = have && leaf, split = have && !leaf;
            const unsigned long long ml = __ballot(is_leaf), ms = __ballot(split);
            uint32_t leaf_at = 0, push_at = 0;
            if (lane == 0) {
                if (ml)
                    leaf_at = atomicAdd(&s_leaves[par], (uint32_t)__popcll(ml));
                if (ms)
                    push_at = atomicAdd(&s_cnt[to], 2u * (uint32_t)__popcll(ms));
            }
            leaf_at = __shfl(leaf_at, 0) + (uint32_t)__popcll(ml & ((1ull << lane) - 1ull));
            push_at = __shfl(push_at, 0) + 2u * (uint32_t)__popcll(ms & ((1ull << lane) - 1ull));
            if (split) {
                GroupNode c0, c1;
                split_node(n, c0, c1);
                s_level[to * kTreeLevelCap + push_at] = make_uint2(c0.x | (c0.y << 16), c0.w | (c0.h << 16));
                s_level[to * kTreeLevelCap + push_at + 1u] = make_uint2(c1.x | (c1.y << 16), c1.w | (c1.h << 16));
            }
            __syncthreads();
            if (t == 0) {
                const uint32_t nl = s_leaves[par];
                s_base[par] = nl ? atomicAdd(&a.counters[4u * f * kGroupCtrPad], nl) : 0u;
                s_leaves[par ^ 1u] = 0u; // (the next go's; this one's is read no more)
            }
            __syncthreads();
            if (is_leaf)
                a.groups[f][s_base[par] + leaf_at] = g;
            par ^= 1u;
        }
    }
}

// one thread per node index of depth `seed_depth` (seed_node)
__global__ __launch_bounds__(256) void adaptive_seed_kernel(AdaptiveArgs a)
{
    const uint32_t f = blockIdx.y;
    const uint64_t total = 1ull << a.seed_depth;
    const uint64_t rounds = (total + (uint64_t)gridDim.x * 256 - 1) / ((uint64_t)gridDim.x * 256);
    for (uint64_t r = 0; r < rounds; ++r) { // whole waves stay together for the ballots
        GroupNode n;
        uint32_t depth;
        const bool live = seed_node(a, (r * gridDim.x + blockIdx.x) * 256 + threadIdx.x, n, depth);
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
