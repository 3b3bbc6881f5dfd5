// engine.hip -- C-ABI of include/stepth_amd.h on top of the gfx950 kernels.
//
// Replaces the body of DepthImage::load_depth_from_additional,
// /root/reference/src/depth_image.rs:96-132 (everything but the Gaussian resample tail).
// No CPU fallback exists: without a HIP device every entry point fails with
// STEPTH_ERR_NO_DEVICE / STEPTH_ERR_HIP.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <new>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/stepth_amd.h"
#include "ring_kernels.h"
#include "tile_kernels.h"
#include "sad_kernels.h"
#include "zone_kernels.h"
#include "group_kernels.h"

using namespace stepth;

static_assert(sizeof(stepth_group) == 20 && sizeof(GroupRec) == 20, "group layout");
static_assert(sizeof(Query) == 16, "query layout");
static_assert(sizeof(stepth_params) == 56 && sizeof(stepth_image) == 32 && sizeof(stepth_stats) == 56,
              "C-ABI struct layout (mirrored by stepth_amd/_lib.py)");

namespace {

struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
};

} // namespace

// words: [0..63] max per frame of a batch (a single call uses [0]), [64] host max staged for
// normalise_rows, then from word 128 on one counter per frame (the list search's queue of
// rectangles too large for its own lanes to paint)
constexpr uint32_t kWordHostMax = 64, kWordCounters = 128;
constexpr size_t kStatsBytes = (size_t)kStatsWords * kStatsReplicas * 8; // replicated counters (ring_kernels.h)
constexpr size_t kWordsBytes = (kWordCounters + kMaxBatch) * 4;
static_assert(kMaxBatch <= (int)kWordHostMax, "per-frame max words");

struct stepth_engine {
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    std::string err;

    DevBuf T, TT, LB, BBp, SBp; // packed additional image: cells row- and column-major, line bounds, padded block / superblock bounds
    uint32_t bbp_layout[3] = {0, 0, 0}; // bw, bh, frames the zero border of BBp was laid out for
    DevBuf stage_main, stage_add, stage_out, stage_groups;
    DevBuf queries_sorted, query_ids, query_bins; // group lists sorted by search-centre bin
    DevBuf queries, raw, aux32, aux_pos, aux_found, pending;
    DevBuf gauss_tmp, gauss_io;
    DevBuf zone_counts, zone_tables; // depth_split: 256 counts and one ZoneTable per frame
    DevBuf group_out, group_list[3], group_counters; // adaptive grouping (group_entries.h)
    // host-image batches: two slots of device staging, copy streams and events (batch_host)
    DevBuf pipe_main[2], pipe_add[2], pipe_out[2], pipe_max;
    hipStream_t copy_in = nullptr, copy_out = nullptr;
    hipEvent_t ev_switch = nullptr; // orders a newly bound stream after the old one's work
    hipEvent_t ev_in[2] = {nullptr, nullptr}, ev_compute[2] = {nullptr, nullptr},
               ev_out[2] = {nullptr, nullptr};
    uint32_t *words = nullptr; // layout: kWord* above
    unsigned long long *stats = nullptr; // [ring0_hits, probes]
    uint64_t stats_groups = 0;
    bool stats_enabled = false;

    DevBuf order_bytes, order_words; // heavy-first launch order of the dense search (tile_cost_kernel)
    bool timing = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> events;
    size_t events_used = 0;
};

namespace {

int fail(stepth_engine *e, int status, const char *fmt, ...)
{
    if (e) {
        char buf[512];
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(buf, sizeof buf, fmt, ap);
        va_end(ap);
        e->err = buf;
    }
    return status;
}

#define HIP_TRY(e, call)                                                                      \
    do {                                                                                      \
        hipError_t _r = (call);                                                               \
        if (_r != hipSuccess)                                                                 \
            return fail((e), _r == hipErrorOutOfMemory ? STEPTH_ERR_OUT_OF_MEMORY             \
                                                       : STEPTH_ERR_HIP,                      \
                        "%s: %s (%s:%d)", #call, hipGetErrorString(_r), __FILE__, __LINE__);  \
    } while (0)

int reserve(stepth_engine *e, DevBuf &b, size_t bytes)
{
    if (bytes <= b.cap)
        return STEPTH_OK;
    if (b.p) {
        HIP_TRY(e, hipStreamSynchronize(e->stream));
        HIP_TRY(e, hipFree(b.p));
        b.p = nullptr;
        b.cap = 0;
    }
    const size_t want = (bytes + 255) & ~(size_t)255;
    HIP_TRY(e, hipMalloc(&b.p, want));
    b.cap = want;
    return STEPTH_OK;
}

void release(DevBuf &b)
{
    if (b.p)
        (void)hipFree(b.p);
    b.p = nullptr;
    b.cap = 0;
}

// The C-ABI promises that no call throws or aborts (include/stepth_amd.h): every extern "C"
// entry with a status runs inside this guard -- a host allocation sized by caller input
// (std::vector, std::string) that fails becomes STEPTH_ERR_OUT_OF_MEMORY instead of a
// std::terminate in the caller's process.
int guard_oom(stepth_engine *e) noexcept
{
    if (e) {
        try {
            e->err = "out of memory"; // (fits the small-string buffer: no allocation)
        } catch (...) {
        }
    }
    return STEPTH_ERR_OUT_OF_MEMORY;
}

int guard_other(stepth_engine *e, const char *what) noexcept
{
    try {
        return fail(e, STEPTH_ERR_HIP, "unexpected C++ exception: %s", what);
    } catch (...) {
        return STEPTH_ERR_HIP;
    }
}

#define STEPTH_GUARD_BEGIN try {
#define STEPTH_GUARD_END(e)                                                                   \
    }                                                                                         \
    catch (const std::bad_alloc &) { return guard_oom(e); }                                   \
    catch (const std::length_error &) { return guard_oom(e); }                                \
    catch (const std::exception &ex_) { return guard_other((e), ex_.what()); }                \
    catch (...) { return guard_other((e), "unknown"); }

#define TRY(expr)                                                                             \
    do {                                                                                      \
        int _s = (expr);                                                                      \
        if (_s != STEPTH_OK)                                                                  \
            return _s;                                                                        \
    } while (0)

// Brings a caller image onto the device if it is host memory.  Returns the device view.
// (r0, r1, cols: the rectangle of a host image that travels -- rows [r0, r1), columns [0, cols);
// the staged copy keeps the full geometry, what lies outside the rectangle is never read)
int to_device(stepth_engine *e, const stepth_image &im, DevBuf &stage, const uint8_t *&dptr,
              size_t &dstride, uint32_t r0 = 0, uint32_t r1 = 0xFFFFFFFFu, uint32_t cols = 0xFFFFFFFFu)
{
    const size_t row = (size_t)im.width * im.channels;
    if (im.memory == STEPTH_MEM_DEVICE) {
        dptr = static_cast<const uint8_t *>(im.data);
        dstride = im.stride;
        return STEPTH_OK;
    }
    const size_t pitch = (row + 15) & ~(size_t)15;
    TRY(reserve(e, stage, pitch * im.height));
    r1 = std::min(r1, im.height);
    r0 = std::min(r0, r1);
    cols = std::min(cols, im.width);
    if (r1 > r0 && cols)
        HIP_TRY(e, hipMemcpy2DAsync(static_cast<uint8_t *>(stage.p) + (size_t)r0 * pitch, pitch,
                                    static_cast<const uint8_t *>(im.data) + (size_t)r0 * im.stride, im.stride,
                                    (size_t)cols * im.channels, r1 - r0, hipMemcpyHostToDevice, e->stream));
    dptr = static_cast<const uint8_t *>(stage.p);
    dstride = pitch;
    return STEPTH_OK;
}

int check_image(stepth_engine *e, const stepth_image *im, const char *name, uint32_t c0,
                uint32_t c1)
{
    if (!im || !im->data)
        return fail(e, STEPTH_ERR_BAD_ARG, "%s image is null", name);
    if (im->width == 0 || im->height == 0 || im->width > 65535u || im->height > 65535u)
        return fail(e, STEPTH_ERR_BAD_ARG, "%s image %ux%u: sides must be in 1..65535", name,
                    im->width, im->height);
    if (im->channels != c0 && im->channels != c1)
        return fail(e, STEPTH_ERR_BAD_ARG, "%s image has %u channels", name, im->channels);
    if (im->stride < (size_t)im->width * im->channels)
        return fail(e, STEPTH_ERR_BAD_ARG, "%s image stride %zu < row bytes", name, im->stride);
    if (im->memory != STEPTH_MEM_HOST && im->memory != STEPTH_MEM_DEVICE)
        return fail(e, STEPTH_ERR_BAD_ARG, "%s image memory kind %u", name, im->memory);
    return STEPTH_OK;
}

struct KernelTimer { // brackets the dominant kernel with events when timing is enabled
    stepth_engine *e;
    size_t slot = (size_t)-1;
    explicit KernelTimer(stepth_engine *eng) : e(eng)
    {
        if (!e->timing)
            return;
        if (e->events_used == e->events.size()) {
            hipEvent_t a = nullptr, b = nullptr;
            if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess)
                return;
            e->events.emplace_back(a, b);
        }
        slot = e->events_used++;
        (void)hipEventRecord(e->events[slot].first, e->stream);
    }
    ~KernelTimer()
    {
        if (slot != (size_t)-1)
            (void)hipEventRecord(e->events[slot].second, e->stream);
    }
};


// The dense search (tile_kernels.h) serves one group per pixel when the search fits its 10 x 10
// superblock slots: rings up to 263 (the reference searches 255, depth_image.rs:119).  Beyond
// that one group per pixel runs as 1x1 tiles of the prepared-query search, whose per-lane ring
// walk has no such limit.
bool tile_search_fits(uint32_t max_steps, uint32_t wa, uint32_t ha)
{
    return ((std::min(max_steps, std::max(wa, ha)) + 6u) >> 3) <= 33u;
}

// RGB8 additional images -> T, TT, padded block bounds, superblock bounds (tile_kernels.h)
int pack_additional_tiles(stepth_engine *e, uint32_t n_frames, const uint8_t *const *d_adds,
                          size_t stride, uint32_t wa, uint32_t ha, uint32_t &bw, uint32_t &bh,
                          uint32_t row_lo, uint32_t row_hi, uint32_t col_hi = 0xFFFFFFFFu)
{
    bw = (wa + 7u) / 8u;
    bh = (ha + 7u) / 8u;
    const size_t nblk = (size_t)bw * bh;
    if (nblk * 64 * n_frames > 0x3FFFFFFFull)
        return fail(e, STEPTH_ERR_BAD_ARG, "additional image %ux%u x %u frames exceeds 2^30 cells",
                    wa, ha, n_frames);
    const uint32_t sbw = (bw + 7u) / 8u, sbh = (bh + 7u) / 8u;
    const uint32_t bwp = tile_bwp(bw), bhp = tile_bhp(bh), sbwp = tile_sbwp(sbw);
    const size_t bb_frame = (size_t)bwp * bhp, sb_frame = (size_t)sbwp * sbh;
    TRY(reserve(e, e->T, nblk * 64 * 4 * n_frames));
    TRY(reserve(e, e->TT, nblk * 64 * 4 * n_frames));
    TRY(reserve(e, e->LB, nblk * 16 * 8 * n_frames));
    const size_t bbp_cap = e->BBp.cap;
    TRY(reserve(e, e->BBp, bb_frame * 8 * n_frames));
    TRY(reserve(e, e->SBp, sb_frame * 8 * n_frames));
    if (e->BBp.cap != bbp_cap || e->bbp_layout[0] != bw || e->bbp_layout[1] != bh ||
        e->bbp_layout[2] < n_frames) {
        // the border around the image's blocks must be empty bounds (min = kNoCell, never matching)
        HIP_TRY(e, hipMemsetD32Async((hipDeviceptr_t)e->BBp.p, (int)kNoCell, bb_frame * 2 * n_frames, e->stream));
        // and so is every superblock until tile_super_kernel has folded it: a strip call leaves
        // the rows outside its reach as they are -- empty, or an earlier image's bounds, never
        // allocator contents (the search's own dmin tests keep both from being probed)
        HIP_TRY(e, hipMemsetD32Async((hipDeviceptr_t)e->SBp.p, (int)kNoCell, sb_frame * 2 * n_frames, e->stream));
        e->bbp_layout[0] = bw;
        e->bbp_layout[1] = bh;
        e->bbp_layout[2] = n_frames;
    }
    row_hi = std::min(row_hi, ha);
    row_lo = std::min(row_lo, row_hi);
    // (a strip that reaches no row at all still packs one, so that the launches are never empty)
    const uint32_t by0 = std::min(row_lo / 8u, bh - 1u);
    const uint32_t by1 = std::min(std::max((row_hi + 7u) / 8u, by0 + 1u), bh);
    TilePackArgs pa{};
    for (uint32_t f = 0; f < n_frames; ++f)
        pa.src[f] = d_adds[f];
    pa.stride = stride;
    pa.wa = wa;
    pa.ha = ha;
    pa.bw = bw;
    pa.bh = bh;
    pa.bwp = bwp;
    pa.T = (uint32_t *)e->T.p;
    pa.TT = (uint32_t *)e->TT.p;
    pa.LB = (uint2 *)e->LB.p;
    pa.BBp = (uint2 *)e->BBp.p;
    pa.by0 = by0;
    pa.nby = by1 - by0;
    pa.zero = e->words;
    pa.n_zero = (uint32_t)(kWordsBytes / 4);
    pa.t_stride = (uint32_t)(nblk * 64);
    pa.bb_stride = (uint32_t)bb_frame;
    // columns the searches can reach (centres lie in the left half of the array, like the rows in
    // its upper half: middle = ((x + w) / 2, (y + h) / 2), src/depth_image.rs:114-117): blocks
    // further right keep whatever bounds they had -- empty or an earlier image's -- and are never
    // probed, exactly like the rows outside [by0, by1) (their nearest cell is beyond every limit)
    const uint32_t bcols = std::min<uint64_t>(bw, ((uint64_t)col_hi + 7u) / 8u);
    hipLaunchKernelGGL(tile_pack_kernel, dim3((std::max(bcols, 1u) + kPackBlocks - 1u) / kPackBlocks, pa.nby, n_frames),
                       dim3(256), 0, e->stream, pa);
    TileSuperArgs sa{};
    sa.BBp = pa.BBp;
    sa.bw = bw;
    sa.bh = bh;
    sa.bwp = bwp;
    sa.sbw = sbw;
    sa.sbwp = sbwp;
    sa.SBp = (uint2 *)e->SBp.p;
    sa.sby0 = by0 / 8u;
    sa.nsby = (by1 + 7u) / 8u - by0 / 8u;
    sa.by0 = by0;
    sa.by1 = by1;
    sa.bb_stride = pa.bb_stride;
    sa.sb_stride = (uint32_t)sb_frame;
    hipLaunchKernelGGL(tile_super_kernel, dim3((sbw * sa.nsby + 3u) / 4u, n_frames), dim3(256), 0,
                       e->stream, sa);
    HIP_TRY(e, hipGetLastError());
    return STEPTH_OK;
}

// PASS 1 for one group per pixel, all frames of a chunk in one launch (tile_kernels.h).
// `ra` carries the geometry the ring0/search path would use (main images, strip, raw).
int launch_tile(stepth_engine *e, const RingArgs &ra, uint32_t n_frames)
{
    const uint32_t rows = ra.row1 - ra.row0;
    if (rows == 0 || ra.w == 0)
        return STEPTH_OK;
    TileArgs ta{};
    ta.T = (const uint32_t *)e->T.p;
    ta.TT = (const uint32_t *)e->TT.p;
    ta.LB = (const uint2 *)e->LB.p;
    ta.BBp = (const uint2 *)e->BBp.p;
    ta.SBp = (const uint2 *)e->SBp.p;
    ta.wa = ra.wa;
    ta.ha = ra.ha;
    ta.bw = ra.bw;
    ta.bh = ra.bh;
    ta.bwp = tile_bwp(ra.bw);
    ta.sbwp = tile_sbwp(ra.sbw);
    ta.t_stride = ra.bw * ra.bh * 64u;
    ta.bb_stride = ta.bwp * tile_bhp(ra.bh);
    ta.sb_stride = ta.sbwp * ra.sbh;
    for (uint32_t f = 0; f < n_frames; ++f)
        ta.main_px[f] = ra.main_px[f];
    ta.mstride = ra.mstride;
    ta.mch = ra.mch;
    ta.w = ra.w;
    ta.row0 = ra.row0;
    ta.row1 = ra.row1;
    ta.rp = ra.rp;
    // centres of the strip's pixels: cx = (x+1)/2 in 0..w/2, cy = (y+1)/2 in (row0+1)/2..row1/2
    ta.cbx_n = ((ra.w >> 1) + 8u) >> 3;
    ta.cby0 = ((ra.row0 + 1u) >> 1) >> 3;
    ta.cby_n = ((ra.row1 >> 1) >> 3) - ta.cby0 + 1u;
    ta.n_rows = ta.cby_n * n_frames;
    const uint64_t n_wg = (uint64_t)((ta.n_rows + 7u) / 8u) * 8u * ta.cbx_n; // rows dealt to 8 XCDs
    if (n_wg > 0x1FFFFFF0ull)
        return fail(e, STEPTH_ERR_BAD_ARG, "too many centre blocks in one launch");
    // a search never leaves the array: more steps than its longer side change nothing
    ta.max_steps = std::min(ra.max_steps, std::max(ra.wa, ra.ha));
    {
        const uint32_t top = std::max(ta.max_steps, 2u) - 1u;
        uint32_t pb = 1;
        while ((2u * top) >> pb)
            ++pb;
        ta.pos_bits = pb;
    }
    memcpy(ta.prec, ra.prec, 3);
    ta.raw = ra.raw;
    ta.raw_stride = (uint32_t)(((size_t)ra.n + 3u) & ~(size_t)3u);
    ta.max_word = ra.max_word;
    ta.stats = e->stats_enabled ? e->stats : nullptr;
    const dim3 grid((uint32_t)n_wg * 4u), block(64); // one wave per workgroup, four per tile
    // heavy-first launch order (STEPTH_FLAG_HEAVY_FIRST): worth its three small launches only
    // when the search kernel runs many rounds of workgroups
    const uint32_t *verdict = nullptr; // the scan kernel's: 1 = identity order, plain variant
    if (ra.heavy_first && n_wg >= 8192u && !e->stats_enabled) { // (a 1080p frame has 8228)
        TileOrderBufs ob{};
        ob.n_wg = (uint32_t)n_wg;
        ob.n_groups = (uint32_t)((n_wg + 63u) / 64u);
        TRY(reserve(e, e->order_bytes, (size_t)ob.n_groups * (64u + 64u + 16u)));
        TRY(reserve(e, e->order_words, ((size_t)ob.n_groups * 16u + 32u + n_wg) * 4u));
        ob.cls = (uint8_t *)e->order_bytes.p;
        ob.rank = ob.cls + (size_t)ob.n_groups * 64u;
        ob.cnt = ob.rank + (size_t)ob.n_groups * 64u;
        ob.off = (uint32_t *)e->order_words.p;
        ob.order = ob.off + (size_t)ob.n_groups * 16u + 32u; // (17 words of class bases in between)
        hipLaunchKernelGGL(tile_cost_kernel, dim3(ob.n_groups), dim3(1024), 0, e->stream, ta, ob);
        hipLaunchKernelGGL(tile_scan_kernel, dim3(1), dim3(1024), 0, e->stream, ob);
        hipLaunchKernelGGL(tile_scatter_kernel, dim3((uint32_t)((n_wg + 255u) / 256u)), dim3(256), 0, e->stream, ob, 0u);
        ta.order = ob.order;
        verdict = ob.off + (size_t)ob.n_groups * 16u + 16u;
    }
    {
        KernelTimer t(e); // the dominant kernel (bench.py's roofline leg)
        // (STEPTH_FLAG_HEAVY_FIRST = natural images: the pooled-probe variant goes with the order)
        if (e->stats_enabled)
            hipLaunchKernelGGL((ring_tile_kernel<true, false>), grid, block, 0, e->stream, ta);
        else if (verdict)
            hipLaunchKernelGGL(ring_tile_auto_kernel, grid, block, 0, e->stream, ta, verdict);
        else if (ra.heavy_first) // (a small launch: no order, probe tasks all the same -- latency is all there is)
            hipLaunchKernelGGL((ring_tile_kernel<false, true>), grid, block, 0, e->stream, ta);
        else
            hipLaunchKernelGGL((ring_tile_kernel<false, false>), grid, block, 0, e->stream, ta);
    }
    HIP_TRY(e, hipGetLastError());
    return STEPTH_OK;
}

// PASS 1 for prepared queries (group lists, tiles > 1x1, distance_dot_array_batch) on the
// per-lane kernel (tile_kernels.h: ring_list_kernel); `ra` as for launch_ring with qmode 1.
int launch_list(stepth_engine *e, const RingArgs &ra, uint32_t n_frames, const PaintArgs *paint = nullptr)
{
    if (ra.n == 0)
        return STEPTH_OK;
    ListArgs la{};
    if (paint) { // the search paints raw values itself (small rectangles; big ones are queued)
        TRY(reserve(e, e->pending, (size_t)ra.n * n_frames * 4));
        for (uint32_t f = 0; f < n_frames; ++f) {
            la.paint_groups[f] = paint->groups_list[f];
            la.paint_out[f] = paint->out[f];
        }
        la.paint_stride = paint->ostride;
        la.paint_w = paint->w;
        la.paint_h = paint->h;
        la.paint_row0 = paint->row_begin;
        la.paint_row1 = paint->row_end;
        la.big_count = e->words + kWordCounters; // (zeroed with the other words by the pack kernel)
        la.big_list = (uint32_t *)e->pending.p;
        la.big_cap = ra.n;
    }
    la.T = (const uint32_t *)e->T.p;
    la.TT = (const uint32_t *)e->TT.p;
    la.LB = (const uint2 *)e->LB.p;
    la.BBp = (const uint2 *)e->BBp.p;
    la.SBp = (const uint2 *)e->SBp.p;
    la.wa = ra.wa;
    la.ha = ra.ha;
    la.bw = ra.bw;
    la.bh = ra.bh;
    la.bwp = tile_bwp(ra.bw);
    la.sbwp = tile_sbwp(ra.sbw);
    la.t_stride = ra.bw * ra.bh * 64u;
    la.bb_stride = la.bwp * tile_bhp(ra.bh);
    la.sb_stride = la.sbwp * ra.sbh;
    la.queries = ra.queries;
    la.ids = ra.query_ids;
    la.q_stride = ra.q_stride;
    for (uint32_t f = 0; f < n_frames; ++f)
        la.nq[f] = ra.nq[f];
    la.max_steps = std::min(ra.max_steps, std::max(ra.wa, ra.ha));
    {
        const uint32_t top = std::max(la.max_steps, 2u) - 1u;
        uint32_t pb = 1;
        while ((2u * top) >> pb)
            ++pb;
        la.pos_bits = pb;
    }
    la.raw = ra.raw;
    la.raw_stride = (uint32_t)(((size_t)ra.n + 3u) & ~(size_t)3u);
    la.raw32 = ra.raw32;
    la.pos = ra.pos;
    la.found = ra.found;
    la.max_word = ra.max_word;
    la.stats = e->stats_enabled ? e->stats : nullptr;
    // few queries (one frame's group list, a thin strip): fewer queries per wave, more waves
    const uint64_t total = (uint64_t)ra.n * n_frames;
    la.lanes = total >= (1u << 18) ? 64u : total >= (1u << 16) ? 32u : 16u;
    la.wpf = (ra.n + la.lanes - 1u) / la.lanes;
    const uint64_t n_waves = (uint64_t)la.wpf * n_frames;
    if (n_waves > 0x7FFFFFF0ull)
        return fail(e, STEPTH_ERR_BAD_ARG, "too many queries in one launch");
    const dim3 grid((uint32_t)n_waves), block(64); // one wave per workgroup, all frames in one launch order
    // heavy-first order of the waves: an estimate per wave, classes, scan, scatter -- four small
    // launches (~30 us).  Measured on 8 x 1080p group lists: reach 255 0.574 -> 0.480 ms (the
    // waves that search far start first); reach 128 nothing (0.291 / 0.364 ms either way: there
    // the kernel is as long as its longest wave wherever that starts) -- so only for long reaches.
    if (la.lanes == 64u && n_waves >= 8192u && la.max_steps > 160u && !e->stats_enabled) {
        TileOrderBufs ob{};
        ob.n_wg = (uint32_t)n_waves;
        ob.n_groups = (uint32_t)((n_waves + 63u) / 64u);
        TRY(reserve(e, e->order_bytes, (size_t)ob.n_groups * (64u + 64u + 16u)));
        TRY(reserve(e, e->order_words, ((size_t)ob.n_groups * 16u + 32u + n_waves) * 4u));
        ob.cls = (uint8_t *)e->order_bytes.p;
        ob.rank = ob.cls + (size_t)ob.n_groups * 64u;
        ob.cnt = ob.rank + (size_t)ob.n_groups * 64u;
        ob.off = (uint32_t *)e->order_words.p;
        ob.order = ob.off + (size_t)ob.n_groups * 16u + 32u;
        ListCostArgs ca{};
        ca.BBp = la.BBp;
        ca.SBp = la.SBp;
        ca.wa = la.wa;
        ca.ha = la.ha;
        ca.bwp = la.bwp;
        ca.sbwp = la.sbwp;
        ca.bb_stride = la.bb_stride;
        ca.sb_stride = la.sb_stride;
        ca.queries = la.queries;
        ca.q_stride = la.q_stride;
        ca.wpf = la.wpf;
        ca.max_steps = la.max_steps;
        for (uint32_t f = 0; f < n_frames; ++f)
            ca.nq[f] = la.nq[f];
        hipLaunchKernelGGL(list_cost_kernel, dim3((uint32_t)((n_waves + 3u) / 4u)), dim3(256), 0, e->stream, ca, ob);
        hipLaunchKernelGGL(list_rank_kernel, dim3(ob.n_groups), dim3(64), 0, e->stream, ob);
        hipLaunchKernelGGL(tile_scan_kernel, dim3(1), dim3(1024), 0, e->stream, ob);
        hipLaunchKernelGGL(tile_scatter_kernel, dim3((uint32_t)((n_waves + 255u) / 256u)), dim3(256), 0, e->stream, ob, 1u);
        la.order = ob.order;
    }
    {
        KernelTimer t(e);
        if (e->stats_enabled)
            hipLaunchKernelGGL(ring_list_kernel<true>, grid, block, 0, e->stream, la);
        else
            hipLaunchKernelGGL(ring_list_kernel<false>, grid, block, 0, e->stream, la);
    }
    HIP_TRY(e, hipGetLastError());
    return STEPTH_OK;
}


// image::imageops::sample::gaussian(x, 0.5) and the per-border-class tap weights of a
// same-size resample (support 3): taps with offsets cl-3 .. 3-cr, normalised by their sum
void gaussian_weights(float w[4][4][7])
{
    for (int cl = 0; cl < 4; ++cl)
        for (int cr = 0; cr < 4; ++cr) {
            float sum = 0.0f;
            for (int k = 0; k < 7; ++k)
                w[cl][cr][k] = 0.0f;
            for (int k = cl - 3; k <= 3 - cr; ++k) {
                // in double, rounded once: independent of any particular expf()
                const double x = (double)k, r = 0.5;
                const float g = (float)((1.0 / (std::sqrt(2.0 * 3.14159265358979323846) * r)) *
                                        std::exp(-(x * x) / (2.0 * r * r)));
                w[cl][cr][k + 3] = g;
                sum += g;
            }
            for (int k = cl - 3; k <= 3 - cr; ++k)
                w[cl][cr][k + 3] /= sum;
        }
}

// vertical pass src -> gauss_tmp, horizontal pass gauss_tmp -> dst (device pointers)
// same-size Gaussian resample of nf planes of one shape (src[f] -> dst[f], may be the same)
int gaussian_device_batch(stepth_engine *e, uint32_t nf, const uint8_t *const *src, size_t sstride,
                          uint32_t w, uint32_t h, uint8_t *const *dst, size_t dstride)
{
    const size_t plane = ((size_t)w * h + 15) & ~(size_t)15;
    TRY(reserve(e, e->gauss_tmp, plane * nf));
    // four pixels per thread when everything is dword aligned (the usual case)
    bool vec4 = (w & 3u) == 0 && (sstride & 3u) == 0 && (dstride & 3u) == 0;
    for (uint32_t f = 0; f < nf; ++f)
        vec4 = vec4 && ((reinterpret_cast<uintptr_t>(src[f]) | reinterpret_cast<uintptr_t>(dst[f])) & 3u) == 0;
    if (vec4) {
        Gauss4Args g4{};
        gaussian_weights(g4.wt);
        g4.w = w;
        g4.h = h;
        uint8_t *tmp4 = static_cast<uint8_t *>(e->gauss_tmp.p);
        const dim3 grid4((uint32_t)std::min<size_t>(((size_t)(w / 4) * h + 255) / 256, 4096), nf);
        for (uint32_t f = 0; f < nf; ++f) {
            g4.src[f] = src[f];
            g4.dst[f] = tmp4 + plane * f;
        }
        g4.sstride = sstride;
        g4.dstride = w;
        hipLaunchKernelGGL(gauss_vertical4_kernel, grid4, dim3(256), 0, e->stream, g4);
        for (uint32_t f = 0; f < nf; ++f) {
            g4.src[f] = tmp4 + plane * f;
            g4.dst[f] = dst[f];
        }
        g4.sstride = w;
        g4.dstride = dstride;
        hipLaunchKernelGGL(gauss_horizontal4_kernel, grid4, dim3(256), 0, e->stream, g4);
        HIP_TRY(e, hipGetLastError());
        return STEPTH_OK;
    }
    GaussArgs ga{};
    gaussian_weights(ga.w);
    const size_t total = (size_t)w * h;
    const dim3 grid((uint32_t)std::min<size_t>((total + 255) / 256, 4096), nf);
    uint8_t *tmp = static_cast<uint8_t *>(e->gauss_tmp.p);
    for (uint32_t f = 0; f < nf; ++f) {
        ga.src[f] = src[f];
        ga.dst[f] = tmp + plane * f;
    }
    ga.s_elem = sstride; // vertical_sample: along rows
    ga.s_line = 1;
    ga.n = h;
    ga.lines = w;
    ga.d_elem = w;
    ga.d_line = 1;
    hipLaunchKernelGGL(gauss_pass_kernel, grid, dim3(256), 0, e->stream, ga);
    for (uint32_t f = 0; f < nf; ++f) {
        ga.src[f] = tmp + plane * f; // horizontal_sample
        ga.dst[f] = dst[f];
    }
    ga.s_elem = 1;
    ga.s_line = w;
    ga.n = w;
    ga.lines = h;
    ga.d_elem = 1;
    ga.d_line = dstride;
    hipLaunchKernelGGL(gauss_pass_kernel, grid, dim3(256), 0, e->stream, ga);
    HIP_TRY(e, hipGetLastError());
    return STEPTH_OK;
}

int gaussian_device(stepth_engine *e, const uint8_t *src, size_t sstride, uint32_t w, uint32_t h,
                    uint8_t *dst, size_t dstride)
{
    return gaussian_device_batch(e, 1, &src, sstride, w, h, &dst, dstride);
}

} // namespace

extern "C" {

uint32_t stepth_abi_version(void) { return STEPTH_ABI_VERSION; }

const char *stepth_status_string(int status)
{
    switch (status) {
    case STEPTH_OK: return "ok";
    case STEPTH_ERR_BAD_ARG: return "bad argument";
    case STEPTH_ERR_SIZE_MISMATCH: return "Sizes don't match"; // src/depth_image.rs:45-48
    case STEPTH_ERR_HIP: return "HIP runtime error";
    case STEPTH_ERR_MAX_ZERO: return "every distance is zero (reference would divide by zero)";
    case STEPTH_ERR_NO_DEVICE: return "no HIP device";
    case STEPTH_ERR_OUT_OF_MEMORY: return "out of device memory";
    case STEPTH_ERR_SPLIT_STEP:
        return "depth range too small for that many zones (reference panics in step_by)";
    case STEPTH_ERR_SPLIT_EMPTY_ZONE: return "a zone ended up empty (reference panics on unwrap)";
    case STEPTH_ERR_SPLIT_DIVERGED: return "zone centroids did not settle";
    default: return "unknown status";
    }
}

int stepth_engine_create(int device_id, stepth_engine **out)
{
    STEPTH_GUARD_BEGIN
    if (!out)
        return STEPTH_ERR_BAD_ARG;
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0)
        return STEPTH_ERR_NO_DEVICE;
    if (device_id < 0 && hipGetDevice(&device_id) != hipSuccess)
        return STEPTH_ERR_NO_DEVICE;
    if (device_id >= count)
        return STEPTH_ERR_NO_DEVICE;
    stepth_engine *e = new (std::nothrow) stepth_engine();
    if (!e)
        return STEPTH_ERR_OUT_OF_MEMORY;
    e->device = device_id;
    if (hipSetDevice(device_id) != hipSuccess ||
        hipStreamCreateWithFlags(&e->own_stream, hipStreamNonBlocking) != hipSuccess ||
        hipMalloc((void **)&e->words, kWordsBytes) != hipSuccess ||
        hipMalloc((void **)&e->stats, kStatsBytes) != hipSuccess) {
        stepth_engine_destroy(e);
        return STEPTH_ERR_HIP;
    }
    e->stream = e->own_stream;
    (void)hipMemsetAsync(e->words, 0, kWordsBytes, e->stream);
    (void)hipMemsetAsync(e->stats, 0, kStatsBytes, e->stream);
    *out = e;
    return STEPTH_OK;
    STEPTH_GUARD_END(nullptr)
}

void stepth_engine_destroy(stepth_engine *e)
{
    if (!e)
        return;
    (void)hipSetDevice(e->device);
    (void)hipStreamSynchronize(e->stream);
    for (DevBuf *b : {&e->T, &e->TT, &e->LB, &e->BBp, &e->SBp, &e->stage_main, &e->stage_add, &e->stage_out,
                      &e->stage_groups, &e->queries, &e->queries_sorted, &e->query_ids, &e->query_bins, &e->raw, &e->aux32, &e->aux_pos,
                      &e->aux_found, &e->pending, &e->gauss_tmp, &e->gauss_io,
                      &e->zone_counts, &e->zone_tables, &e->pipe_main[0], &e->pipe_main[1],
                      &e->pipe_add[0], &e->pipe_add[1], &e->pipe_out[0], &e->pipe_out[1], &e->pipe_max, &e->group_out,
                      &e->group_list[0], &e->group_list[1], &e->group_list[2], &e->group_counters, &e->order_bytes, &e->order_words})
        release(*b);
    for (hipEvent_t ev : {e->ev_in[0], e->ev_in[1], e->ev_compute[0], e->ev_compute[1], e->ev_out[0],
                          e->ev_out[1], e->ev_switch})
        if (ev)
            (void)hipEventDestroy(ev);
    for (hipStream_t st : {e->copy_in, e->copy_out})
        if (st)
            (void)hipStreamDestroy(st);
    if (e->words)
        (void)hipFree(e->words);
    if (e->stats)
        (void)hipFree(e->stats);
    for (auto &ev : e->events) {
        (void)hipEventDestroy(ev.first);
        (void)hipEventDestroy(ev.second);
    }
    if (e->own_stream)
        (void)hipStreamDestroy(e->own_stream);
    delete e;
}

int stepth_host_alloc(void **out, size_t bytes)
{
    STEPTH_GUARD_BEGIN
    if (!out || bytes == 0)
        return STEPTH_ERR_BAD_ARG;
    *out = nullptr;
    const hipError_t r = hipHostMalloc(out, bytes, hipHostMallocDefault);
    if (r == hipErrorOutOfMemory)
        return STEPTH_ERR_OUT_OF_MEMORY;
    if (r == hipErrorNoDevice || r == hipErrorInvalidDevice)
        return STEPTH_ERR_NO_DEVICE;
    return r == hipSuccess ? STEPTH_OK : STEPTH_ERR_HIP;
    STEPTH_GUARD_END(nullptr)
}

void stepth_host_free(void *ptr)
{
    if (ptr)
        (void)hipHostFree(ptr);
}

// The engine's work moves to another stream: what the old stream still holds is ordered before
// whatever the new one gets (an event the new stream waits for -- the host is not blocked).
static int switch_stream(stepth_engine *e, hipStream_t to)
{
    if (to == e->stream)
        return STEPTH_OK;
    HIP_TRY(e, hipSetDevice(e->device));
    if (!e->ev_switch)
        HIP_TRY(e, hipEventCreateWithFlags(&e->ev_switch, hipEventDisableTiming));
    HIP_TRY(e, hipEventRecord(e->ev_switch, e->stream));
    HIP_TRY(e, hipStreamWaitEvent(to, e->ev_switch, 0));
    e->stream = to;
    return STEPTH_OK;
}

int stepth_engine_set_stream(stepth_engine *e, void *hip_stream)
{
    STEPTH_GUARD_BEGIN
    if (!e)
        return STEPTH_ERR_BAD_ARG;
    return switch_stream(e, static_cast<hipStream_t>(hip_stream)); // nullptr = HIP's default stream
    STEPTH_GUARD_END(e)
}

int stepth_engine_reset_stream(stepth_engine *e)
{
    STEPTH_GUARD_BEGIN
    if (!e)
        return STEPTH_ERR_BAD_ARG;
    return switch_stream(e, e->own_stream);
    STEPTH_GUARD_END(e)
}

int stepth_engine_sync(stepth_engine *e)
{
    STEPTH_GUARD_BEGIN
    if (!e)
        return STEPTH_ERR_BAD_ARG;
    HIP_TRY(e, hipStreamSynchronize(e->stream));
    return STEPTH_OK;
    STEPTH_GUARD_END(e)
}

const char *stepth_last_error(const stepth_engine *e) { return e ? e->err.c_str() : "null engine"; }

static int ring_queries_batch(stepth_engine *e, uint32_t nf, const stepth_image *mains,
                              const stepth_image *adds, const stepth_params &p, uint32_t row0,
                              uint32_t row1, RowPattern rp, const void *const *d_groups,
                              const uint32_t *n_groups, const stepth_image *depths,
                              uint32_t *max_out, bool &any_zero, int (*late_counts)(stepth_engine *, uint32_t, uint32_t *) = nullptr,
                              uint32_t *late_buf = nullptr);

// What the three modes of stepth_depth_from_additional share: the validated call, the device
// views of its images (staged copies for host images), the strip.
struct SingleCall {
    stepth_params p;
    const stepth_image *main_im, *add_im;
    uint32_t W, H, row0, row1;
    RowPattern rp;
    bool async, raw_only;
    const uint8_t *d_main, *d_add;
    size_t ms, as, os;
    uint8_t *d_out;
    uint32_t *max_out;
    bool max_published; // a paint kernel already stored the max at max_out
};

// windowed-SAD mode (not in the reference; sad_kernels.h)
static int single_sad(stepth_engine *e, SingleCall &c, const stepth_group *groups)
{
    const stepth_params &p = c.p;
    const stepth_image *main_im = c.main_im, *add_im = c.add_im;
    const uint32_t W = c.W, H = c.H, row0 = c.row0, row1 = c.row1;
    const RowPattern rp = c.rp;
    const bool async = c.async, raw_only = c.raw_only;
    const uint8_t *d_main = c.d_main, *d_add = c.d_add;
    const size_t ms = c.ms, as = c.as, os = c.os;
    uint8_t *d_out = c.d_out;
    uint32_t *max_out = c.max_out;
    bool &max_published = c.max_published;
    (void)main_im, (void)add_im, (void)H, (void)rp, (void)d_main, (void)ms, (void)as, (void)raw_only;
    HIP_TRY(e, hipMemsetAsync(e->words, 0, kWordsBytes, e->stream));
    if (groups)
        return fail(e, STEPTH_ERR_BAD_ARG, "SAD mode takes no group list");
    if (add_im->width != W || add_im->height != H)
        return fail(e, STEPTH_ERR_SIZE_MISMATCH,
                    "Sizes don't match: SAD mode needs equal image sizes");
    if ((p.win & 1u) == 0 || p.win > 15 || p.d_max == 0 || p.d_max > 256)
        return fail(e, STEPTH_ERR_BAD_ARG, "SAD mode: win must be odd <= 15, d_max in 1..256");
    const uint32_t rows = row1 - row0;
    TRY(reserve(e, e->raw, (size_t)W * std::max(rows, 1u)));
    if (rows) {
        SadArgs sa{};
        sa.left[0] = d_main;
        sa.lstride = ms;
        sa.lch = main_im->channels;
        sa.right[0] = d_add;
        sa.rstride = as;
        sa.n_frames = 1;
        sa.w = W;
        sa.h = H;
        sa.row0 = row0;
        sa.row1 = row1;
        sa.d_max = p.d_max;
        sa.win = p.win;
        sa.thr = p.win * p.win * ((uint32_t)p.precision[0] + p.precision[1] + p.precision[2]);
        sa.raw = static_cast<uint8_t *>(e->raw.p);
        sa.max_word = e->words;
        {
            KernelTimer t(e);
            int s = launch_sad(sa, e->stream);
            if (s != STEPTH_OK)
                return fail(e, s, "sad_match launch failed: %s",
                            hipGetErrorString(hipGetLastError()));
        }
        PaintArgs pa{};
        pa.raw = sa.raw;
        pa.max_word = e->words;
        pa.max_out = async ? max_out : nullptr;
        max_published = async && max_out;
        pa.identity = raw_only;
        pa.out[0] = d_out;
        pa.ostride = os;
        pa.w = W;
        pa.h = H;
        pa.row_begin = row0;
        pa.row_end = row1;
        pa.tile_w = pa.tile_h = 1;
        pa.gx = W;
        const size_t total = (size_t)rows * W;
        const uint32_t blocks = (uint32_t)std::min<size_t>((total + 255) / 256, 2048);
        hipLaunchKernelGGL(paint_tiles_kernel, dim3(blocks), dim3(256), 0, e->stream, pa);
        HIP_TRY(e, hipGetLastError());
    }
    return STEPTH_OK;
}

// prepared queries (an explicit group list, or uniform tiles larger than one pixel): the same
// code as a batch of one frame, on the device copies of the images
static int single_prepared(stepth_engine *e, SingleCall &c, const stepth_group *groups, size_t n_groups,
                           uint32_t groups_memory)
{
    const stepth_params &p = c.p;
    const stepth_image *main_im = c.main_im, *add_im = c.add_im;
    const uint32_t W = c.W, H = c.H, row0 = c.row0, row1 = c.row1;
    const RowPattern rp = c.rp;
    const bool async = c.async, raw_only = c.raw_only;
    const uint8_t *d_main = c.d_main, *d_add = c.d_add;
    const size_t ms = c.ms, as = c.as, os = c.os;
    uint8_t *d_out = c.d_out;
    uint32_t *max_out = c.max_out;
    bool &max_published = c.max_published;
    (void)main_im, (void)add_im, (void)H, (void)rp, (void)d_main, (void)ms, (void)as, (void)raw_only;
    // prepared queries (an explicit group list, or uniform tiles larger than one pixel): the
    // same code as a batch of one frame, on the device copies of the images
    if (!groups && (p.tile_w == 0 || p.tile_h == 0))
        return fail(e, STEPTH_ERR_BAD_ARG, "tile size 0");
    if (!groups && (row0 % p.tile_h != 0 || (row1 % p.tile_h != 0 && row1 != H)))
        return fail(e, STEPTH_ERR_BAD_ARG, "row strip [%u,%u) not aligned to tile_h %u", row0, row1,
                    p.tile_h);
    const void *d_groups = nullptr;
    uint32_t ng = 0;
    if (groups) {
        if (n_groups == 0 || n_groups > 0xFFFFFF00ull)
            return fail(e, STEPTH_ERR_BAD_ARG, "group list of %zu entries", n_groups);
        ng = (uint32_t)n_groups;
        if (groups_memory == STEPTH_MEM_DEVICE) {
            d_groups = groups;
        } else {
            TRY(reserve(e, e->stage_groups, n_groups * sizeof(stepth_group)));
            HIP_TRY(e, hipMemcpyAsync(e->stage_groups.p, groups, n_groups * sizeof(stepth_group),
                                      hipMemcpyHostToDevice, e->stream));
            d_groups = e->stage_groups.p;
        }
    }
    const stepth_image dm{d_main, W, H, main_im->channels, STEPTH_MEM_DEVICE, ms};
    const stepth_image da{d_add, add_im->width, add_im->height, 3, STEPTH_MEM_DEVICE, as};
    const stepth_image dd{d_out, W, H, 1, STEPTH_MEM_DEVICE, os};
    stepth_params ap = p;
    ap.flags |= STEPTH_FLAG_ASYNC; // this function reads the max back itself
    bool any_zero = false;
    TRY(ring_queries_batch(e, 1, &dm, &da, ap, row0, row1, rp, groups ? &d_groups : nullptr, &ng, &dd,
                           async ? max_out : nullptr, any_zero));
    max_published = async && max_out && row1 > row0;
    return STEPTH_OK;
}

// one group per pixel: the dense lane = query search (tile_kernels.h).  (Beyond its reach --
// more than 263 rings; the reference always searches 255 -- the caller goes to single_prepared
// with 1x1 tiles.)
static int single_dense(stepth_engine *e, SingleCall &c)
{
    const stepth_params &p = c.p;
    const stepth_image *main_im = c.main_im, *add_im = c.add_im;
    const uint32_t W = c.W, H = c.H, row0 = c.row0, row1 = c.row1;
    const RowPattern rp = c.rp;
    const bool async = c.async, raw_only = c.raw_only;
    const uint8_t *d_main = c.d_main, *d_add = c.d_add;
    const size_t ms = c.ms, as = c.as, os = c.os;
    uint8_t *d_out = c.d_out;
    uint32_t *max_out = c.max_out;
    bool &max_published = c.max_published;
    (void)main_im, (void)add_im, (void)H, (void)rp, (void)d_main, (void)ms, (void)as, (void)raw_only;
    // one group per pixel
    uint32_t bw = 0, bh = 0;
    uint32_t lo = 0, hi = 0xFFFFFFFFu, col_hi = 0xFFFFFFFFu;
    if (p.max_steps < 0x40000000u) {
        // rows of the additional image the strip's searches can reach: the centres lie in
        // rows [row0/2, (row1+1)/2], every search stays within max_steps-1 rings of its centre
        const uint32_t reach = p.max_steps ? p.max_steps - 1u : 0u;
        const uint32_t c_lo = row0 / 2u, c_hi = (row1 + 1u) / 2u;
        lo = c_lo > reach ? c_lo - reach : 0u;
        hi = c_hi + reach + 1u;
        col_hi = (W + 1u) / 2u + reach + 1u;
    }
    TRY(pack_additional_tiles(e, 1, &d_add, as, add_im->width, add_im->height, bw, bh, lo, hi, col_hi));

    RingArgs ra{};
    ra.T = (const uint32_t *)e->T.p;
    ra.sbw = (bw + 7u) / 8u;
    ra.sbh = (bh + 7u) / 8u;
    ra.wa = add_im->width;
    ra.ha = add_im->height;
    ra.bw = bw;
    ra.bh = bh;
    ra.max_steps = p.max_steps;
    memcpy(ra.prec, p.precision, 3);
    ra.max_word = e->words;
    const uint32_t rows = row1 - row0;
    ra.n = W * rows;
    ra.main_px[0] = d_main;
    ra.mstride = ms;
    ra.mch = main_im->channels;
    ra.w = W;
    ra.row0 = row0;
    ra.row1 = row1;
    ra.rp = rp;
    TRY(reserve(e, e->raw, std::max<size_t>(ra.n, 4)));
    ra.raw = static_cast<uint8_t *>(e->raw.p);
    if (e->stats_enabled) {
        HIP_TRY(e, hipMemsetAsync(e->stats, 0, kStatsBytes, e->stream));
        e->stats_groups = ra.n;
    }
    ra.heavy_first = (p.flags & STEPTH_FLAG_HEAVY_FIRST) != 0;
    TRY(launch_tile(e, ra, 1));

    if (rows) {
        PaintArgs pa{};
        pa.max_word = e->words;
        pa.max_out = async ? max_out : nullptr;
        pa.identity = raw_only;
        pa.out[0] = d_out;
        pa.ostride = os;
        pa.w = W;
        pa.h = H;
        pa.row_begin = row0;
        pa.row_end = row1;
        pa.rp = rp;
        pa.tile_w = pa.tile_h = 1;
        pa.gx = W;
        pa.raw = ra.raw;
        pa.n = ra.n;
        max_published = async && max_out;
        if ((W & 3u) == 0 && (os & 3u) == 0 && (reinterpret_cast<uintptr_t>(d_out) & 3u) == 0) {
            const size_t total = (size_t)rows * (W / 4);
            const uint32_t blocks = (uint32_t)std::min<size_t>((total + 255) / 256, 2048);
            hipLaunchKernelGGL(paint_1x1_vec4_kernel, dim3(blocks), dim3(256), 0, e->stream, pa);
        } else {
            const size_t total = (size_t)rows * W;
            const uint32_t blocks = (uint32_t)std::min<size_t>((total + 255) / 256, 2048);
            hipLaunchKernelGGL(paint_tiles_kernel, dim3(blocks), dim3(256), 0, e->stream, pa);
        }
        HIP_TRY(e, hipGetLastError());
    }
    return STEPTH_OK;
}

int stepth_depth_from_additional(stepth_engine *e, const stepth_image *main_im,
                                 const stepth_image *add_im, const stepth_params *pp,
                                 const stepth_group *groups, size_t n_groups,
                                 uint32_t groups_memory, const stepth_image *depth_im,
                                 uint32_t *max_out)
{
    STEPTH_GUARD_BEGIN
    if (!e)
        return STEPTH_ERR_BAD_ARG;
    if (!pp || pp->struct_size != sizeof(stepth_params))
        return fail(e, STEPTH_ERR_BAD_ARG, "params null or struct_size mismatch");
    TRY(check_image(e, main_im, "main", 3, 4));
    TRY(check_image(e, add_im, "additional", 3, 3));
    TRY(check_image(e, depth_im, "depth", 1, 1));
    stepth_params p = *pp;
    const uint32_t W = main_im->width, H = main_im->height;
    // a tile larger than the image is the whole image: keeps the 32-bit tile arithmetic below
    // (strip reach, tile counts) from wrapping
    p.tile_w = std::min(p.tile_w, W);
    p.tile_h = std::min(p.tile_h, H);
    // DepthImage::load_depth, src/depth_image.rs:37-49
    if (depth_im->width != W || depth_im->height != H)
        return fail(e, STEPTH_ERR_SIZE_MISMATCH, "Sizes don't match: depth %ux%u vs image %ux%u",
                    depth_im->width, depth_im->height, W, H);
    const bool async = p.flags & STEPTH_FLAG_ASYNC;
    const bool raw_only = p.flags & STEPTH_FLAG_RAW_ONLY;
    if (async && (main_im->memory != STEPTH_MEM_DEVICE || add_im->memory != STEPTH_MEM_DEVICE ||
                  depth_im->memory != STEPTH_MEM_DEVICE ||
                  (groups && groups_memory != STEPTH_MEM_DEVICE)))
        return fail(e, STEPTH_ERR_BAD_ARG, "STEPTH_FLAG_ASYNC needs device memory everywhere");
    uint32_t row0 = p.row_begin, row1 = p.row_end;
    if (row0 == 0 && row1 == 0)
        row1 = H;
    if (row0 > row1 || row1 > H)
        return fail(e, STEPTH_ERR_BAD_ARG, "row strip [%u,%u) outside 0..%u", row0, row1, H);
    RowPattern rp{0u, 1u, 0u};
    if (p.row_block) {
        if (p.mode != STEPTH_MODE_RING_FIRST_HIT || groups || (p.row_block & 1u) ||
            p.row_mod == 0 || p.row_rem >= p.row_mod || p.tile_h == 0 ||
            p.row_block % p.tile_h != 0 || (p.flags & STEPTH_FLAG_GAUSSIAN_TAIL))
            return fail(e, STEPTH_ERR_BAD_ARG,
                        "row pattern: ring mode with uniform tiles only, even row_block that is "
                        "a multiple of tile_h, row_rem < row_mod, no Gaussian tail");
        rp = RowPattern{p.row_block, p.row_mod, p.row_rem};
    }
    HIP_TRY(e, hipSetDevice(e->device));

    const uint8_t *d_main = nullptr, *d_add = nullptr;
    size_t ms = 0, as = 0;
    if (p.mode == STEPTH_MODE_RING_FIRST_HIT && !groups) // (rows of the strip only; a caller's groups may lie anywhere)
        TRY(to_device(e, *main_im, e->stage_main, d_main, ms, row0, row1));
    else
        TRY(to_device(e, *main_im, e->stage_main, d_main, ms));
    {
        // ring mode without a caller's group list reads the additional image only where a search
        // can reach (see batch_host): the rest of a host image does not travel
        uint32_t ar0 = 0, ar1 = 0xFFFFFFFFu, ac1 = 0xFFFFFFFFu;
        if (p.mode == STEPTH_MODE_RING_FIRST_HIT && !groups && p.max_steps < 0x40000000u && p.tile_w && p.tile_h) {
            const uint32_t reach = p.max_steps ? p.max_steps - 1u : 0u, c_lo = row0 / 2u;
            ar0 = (c_lo > reach ? c_lo - reach : 0u) & ~7u;
            ar1 = (uint32_t)std::min<uint64_t>(0xFFFFFFF8ull, ((uint64_t)(row1 + std::min(p.tile_h, H)) / 2u + reach + 1u + 7u) & ~(uint64_t)7);
            ac1 = (uint32_t)std::min<uint64_t>(0xFFFFFFF8ull, ((uint64_t)(W + std::min(p.tile_w, W)) / 2u + reach + 1u + 7u) & ~(uint64_t)7);
        }
        TRY(to_device(e, *add_im, e->stage_add, d_add, as, ar0, ar1, ac1));
    }

    uint8_t *d_out = nullptr;
    size_t os = 0;
    if (depth_im->memory == STEPTH_MEM_DEVICE) {
        d_out = static_cast<uint8_t *>(const_cast<void *>(depth_im->data));
        os = depth_im->stride;
    } else {
        os = (W + 15u) & ~15u;
        TRY(reserve(e, e->stage_out, os * H));
        d_out = static_cast<uint8_t *>(e->stage_out.p);
        if (rp.block && row1 > row0) // rows of other ranks must come back unchanged
            HIP_TRY(e, hipMemcpy2DAsync(d_out + (size_t)row0 * os, os,
                                        static_cast<const uint8_t *>(depth_im->data) +
                                            (size_t)row0 * depth_im->stride,
                                        depth_im->stride, W, row1 - row0, hipMemcpyHostToDevice,
                                        e->stream));
    }

    SingleCall c{p, main_im, add_im, W, H, row0, row1, rp, async, raw_only, d_main, d_add, ms, as, os, d_out, max_out, false};
    if (p.mode == STEPTH_MODE_SAD_WINDOW)
        TRY(single_sad(e, c, groups));
    else if (p.mode == STEPTH_MODE_RING_FIRST_HIT &&
             (groups || p.tile_w != 1 || p.tile_h != 1 || !tile_search_fits(p.max_steps, add_im->width, add_im->height)))
        TRY(single_prepared(e, c, groups, n_groups, groups_memory)); // (1x1 tiles beyond the dense search's reach too)
    else if (p.mode == STEPTH_MODE_RING_FIRST_HIT)
        TRY(single_dense(e, c));
    else
        return fail(e, STEPTH_ERR_BAD_ARG, "unknown mode %u", p.mode);
    const bool max_published = c.max_published;

    if (p.flags & STEPTH_FLAG_GAUSSIAN_TAIL) { // src/depth_image.rs:133-134
        if (raw_only || row0 != 0 || row1 != H)
            return fail(e, STEPTH_ERR_BAD_ARG,
                        "STEPTH_FLAG_GAUSSIAN_TAIL needs a full-frame, normalised call");
        TRY(gaussian_device(e, d_out, os, W, H, d_out, os));
    }

    if (async) {
        if (max_out && !max_published)
            HIP_TRY(e, hipMemcpyAsync(max_out, e->words, 4, hipMemcpyDeviceToDevice, e->stream));
        return STEPTH_OK;
    }
    uint32_t mx = 0;
    HIP_TRY(e, hipMemcpyAsync(&mx, e->words, 4, hipMemcpyDeviceToHost, e->stream));
    if (depth_im->memory == STEPTH_MEM_HOST && row1 > row0)
        HIP_TRY(e, hipMemcpy2DAsync(
                       static_cast<uint8_t *>(const_cast<void *>(depth_im->data)) +
                           (size_t)row0 * depth_im->stride,
                       depth_im->stride, d_out + (size_t)row0 * os, os, W, row1 - row0,
                       hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(e, hipStreamSynchronize(e->stream));
    if (max_out)
        *max_out = mx;
    if (mx == 0 && !raw_only)
        return fail(e, STEPTH_ERR_MAX_ZERO,
                    "every group's distance is 0; output is all zero (reference: divide by zero, "
                    "src/depth_image.rs:127)");
    return STEPTH_OK;
    STEPTH_GUARD_END(e)
}

static int batch_host(stepth_engine *e, uint32_t n_frames, const stepth_image *mains,
                      const stepth_image *adds, const stepth_params &p, const stepth_image *depths,
                      uint32_t *max_out);

// windowed-SAD mode for a batch of device frames: packs, searches and paints all frames of a
// chunk in one launch each (blockIdx.z / .y = frame)
static int batch_sad(stepth_engine *e, uint32_t n_frames, const stepth_image *mains,
                     const stepth_image *adds, const stepth_params &p, const stepth_image *depths,
                     uint32_t *max_out, uint32_t row0, uint32_t row1)
{
    const uint32_t W = mains[0].width, H = mains[0].height, rows = row1 - row0;
    const bool async = p.flags & STEPTH_FLAG_ASYNC, raw_only = p.flags & STEPTH_FLAG_RAW_ONLY;
    if (p.row_block)
        return fail(e, STEPTH_ERR_BAD_ARG, "SAD mode: contiguous row strips only");
    if (adds[0].width != W || adds[0].height != H)
        return fail(e, STEPTH_ERR_SIZE_MISMATCH, "Sizes don't match: SAD mode needs equal image sizes");
    if ((p.win & 1u) == 0 || p.win > 15 || p.d_max == 0 || p.d_max > 256)
        return fail(e, STEPTH_ERR_BAD_ARG, "SAD mode: win must be odd <= 15, d_max in 1..256");
    bool any_zero = false;
    for (uint32_t f0 = 0; f0 < n_frames; f0 += kMaxBatch) {
        const uint32_t nf = std::min<uint32_t>(kMaxBatch, n_frames - f0);
        HIP_TRY(e, hipMemsetAsync(e->words, 0, kWordsBytes, e->stream));
        SadArgs sa{};
        for (uint32_t f = 0; f < nf; ++f) {
            sa.left[f] = static_cast<const uint8_t *>(mains[f0 + f].data);
            sa.right[f] = static_cast<const uint8_t *>(adds[f0 + f].data);
        }
        sa.lstride = mains[0].stride;
        sa.lch = mains[0].channels;
        sa.rstride = adds[0].stride;
        sa.n_frames = nf;
        sa.w = W;
        sa.h = H;
        sa.row0 = row0;
        sa.row1 = row1;
        sa.d_max = p.d_max;
        sa.win = p.win;
        sa.thr = p.win * p.win * ((uint32_t)p.precision[0] + p.precision[1] + p.precision[2]);
        sa.raw_stride = ((size_t)W * rows + 3u) & ~(size_t)3u;
        TRY(reserve(e, e->raw, sa.raw_stride * nf));
        sa.raw = static_cast<uint8_t *>(e->raw.p);
        sa.max_word = e->words;
        {
            KernelTimer t(e);
            const int s = launch_sad(sa, e->stream);
            if (s != STEPTH_OK)
                return fail(e, s, "sad_match launch failed: %s", hipGetErrorString(hipGetLastError()));
        }
        PaintArgs pa{};
        pa.raw = sa.raw;
        pa.raw_stride = (uint32_t)sa.raw_stride;
        pa.max_word = e->words;
        pa.max_out = (async && max_out) ? max_out + f0 : nullptr;
        pa.identity = raw_only;
        for (uint32_t f = 0; f < nf; ++f)
            pa.out[f] = static_cast<uint8_t *>(const_cast<void *>(depths[f0 + f].data));
        pa.ostride = depths[0].stride;
        pa.w = W;
        pa.h = H;
        pa.row_begin = row0;
        pa.row_end = row1;
        pa.rp = RowPattern{0u, 1u, 0u};
        pa.tile_w = pa.tile_h = 1;
        pa.gx = W;
        pa.n = W * rows;
        bool vec4 = (W & 3u) == 0 && (pa.ostride & 3u) == 0; // four pixels per thread where they align
        for (uint32_t f = 0; f < nf; ++f)
            vec4 = vec4 && (reinterpret_cast<uintptr_t>(pa.out[f]) & 3u) == 0;
        const size_t total = (size_t)rows * (vec4 ? W / 4 : W);
        const dim3 pgrid((uint32_t)std::min<size_t>((total + 255) / 256, 1024), nf);
        if (vec4)
            hipLaunchKernelGGL(paint_1x1_vec4_kernel, pgrid, dim3(256), 0, e->stream, pa);
        else
            hipLaunchKernelGGL(paint_tiles_kernel, pgrid, dim3(256), 0, e->stream, pa);
        HIP_TRY(e, hipGetLastError());
        if (!async) {
            uint32_t mx[kMaxBatch];
            HIP_TRY(e, hipMemcpyAsync(mx, e->words, 4 * nf, hipMemcpyDeviceToHost, e->stream));
            HIP_TRY(e, hipStreamSynchronize(e->stream));
            for (uint32_t f = 0; f < nf; ++f) {
                if (max_out)
                    max_out[f0 + f] = mx[f];
                any_zero |= mx[f] == 0;
            }
        }
    }
    if (!async && any_zero && !raw_only)
        return fail(e, STEPTH_ERR_MAX_ZERO, "a frame's distances are all 0; its output is all zero");
    return STEPTH_OK;
}

// PASS 1 + PASS 2 + paint for the prepared-query modes -- uniform tiles larger than 1x1
// (d_groups == nullptr) or one group list per frame -- for nf <= kMaxBatch device frames in
// one set of launches (src/depth_image.rs:104-132 for nf images at once).
// `late_counts`: the group counts are still on their way from the device (adaptive flow); they
// are fetched -- the one host synchronisation -- after the pack kernels, which do not need them,
// have been enqueued, so the device works on while the host waits.
static int ring_queries_batch(stepth_engine *e, uint32_t nf, const stepth_image *mains,
                              const stepth_image *adds, const stepth_params &p, uint32_t row0,
                              uint32_t row1, RowPattern rp, const void *const *d_groups,
                              const uint32_t *n_groups, const stepth_image *depths,
                              uint32_t *max_out, bool &any_zero, int (*late_counts)(stepth_engine *, uint32_t, uint32_t *),
                              uint32_t *late_buf)
{
    const uint32_t W = mains[0].width, H = mains[0].height, rows = row1 - row0;
    const bool async = p.flags & STEPTH_FLAG_ASYNC, raw_only = p.flags & STEPTH_FLAG_RAW_ONLY;
    const uint8_t *d_adds[kMaxBatch];
    for (uint32_t f = 0; f < nf; ++f)
        d_adds[f] = static_cast<const uint8_t *>(adds[f].data);
    uint32_t bw = 0, bh = 0, lo = 0, hi = 0xFFFFFFFFu, col_hi = 0xFFFFFFFFu;
    if (!d_groups && p.max_steps < 0x40000000u) { // rows and columns the tiles' searches can reach
        const uint32_t reach = p.max_steps ? p.max_steps - 1u : 0u;
        const uint32_t c_lo = row0 / 2u, c_hi = (row1 + p.tile_h) / 2u;
        lo = c_lo > reach ? c_lo - reach : 0u;
        hi = c_hi + reach + 1u;
        col_hi = (W + p.tile_w) / 2u + reach + 1u;
    } else if (late_counts && p.max_steps < 0x40000000u) {
        // the engine's own groups (adaptive flow) lie inside the main image: x + w <= W, y + h <= H.
        // A caller's list may hold anything: everything is packed for it.
        const uint32_t reach = p.max_steps ? p.max_steps - 1u : 0u;
        lo = row0 / 2u > reach ? row0 / 2u - reach : 0u; // (a group's top row lies in the strip, its bottom anywhere)
        hi = H / 2u + reach + 1u;
        col_hi = W / 2u + reach + 1u;
    }
    TRY(pack_additional_tiles(e, nf, d_adds, adds[0].stride, adds[0].width, adds[0].height, bw, bh, lo, hi, col_hi));
    if (late_counts) {
        TRY(late_counts(e, nf, late_buf));
        n_groups = late_buf;
    }

    RingArgs ra{};
    ra.T = (const uint32_t *)e->T.p;
    ra.sbw = (bw + 7u) / 8u;
    ra.sbh = (bh + 7u) / 8u;
    ra.wa = adds[0].width;
    ra.ha = adds[0].height;
    ra.bw = bw;
    ra.bh = bh;
    ra.max_steps = p.max_steps;
    memcpy(ra.prec, p.precision, 3);
    ra.max_word = e->words;

    PrepArgs pr{};
    pr.row_begin = row0;
    pr.row_end = row1;
    pr.rp = rp;
    memcpy(pr.prec, p.precision, 3);
    uint32_t gx = 0, n_max = 0;
    if (d_groups) {
        for (uint32_t f = 0; f < nf; ++f) {
            pr.groups[f] = d_groups[f];
            pr.n[f] = n_groups[f];
            n_max = std::max(n_max, n_groups[f]);
        }
    } else {
        gx = (W + p.tile_w - 1) / p.tile_w;
        n_max = gx * ((rows + p.tile_h - 1) / p.tile_h);
        for (uint32_t f = 0; f < nf; ++f) {
            pr.main_px[f] = static_cast<const uint8_t *>(mains[f].data);
            pr.n[f] = n_max;
        }
        pr.mstride = mains[0].stride;
        pr.mch = mains[0].channels;
        pr.w = W;
        pr.h = H;
        pr.tile_w = p.tile_w;
        pr.tile_h = p.tile_h;
        pr.gx = gx;
        pr.gy = (rows + p.tile_h - 1) / p.tile_h;
    }
    if (n_max == 0)
        return STEPTH_OK;
    pr.q_stride = n_max;
    TRY(reserve(e, e->queries, (size_t)n_max * nf * sizeof(Query)));
    pr.out = (Query *)e->queries.p;
    const dim3 pgrid((n_max + 255) / 256, nf);
    const bool sort_queries = d_groups != nullptr; // tiles come in row-major order already
    BinArgs ba{};
    if (sort_queries) {
        ba.bins_x = (adds[0].width + 31u) / 32u;
        ba.bins_y = (adds[0].height + 31u) / 32u;
        ba.n_bins = ba.bins_x * ba.bins_y + 1u;
        TRY(reserve(e, e->query_bins, (size_t)ba.n_bins * nf * 4));
        TRY(reserve(e, e->queries_sorted, (size_t)n_max * nf * sizeof(Query)));
        TRY(reserve(e, e->query_ids, (size_t)n_max * nf * 4));
        HIP_TRY(e, hipMemsetAsync(e->query_bins.p, 0, (size_t)ba.n_bins * nf * 4, e->stream));
        pr.bin_counts = (uint32_t *)e->query_bins.p;
        pr.bins_x = ba.bins_x;
        pr.bins_y = ba.bins_y;
        pr.n_bins = ba.n_bins;
    }
    if (d_groups)
        hipLaunchKernelGGL(prep_list_kernel, pgrid, dim3(256), 0, e->stream, pr);
    else
        hipLaunchKernelGGL(prep_tiles_kernel, pgrid, dim3(256), 0, e->stream, pr);
    ra.queries = pr.out;
    if (sort_queries) {
        ba.in = pr.out;
        ba.out = (Query *)e->queries_sorted.p;
        ba.ids = (uint32_t *)e->query_ids.p;
        ba.bins = pr.bin_counts;
        ba.q_stride = n_max;
        for (uint32_t f = 0; f < nf; ++f)
            ba.n[f] = pr.n[f];
        hipLaunchKernelGGL(bin_scan_kernel, dim3(nf), dim3(256), 0, e->stream, ba);
        hipLaunchKernelGGL(bin_scatter_kernel, pgrid, dim3(256), 0, e->stream, ba);
        ra.queries = ba.out;
        ra.query_ids = ba.ids;
    }
    HIP_TRY(e, hipGetLastError());
    ra.q_stride = n_max;
    for (uint32_t f = 0; f < nf; ++f)
        ra.nq[f] = pr.n[f];
    ra.n = n_max;
    const size_t raw_stride = ((size_t)n_max + 3u) & ~(size_t)3u;
    TRY(reserve(e, e->raw, raw_stride * nf));
    ra.raw = static_cast<uint8_t *>(e->raw.p);
    if (e->stats_enabled) {
        HIP_TRY(e, hipMemsetAsync(e->stats, 0, kStatsBytes, e->stream));
        e->stats_groups = 0;
        for (uint32_t f = 0; f < nf; ++f)
            e->stats_groups += pr.n[f];
    }
    PaintArgs pa{};
    pa.raw = ra.raw;
    pa.raw_stride = (uint32_t)raw_stride;
    pa.max_word = e->words;
    pa.max_out = (async && max_out) ? max_out : nullptr;
    pa.identity = raw_only;
    for (uint32_t f = 0; f < nf; ++f)
        pa.out[f] = static_cast<uint8_t *>(const_cast<void *>(depths[f].data));
    pa.ostride = depths[0].stride;
    pa.w = W;
    pa.h = H;
    pa.row_begin = row0;
    pa.row_end = row1;
    pa.rp = rp;
    pa.n = n_max;
    if (d_groups) {
        // groups not in the strip paint nothing; unpainted pixels must read 0 like
        // ImageBuffer::from_pixel(.., Luma([0])) (src/depth_image.rs:20)
        for (uint32_t f = 0; f < nf; ++f) {
            pa.groups_list[f] = d_groups[f];
            pa.n_list[f] = n_groups[f];
        }
        hipLaunchKernelGGL(zero_rows_kernel, dim3((uint32_t)std::min<size_t>(((size_t)rows * W + 255) / 256, 512), nf),
                           dim3(256), 0, e->stream, pa);
    }
    // group lists on the per-lane search: it paints the raw values itself, PASS 2 follows in place
    const bool fused_paint = d_groups != nullptr;
    TRY(launch_list(e, ra, nf, fused_paint ? &pa : nullptr));

    if (fused_paint) {
        hipLaunchKernelGGL(paint_big_kernel, dim3(64, nf), dim3(256), 0, e->stream, pa,
                           (const uint32_t *)(e->words + kWordCounters), (const uint32_t *)e->pending.p, n_max);
        // (with RAW_ONLY the LUT is the identity: the launch only publishes the max)
        const size_t total16 = raw_only ? 1 : (size_t)rows * ((W + 15) / 16);
        hipLaunchKernelGGL(lut_rows_kernel, dim3((uint32_t)std::min<size_t>((total16 + 255) / 256, 1024), nf), dim3(256), 0,
                           e->stream, pa);
    } else if (d_groups) {
        hipLaunchKernelGGL(paint_list_kernel, dim3((n_max + 15) / 16, nf), dim3(256), 0, e->stream, pa);
    } else {
        pa.tile_w = p.tile_w;
        pa.tile_h = p.tile_h;
        pa.gx = gx;
        const size_t total = (size_t)rows * W;
        hipLaunchKernelGGL(paint_tiles_kernel, dim3((uint32_t)std::min<size_t>((total + 255) / 256, 1024), nf),
                           dim3(256), 0, e->stream, pa);
    }
    HIP_TRY(e, hipGetLastError());
    if (!async) {
        uint32_t mx[kMaxBatch];
        HIP_TRY(e, hipMemcpyAsync(mx, e->words, 4 * nf, hipMemcpyDeviceToHost, e->stream));
        HIP_TRY(e, hipStreamSynchronize(e->stream));
        for (uint32_t f = 0; f < nf; ++f) {
            if (max_out)
                max_out[f] = mx[f];
            any_zero |= mx[f] == 0;
        }
    }
    return STEPTH_OK;
}

int stepth_depth_from_additional_batch(stepth_engine *e, uint32_t n_frames,
                                       const stepth_image *mains, const stepth_image *adds,
                                       const stepth_params *pp, const stepth_image *depths,
                                       uint32_t *max_out)
{
    STEPTH_GUARD_BEGIN
    if (!e)
        return STEPTH_ERR_BAD_ARG;
    if (!pp || pp->struct_size != sizeof(stepth_params))
        return fail(e, STEPTH_ERR_BAD_ARG, "params null or struct_size mismatch");
    if (n_frames == 0 || !mains || !adds || !depths)
        return fail(e, STEPTH_ERR_BAD_ARG, "empty batch or null image arrays");
    const stepth_params p = *pp;
    const bool sad = p.mode == STEPTH_MODE_SAD_WINDOW;
    if (!sad && (p.mode != STEPTH_MODE_RING_FIRST_HIT || p.tile_w == 0 || p.tile_h == 0))
        return fail(e, STEPTH_ERR_BAD_ARG, "batch: ring mode with uniform tiles or SAD mode");
    // (beyond the dense search's reach one group per pixel is a 1x1 tile of the prepared-query search)
    const bool tiles = !sad && (p.tile_w != 1 || p.tile_h != 1 || !tile_search_fits(p.max_steps, adds[0].width, adds[0].height));
    const bool gauss_tail = p.flags & STEPTH_FLAG_GAUSSIAN_TAIL;
    if (gauss_tail && (sad || p.tile_w != 1 || p.tile_h != 1 || (p.flags & STEPTH_FLAG_RAW_ONLY) || p.row_block ||
                       !(p.row_begin == 0 && (p.row_end == 0 || p.row_end == mains[0].height))))
        return fail(e, STEPTH_ERR_BAD_ARG,
                    "STEPTH_FLAG_GAUSSIAN_TAIL needs whole normalised ring-mode frames");
    uint32_t n_host = 0;
    for (uint32_t f = 0; f < n_frames; ++f) {
        TRY(check_image(e, &mains[f], "main", 3, 4));
        TRY(check_image(e, &adds[f], "additional", 3, 3));
        TRY(check_image(e, &depths[f], "depth", 1, 1));
        n_host += (mains[f].memory == STEPTH_MEM_HOST) + (adds[f].memory == STEPTH_MEM_HOST) +
                  (depths[f].memory == STEPTH_MEM_HOST);
        if (mains[f].width != mains[0].width || mains[f].height != mains[0].height ||
            mains[f].channels != mains[0].channels || adds[f].width != adds[0].width ||
            adds[f].height != adds[0].height)
            return fail(e, STEPTH_ERR_BAD_ARG, "batch: frame %u differs in size", f);
        if (depths[f].width != mains[0].width || depths[f].height != mains[0].height)
            return fail(e, STEPTH_ERR_SIZE_MISMATCH, "Sizes don't match");
    }
    if (n_host == 3 * n_frames) { // every image in host memory: staged and pipelined
        if (!sad && (p.tile_w != 1 || p.tile_h != 1))
            return fail(e, STEPTH_ERR_BAD_ARG, "batch: host-memory frames take 1x1 tiles or the SAD mode");
        return batch_host(e, n_frames, mains, adds, p, depths, max_out);
    }
    for (uint32_t f = 0; f < n_frames; ++f) {
        if (mains[f].memory != STEPTH_MEM_DEVICE || adds[f].memory != STEPTH_MEM_DEVICE ||
            depths[f].memory != STEPTH_MEM_DEVICE)
            return fail(e, STEPTH_ERR_BAD_ARG,
                        "batch: images must be all device memory or all host memory");
        if (mains[f].width != mains[0].width || mains[f].height != mains[0].height ||
            mains[f].channels != mains[0].channels || mains[f].stride != mains[0].stride ||
            adds[f].width != adds[0].width || adds[f].height != adds[0].height ||
            adds[f].stride != adds[0].stride || depths[f].stride != depths[0].stride)
            return fail(e, STEPTH_ERR_BAD_ARG, "batch: frame %u differs in size or stride", f);
        if (depths[f].width != mains[0].width || depths[f].height != mains[0].height)
            return fail(e, STEPTH_ERR_SIZE_MISMATCH, "Sizes don't match");
    }
    const uint32_t W = mains[0].width, H = mains[0].height;
    const bool async = p.flags & STEPTH_FLAG_ASYNC, raw_only = p.flags & STEPTH_FLAG_RAW_ONLY;
    uint32_t row0 = p.row_begin, row1 = p.row_end;
    if (row0 == 0 && row1 == 0)
        row1 = H;
    if (row0 >= row1 || row1 > H)
        return fail(e, STEPTH_ERR_BAD_ARG, "row strip [%u,%u) outside 0..%u", row0, row1, H);
    RowPattern rp{0u, 1u, 0u};
    if (p.row_block) {
        if ((p.row_block & 1u) || p.row_mod == 0 || p.row_rem >= p.row_mod)
            return fail(e, STEPTH_ERR_BAD_ARG, "row pattern: even row_block, row_rem < row_mod");
        rp = RowPattern{p.row_block, p.row_mod, p.row_rem};
    }
    HIP_TRY(e, hipSetDevice(e->device));
    const uint32_t rows = row1 - row0;
    bool any_zero = false;
    if (sad)
        return batch_sad(e, n_frames, mains, adds, p, depths, max_out, row0, row1);
    if (tiles) {
        // uniform tiles larger than one pixel (value = floor mean colour): prepared queries
        stepth_params tp = p;
        tp.tile_w = std::min(tp.tile_w, W);
        tp.tile_h = std::min(tp.tile_h, H);
        if (row0 % tp.tile_h != 0 || (row1 % tp.tile_h != 0 && row1 != H))
            return fail(e, STEPTH_ERR_BAD_ARG, "row strip [%u,%u) not aligned to tile_h %u", row0, row1, tp.tile_h);
        if (p.row_block && p.row_block % tp.tile_h != 0)
            return fail(e, STEPTH_ERR_BAD_ARG, "row pattern: row_block must be a multiple of tile_h");
        const uint64_t cells_t = (uint64_t)((adds[0].width + 7u) / 8u) * ((adds[0].height + 7u) / 8u) * 64u;
        const uint32_t chunk_t = (uint32_t)std::max<uint64_t>(
            1, std::min<uint64_t>(kMaxBatch, 0x3FFFFFFFull / std::max<uint64_t>(cells_t, 1)));
        for (uint32_t f0 = 0; f0 < n_frames; f0 += chunk_t) {
            const uint32_t nf = std::min<uint32_t>(chunk_t, n_frames - f0);
            TRY(ring_queries_batch(e, nf, mains + f0, adds + f0, tp, row0, row1, rp, nullptr, nullptr,
                                   depths + f0, max_out ? max_out + f0 : nullptr, any_zero));
            if (gauss_tail) { // (one group per pixel beyond the dense search's reach; whole device frames)
                uint8_t *outs[kMaxBatch];
                for (uint32_t f = 0; f < nf; ++f)
                    outs[f] = static_cast<uint8_t *>(const_cast<void *>(depths[f0 + f].data));
                TRY(gaussian_device_batch(e, nf, outs, depths[f0].stride, W, H, outs, depths[f0].stride));
            }
        }
        if (!async && any_zero && !raw_only)
            return fail(e, STEPTH_ERR_MAX_ZERO, "a frame's distances are all 0; its output is all zero");
        return STEPTH_OK;
    }

    // frames per set of launches: at most kMaxBatch, and few enough that all their cells can be
    // indexed with 32 bits
    const uint64_t cells = (uint64_t)((adds[0].width + 7u) / 8u) * ((adds[0].height + 7u) / 8u) * 64u;
    uint32_t chunk = (uint32_t)std::max<uint64_t>(
        1, std::min<uint64_t>(kMaxBatch, 0x3FFFFFFFull / std::max<uint64_t>(cells, 1)));
    for (uint32_t f0 = 0; f0 < n_frames; f0 += chunk) {
        const uint32_t nf = std::min<uint32_t>(chunk, n_frames - f0);
        const uint8_t *d_adds[kMaxBatch];
        for (uint32_t f = 0; f < nf; ++f)
            d_adds[f] = static_cast<const uint8_t *>(adds[f0 + f].data);
        uint32_t bw = 0, bh = 0, lo = 0, hi = 0xFFFFFFFFu, col_hi = 0xFFFFFFFFu;
        if (p.max_steps < 0x40000000u) { // rows of the additional image the searches can reach
            const uint32_t reach = p.max_steps ? p.max_steps - 1u : 0u;
            lo = row0 / 2u > reach ? row0 / 2u - reach : 0u;
            hi = (row1 + 1u) / 2u + reach + 1u;
            col_hi = (W + 1u) / 2u + reach + 1u;
        }
        TRY(pack_additional_tiles(e, nf, d_adds, adds[0].stride, adds[0].width, adds[0].height,
                                  bw, bh, lo, hi, col_hi));

        RingArgs ra{};
        ra.T = (const uint32_t *)e->T.p;
        ra.sbw = (bw + 7u) / 8u;
        ra.sbh = (bh + 7u) / 8u;
        ra.wa = adds[0].width;
        ra.ha = adds[0].height;
        ra.bw = bw;
        ra.bh = bh;
        ra.max_steps = p.max_steps;
        memcpy(ra.prec, p.precision, 3);
        ra.max_word = e->words;
        for (uint32_t f = 0; f < nf; ++f)
            ra.main_px[f] = static_cast<const uint8_t *>(mains[f0 + f].data);
        ra.mstride = mains[0].stride;
        ra.mch = mains[0].channels;
        ra.w = W;
        ra.row0 = row0;
        ra.row1 = row1;
        ra.rp = rp;
        ra.n = W * rows;
        const size_t raw_stride = ((size_t)ra.n + 3u) & ~(size_t)3u;
        TRY(reserve(e, e->raw, raw_stride * nf));
        ra.raw = static_cast<uint8_t *>(e->raw.p);
        if (e->stats_enabled) {
            HIP_TRY(e, hipMemsetAsync(e->stats, 0, kStatsBytes, e->stream));
            e->stats_groups = (uint64_t)ra.n * nf;
        }
        ra.heavy_first = (p.flags & STEPTH_FLAG_HEAVY_FIRST) != 0;
        TRY(launch_tile(e, ra, nf));

        PaintArgs pa{};
        pa.raw = ra.raw;
        pa.raw_stride = (uint32_t)raw_stride;
        pa.max_word = e->words;
        pa.max_out = (async && max_out) ? max_out + f0 : nullptr;
        pa.identity = raw_only;
        for (uint32_t f = 0; f < nf; ++f)
            pa.out[f] = static_cast<uint8_t *>(const_cast<void *>(depths[f0 + f].data));
        pa.ostride = depths[0].stride;
        pa.w = W;
        pa.h = H;
        pa.row_begin = row0;
        pa.row_end = row1;
        pa.rp = rp;
        pa.tile_w = pa.tile_h = 1;
        pa.gx = W;
        pa.n = ra.n;
        bool vec4 = (W & 3u) == 0 && (pa.ostride & 3u) == 0;
        for (uint32_t f = 0; f < nf; ++f)
            vec4 = vec4 && (reinterpret_cast<uintptr_t>(pa.out[f]) & 3u) == 0;
        const size_t total = (size_t)rows * (vec4 ? W / 4 : W);
        const dim3 grid((uint32_t)std::min<size_t>((total + 255) / 256, 1024), nf);
        if (vec4)
            hipLaunchKernelGGL(paint_1x1_vec4_kernel, grid, dim3(256), 0, e->stream, pa);
        else
            hipLaunchKernelGGL(paint_tiles_kernel, grid, dim3(256), 0, e->stream, pa);
        HIP_TRY(e, hipGetLastError());
        if (gauss_tail) // src/depth_image.rs:133-134, in place on the painted maps
            TRY(gaussian_device_batch(e, nf, pa.out, pa.ostride, W, H, pa.out, pa.ostride));

        if (!async) {
            uint32_t mx[kMaxBatch];
            HIP_TRY(e, hipMemcpyAsync(mx, e->words, 4 * nf, hipMemcpyDeviceToHost, e->stream));
            HIP_TRY(e, hipStreamSynchronize(e->stream));
            for (uint32_t f = 0; f < nf; ++f) {
                if (max_out)
                    max_out[f0 + f] = mx[f];
                any_zero |= mx[f] == 0;
            }
        }
    }
    if (!async && any_zero && !raw_only)
        return fail(e, STEPTH_ERR_MAX_ZERO, "a frame's distances are all 0; its output is all zero");
    return STEPTH_OK;
    STEPTH_GUARD_END(e)
}

// Host-memory batch (what a caller holding decoded images in RAM has, src/depth_image.rs:8-9,
// :96): frames go through two slots of device staging in chunks, uploads on their own stream,
// downloads on another, so the upload of chunk k+1 and the download of chunk k-1 overlap the
// kernels of chunk k.  Page-locked caller memory (stepth_host_alloc) makes
// the copies truly asynchronous; pageable memory still overlaps because the kernels of the
// previous chunk are already enqueued when the host blocks in the next copy.
static int batch_host(stepth_engine *e, uint32_t n_frames, const stepth_image *mains,
                      const stepth_image *adds, const stepth_params &p, const stepth_image *depths,
                      uint32_t *max_out)
{
    if (p.flags & STEPTH_FLAG_ASYNC)
        return fail(e, STEPTH_ERR_BAD_ARG, "STEPTH_FLAG_ASYNC needs device memory everywhere");
    HIP_TRY(e, hipSetDevice(e->device));
    if (!e->copy_in) {
        HIP_TRY(e, hipStreamCreateWithFlags(&e->copy_in, hipStreamNonBlocking));
        HIP_TRY(e, hipStreamCreateWithFlags(&e->copy_out, hipStreamNonBlocking));
        for (int s = 0; s < 2; ++s) {
            HIP_TRY(e, hipEventCreateWithFlags(&e->ev_in[s], hipEventDisableTiming));
            HIP_TRY(e, hipEventCreateWithFlags(&e->ev_compute[s], hipEventDisableTiming));
            HIP_TRY(e, hipEventCreateWithFlags(&e->ev_out[s], hipEventDisableTiming));
        }
    }
    const uint32_t W = mains[0].width, H = mains[0].height, mch = mains[0].channels;
    const uint32_t Wa = adds[0].width, Ha = adds[0].height;
    const size_t mpitch = ((size_t)W * mch + 15) & ~(size_t)15, apitch = ((size_t)Wa * 3 + 15) & ~(size_t)15,
                 opitch = ((size_t)W + 15) & ~(size_t)15;
    const size_t mbytes = mpitch * H, abytes = apitch * Ha, obytes = opitch * H;
    // chunk: about 64 MB of input, at most 8 frames (the kernels are near their batched
    // efficiency from 4 frames on and a small chunk starts the pipeline sooner)
    const uint32_t chunk = (uint32_t)std::max<size_t>(
        1, std::min<size_t>({(size_t)8, (size_t)n_frames, ((size_t)64 << 20) / (mbytes + abytes) + 1}));
    for (int s = 0; s < 2; ++s) {
        TRY(reserve(e, e->pipe_main[s], mbytes * chunk));
        TRY(reserve(e, e->pipe_add[s], abytes * chunk));
        TRY(reserve(e, e->pipe_out[s], obytes * chunk));
    }
    TRY(reserve(e, e->pipe_max, (size_t)n_frames * 4));
    uint32_t *d_max = static_cast<uint32_t *>(e->pipe_max.p);
    stepth_params dp = p;
    dp.flags |= STEPTH_FLAG_ASYNC;
    // the first upload and the last download cannot hide behind kernels: start with two
    // frames and double up to `chunk`, end symmetric (short last chunks come by themselves)
    std::vector<std::pair<uint32_t, uint32_t>> chunks; // (first frame, frames)
    for (uint32_t f0 = 0, c = std::min(2u, chunk); f0 < n_frames;) {
        const uint32_t nf = std::min(c, n_frames - f0);
        chunks.emplace_back(f0, nf);
        f0 += nf;
        c = std::min(chunk, c * 2u);
    }
    const uint32_t n_chunks = (uint32_t)chunks.size();
    std::vector<stepth_image> dm(chunk), da(chunk), dd(chunk);

    // rows the call computes: only those go back to the caller (the staging buffers hold
    // nothing meaningful elsewhere); with an interleaved row pattern the other ranks' rows of
    // the strip travel up first, so that they come back unchanged
    uint32_t row0 = p.row_begin, row1 = p.row_end;
    if (row0 == 0 && row1 == 0)
        row1 = H;
    if (row0 >= row1 || row1 > H)
        return fail(e, STEPTH_ERR_BAD_ARG, "row strip [%u,%u) outside 0..%u", row0, row1, H);
    const bool pattern = p.row_block != 0;
    // Ring mode reads the additional image only where a search can reach: centres lie at half the
    // main image's coordinates (middle = ((x + w) / 2, (y + h) / 2), src/depth_image.rs:114-117),
    // so rows beyond (row1 + tile_h) / 2 + reach and columns beyond (W + tile_w) / 2 + reach -- two
    // thirds of a 1080p frame at reach 127 -- never travel.  (Block-aligned, the pack kernels' own
    // rectangle: what lies outside it in the staging buffer is an earlier frame's and, like the
    // unpacked blocks, beyond every query's limit.)
    uint32_t ar0 = 0, ar1 = Ha, ac1 = Wa;
    if (p.mode == STEPTH_MODE_RING_FIRST_HIT && p.max_steps < 0x40000000u && p.tile_w && p.tile_h) {
        const uint32_t reach = p.max_steps ? p.max_steps - 1u : 0u, c_lo = row0 / 2u;
        ar0 = std::min((c_lo > reach ? c_lo - reach : 0u) & ~7u, Ha);
        ar1 = (uint32_t)std::min<uint64_t>(Ha, ((uint64_t)(row1 + std::min(p.tile_h, H)) / 2u + reach + 1u + 7u) & ~(uint64_t)7);
        ac1 = (uint32_t)std::min<uint64_t>(Wa, ((uint64_t)(W + std::min(p.tile_w, W)) / 2u + reach + 1u + 7u) & ~(uint64_t)7);
        ar1 = std::max(ar1, ar0);
    }

    auto download = [&](uint32_t k) -> int { // chunk k's depth rows -> caller memory
        const uint32_t s = k & 1u, f0 = chunks[k].first, nf = chunks[k].second;
        HIP_TRY(e, hipStreamWaitEvent(e->copy_out, e->ev_compute[s], 0));
        for (uint32_t f = 0; f < nf; ++f)
            HIP_TRY(e, hipMemcpy2DAsync(static_cast<uint8_t *>(const_cast<void *>(depths[f0 + f].data)) +
                                            (size_t)row0 * depths[f0 + f].stride,
                                        depths[f0 + f].stride,
                                        static_cast<uint8_t *>(e->pipe_out[s].p) + obytes * f +
                                            (size_t)row0 * opitch,
                                        opitch, W, row1 - row0, hipMemcpyDeviceToHost, e->copy_out));
        HIP_TRY(e, hipEventRecord(e->ev_out[s], e->copy_out));
        return STEPTH_OK;
    };

    for (uint32_t k = 0; k < n_chunks; ++k) {
        const uint32_t s = k & 1u, f0 = chunks[k].first, nf = chunks[k].second;
        if (k >= 2) { // slot s: inputs consumed by chunk k-2's kernels, outputs by its download
            HIP_TRY(e, hipStreamWaitEvent(e->copy_in, e->ev_compute[s], 0));
            HIP_TRY(e, hipStreamWaitEvent(e->stream, e->ev_out[s], 0));
            if (pattern)
                HIP_TRY(e, hipStreamWaitEvent(e->copy_in, e->ev_out[s], 0));
        }
        for (uint32_t f = 0; f < nf; ++f) {
            uint8_t *m = static_cast<uint8_t *>(e->pipe_main[s].p) + mbytes * f;
            uint8_t *a = static_cast<uint8_t *>(e->pipe_add[s].p) + abytes * f;
            // (ring mode reads the main image's rows of the strip only; the SAD window reaches beyond them)
            const uint32_t mr0 = p.mode == STEPTH_MODE_RING_FIRST_HIT ? row0 : 0u, mr1 = p.mode == STEPTH_MODE_RING_FIRST_HIT ? row1 : H;
            HIP_TRY(e, hipMemcpy2DAsync(m + (size_t)mr0 * mpitch, mpitch,
                                        static_cast<const uint8_t *>(mains[f0 + f].data) + (size_t)mr0 * mains[f0 + f].stride,
                                        mains[f0 + f].stride, (size_t)W * mch, mr1 - mr0, hipMemcpyHostToDevice, e->copy_in));
            if (ar1 > ar0 && ac1)
                HIP_TRY(e, hipMemcpy2DAsync(a + (size_t)ar0 * apitch, apitch,
                                            static_cast<const uint8_t *>(adds[f0 + f].data) + (size_t)ar0 * adds[f0 + f].stride,
                                            adds[f0 + f].stride, (size_t)ac1 * 3, ar1 - ar0, hipMemcpyHostToDevice,
                                            e->copy_in));
            dm[f] = stepth_image{m, W, H, mch, STEPTH_MEM_DEVICE, mpitch};
            da[f] = stepth_image{a, Wa, Ha, 3, STEPTH_MEM_DEVICE, apitch};
            dd[f] = stepth_image{static_cast<uint8_t *>(e->pipe_out[s].p) + obytes * f, W, H, 1,
                                 STEPTH_MEM_DEVICE, opitch};
            if (pattern)
                HIP_TRY(e, hipMemcpy2DAsync(static_cast<uint8_t *>(e->pipe_out[s].p) + obytes * f +
                                                (size_t)row0 * opitch,
                                            opitch,
                                            static_cast<const uint8_t *>(depths[f0 + f].data) +
                                                (size_t)row0 * depths[f0 + f].stride,
                                            depths[f0 + f].stride, W, row1 - row0, hipMemcpyHostToDevice,
                                            e->copy_in));
        }
        HIP_TRY(e, hipEventRecord(e->ev_in[s], e->copy_in));
        HIP_TRY(e, hipStreamWaitEvent(e->stream, e->ev_in[s], 0));
        TRY(stepth_depth_from_additional_batch(e, nf, dm.data(), da.data(), &dp, dd.data(), d_max + f0));
        HIP_TRY(e, hipEventRecord(e->ev_compute[s], e->stream));
        if (k >= 1)
            TRY(download(k - 1));
    }
    TRY(download(n_chunks - 1));
    std::vector<uint32_t> mx(n_frames);
    HIP_TRY(e, hipStreamSynchronize(e->stream));
    HIP_TRY(e, hipMemcpyAsync(mx.data(), d_max, (size_t)n_frames * 4, hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(e, hipStreamSynchronize(e->stream));
    HIP_TRY(e, hipStreamSynchronize(e->copy_out));
    bool any_zero = false;
    for (uint32_t f = 0; f < n_frames; ++f) {
        if (max_out)
            max_out[f] = mx[f];
        any_zero |= mx[f] == 0;
    }
    if (any_zero && !(p.flags & STEPTH_FLAG_RAW_ONLY))
        return fail(e, STEPTH_ERR_MAX_ZERO, "a frame's distances are all 0; its output is all zero");
    return STEPTH_OK;
}

int stepth_normalise_rows_pattern(stepth_engine *e, const stepth_image *rows, uint32_t first_row,
                                  uint32_t row_block, uint32_t row_mod, uint32_t row_rem,
                                  const uint32_t *max_ptr, uint32_t max_memory, uint32_t flags)
{
    STEPTH_GUARD_BEGIN
    if (!e)
        return STEPTH_ERR_BAD_ARG;
    TRY(check_image(e, rows, "rows", 1, 1));
    if (!max_ptr)
        return fail(e, STEPTH_ERR_BAD_ARG, "max_ptr is null");
    const bool async = flags & STEPTH_FLAG_ASYNC;
    if (async && (rows->memory != STEPTH_MEM_DEVICE || max_memory != STEPTH_MEM_DEVICE))
        return fail(e, STEPTH_ERR_BAD_ARG, "STEPTH_FLAG_ASYNC needs device memory everywhere");
    HIP_TRY(e, hipSetDevice(e->device));
    uint32_t mx_host = 0;
    const uint32_t *d_max = nullptr;
    if (max_memory == STEPTH_MEM_DEVICE) {
        d_max = max_ptr;
        if (!async)
            HIP_TRY(e, hipMemcpyAsync(&mx_host, max_ptr, 4, hipMemcpyDeviceToHost, e->stream));
    } else {
        mx_host = *max_ptr;
        HIP_TRY(e, hipMemcpyAsync(e->words + kWordHostMax, &mx_host, 4, hipMemcpyHostToDevice,
                                  e->stream));
        d_max = e->words + kWordHostMax;
    }
    if (row_block && (row_mod == 0 || row_rem >= row_mod))
        return fail(e, STEPTH_ERR_BAD_ARG, "row pattern: row_rem must be < row_mod");
    NormArgs na{};
    na.first_row = first_row;
    na.rp = RowPattern{row_block, row_block ? row_mod : 1u, row_block ? row_rem : 0u};
    na.w = rows->width;
    na.rows = rows->height;
    na.max_word = d_max;
    if (rows->memory == STEPTH_MEM_DEVICE) {
        na.data[0] = static_cast<uint8_t *>(const_cast<void *>(rows->data));
        na.stride = rows->stride;
    } else {
        const uint8_t *dp = nullptr;
        size_t ds = 0;
        TRY(to_device(e, *rows, e->stage_out, dp, ds));
        na.data[0] = const_cast<uint8_t *>(dp);
        na.stride = ds;
    }
    const size_t total = (size_t)na.rows * na.w;
    const uint32_t blocks = (uint32_t)std::min<size_t>((total + 255) / 256, 2048);
    hipLaunchKernelGGL(normalise_rows_kernel, dim3(blocks), dim3(256), 0, e->stream, na);
    HIP_TRY(e, hipGetLastError());
    if (async)
        return STEPTH_OK;
    if (rows->memory == STEPTH_MEM_HOST)
        HIP_TRY(e, hipMemcpy2DAsync(const_cast<void *>(rows->data), rows->stride, na.data[0],
                                    na.stride, na.w, na.rows, hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(e, hipStreamSynchronize(e->stream));
    if (mx_host == 0)
        return fail(e, STEPTH_ERR_MAX_ZERO, "max is 0; rows zeroed");
    return STEPTH_OK;
    STEPTH_GUARD_END(e)
}

int stepth_normalise_rows_batch(stepth_engine *e, uint32_t n_frames, const stepth_image *rows,
                                uint32_t first_row, uint32_t row_block, uint32_t row_mod,
                                uint32_t row_rem, const uint32_t *max_ptr, uint32_t flags)
{
    STEPTH_GUARD_BEGIN
    if (!e)
        return STEPTH_ERR_BAD_ARG;
    if (n_frames == 0 || !rows || !max_ptr)
        return fail(e, STEPTH_ERR_BAD_ARG, "empty batch or null pointer");
    if (row_block && (row_mod == 0 || row_rem >= row_mod))
        return fail(e, STEPTH_ERR_BAD_ARG, "row pattern: row_rem must be < row_mod");
    for (uint32_t f = 0; f < n_frames; ++f) {
        TRY(check_image(e, &rows[f], "rows", 1, 1));
        if (rows[f].memory != STEPTH_MEM_DEVICE || rows[f].width != rows[0].width ||
            rows[f].height != rows[0].height || rows[f].stride != rows[0].stride)
            return fail(e, STEPTH_ERR_BAD_ARG, "batch: device images of one shape only");
    }
    HIP_TRY(e, hipSetDevice(e->device));
    for (uint32_t f0 = 0; f0 < n_frames; f0 += kMaxBatch) {
        const uint32_t nf = std::min<uint32_t>(kMaxBatch, n_frames - f0);
        NormArgs na{};
        for (uint32_t f = 0; f < nf; ++f)
            na.data[f] = static_cast<uint8_t *>(const_cast<void *>(rows[f0 + f].data));
        na.stride = rows[0].stride;
        na.w = rows[0].width;
        na.rows = rows[0].height;
        na.max_word = max_ptr + f0;
        na.first_row = first_row;
        na.rp = RowPattern{row_block, row_block ? row_mod : 1u, row_block ? row_rem : 0u};
        const size_t total = (size_t)na.rows * na.w;
        const dim3 grid((uint32_t)std::min<size_t>((total + 255) / 256, 1024), nf);
        hipLaunchKernelGGL(normalise_rows_kernel, grid, dim3(256), 0, e->stream, na);
    }
    HIP_TRY(e, hipGetLastError());
    if (!(flags & STEPTH_FLAG_ASYNC))
        HIP_TRY(e, hipStreamSynchronize(e->stream));
    return STEPTH_OK;
    STEPTH_GUARD_END(e)
}

int stepth_normalise_rows(stepth_engine *e, const stepth_image *rows, const uint32_t *max_ptr,
                          uint32_t max_memory, uint32_t flags)
{
    STEPTH_GUARD_BEGIN
    return stepth_normalise_rows_pattern(e, rows, 0, 0, 1, 0, max_ptr, max_memory, flags);
    STEPTH_GUARD_END(e)
}

int stepth_gaussian_same_size(stepth_engine *e, const stepth_image *src, const stepth_image *dst,
                              uint32_t flags)
{
    STEPTH_GUARD_BEGIN
    if (!e)
        return STEPTH_ERR_BAD_ARG;
    TRY(check_image(e, src, "src", 1, 1));
    TRY(check_image(e, dst, "dst", 1, 1));
    if (src->width != dst->width || src->height != dst->height)
        return fail(e, STEPTH_ERR_SIZE_MISMATCH, "Sizes don't match");
    const bool async = flags & STEPTH_FLAG_ASYNC;
    if (async && (src->memory != STEPTH_MEM_DEVICE || dst->memory != STEPTH_MEM_DEVICE))
        return fail(e, STEPTH_ERR_BAD_ARG, "STEPTH_FLAG_ASYNC needs device memory everywhere");
    HIP_TRY(e, hipSetDevice(e->device));
    const uint32_t W = src->width, H = src->height;
    const uint8_t *d_src = nullptr;
    size_t ss = 0;
    TRY(to_device(e, *src, e->gauss_io, d_src, ss));
    uint8_t *d_dst;
    size_t ds;
    if (dst->memory == STEPTH_MEM_DEVICE) {
        d_dst = static_cast<uint8_t *>(const_cast<void *>(dst->data));
        ds = dst->stride;
    } else {
        ds = (W + 15u) & ~(size_t)15;
        TRY(reserve(e, e->stage_out, ds * H));
        d_dst = static_cast<uint8_t *>(e->stage_out.p);
    }
    TRY(gaussian_device(e, d_src, ss, W, H, d_dst, ds));
    if (async)
        return STEPTH_OK;
    if (dst->memory == STEPTH_MEM_HOST)
        HIP_TRY(e, hipMemcpy2DAsync(const_cast<void *>(dst->data), dst->stride, d_dst, ds, W, H,
                                    hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(e, hipStreamSynchronize(e->stream));
    return STEPTH_OK;
    STEPTH_GUARD_END(e)
}

int stepth_distance_dot_array_batch(stepth_engine *e, const stepth_image *array,
                                    const uint8_t *what, const uint32_t *from_xy, size_t n,
                                    uint32_t max, const uint8_t precision[3], uint8_t *found,
                                    uint32_t *dist, uint32_t *pos_xy)
{
    STEPTH_GUARD_BEGIN
    if (!e)
        return STEPTH_ERR_BAD_ARG;
    TRY(check_image(e, array, "array", 3, 3));
    if (!what || !from_xy || !precision || !found || !dist || !pos_xy)
        return fail(e, STEPTH_ERR_BAD_ARG, "null pointer argument");
    if (n == 0)
        return STEPTH_OK;
    if (n > 0x7FFFFFFFull)
        return fail(e, STEPTH_ERR_BAD_ARG, "too many queries");
    HIP_TRY(e, hipSetDevice(e->device));
    const uint8_t *d_add = nullptr;
    size_t as = 0;
    TRY(to_device(e, *array, e->stage_add, d_add, as));
    uint32_t bw = 0, bh = 0;
    TRY(pack_additional_tiles(e, 1, &d_add, as, array->width, array->height, bw, bh, 0, 0xFFFFFFFFu));

    std::vector<Query> qs(n);
    for (size_t i = 0; i < n; ++i) {
        qs[i].cx = from_xy[2 * i]; // helpers.rs:16: from.tuplexy()
        qs[i].cy = from_xy[2 * i + 1];
        make_box(what + 3 * i, precision, qs[i].lo, qs[i].hi);
    }
    TRY(reserve(e, e->queries, n * sizeof(Query)));
    TRY(reserve(e, e->raw, std::max<size_t>(n, 4)));
    TRY(reserve(e, e->aux32, n * 4));
    TRY(reserve(e, e->aux_pos, n * 8));
    TRY(reserve(e, e->aux_found, std::max<size_t>(n, 4)));
    HIP_TRY(e, hipMemcpyAsync(e->queries.p, qs.data(), n * sizeof(Query), hipMemcpyHostToDevice,
                              e->stream));
    RingArgs ra{};
    ra.T = (const uint32_t *)e->T.p;
    ra.sbw = (bw + 7u) / 8u;
    ra.sbh = (bh + 7u) / 8u;
    ra.wa = array->width;
    ra.ha = array->height;
    ra.bw = bw;
    ra.bh = bh;
    ra.queries = (const Query *)e->queries.p;
    ra.n = (uint32_t)n;
    ra.nq[0] = (uint32_t)n;
    ra.max_steps = max;
    memcpy(ra.prec, precision, 3);
    ra.raw = (uint8_t *)e->raw.p;
    ra.raw32 = (uint32_t *)e->aux32.p;
    ra.pos = (uint32_t *)e->aux_pos.p;
    ra.found = (uint8_t *)e->aux_found.p;
    ra.max_word = e->words;
    if (e->stats_enabled) {
        HIP_TRY(e, hipMemsetAsync(e->stats, 0, kStatsBytes, e->stream));
        e->stats_groups = n;
    }
    TRY(launch_list(e, ra, 1));
    HIP_TRY(e, hipMemcpyAsync(dist, ra.raw32, n * 4, hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(e, hipMemcpyAsync(pos_xy, ra.pos, n * 8, hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(e, hipMemcpyAsync(found, ra.found, n, hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(e, hipStreamSynchronize(e->stream));
    return STEPTH_OK;
    STEPTH_GUARD_END(e)
}

int stepth_stats_enable(stepth_engine *e, int enable)
{
    STEPTH_GUARD_BEGIN
    if (!e)
        return STEPTH_ERR_BAD_ARG;
    e->stats_enabled = enable != 0;
    e->stats_groups = 0;
    HIP_TRY(e, hipMemsetAsync(e->stats, 0, kStatsBytes, e->stream));
    return STEPTH_OK;
    STEPTH_GUARD_END(e)
}

// the sum of the counters' replicas (ring_kernels.h stats_replica); synchronises the stream
static int read_stats(stepth_engine *e, unsigned long long (&sum)[kStatsWords])
{
    std::vector<unsigned long long> h((size_t)kStatsWords * kStatsReplicas);
    HIP_TRY(e, hipMemcpyAsync(h.data(), e->stats, kStatsBytes, hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(e, hipStreamSynchronize(e->stream));
    for (uint32_t w = 0; w < kStatsWords; ++w) {
        sum[w] = 0;
        for (uint32_t r = 0; r < kStatsReplicas; ++r)
            sum[w] += h[(size_t)r * kStatsWords + w];
    }
    return STEPTH_OK;
}

int stepth_get_stats(stepth_engine *e, stepth_stats *out)
{
    STEPTH_GUARD_BEGIN
    if (!e || !out)
        return STEPTH_ERR_BAD_ARG;
    if (!e->stats_enabled)
        return fail(e, STEPTH_ERR_BAD_ARG, "stats are not enabled (stepth_stats_enable)");
    unsigned long long h[kStatsWords];
    TRY(read_stats(e, h));
    out->groups = e->stats_groups;
    out->ring0_hits = h[0];
    out->probes = h[1];
    out->near_steps = h[2];
    out->survey_steps = h[3];
    out->superblock_steps = h[4];
    out->blocks_probed = h[5];
    return STEPTH_OK;
    STEPTH_GUARD_END(e)
}

int stepth_get_stats_raw(stepth_engine *e, uint64_t *out, uint32_t n)
{
    STEPTH_GUARD_BEGIN
    if (!e || !out || n > kStatsWords)
        return STEPTH_ERR_BAD_ARG;
    unsigned long long h[kStatsWords];
    TRY(read_stats(e, h));
    for (uint32_t w = 0; w < n; ++w)
        out[w] = h[w];
    return STEPTH_OK;
    STEPTH_GUARD_END(e)
}

int stepth_kernel_time_enable(stepth_engine *e, int enable)
{
    STEPTH_GUARD_BEGIN
    if (!e)
        return STEPTH_ERR_BAD_ARG;
    HIP_TRY(e, hipStreamSynchronize(e->stream));
    e->timing = enable != 0;
    e->events_used = 0;
    return STEPTH_OK;
    STEPTH_GUARD_END(e)
}

int stepth_kernel_time_read(stepth_engine *e, double *total_ms, uint64_t *launches)
{
    STEPTH_GUARD_BEGIN
    if (!e || !total_ms || !launches)
        return STEPTH_ERR_BAD_ARG;
    HIP_TRY(e, hipStreamSynchronize(e->stream));
    double sum = 0;
    for (size_t i = 0; i < e->events_used; ++i) {
        float ms = 0;
        HIP_TRY(e, hipEventElapsedTime(&ms, e->events[i].first, e->events[i].second));
        sum += ms;
    }
    *total_ms = sum;
    *launches = e->events_used;
    e->events_used = 0;
    return STEPTH_OK;
    STEPTH_GUARD_END(e)
}

#include "zone_entries.h"
#include "group_entries.h"

int stepth_selftest_exception_guard(stepth_engine *e, int kind)
{
    STEPTH_GUARD_BEGIN
    if (kind == 0)
        throw std::bad_alloc();
    if (kind == 1)
        throw std::runtime_error("selftest");
    if (kind == 2)
        throw 42;
    return STEPTH_OK;
    STEPTH_GUARD_END(e)
}

} // extern "C"
