// zone_entries.h -- C-ABI entry points of the depth-map consumers (stepth_amd.h, "consumers
// of a finished depth map"); included by engine.hip inside its extern "C" block.
// Reference: /root/reference/src/depth_image.rs:162-247, /root/reference/src/mask_image.rs:205-213.

static_assert(sizeof(stepth_zone) == 4, "C-ABI struct layout (mirrored by stepth_amd/_lib.py)");
static_assert(kSplitErrStep == STEPTH_ERR_SPLIT_STEP && kSplitErrEmpty == STEPTH_ERR_SPLIT_EMPTY_ZONE &&
                  kSplitErrDiverged == STEPTH_ERR_SPLIT_DIVERGED,
              "split status codes");

// A [H][W] plane as the streaming kernels want it: contiguous planes become one long row so
// that every 16-byte segment stays aligned.
static void plane_shape(uint32_t w, uint32_t h, size_t sstride, size_t dstride, PlaneArgs &a)
{
    a.src_stride = sstride;
    a.dst_stride = dstride;
    if (sstride == w && dstride == w && (uint64_t)w * h <= 0xFFFFFFF0ull) {
        a.W = w * h;
        a.H = 1;
    } else {
        a.W = w;
        a.H = h;
    }
}

static dim3 plane_grid(const PlaneArgs &a, uint32_t frames)
{
    const uint64_t segs = (uint64_t)((a.W + 15u) >> 4) * a.H;
    // 16 bytes per lane per trip; enough workgroups for every CU several times over
    return dim3((uint32_t)std::min<uint64_t>((segs + 255) / 256, 4096), frames);
}

static int check_same_size(stepth_engine *e, const stepth_image *a, const stepth_image *b)
{
    if (a->width != b->width || a->height != b->height)
        return fail(e, STEPTH_ERR_SIZE_MISMATCH, "Sizes don't match");
    return STEPTH_OK;
}

static int check_plane_batch(stepth_engine *e, uint32_t n, const stepth_image *ims, const char *name,
                             uint32_t channels)
{
    for (uint32_t f = 0; f < n; ++f) {
        TRY(check_image(e, &ims[f], name, channels, channels));
        if (ims[f].memory != STEPTH_MEM_DEVICE || ims[f].width != ims[0].width ||
            ims[f].height != ims[0].height || ims[f].stride != ims[0].stride)
            return fail(e, STEPTH_ERR_BAD_ARG, "batch: device images of one shape only");
    }
    return STEPTH_OK;
}

// histograms of up to kMaxBatch device planes -> e->zone_counts[0..nf)[256]
static int histogram_device(stepth_engine *e, uint32_t nf, const uint8_t *const *planes, size_t stride,
                            uint32_t w, uint32_t h)
{
    TRY(reserve(e, e->zone_counts, (size_t)kMaxBatch * 256 * sizeof(uint32_t)));
    TRY(reserve(e, e->zone_tables, (size_t)kMaxBatch * sizeof(ZoneTable)));
    PlaneArgs a{};
    for (uint32_t f = 0; f < nf; ++f)
        a.src[f] = planes[f];
    plane_shape(w, h, stride, stride, a);
    HIP_TRY(e, hipMemsetAsync(e->zone_counts.p, 0, (size_t)nf * 256 * sizeof(uint32_t), e->stream));
    // every workgroup ends with up to 256 atomics on the frame's counts, and atomics on one
    // address serialise in L2: about a thousand workgroups in all keep that tail short
    const uint64_t segs = (uint64_t)((a.W + 15u) >> 4) * a.H;
    const uint32_t per_frame = std::max<uint32_t>(1024u / nf, 8u);
    const dim3 hgrid((uint32_t)std::min<uint64_t>((segs + 256 * kHistUnroll - 1) / (256 * kHistUnroll), per_frame), nf);
    hipLaunchKernelGGL(depth_histogram_kernel, hgrid, dim3(256), 0, e->stream, a,
                       static_cast<uint32_t *>(e->zone_counts.p));
    HIP_TRY(e, hipGetLastError());
    return STEPTH_OK;
}

// histogram + split of up to kMaxBatch device planes -> e->zone_tables[0..nf)
static int split_device(stepth_engine *e, uint32_t nf, const uint8_t *const *planes, size_t stride,
                        uint32_t w, uint32_t h, uint32_t zones)
{
    TRY(histogram_device(e, nf, planes, stride, w, h));
    hipLaunchKernelGGL(depth_split_kernel, dim3(nf), dim3(256), 0, e->stream,
                       static_cast<const uint32_t *>(e->zone_counts.p), zones,
                       static_cast<ZoneTable *>(e->zone_tables.p));
    HIP_TRY(e, hipGetLastError());
    return STEPTH_OK;
}

static int slice_device(stepth_engine *e, uint32_t nf, const uint8_t *const *planes, size_t sstride,
                        uint8_t *const *masks, size_t mstride, uint32_t w, uint32_t h, uint32_t from,
                        uint32_t to, bool from_tables)
{
    PlaneArgs a{};
    for (uint32_t f = 0; f < nf; ++f) {
        a.src[f] = planes[f];
        a.dst[f] = masks[f];
    }
    plane_shape(w, h, sstride, mstride, a);
    hipLaunchKernelGGL(slice_kernel, plane_grid(a, nf), dim3(256), 0, e->stream, a, from, to,
                       from_tables ? static_cast<const ZoneTable *>(e->zone_tables.p) : nullptr, 0u);
    HIP_TRY(e, hipGetLastError());
    return STEPTH_OK;
}

// a host or device output plane: the device view to write and what to copy back afterwards
struct OutPlane {
    uint8_t *d = nullptr;
    size_t stride = 0;
    bool staged = false;
};
static int out_plane(stepth_engine *e, const stepth_image &im, DevBuf &stage, OutPlane &o)
{
    const size_t row = (size_t)im.width * im.channels;
    if (im.memory == STEPTH_MEM_DEVICE) {
        o.d = static_cast<uint8_t *>(const_cast<void *>(im.data));
        o.stride = im.stride;
        return STEPTH_OK;
    }
    o.stride = (row + 15) & ~(size_t)15;
    TRY(reserve(e, stage, o.stride * im.height));
    o.d = static_cast<uint8_t *>(stage.p);
    o.staged = true;
    return STEPTH_OK;
}

static int finish_plane(stepth_engine *e, const stepth_image &im, const OutPlane &o, bool async)
{
    if (async)
        return STEPTH_OK;
    if (o.staged)
        HIP_TRY(e, hipMemcpy2DAsync(const_cast<void *>(im.data), im.stride, o.d, o.stride,
                                    (size_t)im.width * im.channels, im.height,
                                    hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(e, hipStreamSynchronize(e->stream));
    return STEPTH_OK;
}

static int async_needs_device(stepth_engine *e, uint32_t flags, std::initializer_list<const stepth_image *> ims)
{
    if (flags & STEPTH_FLAG_ASYNC)
        for (const stepth_image *im : ims)
            if (im->memory != STEPTH_MEM_DEVICE)
                return fail(e, STEPTH_ERR_BAD_ARG, "STEPTH_FLAG_ASYNC needs device memory everywhere");
    return STEPTH_OK;
}

int stepth_invert_depth(stepth_engine *e, const stepth_image *depth, uint32_t flags)
{
    STEPTH_GUARD_BEGIN
    if (!e)
        return STEPTH_ERR_BAD_ARG;
    TRY(check_image(e, depth, "depth", 1, 1));
    TRY(async_needs_device(e, flags, {depth}));
    HIP_TRY(e, hipSetDevice(e->device));
    const uint8_t *d_src = nullptr;
    size_t ss = 0;
    TRY(to_device(e, *depth, e->gauss_io, d_src, ss));
    PlaneArgs a{};
    a.src[0] = d_src;
    a.dst[0] = const_cast<uint8_t *>(d_src); // in place (the staging copy for host memory)
    plane_shape(depth->width, depth->height, ss, ss, a);
    hipLaunchKernelGGL(invert_kernel, plane_grid(a, 1), dim3(256), 0, e->stream, a);
    HIP_TRY(e, hipGetLastError());
    OutPlane o{const_cast<uint8_t *>(d_src), ss, depth->memory == STEPTH_MEM_HOST};
    return finish_plane(e, *depth, o, flags & STEPTH_FLAG_ASYNC);
    STEPTH_GUARD_END(e)
}

static void zone_from_table(const ZoneTable &t, uint32_t i, stepth_zone &z)
{
    z.has_from = z.has_to = 1;
    z.from = t.lo[i];
    z.to = t.hi[i];
}

static int zones_to_host(stepth_engine *e, uint32_t zones, stepth_zone *out, uint32_t capacity,
                         uint32_t *n_out)
{
    ZoneTable t;
    HIP_TRY(e, hipMemcpyAsync(&t, e->zone_tables.p, sizeof t, hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(e, hipStreamSynchronize(e->stream));
    if (t.status != 0)
        return fail(e, t.status, "depth_split(%u): %s", zones, stepth_status_string(t.status));
    if (t.n > capacity)
        return fail(e, STEPTH_ERR_BAD_ARG, "depth_split found %u zones, capacity is %u", t.n, capacity);
    for (uint32_t i = 0; i < t.n; ++i)
        zone_from_table(t, i, out[i]);
    *n_out = t.n;
    return STEPTH_OK;
}

int stepth_depth_split(stepth_engine *e, const stepth_image *depth, uint32_t zones, stepth_zone *out,
                       uint32_t capacity, uint32_t *n_out)
{
    STEPTH_GUARD_BEGIN
    if (!e)
        return STEPTH_ERR_BAD_ARG;
    TRY(check_image(e, depth, "depth", 1, 1));
    if (!out || !n_out || capacity == 0)
        return fail(e, STEPTH_ERR_BAD_ARG, "null output");
    *n_out = 0;
    if (zones > 255u)
        return fail(e, STEPTH_ERR_BAD_ARG, "zones is a u8 in the reference (got %u)", zones);
    if (zones < 2u) { // src/depth_image.rs:163-165
        out[0] = stepth_zone{0, 0, 0, 0};
        *n_out = 1;
        return STEPTH_OK;
    }
    HIP_TRY(e, hipSetDevice(e->device));
    const uint8_t *d_src = nullptr;
    size_t ss = 0;
    TRY(to_device(e, *depth, e->gauss_io, d_src, ss));
    TRY(split_device(e, 1, &d_src, ss, depth->width, depth->height, zones));
    return zones_to_host(e, zones, out, capacity, n_out);
    STEPTH_GUARD_END(e)
}

int stepth_depth_histogram(stepth_engine *e, const stepth_image *depth, uint32_t *counts,
                           uint32_t counts_memory, uint32_t flags)
{
    STEPTH_GUARD_BEGIN
    if (!e)
        return STEPTH_ERR_BAD_ARG;
    TRY(check_image(e, depth, "depth", 1, 1));
    if (!counts || (counts_memory != STEPTH_MEM_HOST && counts_memory != STEPTH_MEM_DEVICE))
        return fail(e, STEPTH_ERR_BAD_ARG, "counts: null pointer or bad memory kind");
    const bool async = flags & STEPTH_FLAG_ASYNC;
    if (async && (depth->memory != STEPTH_MEM_DEVICE || counts_memory != STEPTH_MEM_DEVICE))
        return fail(e, STEPTH_ERR_BAD_ARG, "STEPTH_FLAG_ASYNC needs device memory everywhere");
    HIP_TRY(e, hipSetDevice(e->device));
    const uint8_t *d_src = nullptr;
    size_t ss = 0;
    TRY(to_device(e, *depth, e->gauss_io, d_src, ss));
    TRY(histogram_device(e, 1, &d_src, ss, depth->width, depth->height));
    HIP_TRY(e, hipMemcpyAsync(counts, e->zone_counts.p, 256 * sizeof(uint32_t),
                              counts_memory == STEPTH_MEM_HOST ? hipMemcpyDeviceToHost
                                                               : hipMemcpyDeviceToDevice,
                              e->stream));
    if (!async)
        HIP_TRY(e, hipStreamSynchronize(e->stream));
    return STEPTH_OK;
    STEPTH_GUARD_END(e)
}

int stepth_zones_from_histogram(stepth_engine *e, const uint32_t *counts, uint32_t counts_memory,
                                uint32_t zones, stepth_zone *out, uint32_t capacity, uint32_t *n_out)
{
    STEPTH_GUARD_BEGIN
    if (!e)
        return STEPTH_ERR_BAD_ARG;
    if (!counts || !out || !n_out || capacity == 0 ||
        (counts_memory != STEPTH_MEM_HOST && counts_memory != STEPTH_MEM_DEVICE))
        return fail(e, STEPTH_ERR_BAD_ARG, "null pointer or bad memory kind");
    *n_out = 0;
    if (zones > 255u)
        return fail(e, STEPTH_ERR_BAD_ARG, "zones is a u8 in the reference (got %u)", zones);
    if (zones < 2u) { // src/depth_image.rs:163-165
        out[0] = stepth_zone{0, 0, 0, 0};
        *n_out = 1;
        return STEPTH_OK;
    }
    HIP_TRY(e, hipSetDevice(e->device));
    TRY(reserve(e, e->zone_counts, (size_t)kMaxBatch * 256 * sizeof(uint32_t)));
    TRY(reserve(e, e->zone_tables, (size_t)kMaxBatch * sizeof(ZoneTable)));
    HIP_TRY(e, hipMemcpyAsync(e->zone_counts.p, counts, 256 * sizeof(uint32_t),
                              counts_memory == STEPTH_MEM_HOST ? hipMemcpyHostToDevice
                                                               : hipMemcpyDeviceToDevice,
                              e->stream));
    hipLaunchKernelGGL(depth_split_kernel, dim3(1), dim3(256), 0, e->stream,
                       static_cast<const uint32_t *>(e->zone_counts.p), zones,
                       static_cast<ZoneTable *>(e->zone_tables.p));
    HIP_TRY(e, hipGetLastError());
    return zones_to_host(e, zones, out, capacity, n_out);
    STEPTH_GUARD_END(e)
}

int stepth_slice(stepth_engine *e, const stepth_image *depth, const stepth_zone *range,
                 const stepth_image *mask, uint32_t flags)
{
    STEPTH_GUARD_BEGIN
    if (!e)
        return STEPTH_ERR_BAD_ARG;
    TRY(check_image(e, depth, "depth", 1, 1));
    TRY(check_image(e, mask, "mask", 1, 1));
    if (!range)
        return fail(e, STEPTH_ERR_BAD_ARG, "null range");
    TRY(check_same_size(e, depth, mask));
    TRY(async_needs_device(e, flags, {depth, mask}));
    HIP_TRY(e, hipSetDevice(e->device));
    const uint8_t *d_src = nullptr;
    size_t ss = 0;
    TRY(to_device(e, *depth, e->gauss_io, d_src, ss));
    OutPlane o;
    TRY(out_plane(e, *mask, e->stage_out, o));
    const uint32_t from = range->has_from ? range->from : 0u;  // :231
    const uint32_t to = range->has_to ? range->to : 255u;      // :232
    TRY(slice_device(e, 1, &d_src, ss, &o.d, o.stride, depth->width, depth->height, from, to, false));
    return finish_plane(e, *mask, o, flags & STEPTH_FLAG_ASYNC);
    STEPTH_GUARD_END(e)
}

int stepth_select_foreground(stepth_engine *e, const stepth_image *depth, const stepth_image *mask,
                             stepth_zone *zone_out)
{
    STEPTH_GUARD_BEGIN
    if (!e)
        return STEPTH_ERR_BAD_ARG;
    TRY(check_image(e, depth, "depth", 1, 1));
    TRY(check_image(e, mask, "mask", 1, 1));
    TRY(check_same_size(e, depth, mask));
    HIP_TRY(e, hipSetDevice(e->device));
    const uint8_t *d_src = nullptr;
    size_t ss = 0;
    TRY(to_device(e, *depth, e->gauss_io, d_src, ss));
    OutPlane o;
    TRY(out_plane(e, *mask, e->stage_out, o));
    TRY(split_device(e, 1, &d_src, ss, depth->width, depth->height, 2));
    TRY(slice_device(e, 1, &d_src, ss, &o.d, o.stride, depth->width, depth->height, 0, 255, true));
    ZoneTable t;
    HIP_TRY(e, hipMemcpyAsync(&t, e->zone_tables.p, sizeof t, hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(e, hipStreamSynchronize(e->stream));
    if (t.status != 0)
        return fail(e, t.status, "select_foreground: %s", stepth_status_string(t.status));
    if (zone_out)
        zone_from_table(t, 0, *zone_out);
    return finish_plane(e, *mask, o, false);
    STEPTH_GUARD_END(e)
}

int stepth_select_foreground_batch(stepth_engine *e, uint32_t n_frames, const stepth_image *depths,
                                   const stepth_image *masks, stepth_zone *zones_out, int *statuses,
                                   uint32_t flags)
{
    STEPTH_GUARD_BEGIN
    if (!e)
        return STEPTH_ERR_BAD_ARG;
    if (n_frames == 0 || !depths || !masks)
        return fail(e, STEPTH_ERR_BAD_ARG, "empty batch or null pointer");
    const bool async = flags & STEPTH_FLAG_ASYNC;
    if (async && (zones_out || statuses))
        return fail(e, STEPTH_ERR_BAD_ARG, "STEPTH_FLAG_ASYNC: zones_out and statuses must be NULL");
    TRY(check_plane_batch(e, n_frames, depths, "depth", 1));
    TRY(check_plane_batch(e, n_frames, masks, "mask", 1));
    TRY(check_same_size(e, &depths[0], &masks[0]));
    HIP_TRY(e, hipSetDevice(e->device));
    int first_bad = STEPTH_OK;
    std::vector<ZoneTable> host(async ? 0 : std::min<uint32_t>(n_frames, kMaxBatch));
    for (uint32_t f0 = 0; f0 < n_frames; f0 += kMaxBatch) {
        const uint32_t nf = std::min<uint32_t>(kMaxBatch, n_frames - f0);
        const uint8_t *src[kMaxBatch];
        uint8_t *dst[kMaxBatch];
        for (uint32_t f = 0; f < nf; ++f) {
            src[f] = static_cast<const uint8_t *>(depths[f0 + f].data);
            dst[f] = static_cast<uint8_t *>(const_cast<void *>(masks[f0 + f].data));
        }
        TRY(split_device(e, nf, src, depths[0].stride, depths[0].width, depths[0].height, 2));
        TRY(slice_device(e, nf, src, depths[0].stride, dst, masks[0].stride, depths[0].width,
                         depths[0].height, 0, 255, true));
        if (async)
            continue;
        // the tables are reused by the next chunk: read them back before going on
        HIP_TRY(e, hipMemcpyAsync(host.data(), e->zone_tables.p, nf * sizeof(ZoneTable),
                                  hipMemcpyDeviceToHost, e->stream));
        HIP_TRY(e, hipStreamSynchronize(e->stream));
        for (uint32_t f = 0; f < nf; ++f) {
            if (statuses)
                statuses[f0 + f] = host[f].status;
            if (host[f].status != 0 && first_bad == STEPTH_OK)
                first_bad = host[f].status;
            if (zones_out) {
                zones_out[f0 + f] = stepth_zone{0, 0, 0, 0};
                if (host[f].status == 0)
                    zone_from_table(host[f], 0, zones_out[f0 + f]);
            }
        }
    }
    if (first_bad != STEPTH_OK)
        return fail(e, first_bad, "select_foreground_batch: %s", stepth_status_string(first_bad));
    return STEPTH_OK;
    STEPTH_GUARD_END(e)
}

int stepth_apply_mask(stepth_engine *e, const stepth_image *rgba, const stepth_image *mask,
                      uint32_t flags)
{
    STEPTH_GUARD_BEGIN
    if (!e)
        return STEPTH_ERR_BAD_ARG;
    TRY(check_image(e, rgba, "rgba", 4, 4));
    TRY(check_image(e, mask, "mask", 1, 1));
    TRY(check_same_size(e, rgba, mask));
    TRY(async_needs_device(e, flags, {rgba, mask}));
    HIP_TRY(e, hipSetDevice(e->device));
    const uint8_t *d_rgba = nullptr, *d_mask = nullptr;
    size_t rs = 0, ms = 0;
    TRY(to_device(e, *rgba, e->stage_main, d_rgba, rs));
    TRY(to_device(e, *mask, e->gauss_io, d_mask, ms));
    ApplyArgs a{};
    a.rgba[0] = const_cast<uint8_t *>(d_rgba);
    a.mask[0] = d_mask;
    a.rgba_stride = rs;
    a.mask_stride = ms;
    a.W = rgba->width;
    a.H = rgba->height;
    const uint64_t quads = (uint64_t)((a.W + 3u) >> 2) * a.H;
    hipLaunchKernelGGL(apply_mask_kernel, dim3((uint32_t)std::min<uint64_t>((quads + 255) / 256, 8192)),
                       dim3(256), 0, e->stream, a);
    HIP_TRY(e, hipGetLastError());
    OutPlane o{const_cast<uint8_t *>(d_rgba), rs, rgba->memory == STEPTH_MEM_HOST};
    return finish_plane(e, *rgba, o, flags & STEPTH_FLAG_ASYNC);
    STEPTH_GUARD_END(e)
}
