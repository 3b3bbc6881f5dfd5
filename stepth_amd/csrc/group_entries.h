// group_entries.h -- C-ABI entry points of the adaptive grouping stand-in (group_kernels.h) and
// of the batched group-list flow; included by engine.hip inside its extern "C" block.

// grouping of nf <= kMaxBatch device images of one shape into out[f] (device, `capacity` groups
// each; nullptr = the engine's own buffer); the counts land in e->group_counters[4 f kGroupCtrPad]
static int adaptive_groups_device(stepth_engine *e, uint32_t nf, const uint8_t *const *d_mains, size_t ms,
                                  uint32_t ch, uint32_t W, uint32_t H, const uint8_t precision[3],
                                  uint32_t min_splits, uint32_t max_splits, GroupRec *const *out)
{
    const uint64_t px = (uint64_t)W * H;
    if (px > 0x7FFFFFFFull)
        return fail(e, STEPTH_ERR_BAD_ARG, "adaptive groups: image of %llu pixels is too large",
                    (unsigned long long)px);
    if (max_splits == 0) // (w*h as f32).log2().ceil(), src/depth_image.rs:101-103
        max_splits = (uint32_t)std::ceil(std::log2((float)px));
    if (max_splits > 40u)
        return fail(e, STEPTH_ERR_BAD_ARG, "max_splits %u: at most 40", max_splits);
    // a node has at least one pixel, so no level holds more than W*H nodes; splitting past
    // ceil(log2(px)) + 1 levels cannot happen either
    const uint32_t seed_depth = std::min({min_splits, max_splits, 31u});
    if (!out)
        TRY(reserve(e, e->group_out, (size_t)px * sizeof(GroupRec) * nf));
    // adaptive_tree_kernel (no node lists): 16-bit coordinates, subtrees of at most kTreeLevels levels
    bool one_launch = W <= 0xFFFFu && H <= 0xFFFFu;
    uint32_t levels = 0; // levels below the seed depth
    {
        uint32_t w = W, h = H; // the largest node of a depth: always the ceiling half
        for (uint32_t d = 0; d < max_splits && (uint64_t)w * h > 1; ++d) {
            (w >= h ? w : h) -= (w >= h ? w : h) / 2u;
            levels += d >= seed_depth;
        }
        one_launch = one_launch && levels <= kTreeLevels;
        levels = std::max(levels, 1u);
    }
    for (int k = 0; k < 3 && !one_launch; ++k)
        TRY(reserve(e, e->group_list[k], (size_t)px * sizeof(GroupNode) * nf));
    TRY(reserve(e, e->group_counters, 16 * kGroupCtrPad * kMaxBatch));
    AdaptiveArgs a{};
    for (uint32_t f = 0; f < nf; ++f) {
        a.px[f] = d_mains[f];
        a.groups[f] = out ? out[f] : static_cast<GroupRec *>(e->group_out.p) + (size_t)px * f;
    }
    a.stride = ms;
    a.ch = ch;
    a.W = W;
    a.H = H;
    for (int k = 0; k < 3; ++k)
        a.prec[k] = precision[k];
    a.min_splits = min_splits;
    a.max_splits = max_splits;
    a.seed_depth = seed_depth;
    for (int k = 0; k < 3; ++k)
        a.list[k] = static_cast<GroupNode *>(e->group_list[k].p);
    a.list_stride = (size_t)px;
    a.counters = static_cast<uint32_t *>(e->group_counters.p);
    HIP_TRY(e, hipMemsetAsync(a.counters, 0, 16 * kGroupCtrPad * nf, e->stream));
    const uint64_t seeds = 1ull << seed_depth;
    if (one_launch) { // the whole tree below the seeds in one launch
        hipLaunchKernelGGL(adaptive_tree_kernel, dim3((uint32_t)std::min<uint64_t>((seeds + 255) / 256, 4096), nf),
                           dim3(256), levels * kTreeLevelCap * sizeof(uint2), e->stream, a, levels);
        HIP_TRY(e, hipGetLastError());
        return STEPTH_OK;
    }
    hipLaunchKernelGGL(adaptive_seed_kernel,
                       dim3((uint32_t)std::min<uint64_t>((seeds + 255) / 256, 4096), nf), dim3(256), 0,
                       e->stream, a);
    const uint32_t level_grid = (uint32_t)std::min<uint64_t>((px + 255) / 256, nf > 1 ? 512 : 2048);
    for (uint32_t d = seed_depth + 1; d <= max_splits; ++d) // three lists rotate, see the kernel
        hipLaunchKernelGGL(adaptive_level_kernel, dim3(level_grid, nf), dim3(256), 0, e->stream, a, d);
    HIP_TRY(e, hipGetLastError());
    return STEPTH_OK;
}

int stepth_adaptive_groups(stepth_engine *e, const stepth_image *main_im, const uint8_t precision[3],
                           uint32_t min_splits, uint32_t max_splits, stepth_group *groups,
                           size_t capacity, uint32_t groups_memory, size_t *n_out)
{
    STEPTH_GUARD_BEGIN
    if (!e)
        return STEPTH_ERR_BAD_ARG;
    TRY(check_image(e, main_im, "main", 3, 4));
    if (!precision || !n_out || (groups_memory != STEPTH_MEM_HOST && groups_memory != STEPTH_MEM_DEVICE))
        return fail(e, STEPTH_ERR_BAD_ARG, "null pointer or bad memory kind");
    *n_out = 0;
    HIP_TRY(e, hipSetDevice(e->device));
    const uint8_t *d_main = nullptr;
    size_t ms = 0;
    TRY(to_device(e, *main_im, e->stage_main, d_main, ms));
    TRY(adaptive_groups_device(e, 1, &d_main, ms, main_im->channels, main_im->width, main_im->height,
                               precision, min_splits, max_splits, nullptr));
    uint32_t n = 0;
    HIP_TRY(e, hipMemcpyAsync(&n, e->group_counters.p, 4, hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(e, hipStreamSynchronize(e->stream));
    *n_out = n;
    if (!groups)
        return STEPTH_OK; // count only; the groups stay in the engine's buffer until the next call
    if (n > capacity)
        return fail(e, STEPTH_ERR_BAD_ARG, "adaptive groups: %u groups, capacity is %zu", n, capacity);
    HIP_TRY(e, hipMemcpyAsync(groups, e->group_out.p, (size_t)n * sizeof(GroupRec),
                              groups_memory == STEPTH_MEM_HOST ? hipMemcpyDeviceToHost
                                                               : hipMemcpyDeviceToDevice,
                              e->stream));
    HIP_TRY(e, hipStreamSynchronize(e->stream));
    return STEPTH_OK;
    STEPTH_GUARD_END(e)
}

static int check_frames(stepth_engine *e, uint32_t n_frames, const stepth_image *mains,
                        const stepth_image *adds, const stepth_image *depths)
{
    for (uint32_t f = 0; f < n_frames; ++f) {
        TRY(check_image(e, &mains[f], "main", 3, 4));
        if (mains[f].memory != STEPTH_MEM_DEVICE || mains[f].width != mains[0].width ||
            mains[f].height != mains[0].height || mains[f].channels != mains[0].channels ||
            mains[f].stride != mains[0].stride)
            return fail(e, STEPTH_ERR_BAD_ARG, "batch: main images must be device memory of one shape (frame %u)", f);
        if (adds) {
            TRY(check_image(e, &adds[f], "additional", 3, 3));
            if (adds[f].memory != STEPTH_MEM_DEVICE || adds[f].width != adds[0].width ||
                adds[f].height != adds[0].height || adds[f].stride != adds[0].stride)
                return fail(e, STEPTH_ERR_BAD_ARG, "batch: additional images must be device memory of one shape (frame %u)", f);
        }
        if (depths) {
            TRY(check_image(e, &depths[f], "depth", 1, 1));
            if (depths[f].memory != STEPTH_MEM_DEVICE || depths[f].stride != depths[0].stride)
                return fail(e, STEPTH_ERR_BAD_ARG, "batch: depth maps must be device memory of one stride (frame %u)", f);
            if (depths[f].width != mains[0].width || depths[f].height != mains[0].height)
                return fail(e, STEPTH_ERR_SIZE_MISMATCH, "Sizes don't match");
        }
    }
    return STEPTH_OK;
}

int stepth_adaptive_groups_batch(stepth_engine *e, uint32_t n_frames, const stepth_image *mains,
                                 const uint8_t precision[3], uint32_t min_splits, uint32_t max_splits,
                                 stepth_group *const *groups_out, size_t capacity, size_t *n_out)
{
    STEPTH_GUARD_BEGIN
    if (!e)
        return STEPTH_ERR_BAD_ARG;
    if (n_frames == 0 || !mains || !precision || !groups_out || !n_out)
        return fail(e, STEPTH_ERR_BAD_ARG, "empty batch or null pointer");
    TRY(check_frames(e, n_frames, mains, nullptr, nullptr));
    const uint32_t W = mains[0].width, H = mains[0].height;
    if (capacity < (size_t)W * H)
        return fail(e, STEPTH_ERR_BAD_ARG, "adaptive groups batch: every list needs room for width*height groups");
    HIP_TRY(e, hipSetDevice(e->device));
    for (uint32_t f0 = 0; f0 < n_frames; f0 += kMaxBatch) {
        const uint32_t nf = std::min<uint32_t>(kMaxBatch, n_frames - f0);
        const uint8_t *d_mains[kMaxBatch];
        GroupRec *outs[kMaxBatch];
        for (uint32_t f = 0; f < nf; ++f) {
            d_mains[f] = static_cast<const uint8_t *>(mains[f0 + f].data);
            outs[f] = reinterpret_cast<GroupRec *>(groups_out[f0 + f]);
        }
        TRY(adaptive_groups_device(e, nf, d_mains, mains[0].stride, mains[0].channels, W, H, precision,
                                   min_splits, max_splits, outs));
        std::vector<uint32_t> cnt((size_t)4 * kGroupCtrPad * nf);
        HIP_TRY(e, hipMemcpyAsync(cnt.data(), e->group_counters.p, cnt.size() * 4, hipMemcpyDeviceToHost, e->stream));
        HIP_TRY(e, hipStreamSynchronize(e->stream));
        for (uint32_t f = 0; f < nf; ++f)
            n_out[f0 + f] = cnt[(size_t)4 * kGroupCtrPad * f];
    }
    return STEPTH_OK;
    STEPTH_GUARD_END(e)
}

int stepth_depth_from_groups_batch(stepth_engine *e, uint32_t n_frames, const stepth_image *mains,
                                   const stepth_image *adds, const stepth_params *pp,
                                   const stepth_group *const *groups, const size_t *n_groups,
                                   const stepth_image *depths, uint32_t *max_out)
{
    STEPTH_GUARD_BEGIN
    if (!e)
        return STEPTH_ERR_BAD_ARG;
    if (!pp || pp->struct_size != sizeof(stepth_params))
        return fail(e, STEPTH_ERR_BAD_ARG, "params null or struct_size mismatch");
    if (n_frames == 0 || !mains || !adds || !depths || !groups || !n_groups)
        return fail(e, STEPTH_ERR_BAD_ARG, "empty batch or null pointer");
    const stepth_params p = *pp;
    if (p.mode != STEPTH_MODE_RING_FIRST_HIT || p.row_block)
        return fail(e, STEPTH_ERR_BAD_ARG, "group-list batch: ring mode, contiguous strips");
    const bool gauss_tail = p.flags & STEPTH_FLAG_GAUSSIAN_TAIL; // src/depth_image.rs:133-134
    TRY(check_frames(e, n_frames, mains, adds, depths));
    const uint32_t H = mains[0].height;
    uint32_t row0 = p.row_begin, row1 = p.row_end;
    if (row0 == 0 && row1 == 0)
        row1 = H;
    if (row0 >= row1 || row1 > H)
        return fail(e, STEPTH_ERR_BAD_ARG, "row strip [%u,%u) outside 0..%u", row0, row1, H);
    if (gauss_tail && ((p.flags & STEPTH_FLAG_RAW_ONLY) || row0 != 0 || row1 != H))
        return fail(e, STEPTH_ERR_BAD_ARG, "STEPTH_FLAG_GAUSSIAN_TAIL needs full-frame, normalised calls");
    HIP_TRY(e, hipSetDevice(e->device));
    const uint64_t cells = (uint64_t)((adds[0].width + 7u) / 8u) * ((adds[0].height + 7u) / 8u) * 64u;
    const uint32_t chunk = (uint32_t)std::max<uint64_t>(
        1, std::min<uint64_t>(kMaxBatch, 0x3FFFFFFFull / std::max<uint64_t>(cells, 1)));
    bool any_zero = false;
    for (uint32_t f0 = 0; f0 < n_frames; f0 += chunk) {
        const uint32_t nf = std::min<uint32_t>(chunk, n_frames - f0);
        const void *d_groups[kMaxBatch];
        uint32_t ng[kMaxBatch];
        for (uint32_t f = 0; f < nf; ++f) {
            if (!groups[f0 + f] || n_groups[f0 + f] == 0 || n_groups[f0 + f] > (1u << 26))
                return fail(e, STEPTH_ERR_BAD_ARG, "group list of frame %u: null, empty or beyond 2^26 entries", f0 + f);
            d_groups[f] = groups[f0 + f];
            ng[f] = (uint32_t)n_groups[f0 + f];
        }
        TRY(ring_queries_batch(e, nf, mains + f0, adds + f0, p, row0, row1, RowPattern{0u, 1u, 0u}, d_groups, ng,
                               depths + f0, max_out ? max_out + f0 : nullptr, any_zero));
        if (gauss_tail) { // in place on the painted maps
            uint8_t *planes[kMaxBatch];
            for (uint32_t f = 0; f < nf; ++f)
                planes[f] = static_cast<uint8_t *>(const_cast<void *>(depths[f0 + f].data));
            TRY(gaussian_device_batch(e, nf, planes, depths[0].stride, mains[0].width, H, planes, depths[0].stride));
        }
    }
    if (!(p.flags & STEPTH_FLAG_ASYNC) && any_zero && !(p.flags & STEPTH_FLAG_RAW_ONLY))
        return fail(e, STEPTH_ERR_MAX_ZERO, "a frame's distances are all 0; its output is all zero");
    return STEPTH_OK;
    STEPTH_GUARD_END(e)
}

// fetches the group counts of the grouping launches above (the flow's one host synchronisation)
static int fetch_group_counts(stepth_engine *e, uint32_t nf, uint32_t *out)
{
    std::vector<uint32_t> cnt((size_t)4 * kGroupCtrPad * nf);
    HIP_TRY(e, hipMemcpyAsync(cnt.data(), e->group_counters.p, cnt.size() * 4, hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(e, hipStreamSynchronize(e->stream));
    for (uint32_t f = 0; f < nf; ++f)
        out[f] = cnt[(size_t)4 * kGroupCtrPad * f];
    return STEPTH_OK;
}

int stepth_adaptive_depth_batch(stepth_engine *e, uint32_t n_frames, const stepth_image *mains,
                                const stepth_image *adds, const stepth_params *pp, uint32_t min_splits,
                                uint32_t max_splits, const stepth_image *depths, uint32_t *max_out,
                                size_t *n_groups_out)
{
    STEPTH_GUARD_BEGIN
    if (!e)
        return STEPTH_ERR_BAD_ARG;
    if (!pp || pp->struct_size != sizeof(stepth_params))
        return fail(e, STEPTH_ERR_BAD_ARG, "params null or struct_size mismatch");
    if (n_frames == 0 || !mains || !adds || !depths)
        return fail(e, STEPTH_ERR_BAD_ARG, "empty batch or null pointer");
    const stepth_params p = *pp;
    if (p.mode != STEPTH_MODE_RING_FIRST_HIT || p.row_block || p.row_begin || (p.row_end && p.row_end != mains[0].height))
        return fail(e, STEPTH_ERR_BAD_ARG, "adaptive depth batch: ring mode, whole frames");
    const bool gauss_tail = p.flags & STEPTH_FLAG_GAUSSIAN_TAIL; // src/depth_image.rs:133-134
    if (gauss_tail && (p.flags & STEPTH_FLAG_RAW_ONLY))
        return fail(e, STEPTH_ERR_BAD_ARG, "STEPTH_FLAG_GAUSSIAN_TAIL needs normalised maps");
    TRY(check_frames(e, n_frames, mains, adds, depths));
    const uint32_t W = mains[0].width, H = mains[0].height;
    HIP_TRY(e, hipSetDevice(e->device));
    const uint64_t cells = (uint64_t)((adds[0].width + 7u) / 8u) * ((adds[0].height + 7u) / 8u) * 64u;
    const uint32_t chunk = (uint32_t)std::max<uint64_t>(
        1, std::min<uint64_t>(kMaxBatch, 0x3FFFFFFFull / std::max<uint64_t>(cells, 1)));
    bool any_zero = false;
    for (uint32_t f0 = 0; f0 < n_frames; f0 += chunk) {
        const uint32_t nf = std::min<uint32_t>(chunk, n_frames - f0);
        const uint8_t *d_mains[kMaxBatch];
        const void *d_groups[kMaxBatch];
        uint32_t ng[kMaxBatch];
        for (uint32_t f = 0; f < nf; ++f)
            d_mains[f] = static_cast<const uint8_t *>(mains[f0 + f].data);
        // src/depth_image.rs:101-110 (the engine's own grouping rule, parity with disage unpinned)
        TRY(adaptive_groups_device(e, nf, d_mains, mains[0].stride, mains[0].channels, W, H, p.precision,
                                   min_splits, max_splits, nullptr));
        for (uint32_t f = 0; f < nf; ++f)
            d_groups[f] = static_cast<GroupRec *>(e->group_out.p) + (size_t)W * H * f;
        // :112-132; the counts come back while the pack kernels run
        TRY(ring_queries_batch(e, nf, mains + f0, adds + f0, p, 0, H, RowPattern{0u, 1u, 0u}, d_groups, ng,
                               depths + f0, max_out ? max_out + f0 : nullptr, any_zero, fetch_group_counts, ng));
        if (n_groups_out)
            for (uint32_t f = 0; f < nf; ++f)
                n_groups_out[f0 + f] = ng[f];
        if (gauss_tail) {
            uint8_t *planes[kMaxBatch];
            for (uint32_t f = 0; f < nf; ++f)
                planes[f] = static_cast<uint8_t *>(const_cast<void *>(depths[f0 + f].data));
            TRY(gaussian_device_batch(e, nf, planes, depths[0].stride, W, H, planes, depths[0].stride));
        }
    }
    if (!(p.flags & STEPTH_FLAG_ASYNC) && any_zero && !(p.flags & STEPTH_FLAG_RAW_ONLY))
        return fail(e, STEPTH_ERR_MAX_ZERO, "a frame's distances are all 0; its output is all zero");
    return STEPTH_OK;
    STEPTH_GUARD_END(e)
}
