// group_entries.h -- C-ABI entry point of the adaptive grouping stand-in (group_kernels.h);
// included by engine.hip inside its extern "C" block.

int stepth_adaptive_groups(stepth_engine *e, const stepth_image *main_im, const uint8_t precision[3],
                           uint32_t min_splits, uint32_t max_splits, stepth_group *groups,
                           size_t capacity, uint32_t groups_memory, size_t *n_out)
{
    if (!e)
        return STEPTH_ERR_BAD_ARG;
    TRY(check_image(e, main_im, "main", 3, 4));
    if (!precision || !n_out || (groups_memory != STEPTH_MEM_HOST && groups_memory != STEPTH_MEM_DEVICE))
        return fail(e, STEPTH_ERR_BAD_ARG, "null pointer or bad memory kind");
    *n_out = 0;
    const uint32_t W = main_im->width, H = main_im->height;
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
    if ((1ull << seed_depth) > (1ull << 31))
        return fail(e, STEPTH_ERR_BAD_ARG, "min_splits %u is too deep", min_splits);
    HIP_TRY(e, hipSetDevice(e->device));
    const uint8_t *d_main = nullptr;
    size_t ms = 0;
    TRY(to_device(e, *main_im, e->stage_main, d_main, ms));
    TRY(reserve(e, e->group_out, (size_t)px * sizeof(GroupRec)));
    for (int k = 0; k < 3; ++k)
        TRY(reserve(e, e->group_list[k], (size_t)px * sizeof(GroupNode)));
    TRY(reserve(e, e->group_counters, 16));
    AdaptiveArgs a{};
    a.px = d_main;
    a.stride = ms;
    a.ch = main_im->channels;
    a.W = W;
    a.H = H;
    for (int k = 0; k < 3; ++k)
        a.prec[k] = precision[k];
    a.min_splits = min_splits;
    a.max_splits = max_splits;
    a.seed_depth = seed_depth;
    a.groups = static_cast<GroupRec *>(e->group_out.p);
    for (int k = 0; k < 3; ++k)
        a.list[k] = static_cast<GroupNode *>(e->group_list[k].p);
    a.counters = static_cast<uint32_t *>(e->group_counters.p);
    HIP_TRY(e, hipMemsetAsync(a.counters, 0, 16, e->stream));
    const uint64_t seeds = 1ull << seed_depth;
    hipLaunchKernelGGL(adaptive_seed_kernel, dim3((uint32_t)std::min<uint64_t>((seeds + 255) / 256, 4096)),
                       dim3(256), 0, e->stream, a);
    const uint32_t level_grid = (uint32_t)std::min<uint64_t>((px + 255) / 256, 2048);
    for (uint32_t d = seed_depth + 1; d <= max_splits; ++d) // three lists rotate, see the kernel
        hipLaunchKernelGGL(adaptive_level_kernel, dim3(level_grid), dim3(256), 0, e->stream, a, d);
    HIP_TRY(e, hipGetLastError());
    uint32_t n = 0;
    HIP_TRY(e, hipMemcpyAsync(&n, a.counters, 4, hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(e, hipStreamSynchronize(e->stream));
    *n_out = n;
    if (!groups)
        return STEPTH_OK; // count only; the groups stay in the engine's buffer until the next call
    if (n > capacity)
        return fail(e, STEPTH_ERR_BAD_ARG, "adaptive groups: %u groups, capacity is %zu", n, capacity);
    HIP_TRY(e, hipMemcpyAsync(groups, a.groups, (size_t)n * sizeof(GroupRec),
                              groups_memory == STEPTH_MEM_HOST ? hipMemcpyDeviceToHost
                                                               : hipMemcpyDeviceToDevice,
                              e->stream));
    HIP_TRY(e, hipStreamSynchronize(e->stream));
    return STEPTH_OK;
}
