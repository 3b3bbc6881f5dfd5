// cabi_check.cpp -- the whole C-ABI of include/stepth_amd.h driven from plain C++ (no Python,
// no torch): host-memory batch pipeline vs single calls, adaptive groups kept on the device and
// fed to the list mode, the depth-map consumers, SAD and tiled modes, the Gaussian tail,
// page-locked host memory.  Built by the Makefile as ../stepth_cabi_check; run by
// tests/test_cpp_host_mirror.py on the GPU.  (Also the program used for the host
// AddressSanitizer pass: hipcc -fsanitize=address -fno-gpu-sanitize)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../include/stepth_amd.h"
#define CK(x) do { int _r = (x); if (_r != 0) { printf("FAIL %s -> %d (%s)\n", #x, _r, stepth_last_error(e)); return 1; } } while (0)
int main() {
    stepth_engine *e = nullptr;
    if (stepth_engine_create(-1, &e)) { printf("no engine\n"); return 1; }
    const uint32_t W = 333, H = 201, N = 21;
    std::vector<std::vector<uint8_t>> mains(N), adds(N), outs(N);
    std::vector<stepth_image> mi(N), ai(N), di(N);
    uint32_t x = 12345;
    for (uint32_t f = 0; f < N; ++f) {
        mains[f].resize((size_t)W * H * 3); adds[f].resize((size_t)W * H * 3); outs[f].assign((size_t)W * H, 0);
        for (auto &b : mains[f]) { x = x * 1664525u + 1013904223u; b = (uint8_t)((x >> 24) / 64 * 64); }
        for (size_t i = 0; i < adds[f].size(); ++i) adds[f][i] = mains[f][(i + 3 * 17) % adds[f].size()];
        mi[f] = stepth_image{mains[f].data(), W, H, 3, STEPTH_MEM_HOST, (size_t)W * 3};
        ai[f] = stepth_image{adds[f].data(), W, H, 3, STEPTH_MEM_HOST, (size_t)W * 3};
        di[f] = stepth_image{outs[f].data(), W, H, 1, STEPTH_MEM_HOST, (size_t)W};
    }
    stepth_params p{}; p.struct_size = sizeof p; p.mode = STEPTH_MODE_RING_FIRST_HIT;
    p.precision[0] = p.precision[1] = p.precision[2] = 36; p.max_steps = 100; p.tile_w = p.tile_h = 1;
    std::vector<uint32_t> mx(N);
    CK(stepth_depth_from_additional_batch(e, N, mi.data(), ai.data(), &p, di.data(), mx.data()));
    // single-frame result must equal the batch's
    std::vector<uint8_t> one((size_t)W * H); stepth_image oi{one.data(), W, H, 1, STEPTH_MEM_HOST, (size_t)W};
    uint32_t m1 = 0;
    CK(stepth_depth_from_additional(e, &mi[5], &ai[5], &p, nullptr, 0, STEPTH_MEM_HOST, &oi, &m1));
    if (m1 != mx[5] || one != outs[5]) { printf("batch != single\n"); return 1; }
    // adaptive groups on the device, list mode
    stepth_group *dg = nullptr;
    if (hipMalloc(&dg, (size_t)W * H * sizeof(stepth_group)) != hipSuccess) { printf("hipMalloc\n"); return 1; }
    size_t ng = 0;
    CK(stepth_adaptive_groups(e, &mi[0], p.precision, 16, 0, dg, (size_t)W * H, STEPTH_MEM_DEVICE, &ng));
    CK(stepth_depth_from_additional(e, &mi[0], &ai[0], &p, dg, ng, STEPTH_MEM_DEVICE, &oi, &m1));
    const std::vector<uint8_t> list_depth = one; // frame 0 through the single-frame list mode
    // the same flow batched: three device frames, groups never leave the device
    {
        const uint32_t B = 3;
        uint8_t *dm[B], *da[B], *dd[B];
        stepth_group *gl[B];
        stepth_image bm[B], ba[B], bd[B];
        for (uint32_t f = 0; f < B; ++f) {
            if (hipMalloc(&dm[f], (size_t)W * H * 3) != hipSuccess || hipMalloc(&da[f], (size_t)W * H * 3) != hipSuccess ||
                hipMalloc(&dd[f], (size_t)W * H) != hipSuccess || hipMalloc(&gl[f], (size_t)W * H * sizeof(stepth_group)) != hipSuccess) {
                printf("hipMalloc\n");
                return 1;
            }
            (void)hipMemcpy(dm[f], mains[f].data(), (size_t)W * H * 3, hipMemcpyHostToDevice);
            (void)hipMemcpy(da[f], adds[f].data(), (size_t)W * H * 3, hipMemcpyHostToDevice);
            bm[f] = stepth_image{dm[f], W, H, 3, STEPTH_MEM_DEVICE, (size_t)W * 3};
            ba[f] = stepth_image{da[f], W, H, 3, STEPTH_MEM_DEVICE, (size_t)W * 3};
            bd[f] = stepth_image{dd[f], W, H, 1, STEPTH_MEM_DEVICE, (size_t)W};
        }
        size_t counts[B];
        uint32_t bmx[B];
        CK(stepth_adaptive_groups_batch(e, B, bm, p.precision, 16, 0, gl, (size_t)W * H, counts));
        CK(stepth_depth_from_groups_batch(e, B, bm, ba, &p, gl, counts, bd, bmx));
        std::vector<uint8_t> back((size_t)W * H);
        (void)hipMemcpy(back.data(), dd[0], back.size(), hipMemcpyDeviceToHost);
        if (counts[0] != ng || bmx[0] != m1 || back != list_depth) { printf("batched group flow != single\n"); return 1; }
        // the two calls in one (the counts come back behind the pack kernels): same bytes
        size_t counts2[B];
        CK(stepth_adaptive_depth_batch(e, B, bm, ba, &p, 16, 0, bd, bmx, counts2));
        (void)hipMemcpy(back.data(), dd[0], back.size(), hipMemcpyDeviceToHost);
        if (counts2[0] != ng || bmx[0] != m1 || back != list_depth) { printf("one-call adaptive flow != single\n"); return 1; }
        // one group per pixel with STEPTH_FLAG_HEAVY_FIRST (pooled-probe variant): same bytes as without
        {
            stepth_params hp = p; hp.flags |= STEPTH_FLAG_HEAVY_FIRST;
            uint32_t m_a[B], m_b[B];
            CK(stepth_depth_from_additional_batch(e, B, bm, ba, &p, bd, m_a));
            std::vector<uint8_t> plain((size_t)W * H);
            (void)hipMemcpy(plain.data(), dd[2], plain.size(), hipMemcpyDeviceToHost);
            CK(stepth_depth_from_additional_batch(e, B, bm, ba, &hp, bd, m_b));
            (void)hipMemcpy(back.data(), dd[2], back.size(), hipMemcpyDeviceToHost);
            if (m_a[2] != m_b[2] || back != plain) { printf("heavy-first != image order\n"); return 1; }
        }
        // uniform 4x4 tiles, batched == single
        stepth_params tp = p; tp.tile_w = tp.tile_h = 4;
        CK(stepth_depth_from_additional_batch(e, B, bm, ba, &tp, bd, bmx));
        (void)hipMemcpy(back.data(), dd[1], back.size(), hipMemcpyDeviceToHost);
        CK(stepth_depth_from_additional(e, &mi[1], &ai[1], &tp, nullptr, 0, STEPTH_MEM_HOST, &oi, &m1));
        if (bmx[1] != m1 || back != one) { printf("batched tiles != single\n"); return 1; }
        for (uint32_t f = 0; f < B; ++f) { (void)hipFree(dm[f]); (void)hipFree(da[f]); (void)hipFree(dd[f]); (void)hipFree(gl[f]); }
    }
    std::vector<stepth_group> hg((size_t)W * H);
    CK(stepth_adaptive_groups(e, &mi[0], p.precision, 3, 9, hg.data(), hg.size(), STEPTH_MEM_HOST, &ng));
    // consumers
    stepth_zone z[256]; uint32_t nz = 0;
    CK(stepth_invert_depth(e, &di[0], 0));
    CK(stepth_depth_split(e, &di[0], 4, z, 256, &nz));
    std::vector<uint8_t> mask((size_t)W * H); stepth_image mk{mask.data(), W, H, 1, STEPTH_MEM_HOST, (size_t)W};
    CK(stepth_select_foreground(e, &di[0], &mk, &z[0]));
    CK(stepth_slice(e, &di[0], &z[0], &mk, 0));
    uint32_t counts[256];
    CK(stepth_depth_histogram(e, &di[0], counts, STEPTH_MEM_HOST, 0));
    CK(stepth_zones_from_histogram(e, counts, STEPTH_MEM_HOST, 3, z, 256, &nz));
    std::vector<uint8_t> rgba((size_t)W * H * 4, 200); stepth_image ri{rgba.data(), W, H, 4, STEPTH_MEM_HOST, (size_t)W * 4};
    CK(stepth_apply_mask(e, &ri, &mk, 0));
    CK(stepth_gaussian_same_size(e, &di[0], &oi, 0));
    // SAD mode and tiles
    p.mode = STEPTH_MODE_SAD_WINDOW; p.d_max = 32; p.win = 5;
    CK(stepth_depth_from_additional(e, &mi[1], &ai[1], &p, nullptr, 0, STEPTH_MEM_HOST, &oi, &m1));
    p.mode = STEPTH_MODE_RING_FIRST_HIT; p.tile_w = 4; p.tile_h = 3;
    CK(stepth_depth_from_additional(e, &mi[1], &ai[1], &p, nullptr, 0, STEPTH_MEM_HOST, &oi, &m1));
    // the exception guard (no device needed)
    if (stepth_selftest_exception_guard(e, 0) != STEPTH_ERR_OUT_OF_MEMORY || stepth_selftest_exception_guard(e, 1) != STEPTH_ERR_HIP ||
        stepth_selftest_exception_guard(e, 2) != STEPTH_ERR_HIP || stepth_selftest_exception_guard(e, 9) != STEPTH_OK) {
        printf("exception guard\n");
        return 1;
    }
    void *pin = nullptr; if (stepth_host_alloc(&pin, 1 << 20)) { printf("host_alloc\n"); return 1; } stepth_host_free(pin);
    (void)hipFree(dg);
    stepth_engine_destroy(e);
    printf("CABI-CHECK-OK frames %u groups %zu zones %u\n", N, ng, nz);
    return 0;
}
