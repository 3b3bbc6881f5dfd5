// sad_kernels.h -- windowed-SAD mode (horizontal disparity search).
//
// NOT in the reference: /root/reference/src/operations.rs holds no cost/argmin loop
// (SURVEY.md 8a-0).  This mode implements BASELINE.json north_star's literal wording and is
// checked only against the engine's own CPU statement (oracle_sad_rows):
//   cost(x,y,d) = sum_{win x win} sum_c |L(x+i,y+j)_c - R(x+i-d,y+j)_c|   (clamp-to-edge)
//   raw(x,y)    = smallest d in [0,d_max) with cost < win^2 (p0+p1+p2), else first argmin.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace stepth {

struct SadArgs {
    const uint8_t *left;  // main image, lch bytes per pixel
    size_t lstride;
    uint32_t lch;
    const uint8_t *right; // additional image, RGB8
    size_t rstride;
    uint32_t w, h;
    uint32_t row0, row1;
    uint32_t d_max, win, thr;
    uint8_t *raw; // [(row1-row0)][w]
    uint32_t *max_word;
};

constexpr int kSadTileW = 64, kSadTileH = 4;

__device__ __forceinline__ uint32_t load_rgbx(const uint8_t *img, size_t stride, uint32_t ch,
                                              int x, int y, int w, int h)
{
    x = x < 0 ? 0 : (x >= w ? w - 1 : x);
    y = y < 0 ? 0 : (y >= h ? h - 1 : y);
    const uint8_t *p = img + (size_t)y * stride + (size_t)x * ch;
    return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16);
}

// One thread per output pixel, tile 64 x 4; L and R windows staged in LDS as RGBX dwords so
// that v_sad_u8 does one pixel's three channel differences (plus a zero byte) per issue.
__global__ __launch_bounds__(256) void sad_match_kernel(SadArgs a)
{
    extern __shared__ uint32_t lds[];
    const int rad = (int)a.win / 2;
    const int lw = kSadTileW + 2 * rad;            // L tile width
    const int rw = lw + (int)a.d_max - 1;          // R tile width
    const int th = kSadTileH + 2 * rad;            // tile rows
    uint32_t *Lt = lds, *Rt = lds + lw * th;
    const int x0 = (int)blockIdx.x * kSadTileW;
    const int y0 = (int)a.row0 + (int)blockIdx.y * kSadTileH;

    for (int i = threadIdx.x; i < lw * th; i += blockDim.x) {
        const int r = i / lw, c = i % lw;
        Lt[i] = load_rgbx(a.left, a.lstride, a.lch, x0 - rad + c, y0 - rad + r, a.w, a.h);
    }
    for (int i = threadIdx.x; i < rw * th; i += blockDim.x) {
        const int r = i / rw, c = i % rw;
        Rt[i] = load_rgbx(a.right, a.rstride, 3, x0 - rad - ((int)a.d_max - 1) + c,
                          y0 - rad + r, a.w, a.h);
    }
    __syncthreads();

    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int x = x0 + tx, y = y0 + ty;
    uint32_t best = 0xffffffffu, best_d = 0;
    bool done = false;
    for (uint32_t d = 0; d < a.d_max; ++d) {
        if (__ballot(!done) == 0)
            break;
        uint32_t cost = 0;
        const int roff = (int)a.d_max - 1 - (int)d;
        for (int j = 0; j < (int)a.win; ++j) {
            const uint32_t *lr = Lt + (ty + j) * lw + tx;
            const uint32_t *rr = Rt + (ty + j) * rw + tx + roff;
            for (int i = 0; i < (int)a.win; ++i)
                cost = __builtin_amdgcn_sad_u8(lr[i], rr[i], cost);
        }
        if (!done) {
            if (cost < a.thr) {
                best_d = d;
                done = true;
            } else if (cost < best) {
                best = cost;
                best_d = d;
            }
        }
    }
    uint32_t m = 0;
    if (x < (int)a.w && y < (int)a.row1) {
        a.raw[(size_t)(y - (int)a.row0) * a.w + x] = (uint8_t)best_d;
        m = best_d & 0xffu;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
        m = max(m, (uint32_t)__shfl_xor((int)m, o, 64));
    if ((threadIdx.x & 63) == 0 && m)
        atomicMax(a.max_word, m);
}

inline int launch_sad(const SadArgs &a, hipStream_t stream)
{
    const int rad = (int)a.win / 2;
    const int lw = kSadTileW + 2 * rad, rw = lw + (int)a.d_max - 1, th = kSadTileH + 2 * rad;
    const size_t lds = (size_t)(lw + rw) * th * 4;
    dim3 grid((a.w + kSadTileW - 1) / kSadTileW, (a.row1 - a.row0 + kSadTileH - 1) / kSadTileH);
    hipLaunchKernelGGL(sad_match_kernel, grid, dim3(256), lds, stream, a);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

} // namespace stepth
