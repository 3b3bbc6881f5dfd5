// sad_kernels.h -- windowed-SAD mode (horizontal disparity search).
//
// NOT in the reference: /root/reference/src/operations.rs holds no cost/argmin loop
// (SURVEY.md 8a-0).  This mode implements BASELINE.json north_star's literal wording and is
// checked only against the engine's own CPU statement (oracle_sad_rows):
//   cost(x,y,d) = sum_{win x win} sum_c |L(x+i,y+j)_c - R(x+i-d,y+j)_c|   (clamp-to-edge)
//   raw(x,y)    = smallest d in [0,d_max) with cost < win^2 (p0+p1+p2), else first argmin.
//
// Layout: both images are first packed into RGBX dword planes with an apron that bakes the
// clamp-to-edge rule in (left apron rad [+ d_max-1 for R], right apron rad + 64, rad rows
// above and below), so the search kernel never clamps an index.
//
// sad_search_kernel<WIN, DC, ROWS>: one wave = 64 adjacent columns x ROWS output rows.  Lane x
// owns column x.  For a chunk of DC disparities it walks down the input rows; per row it
// reads the WIN left cells and the DC+WIN-1 right cells its column needs (consecutive lanes
// read consecutive dwords), forms the horizontal window sums H(d) with WIN chained v_sad_u8
// per disparity -- neighbouring disparities reuse the same registers shifted by one -- and
// keeps the vertical window as cost(d) += H_new(d) - H_old(d) with the last WIN rows of H in
// a register ring (the row loop is unrolled WIN times so the ring is indexed statically).
// The per-pixel decision (first d under the threshold, else first argmin) is carried from
// chunk to chunk in LDS; a strip whose pixels are all decided by the threshold stops early.
// No cross-lane traffic apart from that one ballot per chunk.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace stepth {

struct SadArgs {
    // one image pair per frame of a batch (blockIdx.z in the search kernels)
    const uint8_t *left[kMaxBatch];  // main images, lch bytes per pixel
    size_t lstride;
    uint32_t lch;
    const uint8_t *right[kMaxBatch]; // additional images, RGB8
    size_t rstride;
    uint32_t n_frames;
    size_t raw_stride;                // bytes between the frames' raw maps
    size_t lplane_stride, rplane_stride; // dwords between the frames' packed planes
    uint32_t w, h;
    uint32_t row0, row1;
    uint32_t d_max, win, thr;
    uint8_t *raw; // [frame][(row1-row0)][w]
    uint32_t *max_word; // one word per frame
    // packed planes (sad_pack_kernel)
    uint32_t *lp, *rp;
    uint32_t lpitch, rpitch; // dwords
    uint32_t cols, strips;   // sad_search_kernel: waves per strip, strips per frame (launch_sad)
};

__device__ __forceinline__ uint32_t load_rgbx(const uint8_t *img, size_t stride, uint32_t ch,
                                              int x, int y, int w, int h)
{
    x = x < 0 ? 0 : (x >= w ? w - 1 : x);
    y = y < 0 ? 0 : (y >= h ? h - 1 : y);
    const uint8_t *p = img + (size_t)y * stride + (size_t)x * ch;
    return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16);
}

// plane[yy][xx] = RGBX(img[clamp(yy - rad)][clamp(xx - apron_l)])
struct SadPackArgs {
    const uint8_t *img[kMaxBatch]; // blockIdx.y = frame
    size_t plane_stride;           // dwords between the frames' planes
    size_t stride;
    uint32_t ch, w, h;
    int rad, apron_l, row0; // plane row r holds image row row0 - rad + r (clamped)
    uint32_t pitch, rows;
    uint32_t *plane;
};

__global__ __launch_bounds__(256) void sad_pack_kernel(SadPackArgs a)
{
    const size_t total = (size_t)a.pitch * a.rows;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (size_t)gridDim.x * blockDim.x) {
        const int yy = (int)(i / a.pitch), xx = (int)(i % a.pitch);
        a.plane[(size_t)blockIdx.y * a.plane_stride + i] =
            load_rgbx(a.img[blockIdx.y], a.stride, a.ch, xx - a.apron_l, yy - a.rad + a.row0, a.w, a.h);
    }
}

constexpr int kSadPadD = 16; // extra left apron of the R plane: the last chunk of DC
                             // disparities may run past d_max (those costs are ignored)

// ROWS = output rows per wave: 16 costs (16+WIN-1)/16 input rows per output row, 8 costs more
// but gives small frames twice the waves
// One wave per workgroup (round 3): a strip whose pixels are all decided by the threshold stops
// early, and with four waves per workgroup the early ones' LDS stayed allocated until the last
// was done -- five half-empty workgroups held a CU's LDS and nothing new could start (2.0 of 4
// possible waves per SIMD resident, VALU 58 % busy; profiles/r3_sq_counters.txt).
template <int WIN, int DC, int ROWS>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4))) void sad_search_kernel(SadArgs a)
{
    constexpr int RAD = WIN / 2;
    constexpr int kSadRows = ROWS;
    __shared__ uint32_t s_best[1][kSadRows][64]; // running min cost
    __shared__ uint32_t s_pick[1][kSadRows][64]; // bit 31: decided (under threshold); low: d
    const int lane = threadIdx.x & 63, wv = 0;
    // Workgroup ids go round-robin over the 8 XCDs, each with its own L2.  A wave reads the
    // right image's columns [x - d_max - RAD, x + 64 + RAD): two thirds of what its left
    // neighbour reads at d_max 128 -- so the waves of one strip belong on ONE XCD, one after the
    // other in its dispatch order.  XCD k takes the strips k, k + 8, ... (frames stacked).
    // (With blockIdx = (column, strip, frame) the neighbours sat on 8 different XCDs and every L2
    // fetched the whole band for itself: 1.62 GB per 4 x 4K launch, 6 x the planes.)
    const uint32_t xcd = blockIdx.x & 7u, j = blockIdx.x >> 3;
    const uint32_t g = (j / a.cols) * 8u + xcd, bcol = j % a.cols; // strip of the stacked frames, its wave
    if (g >= a.strips * a.n_frames)
        return;
    const uint32_t f = g / a.strips; // frame
    const int x = (int)bcol * 64 + lane;                                // output column
    const int ys = (int)a.row0 + (int)(g - f * a.strips) * kSadRows;    // first output row
    const int nrows = min(kSadRows, (int)a.row1 - ys);
#pragma unroll
    for (int r = 0; r < kSadRows; ++r) {
        s_best[wv][r][lane] = 0xffffffffu;
        s_pick[wv][r][lane] = 0u;
    }
    // plane coordinates: image (x', y') sits at [y' - row0 + RAD][x' + apron_l] (the planes hold
    // only the rows the strip [row0, row1) reads)
    const uint32_t *lcol = a.lp + (size_t)f * a.lplane_stride + x; // + i  -> L(x - RAD + i)
    for (int d0 = 0; d0 < (int)a.d_max; d0 += DC) {
        // + k -> R(x - RAD + i - d) with k = i + (DC-1-j), d = d0 + j
        const uint32_t *rcol = a.rp + (size_t)f * a.rplane_stride + x + kSadPadD + ((int)a.d_max - 1) -
                               d0 - (DC - 1);
        uint32_t cost[DC], ring[WIN][DC];
#pragma unroll
        for (int j = 0; j < DC; ++j) {
            cost[j] = 0;
#pragma unroll
            for (int s = 0; s < WIN; ++s)
                ring[s][j] = 0;
        }
        // input rows ys-RAD .. ys+nrows-1+RAD, i.e. plane rows ys-row0 .. ys-row0+nrows-1+2*RAD
        const int n_in = nrows + 2 * RAD;
        for (int t0 = 0; t0 < n_in; t0 += WIN) {
#pragma unroll
            for (int s = 0; s < WIN; ++s) {
                const int t = t0 + s;
                if (t < n_in) { // uniform
                    const uint32_t *lr = lcol + (size_t)(ys - (int)a.row0 + t) * a.lpitch;
                    const uint32_t *rr = rcol + (size_t)(ys - (int)a.row0 + t) * a.rpitch;
                    uint32_t l[WIN], r[DC + WIN - 1];
#pragma unroll
                    for (int i = 0; i < WIN; ++i)
                        l[i] = lr[i];
#pragma unroll
                    for (int k = 0; k < DC + WIN - 1; ++k)
                        r[k] = rr[k];
#pragma unroll
                    for (int j = 0; j < DC; ++j) {
                        // the sad chain accumulates straight into the sum of the horizontal window
                        // sums of ALL rows so far (< 2^16 per row, a strip has < 2^7 rows), which
                        // is what the previous row left in its ring slot; ring[s][j] = that sum
                        // WIN rows ago
                        uint32_t acc = ring[(s + WIN - 1) % WIN][j];
#pragma unroll
                        for (int i = 0; i < WIN; ++i)
                            acc = __builtin_amdgcn_sad_u8(l[i], r[i + DC - 1 - j], acc);
                        cost[j] = acc - ring[s][j];
                        ring[s][j] = acc;
                    }
                    const int orow = t - 2 * RAD; // output row (relative) completed by this input row
                    if (orow >= 0) {
                        uint32_t best = s_best[wv][orow][lane], pick = s_pick[wv][orow][lane];
                        if (d0 + DC <= (int)a.d_max) {
                            // the chunk's DC disparities at once: the first one under the threshold
                            // decides; else the first minimum (costs < 2^16, so cost << 4 | j orders
                            // by cost, then by disparity) if it beats the best so far
                            static_assert(DC <= 16, "4 bits of disparity in the key");
                            uint32_t mk = 0xffffffffu;
#pragma unroll
                            for (int j = DC - 1; j >= 0; --j)
                                mk = min(mk, (cost[j] << 4) | (uint32_t)j);
                            const bool open = (int)pick >= 0;
                            // the smallest cost tells whether ANY disparity is under the threshold;
                            // which one is first is only worked out in the chunks where some pixel
                            // of the wave needs it (neighbouring pixels decide in the same chunks)
                            if (__builtin_amdgcn_ballot_w64(open && (mk >> 4) < a.thr)) {
                                uint32_t under = 0;
#pragma unroll
                                for (int j = DC - 1; j >= 0; --j)
                                    under = under + under + (cost[j] < a.thr ? 1u : 0u);
                                if (open && under)
                                    pick = 0x80000000u | (uint32_t)(d0 + __builtin_ctz(under));
                            }
                            if ((int)pick >= 0 && (mk >> 4) < best) {
                                best = mk >> 4;
                                pick = (uint32_t)d0 + (mk & 15u);
                            }
                        } else { // the last chunk runs past d_max: disparity by disparity
#pragma unroll
                            for (int j = 0; j < DC; ++j) {
                                const uint32_t d = (uint32_t)(d0 + j), c = cost[j];
                                const bool live = (int)pick >= 0 && d < a.d_max;
                                if (live && c < a.thr)
                                    pick = 0x80000000u | d; // first disparity under the threshold
                                else if (live && c < best) {
                                    best = c;
                                    pick = d;
                                }
                            }
                        }
                        s_best[wv][orow][lane] = best;
                        s_pick[wv][orow][lane] = pick;
                    }
                }
            }
        }
        // every pixel of the strip already decided by the threshold: later disparities
        // cannot change anything
        bool open = false;
        for (int r = 0; r < nrows; ++r)
            open |= (int)s_pick[wv][r][lane] >= 0;
        if (__builtin_amdgcn_ballot_w64(open) == 0)
            break;
    }
    uint32_t m = 0;
    if (x < (int)a.w) {
        for (int r = 0; r < nrows; ++r) {
            const uint32_t d = s_pick[wv][r][lane] & 0xffffu;
            a.raw[(size_t)f * a.raw_stride + (size_t)(ys - (int)a.row0 + r) * a.w + x] = (uint8_t)d;
            m = max(m, d & 0xffu);
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
        m = max(m, (uint32_t)__shfl_xor((int)m, o, 64));
    // (a plain read first: most waves cannot raise the max, and atomics on one line serialise)
    if (lane == 0 && m > __hip_atomic_load(a.max_word + f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
        atomicMax(a.max_word + f, m);
}

// ---- generic fallback (any odd window up to 15): one thread per pixel, windows in LDS ----
constexpr int kSadTileW = 64, kSadTileH = 4;

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
        Lt[i] = load_rgbx(a.left[blockIdx.z], a.lstride, a.lch, x0 - rad + c, y0 - rad + r, a.w, a.h);
    }
    for (int i = threadIdx.x; i < rw * th; i += blockDim.x) {
        const int r = i / rw, c = i % rw;
        Rt[i] = load_rgbx(a.right[blockIdx.z], a.rstride, 3, x0 - rad - ((int)a.d_max - 1) + c,
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
        a.raw[(size_t)blockIdx.z * a.raw_stride + (size_t)(y - (int)a.row0) * a.w + x] = (uint8_t)best_d;
        m = best_d & 0xffu;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
        m = max(m, (uint32_t)__shfl_xor((int)m, o, 64));
    if ((threadIdx.x & 63) == 0 &&
        m > __hip_atomic_load(a.max_word + blockIdx.z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
        atomicMax(a.max_word + blockIdx.z, m);
}

inline bool sad_has_fast_path(uint32_t win) { return win == 3 || win == 5 || win == 7 || win == 9; }

inline void sad_plane_dims(const SadArgs &a, bool right, uint32_t &pitch, uint32_t &rows,
                           int &apron_l)
{
    const int rad = (int)a.win / 2;
    apron_l = rad + (right ? (int)a.d_max - 1 + kSadPadD : 0);
    pitch = (uint32_t)(apron_l + (int)((a.w + 63u) / 64u * 64u) + rad + 64);
    rows = (a.row1 - a.row0) + 2u * (uint32_t)rad; // only what the strip [row0,row1) reads
}

// pack both planes; a.lp/a.rp/pitches must already be set
inline int launch_sad_pack(const SadArgs &a, hipStream_t stream)
{
    for (int side = 0; side < 2; ++side) {
        SadPackArgs p{};
        for (uint32_t f = 0; f < a.n_frames; ++f)
            p.img[f] = side ? a.right[f] : a.left[f];
        p.plane_stride = side ? a.rplane_stride : a.lplane_stride;
        p.stride = side ? a.rstride : a.lstride;
        p.ch = side ? 3u : a.lch;
        p.w = a.w;
        p.h = a.h;
        p.rad = (int)a.win / 2;
        p.row0 = (int)a.row0;
        uint32_t pitch, rows;
        sad_plane_dims(a, side != 0, pitch, rows, p.apron_l);
        p.pitch = pitch;
        p.rows = rows;
        p.plane = side ? a.rp : a.lp;
        const size_t total = (size_t)pitch * rows;
        const uint32_t blocks = (uint32_t)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
        hipLaunchKernelGGL(sad_pack_kernel, dim3(blocks, a.n_frames), dim3(256), 0, stream, p);
    }
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

inline int launch_sad(const SadArgs &a_in, hipStream_t stream)
{
    SadArgs a = a_in;
    if (sad_has_fast_path(a.win)) {
        const uint32_t cols = (a.w + 63) / 64, rows = a.row1 - a.row0; // one wave (64 columns) per workgroup
        // enough waves to fill 256 CUs x 12 wave slots with 16-row strips?
        const bool tall = (size_t)cols * ((rows + 15) / 16) * a.n_frames >= 6144;
        a.cols = cols;
        a.strips = tall ? (rows + 15) / 16 : (rows + 7) / 8;
        // strips of all frames dealt to the 8 XCDs (see the kernel)
        const dim3 grid((a.strips * a.n_frames + 7u) / 8u * 8u * cols);
#define STEPTH_SAD_LAUNCH(W, D)                                                               \
    do {                                                                                      \
        if (tall)                                                                             \
            hipLaunchKernelGGL((sad_search_kernel<W, D, 16>), grid, dim3(64), 0, stream, a);  \
        else                                                                                  \
            hipLaunchKernelGGL((sad_search_kernel<W, D, 8>), grid, dim3(64), 0, stream, a);   \
    } while (0)
        switch (a.win) {
        case 3: STEPTH_SAD_LAUNCH(3, 16); break;
        case 5: STEPTH_SAD_LAUNCH(5, 16); break;
        case 7: STEPTH_SAD_LAUNCH(7, 8); break;
        default: STEPTH_SAD_LAUNCH(9, 8); break;
        }
#undef STEPTH_SAD_LAUNCH
        return hipGetLastError() == hipSuccess ? 0 : -3;
    }
    const int rad = (int)a.win / 2;
    const int lw = kSadTileW + 2 * rad, rw = lw + (int)a.d_max - 1, th = kSadTileH + 2 * rad;
    const size_t lds = (size_t)(lw + rw) * th * 4;
    dim3 grid((a.w + kSadTileW - 1) / kSadTileW, (a.row1 - a.row0 + kSadTileH - 1) / kSadTileH,
              a.n_frames);
    hipLaunchKernelGGL(sad_match_kernel, grid, dim3(256), lds, stream, a);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

} // namespace stepth
