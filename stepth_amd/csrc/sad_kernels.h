// sad_kernels.h -- windowed-SAD mode (horizontal disparity search).
//
// NOT in the reference: /root/reference/src/operations.rs holds no cost/argmin loop
// (SURVEY.md 8a-0).  This mode implements BASELINE.json north_star's literal wording and is
// checked only against the engine's own CPU statement (oracle_sad_rows):
//   cost(x,y,d) = sum_{win x win} sum_c |L(x+i,y+j)_c - R(x+i-d,y+j)_c|   (clamp-to-edge)
//   raw(x,y)    = smallest d in [0,d_max) with cost < win^2 (p0+p1+p2), else first argmin.
//
// sad_search_kernel<WIN, DC, ROWS>: one wave = 64 adjacent columns x ROWS output rows.  Lane x
// owns column x.  For a chunk of DC disparities it walks down the input rows.  A row's pixels --
// 64 + WIN-1 of the left image, 64 + DC+WIN-2 of the right one, clamp-to-edge applied to the
// column and row indices -- are fetched from the RGB(A) images ONCE per wave, one row ahead of
// the arithmetic, and handed to the lanes as RGBX dwords through LDS (until round 3 both images
// were first packed into RGBX planes with aprons, 16 % of a call, and every lane loaded its 25
// dwords itself).  Per row a lane forms the horizontal window sums H(d) with WIN chained v_sad_u8
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
    uint32_t w, h;
    uint32_t row0, row1;
    uint32_t d_max, win, thr;
    uint8_t *raw; // [frame][(row1-row0)][w]
    uint32_t *max_word; // one word per frame
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

// ROWS = output rows per wave: 16 costs (16+WIN-1)/16 input rows per output row, 8 costs more
// but gives small frames twice the waves
// One wave per workgroup (round 3): a strip whose pixels are all decided by the threshold stops
// early, and with four waves per workgroup the early ones' LDS stayed allocated until the last
// was done -- five half-empty workgroups held a CU's LDS and nothing new could start (2.0 of 4
// possible waves per SIMD resident, VALU 58 % busy; profiles/r3_sq_counters.txt).
template <int WIN, int DC, int ROWS>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(3))) void sad_search_kernel(SadArgs a)
{
    constexpr int RAD = WIN / 2;
    constexpr int kSadRows = ROWS;
    __shared__ uint32_t s_best[1][kSadRows][64]; // running min cost
    __shared__ uint32_t s_pick[1][kSadRows][64]; // bit 31: decided (under threshold); low: d
    // the input row a step works on: every pixel of it is fetched from memory ONCE per wave (one
    // row ahead, behind the previous row's arithmetic) and handed to the lanes that need it
    // through LDS -- a lane's WIN + DC+WIN-1 RGBX dwords used to be its own (merged) loads from
    // pre-packed planes, 25 dwords per lane where the wave's lanes need 151 distinct ones between
    // them, issued and waited for at the top of every row with nothing to hide the latency but the
    // other waves
    constexpr int kRowL = 64 + WIN - 1, kRowR = 64 + DC + WIN - 1;
    __shared__ uint32_t s_rowl[kRowL + 1], s_rowr[kRowR]; // (+ 1: where the lanes without a tail pixel write)
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
    // Row fetches.  The lane that stages pixel jj of a row fetches the pixel itself, one row
    // ahead of the arithmetic: the ALIGNED dword that holds its first byte and the one that holds
    // its last (the same dword again when the three bytes do not straddle two) -- loads that touch
    // nothing but dwords holding a byte of the pixel: at an image's very first and last bytes
    // that can be up to 3 bytes of the neighbouring allocation, inside the same aligned dword and
    // therefore the same page, which are never used.  v_alignbyte + mask turn the pair into the
    // RGBX dword that goes to LDS.  Clamp-to-edge is a per-lane, per-chunk column and a scalar
    // row clamp; the byte offsets only depend on the row when the pitches are not multiples of 4.
    // (Measured on the way, 4 x 4K 5x5 d_max 128, search kernel ms: pre-packed RGBX planes with
    // per-lane loads 0.694 (+ 0.142 for the pack kernel), the planes staged through LDS 0.676,
    // byte loads per pixel 0.935, the rows' raw dwords through LDS and unaligned 4-byte LDS reads
    // at a 3-byte stride 1.142, the same with aligned dword pairs 0.794, this scheme with
    // per-lane 64-bit addresses 0.775, with scalar row bases + 32-bit offsets 0.718.)
    const uint8_t *limg = a.left[f], *rimg = a.right[f];
    const uint32_t ulane = threadIdx.x & 63u;
    const int x0 = (int)bcol * 64, wmax = (int)a.w - 1, hmax = (int)a.h - 1;
    auto clampx = [&](int v) __attribute__((always_inline)) { return min(max(v, 0), wmax); };
    // left image: staged index jj holds L(x0 + jj - RAD); lane -> jj = lane, lanes 32.. also jj = lane + 32
    const uint32_t co_l = (uint32_t)clampx(x0 + lane - RAD) * a.lch;       // byte offsets in a row
    const uint32_t co_lt = (uint32_t)clampx(x0 + lane + 32 - RAD) * a.lch;
    const bool same_m = ((a.lstride | a.rstride) & 3u) == 0u; // uniform: a row's alignment is every row's
    for (int d0 = 0; d0 < (int)a.d_max; d0 += DC) {
        // right image: staged index jj holds R(x0 + jj - RAD - d0 - (DC-1)): a lane's r[k] =
        // staged[lane + k] = R(x - RAD + i - d) with k = i + (DC-1-j), d = d0 + j
        const int rshift = RAD + d0 + (DC - 1);
        const uint32_t co_r = (uint32_t)clampx(x0 + lane - rshift) * 3u;
        const uint32_t co_rt = (uint32_t)clampx(x0 + lane + 64 - rshift) * 3u;
        uint32_t cost[DC], ring[WIN][DC];
#pragma unroll
        for (int j = 0; j < DC; ++j) {
            cost[j] = 0;
#pragma unroll
            for (int s = 0; s < WIN; ++s)
                ring[s][j] = 0;
        }
        // input rows ys-RAD .. ys+nrows-1+RAD (clamped to the image)
        const int n_in = nrows + 2 * RAD;
        // staging: every lane its own column of either image, and one pixel of the columns beyond
        // the wave's 64 (lanes 0..DC+WIN-3: right image, the WIN-1 lanes from 32 on: left image;
        // the other lanes stage a pixel into a slot nobody reads: straight-line code)
        static_assert(DC + WIN - 2 <= 32 && WIN - 1 <= 32, "the tail lanes");
        const bool tail_r = ulane < (uint32_t)(DC + WIN - 2), tail_l = ulane >= 32u && ulane < 32u + (uint32_t)(WIN - 1);
        const uint32_t co_t = tail_r ? co_rt : tail_l ? co_lt : co_l;
        uint32_t *dst_t = tail_r ? &s_rowr[64u + ulane] : tail_l ? &s_rowl[32u + ulane] : &s_rowl[kRowL];
        // per pixel: dword index of its first byte and of its last, and the shift (from an aligned
        // base <= 3 bytes before the row)
        uint32_t m_l = 0, m_r = 0; // uniform: how far the rows start inside a dword
        uint32_t i0_l, i1_l, sh_l, i0_r, i1_r, sh_r, i0_t, i1_t, sh_t;
        auto place = [](uint32_t o, uint32_t &i0, uint32_t &i1, uint32_t &sh) __attribute__((always_inline)) {
            i0 = o & ~3u;              // byte offsets of the two dwords (32-bit: the loads below are
            i1 = i0 + ((o & 2u) << 1); // scalar base + vector offset); bytes o..o+2 reach into the
            sh = o;                    // next dword iff o & 3 >= 2
        };
        auto set_m = [&](uint32_t ml, uint32_t mr) __attribute__((always_inline)) {
            m_l = ml;
            m_r = mr;
            place(ml + co_l, i0_l, i1_l, sh_l);
            place(mr + co_r, i0_r, i1_r, sh_r);
            place((tail_r ? mr : ml) + co_t, i0_t, i1_t, sh_t);
        };
        uint32_t pl0 = 0, pl1 = 0, pr0 = 0, pr1 = 0, pt0 = 0, pt1 = 0; // the row in flight
        uint32_t fl, fr, ft;                                             // its shifts
        auto dword_at = [](const uint8_t *base, uint32_t off) __attribute__((always_inline)) -> uint32_t {
            return *reinterpret_cast<const uint32_t *>(base + off);
        };
        // the row pointers step from row to row (one scalar add each where the clamp lets the row
        // advance) instead of being multiplied out afresh: the scalar unit is the busiest one here
        int yy_cur = min(max(ys - RAD, 0), hmax);
        const uint8_t *la = limg + (size_t)yy_cur * a.lstride, *ra = rimg + (size_t)yy_cur * a.rstride;
        auto fetch = [&](int t) __attribute__((always_inline)) {
            const int yy = min(max(ys - RAD + t, 0), hmax); // uniform; yy_cur or yy_cur + 1
            if (yy != yy_cur) {
                la += a.lstride;
                ra += a.rstride;
                yy_cur = yy;
            }
            if (!same_m)
                set_m((uint32_t)reinterpret_cast<uintptr_t>(la) & 3u, (uint32_t)reinterpret_cast<uintptr_t>(ra) & 3u);
            const uint8_t *lq = la - m_l, *rq = ra - m_r; // aligned, uniform (as `la & ~3` the loads lose their scalar base: + 10 %)
            pl0 = dword_at(lq, i0_l);
            pl1 = dword_at(lq, i1_l);
            pr0 = dword_at(rq, i0_r);
            pr1 = dword_at(rq, i1_r);
            if (tail_r) { // (measured equal as two masked pairs)
                pt0 = dword_at(rq, i0_t);
                pt1 = dword_at(rq, i1_t);
            } else {
                pt0 = dword_at(lq, i0_t);
                pt1 = dword_at(lq, i1_t);
            }
            fl = sh_l;
            fr = sh_r;
            ft = sh_t;
        };
        set_m((uint32_t)reinterpret_cast<uintptr_t>(limg) & 3u, (uint32_t)reinterpret_cast<uintptr_t>(rimg) & 3u);
        fetch(0);
        for (int t0 = 0; t0 < n_in; t0 += WIN) {
#pragma unroll
            for (int s = 0; s < WIN; ++s) {
                const int t = t0 + s;
                if (t < n_in) { // uniform
                    // (the previous row's reads are done: LDS operations of a wave complete in
                    // program order; the barriers keep the compiler from moving a lane's reads
                    // of OTHER lanes' pixels across its own writes)
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    s_rowl[ulane] = __builtin_amdgcn_alignbyte(pl1, pl0, fl) & 0xFFFFFFu;
                    s_rowr[ulane] = __builtin_amdgcn_alignbyte(pr1, pr0, fr) & 0xFFFFFFu;
                    *dst_t = __builtin_amdgcn_alignbyte(pt1, pt0, ft) & 0xFFFFFFu;
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    if (t + 1 < n_in)
                        fetch(t + 1); // lands while this row is worked on
                    uint32_t l[WIN], r[DC + WIN - 1];
#pragma unroll
                    for (int i = 0; i < WIN; ++i)
                        l[i] = s_rowl[ulane + i];
#pragma unroll
                    for (int k = 0; k < DC + WIN - 1; ++k)
                        r[k] = s_rowr[ulane + k];
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
