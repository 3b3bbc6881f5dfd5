/*
 * stepth_oracle.c -- CPU restatement of the reference's hot path.  TEST INFRASTRUCTURE ONLY
 * (see stepth_oracle.h for the rules and the parity status).  Every function cites the
 * reference lines it follows, relative to /root/reference.
 *
 * Deliberately literal: the loops below keep the reference's iteration order, duplicate
 * visits (ring 0 is tested 4 times, corners twice) and early exits, so that the HIP path is
 * checked against the reference's *behaviour*, not against a cleverer reformulation.
 */
#include "stepth_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

/* src/helpers.rs:3-7 */
uint32_t oracle_distance_dot_dot(uint32_t fx, uint32_t fy, uint32_t sx, uint32_t sy)
{
    int64_t x1 = (int64_t)fx, y1 = (int64_t)fy;
    int64_t x2 = (int64_t)sx, y2 = (int64_t)sy;
    int64_t dx = x1 - x2, dy = y1 - y2;
    double d = sqrt((double)(dx * dx + dy * dy)); /* IEEE sqrt is correctly rounded */
    return (uint32_t)d;                           /* `as u32` truncates toward zero */
}

/* disage::pixels::PixelOpps for [u8;3] as used at src/helpers.rs:35 (source absent; pinned
 * empirically): substract = per-channel |a-b|, lt = ALL channels strictly <. */
static inline int pixel_sub_lt(const uint8_t *v, const uint8_t *what, const uint8_t *prec)
{
    for (int c = 0; c < 3; ++c) {
        int d = (int)v[c] - (int)what[c];
        if (d < 0)
            d = -d;
        if (!(d < (int)prec[c]))
            return 0;
    }
    return 1;
}

/* src/helpers.rs:9-54 */
int oracle_distance_dot_array(const uint8_t what[3], const uint8_t *array, uint32_t aw,
                              uint32_t ah, size_t stride, uint32_t from_x, uint32_t from_y,
                              uint32_t max, const uint8_t precision[3], uint32_t *dist,
                              uint32_t *pos_x, uint32_t *pos_y, uint64_t *probes,
                              uint64_t *cells)
{
    const int64_t x = (int64_t)from_x; /* helpers.rs:16-18 */
    const int64_t y = (int64_t)from_y;
    uint64_t n_probe = 0, n_cell = 0;
    int found = 0;

    for (int64_t step = 0; step < (int64_t)max; ++step) { /* helpers.rs:26 */
        int still_in_bounds = 0;                          /* helpers.rs:27 */
        /* helpers.rs:28: [(y, x, true), (x, y, false)] */
        for (int pass = 0; pass < 2 && !found; ++pass) {
            const int64_t main_c = pass == 0 ? y : x;
            const int64_t sub_c = pass == 0 ? x : y;
            const int order = pass == 0;
            /* helpers.rs:29: [main + current_step, main - current_step] */
            for (int k = 0; k < 2 && !found; ++k) {
                const int64_t i = k == 0 ? main_c + step : main_c - step;
                /* helpers.rs:30: sub - current_step .. sub + current_step + 1 */
                for (int64_t j = sub_c - step; j < sub_c + step + 1; ++j) {
                    const int64_t point_y = order ? i : j; /* helpers.rs:31 */
                    const int64_t point_x = order ? j : i;
                    ++n_cell;
                    /* helpers.rs:19-25: get2d; negative indices wrap to huge usize -> None */
                    if (point_y < 0 || point_y >= (int64_t)ah || point_x < 0 ||
                        point_x >= (int64_t)aw)
                        continue; /* helpers.rs:44 */
                    still_in_bounds = 1; /* helpers.rs:34 */
                    ++n_probe;
                    const uint8_t *v = array + (size_t)point_y * stride + (size_t)point_x * 3;
                    if (pixel_sub_lt(v, what, precision)) { /* helpers.rs:35 */
                        *pos_x = (uint32_t)point_x;         /* helpers.rs:36 */
                        *pos_y = (uint32_t)point_y;
                        *dist = oracle_distance_dot_dot(from_x, from_y, *pos_x, *pos_y);
                        found = 1; /* helpers.rs:41 return Some(..) */
                        break;
                    }
                }
            }
        }
        if (found)
            break;
        if (!still_in_bounds) /* helpers.rs:49-51 */
            break;
    }
    if (probes)
        *probes += n_probe;
    if (cells)
        *cells += n_cell;
    return found; /* helpers.rs:53: None */
}

size_t oracle_uniform_groups(const uint8_t *main_px, uint32_t w, uint32_t h, size_t stride,
                             uint32_t channels, uint32_t tile_w, uint32_t tile_h,
                             oracle_group *groups)
{
    if (tile_w == 0 || tile_h == 0 || w == 0 || h == 0)
        return 0;
    const uint32_t gx = (w + tile_w - 1) / tile_w, gy = (h + tile_h - 1) / tile_h;
    if (!groups)
        return (size_t)gx * gy;
    size_t n = 0;
    for (uint32_t ty = 0; ty < gy; ++ty) {
        for (uint32_t tx = 0; tx < gx; ++tx) {
            oracle_group g;
            g.x = tx * tile_w;
            g.y = ty * tile_h;
            g.w = (g.x + tile_w <= w) ? tile_w : w - g.x;
            g.h = (g.y + tile_h <= h) ? tile_h : h - g.y;
            uint64_t sum[3] = {0, 0, 0};
            for (uint32_t yy = g.y; yy < g.y + g.h; ++yy)
                for (uint32_t xx = g.x; xx < g.x + g.w; ++xx) {
                    const uint8_t *p = main_px + (size_t)yy * stride + (size_t)xx * channels;
                    sum[0] += p[0];
                    sum[1] += p[1];
                    sum[2] += p[2];
                }
            const uint64_t cnt = (uint64_t)g.w * g.h;
            for (int c = 0; c < 3; ++c)
                g.value[c] = (uint8_t)(sum[c] / cnt);
            g.pad = 0;
            groups[n++] = g;
        }
    }
    return n;
}

typedef struct {
    const oracle_group *groups;
    size_t begin, end; /* static range, or unused for dynamic */
    size_t n;
    const uint8_t *add_px;
    uint32_t aw, ah;
    size_t astride;
    uint32_t max_steps;
    const uint8_t *precision;
    uint8_t *raw;
    uint32_t *raw_u32;
    uint32_t *pos_xy; /* optional: winning cell (x,y) per group, (0,0) for None */
    uint8_t *found;   /* optional: 1 = Some, 0 = None */
    uint64_t probes;
    size_t *next; /* dynamic scheduling cursor (ALLCORES) */
    pthread_mutex_t *lock;
} match_job;

/* the closure body of src/depth_image.rs:113-122 */
static void match_range(match_job *job, size_t b, size_t e)
{
    for (size_t i = b; i < e; ++i) {
        const oracle_group *p = &job->groups[i];
        /* depth_image.rs:114-117 -- u32 arithmetic, floor division */
        const uint32_t mx = (p->x + p->w) / 2u;
        const uint32_t my = (p->y + p->h) / 2u;
        uint32_t d = 0, px = 0, py = 0;
        /* depth_image.rs:118-120: unwrap_or((u32::MIN, Position::new(0,0))) */
        const int hit = oracle_distance_dot_array(p->value, job->add_px, job->aw, job->ah,
                                                  job->astride, mx, my, job->max_steps,
                                                  job->precision, &d, &px, &py, &job->probes,
                                                  NULL);
        if (!hit)
            d = px = py = 0;
        if (job->pos_xy) {
            job->pos_xy[2 * i] = px;
            job->pos_xy[2 * i + 1] = py;
        }
        if (job->found)
            job->found[i] = (uint8_t)hit;
        if (job->raw_u32)
            job->raw_u32[i] = d;
        job->raw[i] = (uint8_t)d; /* depth_image.rs:121: `d as u8` wraps mod 256 */
    }
}

static void *match_static_thread(void *arg)
{
    match_job *job = (match_job *)arg;
    match_range(job, job->begin, job->end);
    return NULL;
}

static void *match_dynamic_thread(void *arg)
{
    match_job *job = (match_job *)arg;
    const size_t grain = 256;
    for (;;) {
        pthread_mutex_lock(job->lock);
        size_t b = *job->next;
        *job->next = b + grain;
        pthread_mutex_unlock(job->lock);
        if (b >= job->n)
            break;
        size_t e = b + grain < job->n ? b + grain : job->n;
        match_range(job, b, e);
    }
    return NULL;
}

static int g_threads = 0; /* 0 = every online core */

void oracle_set_threads(int n) { g_threads = n > 0 ? n : 0; }

int oracle_hardware_threads(void)
{
    if (g_threads > 0)
        return g_threads;
    long n = sysconf(_SC_NPROCESSORS_ONLN);
    return n > 0 ? (int)n : 1;
}

int oracle_match_groups(const oracle_group *groups, size_t n, const uint8_t *add_px,
                        uint32_t aw, uint32_t ah, size_t astride, uint32_t max_steps,
                        const uint8_t precision[3], int threads, uint8_t *raw,
                        uint32_t *raw_u32, uint32_t *pos_xy, uint8_t *found, uint64_t *probes)
{
    if (!groups || !add_px || !raw || !precision)
        return ORACLE_ERR_BAD_ARG;
    match_job base;
    memset(&base, 0, sizeof base);
    base.groups = groups;
    base.n = n;
    base.add_px = add_px;
    base.aw = aw;
    base.ah = ah;
    base.astride = astride;
    base.max_steps = max_steps;
    base.precision = precision;
    base.raw = raw;
    base.raw_u32 = raw_u32;
    base.pos_xy = pos_xy;
    base.found = found;

    if (threads == ORACLE_THREADS_SERIAL || n < 8) {
        /* n < 8: the reference's chunk_size would be 0 and par_chunks_mut panics
         * (depth_image.rs:111-112); the oracle just runs serially. */
        match_range(&base, 0, n);
        if (probes)
            *probes += base.probes;
        return ORACLE_OK;
    }

    if (threads == ORACLE_THREADS_REF8) {
        /* depth_image.rs:111-112: chunk_size = len/8; par_chunks_mut(chunk_size) -> 8 chunks
         * plus one remainder chunk when len % 8 != 0. One thread per chunk, like rayon
         * would at best do with >= 9 workers. */
        const size_t chunk = n / 8;
        const size_t n_chunks = (n + chunk - 1) / chunk;
        match_job *jobs = (match_job *)malloc(n_chunks * sizeof *jobs);
        pthread_t *tids = (pthread_t *)malloc(n_chunks * sizeof *tids);
        for (size_t c = 0; c < n_chunks; ++c) {
            jobs[c] = base;
            jobs[c].begin = c * chunk;
            jobs[c].end = (c + 1) * chunk < n ? (c + 1) * chunk : n;
            pthread_create(&tids[c], NULL, match_static_thread, &jobs[c]);
        }
        for (size_t c = 0; c < n_chunks; ++c) {
            pthread_join(tids[c], NULL);
            if (probes)
                *probes += jobs[c].probes;
        }
        free(jobs);
        free(tids);
        return ORACLE_OK;
    }

    /* ALLCORES: dynamic scheduling over every online core */
    const int nt = oracle_hardware_threads();
    size_t next = 0;
    pthread_mutex_t lock = PTHREAD_MUTEX_INITIALIZER;
    match_job *jobs = (match_job *)malloc((size_t)nt * sizeof *jobs);
    pthread_t *tids = (pthread_t *)malloc((size_t)nt * sizeof *tids);
    for (int t = 0; t < nt; ++t) {
        jobs[t] = base;
        jobs[t].next = &next;
        jobs[t].lock = &lock;
        pthread_create(&tids[t], NULL, match_dynamic_thread, &jobs[t]);
    }
    for (int t = 0; t < nt; ++t) {
        pthread_join(tids[t], NULL);
        if (probes)
            *probes += jobs[t].probes;
    }
    free(jobs);
    free(tids);
    return ORACLE_OK;
}

/* src/depth_image.rs:124-129 */
int oracle_normalise(const uint8_t *raw, size_t n, uint8_t *out, uint8_t *max_out)
{
    uint8_t max = 0;
    for (size_t i = 0; i < n; ++i) /* depth_image.rs:124 */
        if (raw[i] > max)
            max = raw[i];
    if (max_out)
        *max_out = max;
    if (max == 0) { /* depth_image.rs:127 would divide by zero (panic) */
        memset(out, 0, n);
        return ORACLE_ERR_MAX_ZERO;
    }
    for (size_t i = 0; i < n; ++i) /* depth_image.rs:127: u64 math, floor */
        out[i] = (uint8_t)((uint64_t)raw[i] * 255u / (uint64_t)max);
    return ORACLE_OK;
}

void oracle_paint(const oracle_group *groups, const uint8_t *values, size_t n, uint32_t w,
                  uint32_t h, uint8_t *depth, size_t dstride)
{
    for (size_t i = 0; i < n; ++i) {
        const oracle_group *g = &groups[i];
        for (uint64_t yy = g->y; yy < (uint64_t)g->y + g->h && yy < h; ++yy)
            for (uint64_t xx = g->x; xx < (uint64_t)g->x + g->w && xx < w; ++xx)
                depth[yy * dstride + xx] = values[i];
    }
}

int oracle_depth_from_additional(const uint8_t *main_px, uint32_t w, uint32_t h, size_t mstride,
                                 uint32_t mchannels, const uint8_t *add_px, uint32_t aw,
                                 uint32_t ah, size_t astride, const uint8_t precision[3],
                                 uint32_t max_steps, uint32_t tile_w, uint32_t tile_h,
                                 const oracle_group *groups, size_t n_groups, int threads,
                                 uint8_t *depth, uint8_t *raw_out, uint8_t *max_out,
                                 uint64_t *probes)
{
    if (!main_px || !add_px || !depth || !precision || (mchannels != 3 && mchannels != 4))
        return ORACLE_ERR_BAD_ARG;
    oracle_group *own = NULL;
    if (!groups) {
        n_groups = oracle_uniform_groups(main_px, w, h, mstride, mchannels, tile_w, tile_h, NULL);
        if (n_groups == 0)
            return ORACLE_ERR_BAD_ARG;
        own = (oracle_group *)malloc(n_groups * sizeof *own);
        oracle_uniform_groups(main_px, w, h, mstride, mchannels, tile_w, tile_h, own);
        groups = own;
    }
    uint8_t *raw = (uint8_t *)malloc(n_groups);
    uint8_t *norm = (uint8_t *)malloc(n_groups);
    int rc = oracle_match_groups(groups, n_groups, add_px, aw, ah, astride, max_steps, precision,
                                 threads, raw, NULL, NULL, NULL, probes);
    if (rc == ORACLE_OK) {
        memset(depth, 0, (size_t)w * h);
        if (raw_out) {
            memset(raw_out, 0, (size_t)w * h);
            oracle_paint(groups, raw, n_groups, w, h, raw_out, w);
        }
        rc = oracle_normalise(raw, n_groups, norm, max_out);
        oracle_paint(groups, norm, n_groups, w, h, depth, w);
    }
    free(raw);
    free(norm);
    free(own);
    return rc;
}

/* ---- Gaussian same-size resample, src/depth_image.rs:133-134 (image 0.23.14; unpinned) ---- */

static float gauss05(float x)
{ /* image::imageops::sample::gaussian(x, 0.5); evaluated in double and rounded once, so that
   * the value does not depend on whose expf() is used (libm, compile-time folding, GPU) */
    const double r = 0.5, xd = (double)x;
    return (float)((1.0 / (sqrt(2.0 * 3.14159265358979323846) * r)) * exp(-(xd * xd) / (2.0 * r * r)));
}

/* one pass along an axis of length n: taps for output o are inputs [left, right) */
static void gauss_pass(const uint8_t *src, size_t s_elem, size_t s_line, uint32_t n,
                       uint32_t lines, uint8_t *dst, size_t d_elem, size_t d_line)
{
    for (uint32_t o = 0; o < n; ++o) {
        const float in_c = ((float)o + 0.5f) * 1.0f;
        int64_t left = (int64_t)floorf(in_c - 3.0f);
        if (left < 0)
            left = 0;
        if (left > (int64_t)n - 1)
            left = (int64_t)n - 1;
        int64_t right = (int64_t)ceilf(in_c + 3.0f);
        if (right < left + 1)
            right = left + 1;
        if (right > (int64_t)n)
            right = (int64_t)n;
        const float centre = in_c - 0.5f;
        float ws[8], sum = 0.0f;
        const int taps = (int)(right - left);
        for (int i = 0; i < taps; ++i) {
            ws[i] = gauss05((float)(left + i) - centre);
            sum += ws[i];
        }
        for (int i = 0; i < taps; ++i)
            ws[i] /= sum;
        for (uint32_t l = 0; l < lines; ++l) {
            float t = 0.0f;
            for (int i = 0; i < taps; ++i)
                t += (float)src[(size_t)l * s_line + (size_t)(left + i) * s_elem] * ws[i];
            if (t < 0.0f)
                t = 0.0f;
            if (t > 255.0f)
                t = 255.0f;
            dst[(size_t)l * d_line + (size_t)o * d_elem] = (uint8_t)roundf(t);
        }
    }
}

void oracle_gaussian_same_size(const uint8_t *src, uint32_t w, uint32_t h, size_t sstride,
                               uint8_t *dst, size_t dstride)
{
    uint8_t *tmp = (uint8_t *)malloc((size_t)w * h);
    /* vertical_sample: axis = rows, lines = columns */
    gauss_pass(src, sstride, 1, h, w, tmp, w, 1);
    /* horizontal_sample */
    gauss_pass(tmp, 1, w, w, h, dst, 1, dstride);
    free(tmp);
}

/* ---- windowed-SAD mode: not in the reference; the engine's own second mode ---- */

typedef struct {
    const uint8_t *left, *right;
    uint32_t w, h;
    size_t lstride, rstride;
    uint32_t lch, d_max, win;
    uint32_t thr;
    uint32_t row0, row1;
    uint8_t *raw;
} sad_job;

static inline uint32_t clampi(int64_t v, uint32_t hi)
{
    return v < 0 ? 0u : (v > (int64_t)hi ? hi : (uint32_t)v);
}

static void sad_rows(const sad_job *j, uint32_t r0, uint32_t r1)
{
    const int rad = (int)(j->win / 2);
    for (uint32_t y = r0; y < r1; ++y) {
        for (uint32_t x = 0; x < j->w; ++x) {
            uint32_t best = 0xffffffffu, best_d = 0;
            int hit = 0;
            for (uint32_t d = 0; d < j->d_max && !hit; ++d) {
                uint32_t cost = 0;
                for (int dy = -rad; dy <= rad; ++dy) {
                    const uint32_t yy = clampi((int64_t)y + dy, j->h - 1);
                    for (int dx = -rad; dx <= rad; ++dx) {
                        const uint32_t xl = clampi((int64_t)x + dx, j->w - 1);
                        const uint32_t xr = clampi((int64_t)x + dx - (int64_t)d, j->w - 1);
                        const uint8_t *l = j->left + (size_t)yy * j->lstride + (size_t)xl * j->lch;
                        const uint8_t *r = j->right + (size_t)yy * j->rstride + (size_t)xr * 3;
                        for (int c = 0; c < 3; ++c) {
                            int df = (int)l[c] - (int)r[c];
                            cost += (uint32_t)(df < 0 ? -df : df);
                        }
                    }
                }
                if (cost < j->thr) { /* first disparity under the precision threshold */
                    best_d = d;
                    hit = 1;
                } else if (cost < best) { /* strict: smallest d wins ties */
                    best = cost;
                    best_d = d;
                }
            }
            j->raw[(size_t)y * j->w + x] = (uint8_t)best_d;
        }
    }
}

typedef struct {
    const sad_job *job;
    uint32_t *next;
    pthread_mutex_t *lock;
} sad_thread_arg;

static void *sad_thread(void *arg)
{
    sad_thread_arg *a = (sad_thread_arg *)arg;
    for (;;) {
        pthread_mutex_lock(a->lock);
        uint32_t r = *a->next;
        *a->next = r + 1;
        pthread_mutex_unlock(a->lock);
        if (r >= a->job->row1)
            break;
        sad_rows(a->job, r, r + 1);
    }
    return NULL;
}

int oracle_sad_rows(const uint8_t *left, const uint8_t *right, uint32_t w, uint32_t h,
                    size_t lstride, size_t rstride, uint32_t channels, uint32_t d_max,
                    uint32_t win, const uint8_t precision[3], uint32_t row0, uint32_t row1,
                    int threads, uint8_t *raw)
{
    if (!left || !right || !raw || !precision || (channels != 3 && channels != 4) ||
        (win & 1u) == 0 || d_max == 0 || d_max > 256 || row1 > h || row0 > row1 || w == 0)
        return ORACLE_ERR_BAD_ARG;
    sad_job j = {left, right, w, h, lstride, rstride, channels, d_max, win, 0, row0, row1, raw};
    j.thr = win * win * ((uint32_t)precision[0] + precision[1] + precision[2]);
    if (threads == ORACLE_THREADS_SERIAL) {
        sad_rows(&j, row0, row1);
        return ORACLE_OK;
    }
    const int nt = oracle_hardware_threads();
    uint32_t next = row0;
    pthread_mutex_t lock = PTHREAD_MUTEX_INITIALIZER;
    pthread_t *tids = (pthread_t *)malloc((size_t)nt * sizeof *tids);
    sad_thread_arg arg = {&j, &next, &lock};
    for (int t = 0; t < nt; ++t)
        pthread_create(&tids[t], NULL, sad_thread, &arg);
    for (int t = 0; t < nt; ++t)
        pthread_join(tids[t], NULL);
    free(tids);
    return ORACLE_OK;
}

/* ---- consumers of a finished depth map ------------------------------- */

/* src/depth_image.rs:166-211, one call of `inner` per loop trip */
int oracle_depth_split(const uint8_t *depth, size_t n, uint32_t zones, uint8_t *lo, uint8_t *hi,
                       uint8_t *has, uint32_t *n_zones, uint32_t *iters)
{
    if (iters)
        *iters = 0;
    if (zones < 2) { /* :163-165 */
        lo[0] = hi[0] = has[0] = 0;
        *n_zones = 1;
        return ORACLE_SPLIT_OK;
    }
    if (zones > 255 || n == 0) /* zones is a u8; :214 unwraps min() of nothing */
        return ORACLE_SPLIT_ERR_INPUT;
    uint8_t img_min = 255, img_max = 0; /* :214-215 */
    for (size_t i = 0; i < n; ++i) {
        if (depth[i] < img_min)
            img_min = depth[i];
        if (depth[i] > img_max)
            img_max = depth[i];
    }
    /* :217  ((img_max - img_min) / (zones - 1)) as usize - 1 ; step_by(0) asserts */
    const uint32_t q = (uint32_t)(uint8_t)(img_max - img_min) / (zones - 1);
    if (q < 2)
        return ORACLE_SPLIT_ERR_STEP;
    const uint32_t step = q - 1;
    uint8_t prev[256], next[256];
    uint32_t n_prev = 0, n_next;
    for (uint32_t c = img_min; c < img_max; c += step) /* :216-218, img_max exclusive */
        prev[n_prev++] = (uint8_t)c;

    uint64_t *sum = calloc(256, sizeof *sum), *cnt = calloc(256, sizeof *cnt);
    uint8_t key[256], cmin[256], cmax[256];
    int rc = ORACLE_SPLIT_ERR_DIVERGED;
    for (uint32_t it = 1; it <= ORACLE_SPLIT_MAX_ITERS; ++it) {
        if (iters)
            *iters = it;
        memset(sum, 0, 256 * sizeof *sum);
        memset(cnt, 0, 256 * sizeof *cnt);
        memset(key, 0, sizeof key);
        memset(cmin, 255, sizeof cmin);
        memset(cmax, 0, sizeof cmax);
        for (uint32_t i = 0; i < n_prev; ++i) /* :167-170, duplicates share one entry */
            key[prev[i]] = 1;
        for (size_t i = 0; i < n; ++i) { /* :171-183 */
            const int item = depth[i];
            uint32_t best = 0;
            int best_d = abs((int)prev[0] - item);
            for (uint32_t k = 1; k < n_prev; ++k) { /* min_by keeps the first minimum */
                const int d = abs((int)prev[k] - item);
                if (d < best_d) {
                    best_d = d;
                    best = k;
                }
            }
            const uint8_t c = prev[best];
            sum[c] += (uint64_t)item;
            cnt[c] += 1;
            if (depth[i] < cmin[c])
                cmin[c] = depth[i];
            if (depth[i] > cmax[c])
                cmax[c] = depth[i];
        }
        n_next = 0; /* :184-188: means in any order, then sort -- walk the keys and insert */
        for (uint32_t c = 0; c < 256; ++c) {
            if (!key[c])
                continue;
            const uint8_t m = (uint8_t)(sum[c] / (cnt[c] ? cnt[c] : 1));
            uint32_t at = n_next++;
            while (at > 0 && next[at - 1] > m) {
                next[at] = next[at - 1];
                --at;
            }
            next[at] = m;
        }
        int same = 1; /* :189-195, zip stops at the shorter list */
        for (uint32_t i = 0; i < n_prev && i < n_next; ++i)
            if (prev[i] != next[i]) {
                same = 0;
                break;
            }
        if (same) { /* :196-208 */
            rc = ORACLE_SPLIT_OK;
            for (uint32_t i = 0; i < n_next; ++i) {
                const uint8_t c = next[i];
                if (!key[c] || cnt[c] == 0) { /* get().unwrap() / min().unwrap() */
                    rc = ORACLE_SPLIT_ERR_EMPTY;
                    break;
                }
                lo[i] = cmin[c];
                hi[i] = cmax[c];
                has[i] = 1;
            }
            *n_zones = n_next;
            break;
        }
        memcpy(prev, next, n_next); /* :210 */
        n_prev = n_next;
    }
    free(sum);
    free(cnt);
    return rc;
}

void oracle_invert_depth(uint8_t *depth, size_t n)
{
    for (size_t i = 0; i < n; ++i)
        depth[i] = (uint8_t)(255 - depth[i]);
}

void oracle_slice(const uint8_t *depth, size_t n, int has_from, uint8_t from, int has_to,
                  uint8_t to, uint8_t *mask)
{
    const uint8_t f = has_from ? from : 0, t = has_to ? to : 255; /* :231-232 */
    for (size_t i = 0; i < n; ++i)
        mask[i] = (depth[i] < f || depth[i] > t) ? 0 : 255; /* :237-239 */
}

int oracle_select_foreground(const uint8_t *depth, size_t n, uint8_t *mask, uint8_t *from,
                             uint8_t *to)
{
    uint8_t lo[256], hi[256], has[256];
    uint32_t nz = 0;
    const int rc = oracle_depth_split(depth, n, 2, lo, hi, has, &nz, NULL);
    if (rc != ORACLE_SPLIT_OK)
        return rc;
    if (from)
        *from = lo[0];
    if (to)
        *to = hi[0];
    oracle_slice(depth, n, has[0], lo[0], has[0], hi[0], mask);
    return ORACLE_SPLIT_OK;
}

void oracle_apply_mask(uint8_t *rgba, const uint8_t *mask, size_t n)
{
    for (size_t i = 0; i < n; ++i)
        if (mask[i] == 0)
            memset(rgba + 4 * i, 0, 4);
}

/* ---- adaptive grouping stand-in (own design; disage's source is absent) ----------------- */

typedef struct {
    const uint8_t *px;
    uint32_t channels;
    size_t stride;
    uint8_t prec[3];
    uint32_t min_splits, max_splits;
    oracle_group *out;
    size_t n;
} adaptive_ctx;

static void adaptive_node(adaptive_ctx *c, uint32_t x, uint32_t y, uint32_t w, uint32_t h,
                          uint32_t depth)
{
    int leaf = (uint64_t)w * h == 1 || depth >= c->max_splits;
    uint64_t sum[3] = {0, 0, 0};
    if (leaf || depth >= c->min_splits) { /* statistics of the rectangle */
        uint8_t mn[3] = {255, 255, 255}, mx[3] = {0, 0, 0};
        for (uint32_t yy = y; yy < y + h; ++yy)
            for (uint32_t xx = x; xx < x + w; ++xx) {
                const uint8_t *p = c->px + (size_t)yy * c->stride + (size_t)xx * c->channels;
                for (int k = 0; k < 3; ++k) {
                    sum[k] += p[k];
                    if (p[k] < mn[k])
                        mn[k] = p[k];
                    if (p[k] > mx[k])
                        mx[k] = p[k];
                }
            }
        if (!leaf) {
            leaf = 1;
            for (int k = 0; k < 3; ++k)
                if (!((int)mx[k] - (int)mn[k] < (int)c->prec[k]))
                    leaf = 0;
        }
    }
    if (leaf) {
        if (c->out) {
            oracle_group *g = &c->out[c->n];
            g->x = x;
            g->y = y;
            g->w = w;
            g->h = h;
            for (int k = 0; k < 3; ++k)
                g->value[k] = (uint8_t)(sum[k] / ((uint64_t)w * h));
            g->pad = 0;
        }
        c->n++;
        return;
    }
    if (w >= h) {
        adaptive_node(c, x, y, w / 2, h, depth + 1);
        adaptive_node(c, x + w / 2, y, w - w / 2, h, depth + 1);
    } else {
        adaptive_node(c, x, y, w, h / 2, depth + 1);
        adaptive_node(c, x, y + h / 2, w, h - h / 2, depth + 1);
    }
}

size_t oracle_adaptive_groups(const uint8_t *main_px, uint32_t w, uint32_t h, size_t stride,
                              uint32_t channels, const uint8_t precision[3],
                              uint32_t min_splits, uint32_t max_splits, oracle_group *groups)
{
    if (w == 0 || h == 0)
        return 0;
    if (max_splits == 0) /* (w*h as f32).log2().ceil(), src/depth_image.rs:101-103 */
        max_splits = (uint32_t)ceilf(log2f((float)((uint64_t)w * h)));
    adaptive_ctx c = {main_px, channels, stride, {precision[0], precision[1], precision[2]},
                      min_splits, max_splits, groups, 0};
    adaptive_node(&c, 0, 0, w, h, 0);
    return c.n;
}
