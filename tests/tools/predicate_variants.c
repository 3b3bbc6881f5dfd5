// predicate_variants.c -- the reference's ring first-hit search (src/helpers.rs:9-54) with the
// pixel predicate swapped for the alternatives SURVEY.md 0.5 rejected.  Test infrastructure
// for tests/test_oracle_asset.py: disage's PixelOpps::{substract,lt} (src/helpers.rs:35) is not
// in the reference tree, so its meaning is pinned by which variant reproduces assets/depth.jpg.
//   0  |v-w| <  p on all channels      (the pinned reading)
//   1  sat(v-w) < p on all channels    (substract = saturating subtraction)
//   2  |v-w| <  p on any channel
//   3  sum |v-w| < sum p               (channel sum)
//   4  |v-w| <= p on all channels      (lt read as "less or equal")
// build: gcc -O2 -fopenmp -shared -fPIC tests/tools/predicate_variants.c -o <out>.so
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

static int pred(const uint8_t *v, const uint8_t *w, const uint8_t *p, int variant)
{
    int d[3], s = 0, all = 1, any = 0;
    for (int c = 0; c < 3; ++c) {
        int x = (int)v[c] - (int)w[c];
        d[c] = variant == 1 ? (x < 0 ? 0 : x) : abs(x);
        s += d[c];
        const int ok = variant == 4 ? d[c] <= p[c] : d[c] < p[c];
        all &= ok;
        any |= ok;
    }
    if (variant == 2)
        return any;
    if (variant == 3)
        return s < p[0] + p[1] + p[2];
    return all;
}

// 1x1 groups: raw[y*W+x] = (d as u8) for the main pixel (x, y), src/depth_image.rs:112-121
void variant_raw(const uint8_t *main_px, int W, int H, const uint8_t *add, int wa, int ha,
                 const uint8_t *p, int max_steps, int variant, uint8_t *raw)
{
#pragma omp parallel for schedule(dynamic, 4)
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            const uint8_t *what = main_px + ((size_t)y * W + x) * 3;
            const int cx = (x + 1) / 2, cy = (y + 1) / 2; // middle, src/depth_image.rs:114-117
            unsigned d = 0;
            int done = 0;
            for (int s = 0; s < max_steps && !done; ++s) { // helpers.rs:26
                int still = 0;
                for (int side = 0; side < 4 && !done; ++side)
                    for (int j = -s; j <= s && !done; ++j) { // helpers.rs:28-31
                        const int px = side < 2 ? cx + j : (side == 2 ? cx + s : cx - s);
                        const int py = side < 2 ? (side == 0 ? cy + s : cy - s) : cy + j;
                        if (px < 0 || py < 0 || px >= wa || py >= ha)
                            continue;
                        still = 1;
                        if (pred(add + ((size_t)py * wa + px) * 3, what, p, variant)) {
                            const double dx = px - cx, dy = py - cy;
                            d = (unsigned)sqrt(dx * dx + dy * dy); // helpers.rs:3-7
                            done = 1;
                        }
                    }
                if (!still)
                    break; // helpers.rs:49-51
            }
            raw[(size_t)y * W + x] = (uint8_t)d;
        }
}
