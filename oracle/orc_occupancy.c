/*
 * orc_occupancy.c -- CPU ORACLE (TEST INFRASTRUCTURE ONLY, see radnerf_oracle.h) for the occupancy-grid maintenance of
 * NeRFRenderer: mark_untrained_grid (nerf/renderer.py:318-379) and update_extra_state (nerf/renderer.py:383-499), restated as
 * plain loops over the cells.  The Python block loops (`for xs in X: for ys in Y: for zs in Z`) only chunk the work; per cell
 * the arithmetic is the float32 tensor arithmetic of the cited lines, with Python-float scalars (double) rounded to float
 * where they meet a float32 tensor, as PyTorch does.
 *
 * Element order: index i of cascade c is the cell with morton code i (what `tmp_grid[cas, indices] = sigmas` with
 * indices = morton3D(coords) produces, :420-438).
 */
#include "radnerf_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

static inline uint32_t inv_bits(uint32_t x) { /* raymarching.cu:73-81 */
    x = x & 0x49249249u;
    x = (x | (x >> 2)) & 0xc30c30c3u;
    x = (x | (x >> 4)) & 0x0f00f00fu;
    x = (x | (x >> 8)) & 0xff0000ffu;
    x = (x | (x >> 16)) & 0x0000ffffu;
    return x;
}

/* The built-in jitter of the HIP path (no counterpart in the reference, which draws torch.rand_like): 24 bits of a 32-bit
 * mix of (seed, element index).  Integer arithmetic only, so both sides agree bit for bit. */
static inline uint32_t mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
uint32_t orc_hash_u01_bits(uint32_t seed, uint32_t idx) { return mix32(mix32(idx) ^ seed) >> 8; }
static inline float hash_u01(uint32_t seed, uint32_t idx) { return (float)orc_hash_u01_bits(seed, idx) * (1.0f / 16777216.0f); }

static void cascade_consts(uint32_t c, uint32_t H, double bound, float *scale, float *half) {
    double b = (double)(1u << c);          /* bound = min(2 ** cas, self.bound)              :425 */
    if (b > bound) b = bound;
    const double hg = b / (double)H;       /* half_grid_size = bound / self.grid_size        :426 */
    *scale = (float)(b - hg);              /* xyzs * (bound - half_grid_size)                :427 */
    *half = (float)hg;
}

/* renderer.py:421-430: xyzs = 2 * coords.float() / (H - 1) - 1; cas_xyzs = xyzs * (bound - half); cas_xyzs += (rand * 2 - 1) * half */
void orc_occupancy_points(uint32_t C, uint32_t H, float bound, const float *noise, uint32_t seed, float *xyzs) {
    const uint32_t H3 = H * H * H;
    const float hm1 = (float)(H - 1);
    for (uint32_t cas = 0; cas < C; cas++) {
        float scale, half;
        cascade_consts(cas, H, (double)bound, &scale, &half);
#pragma omp parallel for
        for (uint32_t mo = 0; mo < H3; mo++) {
            const uint32_t i = cas * H3 + mo;
            const uint32_t c[3] = {inv_bits(mo), inv_bits(mo >> 1), inv_bits(mo >> 2)};
            for (int d = 0; d < 3; d++) {
                const float base = (2.0f * (float)c[d]) / hm1 - 1.0f;
                const float u = noise ? noise[(size_t)i * 3 + d] : hash_u01(seed, i * 3u + (uint32_t)d);
                const float jit = (u * 2.0f - 1.0f) * half;
                xyzs[(size_t)i * 3 + d] = base * scale + jit;
            }
        }
    }
}

/* renderer.py:437-448: sigmas *= density_scale; tmp_grid[cas, indices] = sigmas; dilation; valid_mask; maximum; mean of the
 * clamped grid; threshold; packbits.  stats = {mean_density, density_thresh actually used}.  The mean is summed in double
 * (torch.mean's float32 pairwise sum cannot be restated; the two agree to a few float ulps). */
void orc_occupancy_update(const float *sigmas, float density_scale, float *grid, uint32_t C, uint32_t H, float decay,
                          float density_thresh, uint8_t *bitfield, float *stats) {
    const size_t total = (size_t)C * H * H * H;
    float *tmp = (float *)calloc(total, sizeof(float));
    float *dil = (float *)malloc(total * sizeof(float));
    for (size_t i = 0; i < total; i++) tmp[i] = sigmas[i] * density_scale;
    orc_morton3D_dilation(tmp, C, H, dil);                                   /* :440 */
    double sum = 0.0;
    for (size_t i = 0; i < total; i++) {
        float v = grid[i];
        if (v >= 0.0f && dil[i] >= 0.0f) {                                   /* :443-444 */
            const float a = v * decay;
            v = a > dil[i] ? a : dil[i];
            grid[i] = v;
        }
        sum += (double)(v > 0.0f ? v : 0.0f);                                /* :445 */
    }
    const float mean = (float)(sum / (double)total);
    const float thresh = mean < density_thresh ? mean : density_thresh;      /* :448 */
    orc_packbits(grid, (uint32_t)(total / 8), thresh, bitfield);             /* :449 */
    if (stats) { stats[0] = mean; stats[1] = thresh; }
    free(tmp);
    free(dil);
}

/* renderer.py:318-379.  poses: [n, 4, 4] (stride 16) or [n, 3, 4] (stride 12) row-major cam2world.  A cell stays trained if
 * any camera has it in front and inside the frustum widened by one cell (count > 0); otherwise density_grid = -1. */
void orc_mark_untrained_grid(const float *poses, uint32_t n_poses, uint32_t pose_stride, double fx, double fy, double cx,
                             double cy, uint32_t C, uint32_t H, float bound, float *grid) {
    const uint32_t H3 = H * H * H;
    const float hm1 = (float)(H - 1);
    const float cx_fx = (float)(cx / fx), cy_fy = (float)(cy / fy);          /* Python floats meeting a float32 tensor :368-369 */
    for (uint32_t cas = 0; cas < C; cas++) {
        float scale, half;
        cascade_consts(cas, H, (double)bound, &scale, &half);
        const float margin = half * 2.0f;                                    /* half_grid_size * 2 */
#pragma omp parallel for
        for (uint32_t mo = 0; mo < H3; mo++) {
            const uint32_t c[3] = {inv_bits(mo), inv_bits(mo >> 1), inv_bits(mo >> 2)};
            float w[3];
            for (int d = 0; d < 3; d++) w[d] = ((2.0f * (float)c[d]) / hm1 - 1.0f) * scale;    /* :347, :355 */
            uint32_t count = 0;
            for (uint32_t p = 0; p < n_poses; p++) {
                const float *M = poses + (size_t)p * pose_stride;
                const float dx = w[0] - M[3], dy = w[1] - M[7], dz = w[2] - M[11];               /* :362 */
                const float camx = dx * M[0] + dy * M[4] + dz * M[8];                             /* :363  (w - t) @ R */
                const float camy = dx * M[1] + dy * M[5] + dz * M[9];
                const float camz = dx * M[2] + dy * M[6] + dz * M[10];
                const int mz = camz > 0.0f;                                                       /* :366 */
                const int mx = fabsf(camx) < cx_fx * camz + margin;                               /* :368 */
                const int my = fabsf(camy) < cy_fy * camz + margin;                               /* :369 */
                count += (uint32_t)(mz & mx & my);
            }
            if (count == 0) grid[(size_t)cas * H3 + mo] = -1.0f;                                  /* :378 */
        }
    }
}

/* renderer.py:464-476: element i = (column x = i % H, row y = i / H), i.e. tmp_grid_torso[y * H + x] belongs to the point
 * (xs[x], ys[y]) -- the reference's "xy transposed" index. */
void orc_torso_grid_points(uint32_t H, const float *noise, uint32_t seed, float *xys) {
    const double hg = 1.0 / (double)H;                                        /* :466 */
    const float scale = (float)(1.0 - hg), half = (float)hg, hm1 = (float)(H - 1);
    for (uint32_t i = 0; i < H * H; i++) {
        const uint32_t c[2] = {i % H, i / H};
        for (int d = 0; d < 2; d++) {
            const float base = ((2.0f * (float)c[d]) / hm1 - 1.0f) * scale;    /* :473-474 */
            const float u = noise ? noise[(size_t)i * 2 + d] : hash_u01(seed, i * 2u + (uint32_t)d);
            xys[(size_t)i * 2 + d] = base + (u * 2.0f - 1.0f) * half;          /* :475 */
        }
    }
}

/* renderer.py:482-490: F.max_pool2d(kernel 5, stride 1, padding 2: -inf outside), maximum with the decayed grid, mean. */
void orc_torso_grid_update(const float *alphas, float *grid, uint32_t H, float decay, float *stats) {
    double sum = 0.0;
    float *out = (float *)malloc((size_t)H * H * sizeof(float));
    for (int y = 0; y < (int)H; y++)
        for (int x = 0; x < (int)H; x++) {
            float m = -INFINITY;
            for (int dy = -2; dy <= 2; dy++)
                for (int dx = -2; dx <= 2; dx++) {
                    const int xx = x + dx, yy = y + dy;
                    if (xx >= 0 && xx < (int)H && yy >= 0 && yy < (int)H) {
                        const float a = alphas[yy * (int)H + xx];
                        m = a > m ? a : m;
                    }
                }
            const float g = grid[y * (int)H + x] * decay;
            const float v = g > m ? g : m;
            out[y * (int)H + x] = v;
            sum += (double)v;
        }
    memcpy(grid, out, (size_t)H * H * sizeof(float));
    if (stats) stats[0] = (float)(sum / (double)((size_t)H * H));
    free(out);
}
