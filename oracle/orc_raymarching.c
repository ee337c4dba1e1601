/*
 * orc_raymarching.c -- CPU oracle (TEST INFRASTRUCTURE, see radnerf_oracle.h)
 * for raymarching/src/raymarching.cu.  Plain loops, one iteration per CUDA
 * thread, float/double/int conversions placed exactly where the C++ usual
 * arithmetic conversions put them in the reference expressions.
 * Compile with -ffp-contract=off: every * and + rounds on its own.
 */
#include "radnerf_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define ORC_SQRT3 1.7320508075688772f  /* raymarching.cu:19 */
#define ORC_RPI 0.3183098861837907f    /* raymarching.cu:22 */

/* raymarching.cu:34-36 */
static inline float clampf(float x, float lo, float hi) { return fminf(hi, fmaxf(lo, x)); }
/* raymarching.cu:30-32 */
static inline float signf_(float x) { return copysignf(1.0f, x); }

/* raymarching.cu:42-47 */
static inline int mip_from_pos(float x, float y, float z, float max_cascade) {
    const float mx = fmaxf(fabsf(x), fmaxf(fabsf(y), fabsf(z)));
    int exponent;
    frexpf(mx, &exponent);
    return (int)fminf(max_cascade - 1, fmaxf(0.0f, (float)exponent));
}

/* raymarching.cu:49-54  (dt * H in float, then * 0.5 in double, stored to float) */
static inline int mip_from_dt(float dt, float H, float max_cascade) {
    const float mx = (float)((double)(dt * H) * 0.5);
    int exponent;
    frexpf(mx, &exponent);
    return (int)fminf(max_cascade - 1, fmaxf(0.0f, (float)exponent));
}

/* raymarching.cu:56-63 */
static inline uint32_t expand_bits(uint32_t v) {
    v = (v * 0x00010001u) & 0xFF0000FFu;
    v = (v * 0x00000101u) & 0x0F00F00Fu;
    v = (v * 0x00000011u) & 0xC30C30C3u;
    v = (v * 0x00000005u) & 0x49249249u;
    return v;
}
/* raymarching.cu:65-71 */
static inline uint32_t morton3D(uint32_t x, uint32_t y, uint32_t z) {
    return expand_bits(x) | (expand_bits(y) << 1) | (expand_bits(z) << 2);
}
/* raymarching.cu:73-81 */
static inline uint32_t morton3D_invert(uint32_t x) {
    x = x & 0x49249249u;
    x = (x | (x >> 2)) & 0xc30c30c3u;
    x = (x | (x >> 4)) & 0x0f00f00fu;
    x = (x | (x >> 8)) & 0xff0000ffu;
    x = (x | (x >> 16)) & 0x0000ffffu;
    return x;
}

/* raymarching.cu:91-145 */
void orc_near_far_from_aabb(const float *rays_o, const float *rays_d, const float *aabb,
                            uint32_t N, float min_near, float *nears, float *fars) {
#pragma omp parallel for schedule(static)
    for (int64_t n = 0; n < (int64_t)N; n++) {
        const float ox = rays_o[n * 3], oy = rays_o[n * 3 + 1], oz = rays_o[n * 3 + 2];
        const float dx = rays_d[n * 3], dy = rays_d[n * 3 + 1], dz = rays_d[n * 3 + 2];
        const float rdx = 1 / dx, rdy = 1 / dy, rdz = 1 / dz;

        float near = (aabb[0] - ox) * rdx;
        float far = (aabb[3] - ox) * rdx;
        if (near > far) { float c = near; near = far; far = c; }

        float near_y = (aabb[1] - oy) * rdy;
        float far_y = (aabb[4] - oy) * rdy;
        if (near_y > far_y) { float c = near_y; near_y = far_y; far_y = c; }

        if (near > far_y || near_y > far) {
            nears[n] = fars[n] = FLT_MAX;
            continue;
        }
        if (near_y > near) near = near_y;
        if (far_y < far) far = far_y;

        float near_z = (aabb[2] - oz) * rdz;
        float far_z = (aabb[5] - oz) * rdz;
        if (near_z > far_z) { float c = near_z; near_z = far_z; far_z = c; }

        if (near > far_z || near_z > far) {
            nears[n] = fars[n] = FLT_MAX;
            continue;
        }
        if (near_z > near) near = near_z;
        if (far_z < far) far = far_z;

        if (near < min_near) near = min_near;

        nears[n] = near;
        fars[n] = far;
    }
}

/* raymarching.cu:162-198 */
void orc_sph_from_ray(const float *rays_o, const float *rays_d, float radius, uint32_t N,
                      float *coords) {
    for (uint32_t n = 0; n < N; n++) {
        const float ox = rays_o[n * 3], oy = rays_o[n * 3 + 1], oz = rays_o[n * 3 + 2];
        const float dx = rays_d[n * 3], dy = rays_d[n * 3 + 1], dz = rays_d[n * 3 + 2];
        const float A = dx * dx + dy * dy + dz * dz;
        const float B = ox * dx + oy * dy + oz * dz;
        const float C = ox * ox + oy * oy + oz * oz - radius * radius;
        const float t = (-B + sqrtf(B * B - A * C)) / A;
        const float x = ox + t * dx, y = oy + t * dy, z = oz + t * dz;
        const float theta = atan2f(sqrtf(x * x + z * z), y);
        const float phi = atan2f(z, x);
        coords[n * 2] = 2 * theta * ORC_RPI - 1;
        coords[n * 2 + 1] = phi * ORC_RPI;
    }
}

/* raymarching.cu:214-226 */
void orc_morton3D(const int32_t *coords, uint32_t N, int32_t *indices) {
    for (uint32_t n = 0; n < N; n++)
        indices[n] = (int32_t)morton3D((uint32_t)coords[n * 3], (uint32_t)coords[n * 3 + 1],
                                       (uint32_t)coords[n * 3 + 2]);
}

/* raymarching.cu:237-254 */
void orc_morton3D_invert(const int32_t *indices, uint32_t N, int32_t *coords) {
    for (uint32_t n = 0; n < N; n++) {
        const int ind = indices[n];
        coords[n * 3] = (int32_t)morton3D_invert((uint32_t)(ind >> 0));
        coords[n * 3 + 1] = (int32_t)morton3D_invert((uint32_t)(ind >> 1));
        coords[n * 3 + 2] = (int32_t)morton3D_invert((uint32_t)(ind >> 2));
    }
}

/* raymarching.cu:267-289 */
void orc_packbits(const float *grid, uint32_t N, float density_thresh, uint8_t *bitfield) {
    for (uint32_t n = 0; n < N; n++) {
        const float *g = grid + (size_t)n * 8;
        uint8_t bits = 0;
        for (uint8_t i = 0; i < 8; i++)
            bits |= (g[i] > density_thresh) ? (uint8_t)((uint8_t)1 << i) : 0;
        bitfield[n] = bits;
    }
}

/* raymarching.cu:304-335 */
void orc_morton3D_dilation(const float *grid, uint32_t C, uint32_t H, float *grid_dilation) {
    const uint32_t H3 = H * H * H;
#pragma omp parallel for schedule(static)
    for (int64_t nn = 0; nn < (int64_t)C * H3; nn++) {
        const uint32_t n = (uint32_t)nn;
        const uint32_t c = n / H3;
        const uint32_t ind = n - c * H3;
        const uint32_t x = morton3D_invert(ind >> 0);
        const uint32_t y = morton3D_invert(ind >> 1);
        const uint32_t z = morton3D_invert(ind >> 2);
        float res = grid[n];
        if (x + 1 < H) res = fmaxf(res, grid[c * H3 + morton3D(x + 1, y, z)]);
        if (x > 0) res = fmaxf(res, grid[c * H3 + morton3D(x - 1, y, z)]);
        if (y + 1 < H) res = fmaxf(res, grid[c * H3 + morton3D(x, y + 1, z)]);
        if (y > 0) res = fmaxf(res, grid[c * H3 + morton3D(x, y - 1, z)]);
        if (z + 1 < H) res = fmaxf(res, grid[c * H3 + morton3D(x, y, z + 1)]);
        if (z > 0) res = fmaxf(res, grid[c * H3 + morton3D(x, y, z - 1)]);
        grid_dilation[n] = res;
    }
}

/* ------------------------------------------------------------------ the DDA
 * One DDA walk shared by the three marching kernels (raymarching.cu:400-441,
 * 466-517, 875-928 are the same loop body).  `emit` == 0 counts only. */
typedef struct {
    float ox, oy, oz, dx, dy, dz, rdx, rdy, rdz;
    float rH, H3, bound, dt_gamma, dt_min, dt_max, far;
    uint32_t C, H;
    const uint8_t *grid;
} dda_t;

static inline void dda_init(dda_t *s, const float *o, const float *d, float bound,
                            float dt_gamma, uint32_t max_steps, uint32_t C, uint32_t H,
                            const uint8_t *grid, float far) {
    s->ox = o[0]; s->oy = o[1]; s->oz = o[2];
    s->dx = d[0]; s->dy = d[1]; s->dz = d[2];
    s->rdx = 1 / s->dx; s->rdy = 1 / s->dy; s->rdz = 1 / s->dz;
    s->rH = 1 / (float)H;                     /* :379 */
    s->H3 = (float)(H * H * H);               /* :380 (uint32 product -> float) */
    s->bound = bound; s->dt_gamma = dt_gamma; s->far = far;
    s->C = C; s->H = H; s->grid = grid;
    s->dt_max = 2 * ORC_SQRT3 * (float)(1 << (C - 1)) / (float)H;     /* :386 */
    s->dt_min = fminf(s->dt_max, 2 * ORC_SQRT3 / (float)max_steps);   /* :387 */
}

/* Walks from *t; writes at most `limit` samples when emit != 0.  Returns the
 * number of occupied steps taken.  */
static uint32_t dda_walk(const dda_t *s, float *t_io, uint32_t limit, int emit, float *xyzs,
                         float *dirs, float *deltas) {
    float t = *t_io;
    uint32_t step = 0;
    const float C = (float)s->C, Hf = (float)s->H;
    while (t < s->far && step < limit) {
        const float x = clampf(s->ox + t * s->dx, -s->bound, s->bound);
        const float y = clampf(s->oy + t * s->dy, -s->bound, s->bound);
        const float z = clampf(s->oz + t * s->dz, -s->bound, s->bound);

        const float dt = clampf(t * s->dt_gamma, s->dt_min, s->dt_max);

        const int lp = mip_from_pos(x, y, z, C), ld = mip_from_dt(dt, Hf, C);
        const int level = lp > ld ? lp : ld;

        const float mip_bound = fminf(scalbnf(1.0f, level), s->bound);
        const float mip_rbound = 1 / mip_bound;

        /* :415-417  0.5 (double) * float * uint32 -> double, narrowed to float by clamp() */
        const int nx = (int)clampf((float)(0.5 * (double)(x * mip_rbound + 1) * (double)s->H), 0.0f, (float)(s->H - 1));
        const int ny = (int)clampf((float)(0.5 * (double)(y * mip_rbound + 1) * (double)s->H), 0.0f, (float)(s->H - 1));
        const int nz = (int)clampf((float)(0.5 * (double)(z * mip_rbound + 1) * (double)s->H), 0.0f, (float)(s->H - 1));

        /* :419  int * float + uint32 evaluates in float */
        const uint32_t index = (uint32_t)((float)level * s->H3 + (float)morton3D((uint32_t)nx, (uint32_t)ny, (uint32_t)nz));
        const int occ = s->grid[index / 8] & (1 << (index % 8));

        if (occ) {
            if (emit) {
                xyzs[0] = x; xyzs[1] = y; xyzs[2] = z;
                dirs[0] = s->dx; dirs[1] = s->dy; dirs[2] = s->dz;
            }
            t += dt;
            if (emit) {
                deltas[0] = dt;
                deltas[1] = t;
                xyzs += 3; dirs += 3; deltas += 2;
            }
            step++;
        } else {
            const float tx = (((nx + 0.5f + 0.5f * signf_(s->dx)) * s->rH * 2 - 1) * mip_bound - x) * s->rdx;
            const float ty = (((ny + 0.5f + 0.5f * signf_(s->dy)) * s->rH * 2 - 1) * mip_bound - y) * s->rdy;
            const float tz = (((nz + 0.5f + 0.5f * signf_(s->dz)) * s->rH * 2 - 1) * mip_bound - z) * s->rdz;
            const float tt = t + fmaxf(0.0f, fminf(tx, fminf(ty, tz)));
            do {
                t += clampf(t * s->dt_gamma, s->dt_min, s->dt_max);
            } while (t < tt);
        }
    }
    *t_io = t;
    return step;
}

/* raymarching.cu:352-518 */
void orc_march_rays_train(const float *rays_o, const float *rays_d, const uint8_t *grid,
                          float bound, float dt_gamma, uint32_t max_steps, uint32_t N,
                          uint32_t C, uint32_t H, uint32_t M, const float *nears,
                          const float *fars, float *xyzs, float *dirs, float *deltas,
                          int32_t *rays, int32_t *counter, const float *noises) {
    for (uint32_t n = 0; n < N; n++) {
        dda_t s;
        dda_init(&s, rays_o + (size_t)n * 3, rays_d + (size_t)n * 3, bound, dt_gamma, max_steps,
                 C, H, grid, fars[n]);
        const float near = nears[n];
        const float noise = noises[n];

        float t0 = near;
        t0 += clampf(t0 * dt_gamma, s.dt_min, s.dt_max) * noise; /* :392 */

        float t = t0;
        const uint32_t num_steps = dda_walk(&s, &t, max_steps, 0, NULL, NULL, NULL);

        /* :446-447 atomicAdd returns the old value */
        const uint32_t point_index = (uint32_t)counter[0];
        counter[0] += (int32_t)num_steps;
        const uint32_t ray_index = (uint32_t)counter[1];
        counter[1] += 1;

        rays[ray_index * 3] = (int32_t)n;
        rays[ray_index * 3 + 1] = (int32_t)point_index;
        rays[ray_index * 3 + 2] = (int32_t)num_steps;

        if (num_steps == 0) continue;
        if (point_index + num_steps > M) continue;

        t = t0;
        dda_walk(&s, &t, num_steps, 1, xyzs + (size_t)point_index * 3,
                 dirs + (size_t)point_index * 3, deltas + (size_t)point_index * 2);
    }
}

/* raymarching.cu:535-583 */
void orc_march_rays_train_backward(const float *grad_xyzs, const float *grad_dirs,
                                   const int32_t *rays, const float *deltas, uint32_t N,
                                   uint32_t M, float *grad_rays_o, float *grad_rays_d) {
    for (uint32_t n = 0; n < N; n++) {
        /* NB (faithful): outputs are indexed by the ROW n of `rays`, not by rays[n,0] (:550-551) */
        float *go = grad_rays_o + (size_t)n * 3;
        float *gd = grad_rays_d + (size_t)n * 3;
        const uint32_t offset = (uint32_t)rays[n * 3 + 1];
        const uint32_t num_steps = (uint32_t)rays[n * 3 + 2];
        if (num_steps == 0 || offset + num_steps > M) continue;
        const float *gx = grad_xyzs + (size_t)offset * 3;
        const float *gdi = grad_dirs + (size_t)offset * 3;
        const float *dl = deltas + (size_t)offset * 2;
        for (uint32_t step = 0; step < num_steps; step++) {
            go[0] += gx[0]; go[1] += gx[1]; go[2] += gx[2];
            gd[0] += gx[0] * dl[1] + gdi[0];
            gd[1] += gx[1] * dl[1] + gdi[1];
            gd[2] += gx[2] * dl[1] + gdi[2];
            gx += 3; gdi += 3; dl += 2;
        }
    }
}

/* raymarching.cu:603-687 */
void orc_composite_rays_train_forward(const float *sigmas, const float *rgbs,
                                      const float *ambient, const float *deltas,
                                      const int32_t *rays, uint32_t M, uint32_t N,
                                      float T_thresh, float *weights_sum, float *ambient_sum,
                                      float *depth, float *image) {
    for (uint32_t n = 0; n < N; n++) {
        const uint32_t index = (uint32_t)rays[n * 3];
        const uint32_t offset = (uint32_t)rays[n * 3 + 1];
        const uint32_t num_steps = (uint32_t)rays[n * 3 + 2];

        if (num_steps == 0 || offset + num_steps > M) {
            weights_sum[index] = 0; ambient_sum[index] = 0; depth[index] = 0;
            image[index * 3] = 0; image[index * 3 + 1] = 0; image[index * 3 + 2] = 0;
            continue;
        }
        const float *sg = sigmas + offset, *rg = rgbs + (size_t)offset * 3;
        const float *am = ambient + offset, *dl = deltas + (size_t)offset * 2;

        uint32_t step = 0;
        float T = 1.0f;
        float r = 0, g = 0, b = 0, ws = 0, d = 0, amb = 0;
        while (step < num_steps) {
            const float alpha = 1.0f - expf(-sg[0] * dl[0]); /* __expf in the reference */
            const float weight = alpha * T;
            r += weight * rg[0]; g += weight * rg[1]; b += weight * rg[2];
            d += weight * dl[1];
            ws += weight;
            amb += am[0];
            T *= 1.0f - alpha;
            if (T < T_thresh) break;
            sg++; rg += 3; am++; dl += 2;
            step++;
        }
        weights_sum[index] = ws; ambient_sum[index] = amb; depth[index] = d;
        image[index * 3] = r; image[index * 3 + 1] = g; image[index * 3 + 2] = b;
    }
}

/* raymarching.cu:711-809 */
void orc_composite_rays_train_backward(const float *grad_weights_sum,
                                       const float *grad_ambient_sum, const float *grad_image,
                                       const float *sigmas, const float *rgbs,
                                       const float *ambient, const float *deltas,
                                       const int32_t *rays, const float *weights_sum,
                                       const float *ambient_sum, const float *image, uint32_t M,
                                       uint32_t N, float T_thresh, float *grad_sigmas,
                                       float *grad_rgbs, float *grad_ambient) {
    (void)ambient; (void)ambient_sum;
    for (uint32_t n = 0; n < N; n++) {
        const uint32_t index = (uint32_t)rays[n * 3];
        const uint32_t offset = (uint32_t)rays[n * 3 + 1];
        const uint32_t num_steps = (uint32_t)rays[n * 3 + 2];
        if (num_steps == 0 || offset + num_steps > M) continue;

        const float gws = grad_weights_sum[index], gas = grad_ambient_sum[index];
        const float *gi = grad_image + (size_t)index * 3;
        const float r_final = image[index * 3], g_final = image[index * 3 + 1],
                    b_final = image[index * 3 + 2], ws_final = weights_sum[index];

        const float *sg = sigmas + offset, *rg = rgbs + (size_t)offset * 3;
        const float *dl = deltas + (size_t)offset * 2;
        float *gs = grad_sigmas + offset, *gr = grad_rgbs + (size_t)offset * 3;
        float *ga = grad_ambient + offset;

        uint32_t step = 0;
        float T = 1.0f;
        float r = 0, g = 0, b = 0, ws = 0;
        while (step < num_steps) {
            const float alpha = 1.0f - expf(-sg[0] * dl[0]);
            const float weight = alpha * T;
            r += weight * rg[0]; g += weight * rg[1]; b += weight * rg[2];
            ws += weight;
            T *= 1.0f - alpha;

            gr[0] = gi[0] * weight; gr[1] = gi[1] * weight; gr[2] = gi[2] * weight;
            ga[0] = gas;
            gs[0] = dl[0] * (gi[0] * (T * rg[0] - (r_final - r)) +
                             gi[1] * (T * rg[1] - (g_final - g)) +
                             gi[2] * (T * rg[2] - (b_final - b)) +
                             gws * (1 - ws_final));
            if (T < T_thresh) break;
            sg++; rg += 3; dl += 2; gs++; gr += 3; ga++;
            step++;
        }
    }
}

/* raymarching.cu:827-929 */
void orc_march_rays(uint32_t n_alive, uint32_t n_step, const int32_t *rays_alive,
                    const float *rays_t, const float *rays_o, const float *rays_d, float bound,
                    float dt_gamma, uint32_t max_steps, uint32_t C, uint32_t H,
                    const uint8_t *grid, const float *nears, const float *fars, float *xyzs,
                    float *dirs, float *deltas, const float *noises) {
    (void)nears;
#pragma omp parallel for schedule(dynamic, 256)
    for (int64_t n = 0; n < (int64_t)n_alive; n++) {
        const int index = rays_alive[n];
        const float noise = noises[n];
        dda_t s;
        dda_init(&s, rays_o + (size_t)index * 3, rays_d + (size_t)index * 3, bound, dt_gamma,
                 max_steps, C, H, grid, fars[index]);
        float t = rays_t[index];
        t += clampf(t * dt_gamma, s.dt_min, s.dt_max) * noise; /* :873 */
        dda_walk(&s, &t, n_step, 1, xyzs + (size_t)n * n_step * 3, dirs + (size_t)n * n_step * 3,
                 deltas + (size_t)n * n_step * 2);
    }
}

/* raymarching.cu:942-1029 */
void orc_composite_rays(uint32_t n_alive, uint32_t n_step, float T_thresh, int32_t *rays_alive,
                        float *rays_t, const float *sigmas, const float *rgbs,
                        const float *deltas, float *weights_sum, float *depth, float *image) {
#pragma omp parallel for schedule(static)
    for (int64_t n = 0; n < (int64_t)n_alive; n++) {
        const int index = rays_alive[n];
        const float *sg = sigmas + (size_t)n * n_step;
        const float *rg = rgbs + (size_t)n * n_step * 3;
        const float *dl = deltas + (size_t)n * n_step * 2;

        float t = rays_t[index];
        float weight_sum = weights_sum[index];
        float d = depth[index];
        float r = image[index * 3], g = image[index * 3 + 1], b = image[index * 3 + 2];

        uint32_t step = 0;
        while (step < n_step) {
            if (dl[0] == 0) break;
            const float alpha = 1.0f - expf(-sg[0] * dl[0]);
            const float T = 1 - weight_sum;
            const float weight = alpha * T;
            weight_sum += weight;
            t = dl[1];
            d += weight * t;
            r += weight * rg[0]; g += weight * rg[1]; b += weight * rg[2];
            if (T < T_thresh) break;
            sg++; rg += 3; dl += 2;
            step++;
        }
        if (step < n_step) rays_alive[n] = -1;
        else rays_t[index] = t;

        weights_sum[index] = weight_sum;
        depth[index] = d;
        image[index * 3] = r; image[index * 3 + 1] = g; image[index * 3 + 2] = b;
    }
}
