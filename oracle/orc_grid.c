/*
 * orc_grid.c -- CPU oracle (TEST INFRASTRUCTURE, see radnerf_oracle.h) for
 * gridencoder/src/gridencoder.cu.  Compile with -ffp-contract=off.
 *
 * The scalar type of the table/outputs is float or c10::Half.  Half follows
 * c10 semantics: `Half op x` converts to float, computes in float, and an
 * assignment back into a Half rounds to nearest-even.
 */
#include "radnerf_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define ORC_MAX_D 5
#define ORC_MAX_C 8

/* ------------------------------------------------------------ binary16 <-> float */
uint16_t orc_float_to_half(float f) {
    uint32_t x;
    memcpy(&x, &f, 4);
    const uint32_t sign = (x >> 16) & 0x8000u;
    x &= 0x7fffffffu;
    if (x >= 0x7f800000u) /* inf / nan */
        return (uint16_t)(sign | 0x7c00u | (x > 0x7f800000u ? (0x200u | ((x >> 13) & 0x3ffu)) : 0u));
    if (x >= 0x477ff000u) /* >= 65520 rounds to inf */
        return (uint16_t)(sign | 0x7c00u);
    if (x < 0x33000001u) /* <= 2^-25 rounds to zero */
        return (uint16_t)sign;
    if (x < 0x38800000u) { /* subnormal half */
        const uint32_t e = x >> 23;            /* biased float exponent, 102..112 */
        const uint32_t mant = (x & 0x7fffffu) | 0x800000u;
        const uint32_t shift = 126u - e;       /* 14..24 */
        uint32_t h = mant >> shift;
        const uint32_t rem = mant & ((1u << shift) - 1u);
        const uint32_t half = 1u << (shift - 1u);
        if (rem > half || (rem == half && (h & 1u))) h++;
        return (uint16_t)(sign | h);
    }
    uint32_t h = ((x - 0x38000000u) >> 13);
    const uint32_t rem = x & 0x1fffu;
    if (rem > 0x1000u || (rem == 0x1000u && (h & 1u))) h++;
    return (uint16_t)(sign | h);
}

float orc_half_to_float(uint16_t h) {
    const uint32_t sign = ((uint32_t)h & 0x8000u) << 16;
    uint32_t e = (h >> 10) & 0x1fu;
    uint32_t m = h & 0x3ffu;
    uint32_t x;
    if (e == 0) {
        if (m == 0) x = sign;
        else {
            e = 113;
            while (!(m & 0x400u)) { m <<= 1; e--; }
            x = sign | (e << 23) | ((m & 0x3ffu) << 13);
        }
    } else if (e == 31) x = sign | 0x7f800000u | (m << 13);
    else x = sign | ((e + 112u) << 23) | (m << 13);
    float f;
    memcpy(&f, &x, 4);
    return f;
}

/* typed load/store of one table scalar as float */
static inline float ld(const void *p, size_t i, int is_half) {
    return is_half ? orc_half_to_float(((const uint16_t *)p)[i]) : ((const float *)p)[i];
}
static inline void st(void *p, size_t i, float v, int is_half) {
    if (is_half) ((uint16_t *)p)[i] = orc_float_to_half(v);
    else ((float *)p)[i] = v;
}
/* value after "assigning into a scalar_t" */
static inline float rnd(float v, int is_half) {
    return is_half ? orc_half_to_float(orc_float_to_half(v)) : v;
}

/* gridencoder.cu:50-63 */
static inline uint32_t fast_hash(uint32_t D, const uint32_t *pos_grid) {
    static const uint32_t primes[7] = {1u, 2654435761u, 805459861u, 3674653429u,
                                       2097192037u, 1434869437u, 2165219737u};
    uint32_t result = 0;
    for (uint32_t i = 0; i < D; ++i) result ^= pos_grid[i] * primes[i];
    return result;
}

/* gridencoder.cu:66-84 */
uint32_t orc_grid_index(uint32_t D, uint32_t C, uint32_t gridtype, int align_corners,
                        uint32_t ch, uint32_t hashmap_size, uint32_t resolution,
                        const uint32_t *pos_grid) {
    uint32_t stride = 1;
    uint32_t index = 0;
    for (uint32_t d = 0; d < D && stride <= hashmap_size; d++) {
        index += pos_grid[d] * stride;
        stride *= align_corners ? resolution : (resolution + 1);
    }
    if (gridtype == 0 && stride > hashmap_size) index = fast_hash(D, pos_grid);
    return (index % hashmap_size) * C + ch;
}

/* gridencoder.cu:40-47 */
static inline float smoothstep(float v) { return v * v * (3.0f - 2.0f * v); }
static inline float smoothstep_derivative(float v) { return 6 * v * (1.0f - v); }

/* per-(level) constants, gridencoder.cu:137-139 */
static inline void level_consts(const int32_t *offsets, uint32_t level, float S, uint32_t H,
                                uint32_t *hashmap_size, float *scale, uint32_t *resolution) {
    *hashmap_size = (uint32_t)(offsets[level + 1] - offsets[level]);
    *scale = exp2f((float)level * S) * (float)H - 1.0f;
    *resolution = (uint32_t)ceilf(*scale) + 1;
}

/* gridencoder.cu:87-244 (one (b, level) work item per loop iteration) */
void orc_grid_encode_forward(const float *inputs, const void *embeddings, const int32_t *offsets,
                             void *outputs, uint32_t B, uint32_t D, uint32_t C, uint32_t L,
                             float S, uint32_t H, void *dy_dx, uint32_t gridtype,
                             int align_corners, uint32_t interp, int is_half) {
    for (uint32_t level = 0; level < L; level++) {
        uint32_t hashmap_size, resolution;
        float scale;
        level_consts(offsets, level, S, H, &hashmap_size, &scale, &resolution);
        const size_t gbase = (size_t)(uint32_t)offsets[level] * C;

#pragma omp parallel for schedule(static)
        for (int64_t bb = 0; bb < (int64_t)B; bb++) {
            const uint32_t b = (uint32_t)bb;
            const float *in = inputs + (size_t)b * D;
            const size_t obase = (size_t)level * B * C + (size_t)b * C;
            const size_t dbase = (size_t)b * D * L * C + (size_t)level * D * C; /* [B, L, D, C] */

            int flag_oob = 0;
            for (uint32_t d = 0; d < D; d++)
                if (in[d] < 0 || in[d] > 1) flag_oob = 1;
            if (flag_oob) {
                for (uint32_t ch = 0; ch < C; ch++) st(outputs, obase + ch, 0.0f, is_half);
                if (dy_dx)
                    for (uint32_t d = 0; d < D; d++)
                        for (uint32_t ch = 0; ch < C; ch++)
                            st(dy_dx, dbase + d * C + ch, 0.0f, is_half);
                continue;
            }

            float pos[ORC_MAX_D], pos_deriv[ORC_MAX_D];
            uint32_t pos_grid[ORC_MAX_D];
            for (uint32_t d = 0; d < D; d++) {
                pos[d] = in[d] * scale + (align_corners ? 0.0f : 0.5f);
                pos_grid[d] = (uint32_t)floorf(pos[d]);
                pos[d] -= (float)pos_grid[d];
                if (interp == 1) {
                    pos_deriv[d] = smoothstep_derivative(pos[d]);
                    pos[d] = smoothstep(pos[d]);
                } else {
                    pos_deriv[d] = 1.0f;
                }
            }

            float results[ORC_MAX_C] = {0};
            for (uint32_t idx = 0; idx < (1u << D); idx++) {
                float w = 1;
                uint32_t pgl[ORC_MAX_D];
                for (uint32_t d = 0; d < D; d++) {
                    if ((idx & (1u << d)) == 0) { w *= 1 - pos[d]; pgl[d] = pos_grid[d]; }
                    else { w *= pos[d]; pgl[d] = pos_grid[d] + 1; }
                }
                const uint32_t index = orc_grid_index(D, C, gridtype, align_corners, 0,
                                                      hashmap_size, resolution, pgl);
                for (uint32_t ch = 0; ch < C; ch++) /* :186  scalar_t += float * scalar_t */
                    results[ch] = rnd(results[ch] + w * ld(embeddings, gbase + index + ch, is_half), is_half);
            }
            for (uint32_t ch = 0; ch < C; ch++) st(outputs, obase + ch, results[ch], is_half);

            if (dy_dx) {
                for (uint32_t gd = 0; gd < D; gd++) {
                    float results_grad[ORC_MAX_C] = {0};
                    for (uint32_t idx = 0; idx < (1u << (D - 1)); idx++) {
                        float w = scale;
                        uint32_t pgl[ORC_MAX_D];
                        for (uint32_t nd = 0; nd < D - 1; nd++) {
                            const uint32_t d = (nd >= gd) ? (nd + 1) : nd;
                            if ((idx & (1u << nd)) == 0) { w *= 1 - pos[d]; pgl[d] = pos_grid[d]; }
                            else { w *= pos[d]; pgl[d] = pos_grid[d] + 1; }
                        }
                        pgl[gd] = pos_grid[gd];
                        const uint32_t il = orc_grid_index(D, C, gridtype, align_corners, 0,
                                                           hashmap_size, resolution, pgl);
                        pgl[gd] = pos_grid[gd] + 1;
                        const uint32_t ir = orc_grid_index(D, C, gridtype, align_corners, 0,
                                                           hashmap_size, resolution, pgl);
                        for (uint32_t ch = 0; ch < C; ch++) {
                            /* :234  (scalar_t - scalar_t) is a scalar_t */
                            const float diff = rnd(ld(embeddings, gbase + ir + ch, is_half) -
                                                   ld(embeddings, gbase + il + ch, is_half), is_half);
                            results_grad[ch] = rnd(results_grad[ch] + w * diff * pos_deriv[gd], is_half);
                        }
                    }
                    for (uint32_t ch = 0; ch < C; ch++)
                        st(dy_dx, dbase + gd * C + ch, results_grad[ch], is_half);
                }
            }
        }
    }
}

/* gridencoder.cu:247-368 */
void orc_grid_encode_backward(const void *grad, const float *inputs, const void *embeddings,
                              const int32_t *offsets, void *grad_embeddings, uint32_t B,
                              uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H,
                              const void *dy_dx, void *grad_inputs, uint32_t gridtype,
                              int align_corners, uint32_t interp, int is_half) {
    (void)embeddings;
    const uint32_t N_C = C < 2 ? C : 2; /* :404 */
    for (uint32_t level = 0; level < L; level++) {
        uint32_t hashmap_size, resolution;
        float scale;
        level_consts(offsets, level, S, H, &hashmap_size, &scale, &resolution);
        const size_t gbase = (size_t)(uint32_t)offsets[level] * C;

        for (uint32_t tid = 0; tid < B * C / N_C; tid++) {
            const uint32_t b = tid * N_C / C;           /* :259 */
            const uint32_t ch = tid * N_C - b * C;      /* :263 */
            const float *in = inputs + (size_t)b * D;
            const size_t grbase = (size_t)level * B * C + (size_t)b * C + ch;

            int oob = 0;
            for (uint32_t d = 0; d < D; d++)
                if (in[d] < 0 || in[d] > 1) oob = 1;
            if (oob) continue;

            float pos[ORC_MAX_D];
            uint32_t pos_grid[ORC_MAX_D];
            for (uint32_t d = 0; d < D; d++) {
                pos[d] = in[d] * scale + (align_corners ? 0.0f : 0.5f);
                pos_grid[d] = (uint32_t)floorf(pos[d]);
                pos[d] -= (float)pos_grid[d];
                if (interp == 1) pos[d] = smoothstep(pos[d]);
            }
            float grad_cur[2] = {0, 0};
            for (uint32_t c = 0; c < N_C; c++) grad_cur[c] = ld(grad, grbase + c, is_half);

            for (uint32_t idx = 0; idx < (1u << D); idx++) {
                float w = 1;
                uint32_t pgl[ORC_MAX_D];
                for (uint32_t d = 0; d < D; d++) {
                    if ((idx & (1u << d)) == 0) { w *= 1 - pos[d]; pgl[d] = pos_grid[d]; }
                    else { w *= pos[d]; pgl[d] = pos_grid[d] + 1; }
                }
                const uint32_t index = orc_grid_index(D, C, gridtype, align_corners, ch,
                                                      hashmap_size, resolution, pgl);
                for (uint32_t c = 0; c < N_C; c++) {
                    /* :328-335  atomicAdd of (scalar_t)(w * grad) */
                    const float v = rnd(w * grad_cur[c], is_half);
                    const float old = ld(grad_embeddings, gbase + index + c, is_half);
                    st(grad_embeddings, gbase + index + c, old + v, is_half);
                }
            }
        }
    }

    if (dy_dx && grad_inputs) { /* :342-368 */
        for (uint32_t t = 0; t < B * D; t++) {
            const uint32_t b = t / D;
            const uint32_t d = t - b * D;
            const size_t dbase = (size_t)b * L * D * C;
            float result = 0;
            for (uint32_t l = 0; l < L; l++)
                for (uint32_t ch = 0; ch < C; ch++) {
                    const float p = rnd(ld(grad, (size_t)l * B * C + (size_t)b * C + ch, is_half) *
                                        ld(dy_dx, dbase + (size_t)l * D * C + d * C + ch, is_half), is_half);
                    result = rnd(result + p, is_half);
                }
            st(grad_inputs, t, result, is_half);
        }
    }
}

/* gridencoder.cu:505-609 */
void orc_grad_total_variation(const float *inputs, const float *embeddings, float *grad,
                              const int32_t *offsets, float weight, uint32_t B, uint32_t D,
                              uint32_t C, uint32_t L, float S, uint32_t H, uint32_t gridtype,
                              int align_corners) {
    for (uint32_t level = 0; level < L; level++) {
        uint32_t hashmap_size, resolution;
        float scale;
        level_consts(offsets, level, S, H, &hashmap_size, &scale, &resolution);
        const float *grid = embeddings + (size_t)(uint32_t)offsets[level] * C;
        float *gr = grad + (size_t)(uint32_t)offsets[level] * C;

        for (uint32_t b = 0; b < B; b++) {
            const float *in = inputs + (size_t)b * D;
            int oob = 0;
            for (uint32_t d = 0; d < D; d++)
                if (in[d] < 0 || in[d] > 1) oob = 1;
            if (oob) continue;

            uint32_t pos_grid[ORC_MAX_D];
            for (uint32_t d = 0; d < D; d++) {
                const float p = in[d] * scale + (align_corners ? 0.0f : 0.5f);
                pos_grid[d] = (uint32_t)floorf(p);
            }
            float results[ORC_MAX_C] = {0}, idelta[ORC_MAX_C] = {0};
            const uint32_t index = orc_grid_index(D, C, gridtype, align_corners, 0, hashmap_size,
                                                  resolution, pos_grid);
            const float w = weight / (float)(2 * D);
            for (uint32_t d = 0; d < D; d++) {
                const uint32_t cur_d = pos_grid[d];
                if (cur_d < resolution) {
                    pos_grid[d] = cur_d + 1;
                    const uint32_t ir = orc_grid_index(D, C, gridtype, align_corners, 0,
                                                       hashmap_size, resolution, pos_grid);
                    for (uint32_t ch = 0; ch < C; ch++) {
                        const float gv = grid[index + ch] - grid[ir + ch];
                        results[ch] += gv;
                        idelta[ch] += gv * gv;
                    }
                }
                if (cur_d > 0) {
                    pos_grid[d] = cur_d - 1;
                    const uint32_t il = orc_grid_index(D, C, gridtype, align_corners, 0,
                                                       hashmap_size, resolution, pos_grid);
                    for (uint32_t ch = 0; ch < C; ch++) {
                        const float gv = grid[index + ch] - grid[il + ch];
                        results[ch] += gv;
                        idelta[ch] += gv * gv;
                    }
                }
                pos_grid[d] = cur_d;
            }
            for (uint32_t ch = 0; ch < C; ch++) /* rsqrtf in the reference */
                gr[index + ch] += w * results[ch] * (1.0f / sqrtf(idelta[ch] + 1e-9f));
        }
    }
}
