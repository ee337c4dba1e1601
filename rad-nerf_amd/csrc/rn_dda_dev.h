// rn_dda_dev.h -- the occupancy-grid DDA walk, shared by the marching kernels of rn_raymarching.hip and
// the device-side inference loop of rn_fused.hip (identical samples from both).
#pragma once

#include "rn_common.h"

namespace rn {

constexpr float kSqrt3 = 1.7320508075688772f;

// The DDA shared by the three marching kernels (raymarching.cu:400-441 == 466-517 == 875-928).

// raymarching.cu:42-54.  frexpf/scalbnf are exact on every target, so `level` is bit-exact.
__device__ __forceinline__ int mip_from_pos(float x, float y, float z, float max_cascade) {
    const float mx = fmaxf(fabsf(x), fmaxf(fabsf(y), fabsf(z)));
    int e;
    frexpf(mx, &e);
    return (int)fminf(max_cascade - 1, fmaxf(0.0f, (float)e));
}
__device__ __forceinline__ int mip_from_dt(float dt, float H, float max_cascade) {
    const float mx = (float)((double)(dt * H) * 0.5);
    int e;
    frexpf(mx, &e);
    return (int)fminf(max_cascade - 1, fmaxf(0.0f, (float)e));
}

struct Dda {
    float ox, oy, oz, dx, dy, dz, rdx, rdy, rdz;
    float rH, H3, bound, dt_gamma, dt_min, dt_max, far, Cf, Hf;
    uint32_t H;
    const uint8_t *grid;

    __device__ __forceinline__ void init(const float *o, const float *d, float bound_, float dt_gamma_,
                                         uint32_t max_steps, uint32_t C, uint32_t H_, const uint8_t *grid_,
                                         float far_) {
        ox = o[0]; oy = o[1]; oz = o[2];
        dx = d[0]; dy = d[1]; dz = d[2];
        rdx = 1 / dx; rdy = 1 / dy; rdz = 1 / dz;
        H = H_; Hf = (float)H_; Cf = (float)C;
        rH = 1 / Hf;
        H3 = (float)(H_ * H_ * H_);
        bound = bound_; dt_gamma = dt_gamma_; far = far_; grid = grid_;
        dt_max = 2 * kSqrt3 * (float)(1 << (C - 1)) / Hf;        // :386
        dt_min = fminf(dt_max, 2 * kSqrt3 / (float)max_steps);   // :387
    }

    // Walk from t, at most `limit` occupied steps.  EMIT writes samples to xyzs/dirs/deltas.
    template <bool EMIT>
    __device__ __forceinline__ uint32_t walk(float &t_io, uint32_t limit, float *xyzs, float *dirs,
                                             float *deltas) const {
        float t = t_io;
        uint32_t step = 0;
        uint32_t guard = 0;  // not in the reference: bounds the walk on degenerate inputs (far = inf)
        while (t < far && step < limit && guard < (1u << 20)) {
            const float x = clampf(ox + t * dx, -bound, bound);
            const float y = clampf(oy + t * dy, -bound, bound);
            const float z = clampf(oz + t * dz, -bound, bound);
            const float dt = clampf(t * dt_gamma, dt_min, dt_max);

            const int lp = mip_from_pos(x, y, z, Cf), ld = mip_from_dt(dt, Hf, Cf);
            const int level = lp > ld ? lp : ld;
            const float mip_bound = fminf(scalbnf(1.0f, level), bound);
            const float mip_rbound = 1 / mip_bound;

            // :415-417 -- the 0.5 literal makes the product double; clamp() narrows it to float.
            const int nx = (int)clampf((float)(0.5 * (double)(x * mip_rbound + 1) * (double)H), 0.0f, (float)(H - 1));
            const int ny = (int)clampf((float)(0.5 * (double)(y * mip_rbound + 1) * (double)H), 0.0f, (float)(H - 1));
            const int nz = (int)clampf((float)(0.5 * (double)(z * mip_rbound + 1) * (double)H), 0.0f, (float)(H - 1));

            // :419 -- evaluated in float (H3 is a float in the reference)
            const uint32_t index = (uint32_t)((float)level * H3 + (float)morton3D((uint32_t)nx, (uint32_t)ny, (uint32_t)nz));
            const bool occ = grid[index >> 3] & (1u << (index & 7u));

            if (occ) {
                if (EMIT) {
                    xyzs[0] = x; xyzs[1] = y; xyzs[2] = z;
                    dirs[0] = dx; dirs[1] = dy; dirs[2] = dz;
                }
                t += dt;
                if (EMIT) {
                    deltas[0] = dt;
                    deltas[1] = t;
                    xyzs += 3; dirs += 3; deltas += 2;
                }
                step++;
            } else {
                const float sx = copysignf(1.0f, dx), sy = copysignf(1.0f, dy), sz = copysignf(1.0f, dz);
                const float tx = ((((float)nx + 0.5f + 0.5f * sx) * rH * 2 - 1) * mip_bound - x) * rdx;
                const float ty = ((((float)ny + 0.5f + 0.5f * sy) * rH * 2 - 1) * mip_bound - y) * rdy;
                const float tz = ((((float)nz + 0.5f + 0.5f * sz) * rH * 2 - 1) * mip_bound - z) * rdz;
                // Clipping the cell-exit time at `far` changes no output: once t >= far the walk is over.
                const float tt = fminf(t + fmaxf(0.0f, fminf(tx, fminf(ty, tz))), far);
                do {
                    t += clampf(t * dt_gamma, dt_min, dt_max);
                    guard++;
                } while (t < tt && guard < (1u << 20));
            }
            guard++;
        }
        t_io = t;
        return step;
    }
};

}  // namespace rn
