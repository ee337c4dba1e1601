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
    float mb1, rmb1;  // one cascade: mip_bound = min(2^0, bound) and its reciprocal are the same for every lattice point
    bool one_cascade;
    bool one_step_skips;  // see init()
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
        mb1 = fminf(1.0f, bound_);
        rmb1 = 1 / mb1;
        one_cascade = C == 1 && H_ <= 256;  // then level * H3 + (float)morton < 2^24 is exact: no float round trip needed
        // An empty cell is left by `do t += dt while (t < tt)` with tt = the cell's exit time (:430-440).  With ONE cascade and
        // dt_min == dt_max (max_steps <= H: every configuration of this repo) dt is the constant 2 sqrt(3) / H = the cell's
        // diagonal, while the ray stays in a cell for at most (cell side) / max|d_i| <= (2 / H) / max|d_i|.  For max|d_i| > 0.58
        // (all rays but those within 0.3 deg of a space diagonal) that is < 0.9955 dt: the loop body runs exactly once whatever
        // tx, ty, tz round to, so the 30-odd operations that compute them are skipped.  Same t, same samples, bit for bit.
        one_step_skips = C == 1 && dt_min == dt_max && fmaxf(fabsf(dx), fmaxf(fabsf(dy), fabsf(dz))) > 0.58f;
    }

    // occupancy-grid cell of the lattice point at parameter t (raymarching.cu:404-419); returns the bit index.
    //
    // :415-417 spell the cell coordinate as 0.5 * (double)(x * mip_rbound + 1) * (double)H, narrowed to float by
    // clamp().  v = x * mip_rbound + 1 has 24 significant bits and H < 2^24, so the double product v * H / 2 is exact
    // and its narrowing is the correctly rounded value of v * H / 2 -- which is what the single fp32 multiply
    // v * (0.5f * H) returns (0.5 * H is exact).  Same bits, no fp64.  With one cascade the level is 0 for every point.
    __device__ __forceinline__ uint32_t cell_of(float t, float &x, float &y, float &z, float &dt, float &mip_bound, int &nx,
                                                int &ny, int &nz) const {
        x = clampf(ox + t * dx, -bound, bound);
        y = clampf(oy + t * dy, -bound, bound);
        z = clampf(oz + t * dz, -bound, bound);
        dt = clampf(t * dt_gamma, dt_min, dt_max);
        if (one_cascade) {  // wave-uniform fast path: same arithmetic with the per-ray constants hoisted (level = 0)
            mip_bound = mb1;
            const float half_h1 = 0.5f * Hf, top1 = (float)(H - 1);
            nx = (int)clampf((x * rmb1 + 1) * half_h1, 0.0f, top1);
            ny = (int)clampf((y * rmb1 + 1) * half_h1, 0.0f, top1);
            nz = (int)clampf((z * rmb1 + 1) * half_h1, 0.0f, top1);
            return morton3D_8((uint32_t)nx, (uint32_t)ny, (uint32_t)nz);    // one_cascade implies H <= 256
        }
        int level = 0;
        if (Cf > 1.0f) {  // wave-uniform
            const int lp = mip_from_pos(x, y, z, Cf), ld = mip_from_dt(dt, Hf, Cf);
            level = lp > ld ? lp : ld;
        }
        mip_bound = fminf(scalbnf(1.0f, level), bound);
        const float mip_rbound = 1 / mip_bound;
        const float half_h = 0.5f * Hf, top = (float)(H - 1);
        nx = (int)clampf((x * mip_rbound + 1) * half_h, 0.0f, top);
        ny = (int)clampf((y * mip_rbound + 1) * half_h, 0.0f, top);
        nz = (int)clampf((z * mip_rbound + 1) * half_h, 0.0f, top);
        // :419 -- evaluated in float (H3 is a float in the reference)
        return (uint32_t)((float)level * H3 + (float)morton3D((uint32_t)nx, (uint32_t)ny, (uint32_t)nz));
    }

    // Walk from t, at most `limit` occupied steps.  EMIT writes samples to xyzs/dirs/deltas.
    // (Measured on MI355X: the walk is bound by its own arithmetic, ~100 VALU ops per lattice point, not by the
    // dependent bitfield loads -- fetching the occupancy of the next 4 / 8 / 16 lattice points together and replaying the
    // decisions on a bit mask made k_head_march 10-17 % slower, so the loop keeps the reference's shape.)
    template <bool EMIT>
    __device__ __forceinline__ uint32_t walk(float &t_io, uint32_t limit, float *xyzs, float *dirs,
                                             float *deltas) const {
        float t = t_io;
        uint32_t step = 0;
        uint32_t guard = 0;  // not in the reference: bounds the walk on degenerate inputs (far = inf)
        while (t < far && step < limit && guard < (1u << 20)) {
            float x, y, z, dt, mip_bound;
            int nx, ny, nz;
            const uint32_t index = cell_of(t, x, y, z, dt, mip_bound, nx, ny, nz);
            if (grid[index >> 3] & (1u << (index & 7u))) {
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
            } else if (one_step_skips) {
                t += dt;                     // = t + clampf(t * dt_gamma, dt_min, dt_max), the loop's single pass
                guard++;
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

    // The same walk, remembering where each sample was taken instead of writing it: rec[k * stride] = t of sample k.
    // emit() rebuilds the sample from that t with the expressions of cell_of() / walk<true>() -- same bits -- so a caller that
    // must know every ray's count before it can place the samples (the training marcher) walks once, not twice.
    __device__ __forceinline__ uint32_t walk_record(float &t_io, uint32_t limit, float *rec, uint32_t stride) const {
        float t = t_io;
        uint32_t step = 0;
        uint32_t guard = 0;
        while (t < far && step < limit && guard < (1u << 20)) {
            float x, y, z, dt, mip_bound;
            int nx, ny, nz;
            const uint32_t index = cell_of(t, x, y, z, dt, mip_bound, nx, ny, nz);
            if (grid[index >> 3] & (1u << (index & 7u))) {
                rec[step * stride] = t;
                t += dt;
                step++;
            } else if (one_step_skips) {
                t += dt;                     // = t + clampf(t * dt_gamma, dt_min, dt_max), the loop's single pass
                guard++;
            } else {
                const float sx = copysignf(1.0f, dx), sy = copysignf(1.0f, dy), sz = copysignf(1.0f, dz);
                const float tx = ((((float)nx + 0.5f + 0.5f * sx) * rH * 2 - 1) * mip_bound - x) * rdx;
                const float ty = ((((float)ny + 0.5f + 0.5f * sy) * rH * 2 - 1) * mip_bound - y) * rdy;
                const float tz = ((((float)nz + 0.5f + 0.5f * sz) * rH * 2 - 1) * mip_bound - z) * rdz;
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

    __device__ __forceinline__ void emit(float t, float *xyz, float *dir, float *delta) const {
        xyz[0] = clampf(ox + t * dx, -bound, bound);
        xyz[1] = clampf(oy + t * dy, -bound, bound);
        xyz[2] = clampf(oz + t * dz, -bound, bound);
        dir[0] = dx; dir[1] = dy; dir[2] = dz;
        const float dt = clampf(t * dt_gamma, dt_min, dt_max);
        delta[0] = dt;
        delta[1] = t + dt;
    }
};

}  // namespace rn
