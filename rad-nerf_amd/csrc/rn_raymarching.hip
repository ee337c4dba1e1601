// rn_raymarching.hip -- occupancy-grid ray marching and compositing for gfx950.
//
// Implements the raymarching half of include/radnerf_hip.h.  Behaviour follows
// raymarching/src/raymarching.cu of the reference (cited per kernel); structure is
// MI355X-first: 64-wide waves, one ray per lane, deterministic scan-based slice
// reservation instead of per-ray atomics, ballot/mbcnt stable compaction, device-side
// live-ray counts so the inference loop never has to read a size back on the host.
#include "rn_dda_dev.h"

#include <float.h>

namespace rn {

constexpr int kBlock = 256;  // 4 waves per workgroup

constexpr float kRPi = 0.3183098861837907f;

// ------------------------------------------------------------------------------------------------
// near / far  (raymarching.cu:91-145)
__device__ __forceinline__ void near_far_of(const float *__restrict__ o, const float *__restrict__ d, const float *__restrict__ aabb,
                                            float min_near, float &near_out, float &far_out) {
    const float ox = o[0], oy = o[1], oz = o[2];
    const float dx = d[0], dy = d[1], dz = d[2];
    const float rdx = 1 / dx, rdy = 1 / dy, rdz = 1 / dz;

    float near = (aabb[0] - ox) * rdx, far = (aabb[3] - ox) * rdx;
    if (near > far) { float c = near; near = far; far = c; }
    float near_y = (aabb[1] - oy) * rdy, far_y = (aabb[4] - oy) * rdy;
    if (near_y > far_y) { float c = near_y; near_y = far_y; far_y = c; }

    bool miss = (near > far_y || near_y > far);
    if (!miss) {
        if (near_y > near) near = near_y;
        if (far_y < far) far = far_y;
        float near_z = (aabb[2] - oz) * rdz, far_z = (aabb[5] - oz) * rdz;
        if (near_z > far_z) { float c = near_z; near_z = far_z; far_z = c; }
        miss = (near > far_z || near_z > far);
        if (!miss) {
            if (near_z > near) near = near_z;
            if (far_z < far) far = far_z;
            if (near < min_near) near = min_near;
        }
    }
    near_out = miss ? FLT_MAX : near;
    far_out = miss ? FLT_MAX : far;
}

__global__ void __launch_bounds__(kBlock)
k_near_far(const float *__restrict__ rays_o, const float *__restrict__ rays_d,
           const float *__restrict__ aabb, uint32_t N, float min_near,
           float *__restrict__ nears, float *__restrict__ fars) {
    const uint32_t n = blockIdx.x * kBlock + threadIdx.x;
    if (n >= N) return;
    float near, far;
    near_far_of(rays_o + (size_t)n * 3, rays_d + (size_t)n * 3, aabb, min_near, near, far);
    nears[n] = near;
    fars[n] = far;
}

// sph_from_ray  (raymarching.cu:162-198)
__global__ void __launch_bounds__(kBlock)
k_sph_from_ray(const float *__restrict__ rays_o, const float *__restrict__ rays_d, float radius,
               uint32_t N, float *__restrict__ coords) {
    const uint32_t n = blockIdx.x * kBlock + threadIdx.x;
    if (n >= N) return;
    const float ox = rays_o[n * 3], oy = rays_o[n * 3 + 1], oz = rays_o[n * 3 + 2];
    const float dx = rays_d[n * 3], dy = rays_d[n * 3 + 1], dz = rays_d[n * 3 + 2];
    const float A = dx * dx + dy * dy + dz * dz;
    const float B = ox * dx + oy * dy + oz * dz;
    const float C = ox * ox + oy * oy + oz * oz - radius * radius;
    const float t = (-B + sqrtf(B * B - A * C)) / A;
    const float x = ox + t * dx, y = oy + t * dy, z = oz + t * dz;
    const float theta = atan2f(sqrtf(x * x + z * z), y);
    const float phi = atan2f(z, x);
    coords[n * 2] = 2 * theta * kRPi - 1;
    coords[n * 2 + 1] = phi * kRPi;
}

// morton3D / invert  (raymarching.cu:214-254)
__global__ void __launch_bounds__(kBlock)
k_morton3D(const int32_t *__restrict__ coords, uint32_t N, int32_t *__restrict__ indices) {
    const uint32_t n = blockIdx.x * kBlock + threadIdx.x;
    if (n >= N) return;
    indices[n] = (int32_t)morton3D((uint32_t)coords[n * 3], (uint32_t)coords[n * 3 + 1], (uint32_t)coords[n * 3 + 2]);
}
__global__ void __launch_bounds__(kBlock)
k_morton3D_invert(const int32_t *__restrict__ indices, uint32_t N, int32_t *__restrict__ coords) {
    const uint32_t n = blockIdx.x * kBlock + threadIdx.x;
    if (n >= N) return;
    const int ind = indices[n];
    coords[n * 3] = (int32_t)morton3D_invert((uint32_t)(ind >> 0));
    coords[n * 3 + 1] = (int32_t)morton3D_invert((uint32_t)(ind >> 1));
    coords[n * 3 + 2] = (int32_t)morton3D_invert((uint32_t)(ind >> 2));
}

// packbits  (raymarching.cu:267-289): one output byte per lane, two 16-byte loads.
__global__ void __launch_bounds__(kBlock)
k_packbits(const float *__restrict__ grid, uint32_t N, float thresh, uint8_t *__restrict__ bitfield) {
    const uint32_t n = blockIdx.x * kBlock + threadIdx.x;
    if (n >= N) return;
    const float4 a = reinterpret_cast<const float4 *>(grid)[(size_t)n * 2];
    const float4 b = reinterpret_cast<const float4 *>(grid)[(size_t)n * 2 + 1];
    uint32_t bits = 0;
    bits |= (a.x > thresh) ? 1u : 0u;
    bits |= (a.y > thresh) ? 2u : 0u;
    bits |= (a.z > thresh) ? 4u : 0u;
    bits |= (a.w > thresh) ? 8u : 0u;
    bits |= (b.x > thresh) ? 16u : 0u;
    bits |= (b.y > thresh) ? 32u : 0u;
    bits |= (b.z > thresh) ? 64u : 0u;
    bits |= (b.w > thresh) ? 128u : 0u;
    bitfield[n] = (uint8_t)bits;
}

// morton3D_dilation  (raymarching.cu:304-335)
__global__ void __launch_bounds__(kBlock)
k_morton3D_dilation(const float *__restrict__ grid, uint32_t C, uint32_t H, float *__restrict__ out) {
    const uint32_t H3 = H * H * H;
    const uint32_t n = blockIdx.x * kBlock + threadIdx.x;
    if (n >= C * H3) return;
    const uint32_t c = n / H3;
    const uint32_t ind = n - c * H3;
    const uint32_t x = morton3D_invert(ind >> 0), y = morton3D_invert(ind >> 1), z = morton3D_invert(ind >> 2);
    const float *g = grid + (size_t)c * H3;
    float res = grid[n];
    if (x + 1 < H) res = fmaxf(res, g[morton3D(x + 1, y, z)]);
    if (x > 0) res = fmaxf(res, g[morton3D(x - 1, y, z)]);
    if (y + 1 < H) res = fmaxf(res, g[morton3D(x, y + 1, z)]);
    if (y > 0) res = fmaxf(res, g[morton3D(x, y - 1, z)]);
    if (z + 1 < H) res = fmaxf(res, g[morton3D(x, y, z + 1)]);
    if (z > 0) res = fmaxf(res, g[morton3D(x, y, z - 1)]);
    out[n] = res;
}

// ------------------------------------------------------------------------------------------------
// Inference marcher  (raymarching.cu:827-929)
__global__ void __launch_bounds__(kBlock)
k_march_rays(uint32_t n_alive_arg, uint32_t n_step, const int32_t *__restrict__ rays_alive,
             const float *__restrict__ rays_t, const float *__restrict__ rays_o,
             const float *__restrict__ rays_d, float bound, float dt_gamma, uint32_t max_steps,
             uint32_t C, uint32_t H, const uint8_t *__restrict__ grid, const float *__restrict__ fars,
             float *__restrict__ xyzs, float *__restrict__ dirs, float *__restrict__ deltas,
             const float *__restrict__ noises, const int32_t *__restrict__ n_alive_dev) {
    uint32_t n_alive = n_alive_arg;
    if (n_alive_dev) { const uint32_t d = (uint32_t)*n_alive_dev; n_alive = d < n_alive ? d : n_alive; }
    const uint32_t n = blockIdx.x * kBlock + threadIdx.x;
    if (n >= n_alive) return;

    const int index = rays_alive[n];
    const float noise = noises ? noises[n] : 0.0f;
    Dda s;
    s.init(rays_o + (size_t)index * 3, rays_d + (size_t)index * 3, bound, dt_gamma, max_steps, C, H, grid,
           fars[index]);
    float t = rays_t[index];
    t += clampf(t * dt_gamma, s.dt_min, s.dt_max) * noise;  // :873
    const size_t base = (size_t)n * n_step;
    s.walk<true>(t, n_step, xyzs + base * 3, dirs + base * 3, deltas + base * 2);
}

// Inference compositor  (raymarching.cu:942-1029)
__global__ void __launch_bounds__(kBlock)
k_composite_rays(uint32_t n_alive_arg, uint32_t n_step, float T_thresh, int32_t *__restrict__ rays_alive,
                 float *__restrict__ rays_t, const float *__restrict__ sigmas,
                 const float *__restrict__ rgbs, const float *__restrict__ deltas,
                 float *__restrict__ weights_sum, float *__restrict__ depth, float *__restrict__ image,
                 const int32_t *__restrict__ n_alive_dev) {
    uint32_t n_alive = n_alive_arg;
    if (n_alive_dev) { const uint32_t d = (uint32_t)*n_alive_dev; n_alive = d < n_alive ? d : n_alive; }
    const uint32_t n = blockIdx.x * kBlock + threadIdx.x;
    if (n >= n_alive) return;

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
        const float alpha = 1.0f - __expf(-sg[0] * dl[0]);
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

// ------------------------------------------------------------------------------------------------
// Stable compaction: out = in[in >= 0].  Two launches: per-block live counts, then every block
// sums the counts of the blocks before it, and scatters with a ballot/mbcnt prefix per wavefront.
constexpr int kCompactItems = 1024;  // elements per workgroup (4 per lane)

__device__ __forceinline__ uint32_t block_reduce_sum(uint32_t v, uint32_t *lds) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    const uint32_t wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) lds[wave] = v;
    __syncthreads();
    uint32_t s = 0;
    for (uint32_t w = 0; w < kBlock / kWave; w++) s += lds[w];
    __syncthreads();
    return s;
}

__global__ void __launch_bounds__(kBlock)
k_compact_count(const int32_t *__restrict__ in, uint32_t n_arg, const int32_t *__restrict__ n_dev,
                uint32_t *__restrict__ block_counts) {
    __shared__ uint32_t lds[kBlock / kWave];
    uint32_t n = n_arg;
    if (n_dev) { const uint32_t d = (uint32_t)*n_dev; n = d < n ? d : n; }
    const uint32_t base = blockIdx.x * kCompactItems;
    uint32_t c = 0;
    for (int i = 0; i < kCompactItems / kBlock; i++) {
        const uint32_t idx = base + i * kBlock + threadIdx.x;
        c += (idx < n && in[idx] >= 0) ? 1u : 0u;
    }
    const uint32_t s = block_reduce_sum(c, lds);
    if (threadIdx.x == 0) block_counts[blockIdx.x] = s;
}

__global__ void __launch_bounds__(kBlock)
k_compact_scatter(const int32_t *__restrict__ in, uint32_t n_arg, const int32_t *__restrict__ n_dev,
                  const uint32_t *__restrict__ block_counts, int32_t *__restrict__ out,
                  int32_t *__restrict__ n_out) {
    __shared__ uint32_t lds[kBlock / kWave];
    __shared__ uint32_t wave_off[kBlock / kWave];
    uint32_t n = n_arg;
    if (n_dev) { const uint32_t d = (uint32_t)*n_dev; n = d < n ? d : n; }

    // exclusive prefix of the blocks before this one (and the grand total in the last block)
    uint32_t part = 0;
    for (uint32_t b = threadIdx.x; b < blockIdx.x; b += kBlock) part += block_counts[b];
    uint32_t offset = block_reduce_sum(part, lds);
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) *n_out = (int32_t)(offset + block_counts[blockIdx.x]);

    const uint32_t base = blockIdx.x * kCompactItems;
    const uint32_t wave = threadIdx.x >> 6;
    for (int i = 0; i < kCompactItems / kBlock; i++) {
        const uint32_t idx = base + i * kBlock + threadIdx.x;
        const int32_t v = (idx < n) ? in[idx] : -1;
        const bool keep = v >= 0;
        const unsigned long long mask = __ballot(keep);
        const uint32_t within = ballot_prefix(mask);
        const uint32_t cnt = (uint32_t)__popcll(mask);
        if ((threadIdx.x & 63) == 0) wave_off[wave] = cnt;
        __syncthreads();
        uint32_t before = 0, total = 0;
        for (uint32_t w = 0; w < kBlock / kWave; w++) {
            const uint32_t cw = wave_off[w];
            before += (w < wave) ? cw : 0u;
            total += cw;
        }
        if (keep) out[offset + before + within] = v;
        offset += total;
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
// Training marcher  (raymarching.cu:352-518): count pass, scan, write pass.
__global__ void __launch_bounds__(kBlock)
k_march_train_count(const float *__restrict__ rays_o, const float *__restrict__ rays_d,
                    const uint8_t *__restrict__ grid, float bound, float dt_gamma, uint32_t max_steps,
                    uint32_t N, uint32_t C, uint32_t H, const float *__restrict__ nears,
                    const float *__restrict__ fars, const float *__restrict__ noises,
                    int32_t *__restrict__ rays, uint32_t *__restrict__ block_sums) {
    __shared__ uint32_t lds[kBlock / kWave];
    const uint32_t n = blockIdx.x * kBlock + threadIdx.x;
    uint32_t num_steps = 0;
    if (n < N) {
        Dda s;
        s.init(rays_o + (size_t)n * 3, rays_d + (size_t)n * 3, bound, dt_gamma, max_steps, C, H, grid, fars[n]);
        float t = nears[n];
        t += clampf(t * dt_gamma, s.dt_min, s.dt_max) * noises[n];  // :392
        num_steps = s.walk<false>(t, max_steps, nullptr, nullptr, nullptr);
        rays[n * 3 + 2] = (int32_t)num_steps;
    }
    const uint32_t sum = block_reduce_sum(num_steps, lds);
    if (threadIdx.x == 0) block_sums[blockIdx.x] = sum;
}

__global__ void __launch_bounds__(kBlock)
k_march_train_write(const float *__restrict__ rays_o, const float *__restrict__ rays_d,
                    const uint8_t *__restrict__ grid, float bound, float dt_gamma, uint32_t max_steps,
                    uint32_t N, uint32_t C, uint32_t H, uint32_t M, const float *__restrict__ nears,
                    const float *__restrict__ fars, const float *__restrict__ noises,
                    float *__restrict__ xyzs, float *__restrict__ dirs, float *__restrict__ deltas,
                    int32_t *__restrict__ rays, int32_t *__restrict__ counter,
                    const uint32_t *__restrict__ block_sums, const int32_t *__restrict__ M_dev) {
    __shared__ uint32_t lds[kBlock / kWave];
    __shared__ uint32_t wave_tot[kBlock / kWave];
    // M_dev (rn_march_rays_train_budget): the sample budget lives on the device and M is only the capacity of the buffers
    if (M_dev) { const uint32_t b = (uint32_t)*M_dev; M = b < M ? b : M; }
    // samples reserved by all earlier blocks, on top of what the counter already holds (:446)
    uint32_t part = 0;
    for (uint32_t b = threadIdx.x; b < blockIdx.x; b += kBlock) part += block_sums[b];
    const uint32_t block_off = block_reduce_sum(part, lds);
    const uint32_t counter0 = (uint32_t)counter[0], counter1 = (uint32_t)counter[1];

    const uint32_t n = blockIdx.x * kBlock + threadIdx.x;
    const uint32_t num_steps = (n < N) ? (uint32_t)rays[n * 3 + 2] : 0u;

    // exclusive scan of num_steps inside the block: wave scan (dpp shuffles) + wave totals
    uint32_t incl = num_steps;
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t o = __shfl_up(incl, off, 64);
        if ((int)(threadIdx.x & 63) >= off) incl += o;
    }
    const uint32_t wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 63) wave_tot[wave] = incl;
    __syncthreads();
    uint32_t before = 0;
    for (uint32_t w = 0; w < wave; w++) before += wave_tot[w];
    const uint32_t point_index = counter0 + block_off + before + (incl - num_steps);

    if (n < N) {
        const uint32_t ray_index = counter1 + n;
        rays[ray_index * 3] = (int32_t)n;
        rays[ray_index * 3 + 1] = (int32_t)point_index;
        // a ray beyond the budget keeps its count in the reference (the compositor repeats the test with the same M); with a
        // device-side budget the compositor only knows the capacity, so the ray is marked empty here -- same outputs
        const bool fits = point_index + num_steps <= M;
        rays[ray_index * 3 + 2] = (M_dev && !fits) ? 0 : (int32_t)num_steps;
        if (num_steps != 0 && fits) {
            Dda s;
            s.init(rays_o + (size_t)n * 3, rays_d + (size_t)n * 3, bound, dt_gamma, max_steps, C, H, grid, fars[n]);
            float t = nears[n];
            t += clampf(t * dt_gamma, s.dt_min, s.dt_max) * noises[n];
            s.walk<true>(t, num_steps, xyzs + (size_t)point_index * 3, dirs + (size_t)point_index * 3,
                         deltas + (size_t)point_index * 2);
        }
    }
    // last block publishes the new counter values once every block has read the old ones:
    // done by a trailing single-thread kernel instead (k_march_train_counter) to stay race-free.
}

__global__ void k_march_train_counter(int32_t *counter, const uint32_t *block_sums, uint32_t n_blocks, uint32_t N) {
    __shared__ uint32_t lds[kBlock / kWave];
    uint32_t part = 0;
    for (uint32_t b = threadIdx.x; b < n_blocks; b += kBlock) part += block_sums[b];
    const uint32_t total = block_reduce_sum(part, lds);
    if (threadIdx.x == 0) {
        counter[0] += (int32_t)total;
        counter[1] += (int32_t)N;
    }
}

// The training marcher of ONE step in ONE launch (rn_march_rays_train_step): near / far (raymarching.cu:91-145), the count
// pass, the ordered slice reservation, the write pass and the step's counters (raymarching.cu:352-518) -- what
// rn_near_far_from_aabb + a memset of the counters + rn_march_rays_train_budget's three kernels do in five launches; at 4 096
// rays these are 16 workgroups whose time is the length of one ray's walk, so the launches cost more than the work.
// The counts cross workgroups inside the launch: workgroup b stores (launch tag, its sample count) as one 64-bit word with an
// agent-scope atomic store (relaxed: the word is the whole message -- a release / acquire pair here would be an L2 write-back and
// an L2 invalidate per workgroup and per poll), and every workgroup waits until all n_blocks words carry this launch's tag (they are its barrier
// and its scan at once: the sum of the words before b is b's offset, the sum of all is the step's sample count).  The tag is a
// launch epoch kept in `state` (zero-initialised once by the caller, then owned by these launches): read by every workgroup
// before it stores its word, bumped by workgroup 0 after it has seen all words -- so nobody can read the new value early.
// All workgroups must be resident together (the host refuses more than one per CU); the wait is bounded all the same: a
// workgroup that gives up counts in state[1], and the step then reports zero samples (counter[0] = 0) instead of rows built on
// stale offsets.
// The counters are SET (counter[0] = samples, counter[1] = N), not added to: a training step starts them from zero
// (renderer.py:209-211), which is the memset this launch replaces.
// RECORD (max_steps <= kRecSteps): the count pass remembers the t of every sample in LDS and the write pass rebuilds the samples
// from them (Dda::emit: the same expressions) instead of walking the grid a second time -- the walk is the launch's whole time.
constexpr uint32_t kStepPolls = 1u << 20;
constexpr uint32_t kRecSteps = 32;

__device__ __forceinline__ uint32_t mix32(uint32_t x) {   // "lowbias32" finaliser (as rn_occupancy.hip's jitter)
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

template <bool RECORD>
__global__ void __launch_bounds__(kBlock)
k_march_train_step(const float *__restrict__ rays_o, const float *__restrict__ rays_d, const uint8_t *__restrict__ grid,
                   const float *__restrict__ aabb, float min_near, float bound, float dt_gamma, uint32_t max_steps, uint32_t N,
                   uint32_t C, uint32_t H, uint32_t M, const int32_t *__restrict__ M_dev, const float *__restrict__ noises,
                   float *__restrict__ nears, float *__restrict__ fars, float *__restrict__ xyzs, float *__restrict__ dirs,
                   float *__restrict__ deltas, int32_t *__restrict__ rays, int32_t *__restrict__ counter, uint32_t *state,
                   unsigned long long *words, uint32_t jitter_seed) {
    __shared__ uint32_t lds[kBlock / kWave];
    __shared__ uint32_t wave_tot[kBlock / kWave];
    __shared__ uint32_t stalled;
    __shared__ float t_rec[RECORD ? kRecSteps * kBlock : 1];      // [sample][lane]: conflict-free
    const uint32_t n_blocks = gridDim.x;
    const uint32_t tag = __hip_atomic_load(&state[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
    if (threadIdx.x == 0) stalled = 0u;
    uint32_t budget = M;                                       // M: rows the buffers hold; *M_dev: this step's sample budget
    if (M_dev) { const uint32_t b = (uint32_t)*M_dev; budget = b < M ? b : M; }

    const uint32_t n = blockIdx.x * kBlock + threadIdx.x;
    uint32_t num_steps = 0;
    float near = FLT_MAX, far = FLT_MAX, t0 = 0.0f;
    if (n < N) {
        near_far_of(rays_o + (size_t)n * 3, rays_d + (size_t)n * 3, aabb, min_near, near, far);
        nears[n] = near;
        fars[n] = far;
        Dda s;
        s.init(rays_o + (size_t)n * 3, rays_d + (size_t)n * 3, bound, dt_gamma, max_steps, C, H, grid, far);
        t0 = near;
        // :392 -- the jitter: the caller's uniform numbers (torch.rand in the reference), or (jitter_seed != 0) 24 random bits
        // of a counter-based hash of (seed, launch epoch, ray): a launch less per step, a new draw per launch even when replayed
        float u = 0.0f;
        if (noises) u = noises[n];
        else if (jitter_seed) u = (float)(mix32(mix32(n) ^ mix32(jitter_seed + tag)) >> 8) * (1.0f / 16777216.0f);
        t0 += clampf(t0 * dt_gamma, s.dt_min, s.dt_max) * u;
        float t = t0;
        if constexpr (RECORD) num_steps = s.walk_record(t, max_steps, t_rec + threadIdx.x, kBlock);
        else num_steps = s.walk<false>(t, max_steps, nullptr, nullptr, nullptr);
    }
    const uint32_t sum = block_reduce_sum(num_steps, lds);
    if (threadIdx.x == 0)
        __hip_atomic_store(&words[blockIdx.x], ((unsigned long long)tag << 32) | sum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);

    // barrier + scan: every workgroup's word of THIS launch
    uint32_t part = 0, all = 0;
    for (uint32_t b = threadIdx.x; b < n_blocks; b += kBlock) {
        unsigned long long w = __hip_atomic_load(&words[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        uint32_t polls = 0;
        while ((uint32_t)(w >> 32) != tag) {
            __builtin_amdgcn_s_sleep(1);
            if (++polls > kStepPolls) { stalled = 1u; break; }
            w = __hip_atomic_load(&words[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        const uint32_t cnt = (uint32_t)(w & 0xffffffffull);
        all += cnt;
        part += b < blockIdx.x ? cnt : 0u;
    }
    const uint32_t block_off = block_reduce_sum(part, lds);
    const uint32_t total = block_reduce_sum(all, lds);         // (the reductions' barriers also publish `stalled`)
    const bool bad = stalled != 0u;

    // exclusive scan of num_steps inside the workgroup (as k_march_train_write)
    uint32_t incl = num_steps;
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t o = __shfl_up(incl, off, 64);
        if ((int)(threadIdx.x & 63) >= off) incl += o;
    }
    const uint32_t wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 63) wave_tot[wave] = incl;
    __syncthreads();
    uint32_t before = 0;
    for (uint32_t w = 0; w < wave; w++) before += wave_tot[w];
    const uint32_t point_index = block_off + before + (incl - num_steps);

    if (n < N) {
        rays[n * 3] = (int32_t)n;
        rays[n * 3 + 1] = (int32_t)point_index;
        const bool fits = !bad && point_index + num_steps <= budget;
        rays[n * 3 + 2] = fits ? (int32_t)num_steps : 0;   // a ray beyond the budget is marked empty (see k_march_train_write)
        if (num_steps != 0 && fits) {
            Dda s;
            s.init(rays_o + (size_t)n * 3, rays_d + (size_t)n * 3, bound, dt_gamma, max_steps, C, H, grid, far);
            if constexpr (RECORD) {
                for (uint32_t k = 0; k < num_steps; k++) {
                    const size_t r = (size_t)point_index + k;
                    s.emit(t_rec[k * kBlock + threadIdx.x], xyzs + r * 3, dirs + r * 3, deltas + r * 2);
                }
            } else {
                float t = t0;
                s.walk<true>(t, num_steps, xyzs + (size_t)point_index * 3, dirs + (size_t)point_index * 3, deltas + (size_t)point_index * 2);
            }
        } else if (num_steps != 0 && !bad) {
            // The slice of a ray that does not fit is cleared (up to the buffers' capacity): the network pass visits rows
            // [0, min(counter[0], capacity)), whose rows then all hold either a sample or zeros -- the caller need not memset
            // the buffers (rn_march_rays_train_budget relies on one).
            for (uint32_t k = 0; k < num_steps && point_index + k < M; k++) {
                const size_t r = (size_t)point_index + k;
                xyzs[r * 3] = 0.0f; xyzs[r * 3 + 1] = 0.0f; xyzs[r * 3 + 2] = 0.0f;
                dirs[r * 3] = 0.0f; dirs[r * 3 + 1] = 0.0f; dirs[r * 3 + 2] = 0.0f;
                deltas[r * 2] = 0.0f; deltas[r * 2 + 1] = 0.0f;
            }
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        counter[0] = bad ? 0 : (int32_t)total;
        counter[1] = (int32_t)N;
        if (bad) atomicAdd(&state[1], 1u);
        __hip_atomic_store(&state[0], tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // every workgroup has read the old epoch
    }
}

// raymarching.cu:535-583
__global__ void __launch_bounds__(kBlock)
k_march_train_backward(const float *__restrict__ grad_xyzs, const float *__restrict__ grad_dirs,
                       const int32_t *__restrict__ rays, const float *__restrict__ deltas, uint32_t N,
                       uint32_t M, float *__restrict__ grad_rays_o, float *__restrict__ grad_rays_d) {
    const uint32_t n = blockIdx.x * kBlock + threadIdx.x;
    if (n >= N) return;
    const uint32_t offset = (uint32_t)rays[n * 3 + 1], num_steps = (uint32_t)rays[n * 3 + 2];
    if (num_steps == 0 || offset + num_steps > M) return;
    const float *gx = grad_xyzs + (size_t)offset * 3, *gdi = grad_dirs + (size_t)offset * 3;
    const float *dl = deltas + (size_t)offset * 2;
    float o0 = grad_rays_o[n * 3], o1 = grad_rays_o[n * 3 + 1], o2 = grad_rays_o[n * 3 + 2];
    float d0 = grad_rays_d[n * 3], d1 = grad_rays_d[n * 3 + 1], d2 = grad_rays_d[n * 3 + 2];
    for (uint32_t step = 0; step < num_steps; step++) {
        o0 += gx[0]; o1 += gx[1]; o2 += gx[2];
        d0 += gx[0] * dl[1] + gdi[0];
        d1 += gx[1] * dl[1] + gdi[1];
        d2 += gx[2] * dl[1] + gdi[2];
        gx += 3; gdi += 3; dl += 2;
    }
    grad_rays_o[n * 3] = o0; grad_rays_o[n * 3 + 1] = o1; grad_rays_o[n * 3 + 2] = o2;
    grad_rays_d[n * 3] = d0; grad_rays_d[n * 3 + 1] = d1; grad_rays_d[n * 3 + 2] = d2;
}

// raymarching.cu:603-687
__global__ void __launch_bounds__(kBlock)
k_composite_train_fwd(const float *__restrict__ sigmas, const float *__restrict__ rgbs,
                      const float *__restrict__ ambient, const float *__restrict__ deltas,
                      const int32_t *__restrict__ rays, uint32_t M, uint32_t N, float T_thresh,
                      float *__restrict__ weights_sum, float *__restrict__ ambient_sum,
                      float *__restrict__ depth, float *__restrict__ image) {
    const uint32_t n = blockIdx.x * kBlock + threadIdx.x;
    if (n >= N) return;
    const uint32_t index = (uint32_t)rays[n * 3], offset = (uint32_t)rays[n * 3 + 1],
                   num_steps = (uint32_t)rays[n * 3 + 2];
    float T = 1.0f, r = 0, g = 0, b = 0, ws = 0, d = 0, amb = 0;
    if (!(num_steps == 0 || offset + num_steps > M)) {
        const float *sg = sigmas + offset, *rg = rgbs + (size_t)offset * 3;
        const float *am = ambient + offset, *dl = deltas + (size_t)offset * 2;
        uint32_t step = 0;
        while (step < num_steps) {
            const float alpha = 1.0f - __expf(-sg[0] * dl[0]);
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
    }
    weights_sum[index] = ws; ambient_sum[index] = amb; depth[index] = d;
    image[index * 3] = r; image[index * 3 + 1] = g; image[index * 3 + 2] = b;
}

// raymarching.cu:711-809
__global__ void __launch_bounds__(kBlock)
k_composite_train_bwd(const float *__restrict__ grad_weights_sum, const float *__restrict__ grad_ambient_sum,
                      const float *__restrict__ grad_image, const float *__restrict__ sigmas,
                      const float *__restrict__ rgbs, const float *__restrict__ deltas,
                      const int32_t *__restrict__ rays, const float *__restrict__ weights_sum,
                      const float *__restrict__ image, uint32_t M, uint32_t N, float T_thresh,
                      float *__restrict__ grad_sigmas, float *__restrict__ grad_rgbs,
                      float *__restrict__ grad_ambient) {
    const uint32_t n = blockIdx.x * kBlock + threadIdx.x;
    if (n >= N) return;
    const uint32_t index = (uint32_t)rays[n * 3], offset = (uint32_t)rays[n * 3 + 1],
                   num_steps = (uint32_t)rays[n * 3 + 2];
    if (num_steps == 0 || offset + num_steps > M) return;

    const float gws = grad_weights_sum[index], gas = grad_ambient_sum[index];
    const float gi0 = grad_image[index * 3], gi1 = grad_image[index * 3 + 1], gi2 = grad_image[index * 3 + 2];
    const float r_final = image[index * 3], g_final = image[index * 3 + 1], b_final = image[index * 3 + 2];
    const float ws_final = weights_sum[index];

    const float *sg = sigmas + offset, *rg = rgbs + (size_t)offset * 3, *dl = deltas + (size_t)offset * 2;
    float *gs = grad_sigmas + offset, *gr = grad_rgbs + (size_t)offset * 3, *ga = grad_ambient + offset;

    uint32_t step = 0;
    float T = 1.0f, r = 0, g = 0, b = 0;
    while (step < num_steps) {
        const float alpha = 1.0f - __expf(-sg[0] * dl[0]);
        const float weight = alpha * T;
        r += weight * rg[0]; g += weight * rg[1]; b += weight * rg[2];
        T *= 1.0f - alpha;
        gr[0] = gi0 * weight; gr[1] = gi1 * weight; gr[2] = gi2 * weight;
        ga[0] = gas;
        gs[0] = dl[0] * (gi0 * (T * rg[0] - (r_final - r)) + gi1 * (T * rg[1] - (g_final - g)) +
                         gi2 * (T * rg[2] - (b_final - b)) + gws * (1 - ws_final));
        if (T < T_thresh) break;
        sg++; rg += 3; dl += 2; gs++; gr += 3; ga++;
        step++;
    }
}

}  // namespace rn

// ================================================================================================
// C ABI
using namespace rn;

extern "C" {

int rn_near_far_from_aabb(const float *rays_o, const float *rays_d, const float *aabb, uint32_t N,
                          float min_near, float *nears, float *fars, rn_stream_t stream) {
    if (N == 0) return RN_OK;
    RN_REQUIRE(rays_o && rays_d && aabb && nears && fars, "near_far_from_aabb: null pointer");
    hipLaunchKernelGGL(k_near_far, dim3(div_up(N, kBlock)), dim3(kBlock), 0, as_stream(stream), rays_o, rays_d,
                       aabb, N, min_near, nears, fars);
    return check_launch("near_far_from_aabb");
}

int rn_sph_from_ray(const float *rays_o, const float *rays_d, float radius, uint32_t N, float *coords,
                    rn_stream_t stream) {
    if (N == 0) return RN_OK;
    RN_REQUIRE(rays_o && rays_d && coords, "sph_from_ray: null pointer");
    hipLaunchKernelGGL(k_sph_from_ray, dim3(div_up(N, kBlock)), dim3(kBlock), 0, as_stream(stream), rays_o,
                       rays_d, radius, N, coords);
    return check_launch("sph_from_ray");
}

int rn_morton3D(const int32_t *coords, uint32_t N, int32_t *indices, rn_stream_t stream) {
    if (N == 0) return RN_OK;
    RN_REQUIRE(coords && indices, "morton3D: null pointer");
    hipLaunchKernelGGL(k_morton3D, dim3(div_up(N, kBlock)), dim3(kBlock), 0, as_stream(stream), coords, N, indices);
    return check_launch("morton3D");
}

int rn_morton3D_invert(const int32_t *indices, uint32_t N, int32_t *coords, rn_stream_t stream) {
    if (N == 0) return RN_OK;
    RN_REQUIRE(coords && indices, "morton3D_invert: null pointer");
    hipLaunchKernelGGL(k_morton3D_invert, dim3(div_up(N, kBlock)), dim3(kBlock), 0, as_stream(stream), indices, N,
                       coords);
    return check_launch("morton3D_invert");
}

int rn_packbits(const float *grid, uint32_t N, float density_thresh, uint8_t *bitfield, rn_stream_t stream) {
    if (N == 0) return RN_OK;
    RN_REQUIRE(grid && bitfield, "packbits: null pointer");
    RN_REQUIRE(((uintptr_t)grid & 15u) == 0, "packbits: grid must be 16-byte aligned");
    hipLaunchKernelGGL(k_packbits, dim3(div_up(N, kBlock)), dim3(kBlock), 0, as_stream(stream), grid, N,
                       density_thresh, bitfield);
    return check_launch("packbits");
}

int rn_morton3D_dilation(const float *grid, uint32_t C, uint32_t H, float *grid_dilation, rn_stream_t stream) {
    RN_REQUIRE(grid && grid_dilation, "morton3D_dilation: null pointer");
    RN_REQUIRE(H > 0 && H <= 1024, "morton3D_dilation: H=%u out of range (1..1024)", H);
    const uint32_t total = C * H * H * H;
    if (total == 0) return RN_OK;
    hipLaunchKernelGGL(k_morton3D_dilation, dim3(div_up(total, kBlock)), dim3(kBlock), 0, as_stream(stream), grid, C,
                       H, grid_dilation);
    return check_launch("morton3D_dilation");
}

size_t rn_march_rays_train_workspace(uint32_t N) { return (size_t)(div_up(N, kBlock) + 1) * sizeof(uint32_t); }

int rn_march_rays_train_budget(const float *rays_o, const float *rays_d, const uint8_t *grid, float bound,
                               float dt_gamma, uint32_t max_steps, uint32_t N, uint32_t C, uint32_t H, uint32_t M,
                               const int32_t *M_dev, const float *nears, const float *fars, float *xyzs, float *dirs,
                               float *deltas, int32_t *rays, int32_t *counter, const float *noises, void *workspace,
                               rn_stream_t stream) {
    if (N == 0) return RN_OK;
    RN_REQUIRE(rays_o && rays_d && grid && nears && fars && xyzs && dirs && deltas && rays && counter && noises,
               "march_rays_train: null pointer");
    RN_REQUIRE(workspace, "march_rays_train: workspace of rn_march_rays_train_workspace(N) bytes required");
    RN_REQUIRE(C >= 1 && C <= 16 && H >= 1 && max_steps >= 1, "march_rays_train: bad C/H/max_steps");
    uint32_t *block_sums = static_cast<uint32_t *>(workspace);
    const uint32_t blocks = div_up(N, kBlock);
    hipLaunchKernelGGL(k_march_train_count, dim3(blocks), dim3(kBlock), 0, as_stream(stream), rays_o, rays_d, grid,
                       bound, dt_gamma, max_steps, N, C, H, nears, fars, noises, rays, block_sums);
    hipLaunchKernelGGL(k_march_train_write, dim3(blocks), dim3(kBlock), 0, as_stream(stream), rays_o, rays_d, grid,
                       bound, dt_gamma, max_steps, N, C, H, M, nears, fars, noises, xyzs, dirs, deltas, rays, counter,
                       block_sums, M_dev);
    hipLaunchKernelGGL(k_march_train_counter, dim3(1), dim3(kBlock), 0, as_stream(stream), counter, block_sums,
                       blocks, N);
    return check_launch("march_rays_train");
}

size_t rn_march_rays_train_step_state(uint32_t N) { return (size_t)(2 + 2 * div_up(N, kBlock)) * sizeof(uint32_t); }

static int step_cus() {
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
        if (cus <= 0) cus = 256;
    }
    return cus;
}

int rn_march_rays_train_step(const float *rays_o, const float *rays_d, const uint8_t *grid, const float *aabb, float min_near,
                             float bound, float dt_gamma, uint32_t max_steps, uint32_t N, uint32_t C, uint32_t H, uint32_t M,
                             const int32_t *M_dev, const float *noises, float *nears, float *fars, float *xyzs, float *dirs,
                             float *deltas, int32_t *rays, int32_t *counter, void *state, uint32_t jitter_seed, rn_stream_t stream) {
    if (N == 0) return RN_OK;
    RN_REQUIRE(rays_o && rays_d && grid && aabb && nears && fars && xyzs && dirs && deltas && rays && counter && state,
               "march_rays_train_step: null pointer");
    RN_REQUIRE(((uintptr_t)state & 7u) == 0, "march_rays_train_step: state must be 8-byte aligned");
    RN_REQUIRE(C >= 1 && C <= 16 && H >= 1 && max_steps >= 1, "march_rays_train_step: bad C/H/max_steps");
    const uint32_t blocks = div_up(N, kBlock);
    RN_REQUIRE(blocks <= (uint32_t)step_cus(), "march_rays_train_step: %u rays need %u workgroups resident together, the device has %d CUs "
               "(use rn_march_rays_train_budget)", N, blocks, step_cus());
    uint32_t *st = static_cast<uint32_t *>(state);
    unsigned long long *words = reinterpret_cast<unsigned long long *>(st + 2);
    if (max_steps <= kRecSteps)
        hipLaunchKernelGGL(k_march_train_step<true>, dim3(blocks), dim3(kBlock), 0, as_stream(stream), rays_o, rays_d, grid, aabb, min_near,
                           bound, dt_gamma, max_steps, N, C, H, M, M_dev, noises, nears, fars, xyzs, dirs, deltas, rays, counter, st, words, jitter_seed);
    else
        hipLaunchKernelGGL(k_march_train_step<false>, dim3(blocks), dim3(kBlock), 0, as_stream(stream), rays_o, rays_d, grid, aabb, min_near,
                           bound, dt_gamma, max_steps, N, C, H, M, M_dev, noises, nears, fars, xyzs, dirs, deltas, rays, counter, st, words, jitter_seed);
    return check_launch("march_rays_train_step");
}

int rn_march_rays_train(const float *rays_o, const float *rays_d, const uint8_t *grid, float bound,
                        float dt_gamma, uint32_t max_steps, uint32_t N, uint32_t C, uint32_t H, uint32_t M,
                        const float *nears, const float *fars, float *xyzs, float *dirs, float *deltas,
                        int32_t *rays, int32_t *counter, const float *noises, void *workspace,
                        rn_stream_t stream) {
    return rn_march_rays_train_budget(rays_o, rays_d, grid, bound, dt_gamma, max_steps, N, C, H, M, nullptr, nears, fars, xyzs, dirs,
                                      deltas, rays, counter, noises, workspace, stream);
}

int rn_march_rays_train_backward(const float *grad_xyzs, const float *grad_dirs, const int32_t *rays,
                                 const float *deltas, uint32_t N, uint32_t M, float *grad_rays_o,
                                 float *grad_rays_d, rn_stream_t stream) {
    if (N == 0) return RN_OK;
    RN_REQUIRE(grad_xyzs && grad_dirs && rays && deltas && grad_rays_o && grad_rays_d,
               "march_rays_train_backward: null pointer");
    hipLaunchKernelGGL(k_march_train_backward, dim3(div_up(N, kBlock)), dim3(kBlock), 0, as_stream(stream),
                       grad_xyzs, grad_dirs, rays, deltas, N, M, grad_rays_o, grad_rays_d);
    return check_launch("march_rays_train_backward");
}

int rn_composite_rays_train_forward(const float *sigmas, const float *rgbs, const float *ambient,
                                    const float *deltas, const int32_t *rays, uint32_t M, uint32_t N,
                                    float T_thresh, float *weights_sum, float *ambient_sum, float *depth,
                                    float *image, rn_stream_t stream) {
    if (N == 0) return RN_OK;
    RN_REQUIRE(sigmas && rgbs && ambient && deltas && rays && weights_sum && ambient_sum && depth && image,
               "composite_rays_train_forward: null pointer");
    hipLaunchKernelGGL(k_composite_train_fwd, dim3(div_up(N, kBlock)), dim3(kBlock), 0, as_stream(stream), sigmas,
                       rgbs, ambient, deltas, rays, M, N, T_thresh, weights_sum, ambient_sum, depth, image);
    return check_launch("composite_rays_train_forward");
}

int rn_composite_rays_train_backward(const float *grad_weights_sum, const float *grad_ambient_sum,
                                     const float *grad_image, const float *sigmas, const float *rgbs,
                                     const float *ambient, const float *deltas, const int32_t *rays,
                                     const float *weights_sum, const float *ambient_sum, const float *image,
                                     uint32_t M, uint32_t N, float T_thresh, float *grad_sigmas,
                                     float *grad_rgbs, float *grad_ambient, rn_stream_t stream) {
    if (N == 0) return RN_OK;
    (void)ambient; (void)ambient_sum;
    RN_REQUIRE(grad_weights_sum && grad_ambient_sum && grad_image && sigmas && rgbs && deltas && rays &&
                   weights_sum && image && grad_sigmas && grad_rgbs && grad_ambient,
               "composite_rays_train_backward: null pointer");
    hipLaunchKernelGGL(k_composite_train_bwd, dim3(div_up(N, kBlock)), dim3(kBlock), 0, as_stream(stream),
                       grad_weights_sum, grad_ambient_sum, grad_image, sigmas, rgbs, deltas, rays, weights_sum, image,
                       M, N, T_thresh, grad_sigmas, grad_rgbs, grad_ambient);
    return check_launch("composite_rays_train_backward");
}

int rn_march_rays(uint32_t n_alive, uint32_t n_step, const int32_t *rays_alive, const float *rays_t,
                  const float *rays_o, const float *rays_d, float bound, float dt_gamma, uint32_t max_steps,
                  uint32_t C, uint32_t H, const uint8_t *grid, const float *nears, const float *fars,
                  float *xyzs, float *dirs, float *deltas, const float *noises, const int32_t *n_alive_dev,
                  rn_stream_t stream) {
    if (n_alive == 0) return RN_OK;
    (void)nears;
    RN_REQUIRE(rays_alive && rays_t && rays_o && rays_d && grid && fars && xyzs && dirs && deltas,
               "march_rays: null pointer");
    RN_REQUIRE(C >= 1 && C <= 16 && H >= 1 && max_steps >= 1 && n_step >= 1, "march_rays: bad C/H/max_steps/n_step");
    hipLaunchKernelGGL(k_march_rays, dim3(div_up(n_alive, kBlock)), dim3(kBlock), 0, as_stream(stream), n_alive,
                       n_step, rays_alive, rays_t, rays_o, rays_d, bound, dt_gamma, max_steps, C, H, grid, fars, xyzs,
                       dirs, deltas, noises, n_alive_dev);
    return check_launch("march_rays");
}

int rn_composite_rays(uint32_t n_alive, uint32_t n_step, float T_thresh, int32_t *rays_alive, float *rays_t,
                      const float *sigmas, const float *rgbs, const float *deltas, float *weights_sum,
                      float *depth, float *image, const int32_t *n_alive_dev, rn_stream_t stream) {
    if (n_alive == 0) return RN_OK;
    RN_REQUIRE(rays_alive && rays_t && sigmas && rgbs && deltas && weights_sum && depth && image,
               "composite_rays: null pointer");
    RN_REQUIRE(n_step >= 1, "composite_rays: n_step must be >= 1");
    hipLaunchKernelGGL(k_composite_rays, dim3(div_up(n_alive, kBlock)), dim3(kBlock), 0, as_stream(stream), n_alive,
                       n_step, T_thresh, rays_alive, rays_t, sigmas, rgbs, deltas, weights_sum, depth, image,
                       n_alive_dev);
    return check_launch("composite_rays");
}

size_t rn_compact_rays_workspace(uint32_t n) { return (size_t)(div_up(n, kCompactItems) + 1) * sizeof(uint32_t); }

int rn_compact_rays(const int32_t *rays_alive_in, uint32_t n, const int32_t *n_dev, int32_t *rays_alive_out,
                    int32_t *n_out, void *workspace, rn_stream_t stream) {
    RN_REQUIRE(n_out, "compact_rays: n_out is null");
    if (n == 0) return hipMemsetAsync(n_out, 0, sizeof(int32_t), as_stream(stream)) == hipSuccess ? RN_OK : RN_ERR_LAUNCH;
    RN_REQUIRE(rays_alive_in && rays_alive_out && workspace, "compact_rays: null pointer");
    RN_REQUIRE(rays_alive_in != rays_alive_out, "compact_rays: in-place compaction is not supported");
    uint32_t *block_counts = static_cast<uint32_t *>(workspace);
    const uint32_t blocks = div_up(n, kCompactItems);
    hipLaunchKernelGGL(k_compact_count, dim3(blocks), dim3(kBlock), 0, as_stream(stream), rays_alive_in, n, n_dev,
                       block_counts);
    hipLaunchKernelGGL(k_compact_scatter, dim3(blocks), dim3(kBlock), 0, as_stream(stream), rays_alive_in, n, n_dev,
                       block_counts, rays_alive_out, n_out);
    return check_launch("compact_rays");
}

}  // extern "C"
