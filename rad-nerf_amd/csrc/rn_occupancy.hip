// rn_occupancy.hip -- occupancy-grid maintenance of the renderer as gfx950 kernels.
//
// What is computed: NeRFRenderer.mark_untrained_grid (nerf/renderer.py:318-379) and NeRFRenderer.update_extra_state
// (nerf/renderer.py:383-499).  The reference walks the 128^3 cells in 64^3 / 128^3 Python blocks with ~12 PyTorch launches
// per block and cascade (meshgrid, cat, morton3D, float math, rand_like, the density network, a scatter by index), then
// dilates, takes an elementwise max on a boolean-masked copy, a mean with a host read-back, and packs bits.
//
// Here the cells are enumerated IN MORTON ORDER -- thread i of cascade c works on the cell whose morton code is i -- so
//   * a probe point's slot in the sample buffer IS its cell's slot in the density grid: the network kernel's sigma output is
//     the "tmp_grid" of the reference, no index tensor and no scatter exist;
//   * every read and write of the grid is coalesced.
// Per refresh:  k_occ_points (cell centre + jitter)  ->  fused network kernel, sigma branch only (rn_nerf_fused_forward with
// rgbs = NULL)  ->  k_occ_update (6-neighbour dilation in morton space, decayed running max, per-workgroup partial sums of
// max(grid, 0))  ->  k_occ_mean (one workgroup adds the partials up in index order and publishes mean and threshold)  ->
// k_occ_pack (bit i of byte n = grid[8 n + i] > threshold, threshold read from device memory: no host read-back).
// The 2-D torso grid (128^2 alphas): k_torso_points -> rn_torso_fused with the occupancy test disabled -> k_torso_update
// (5 x 5 max pool with -inf padding, decayed running max, mean; one workgroup, the whole grid staged in LDS).
//
// Jitter: either the caller's uniform numbers (noise != NULL; torch.rand_like in the reference) or a counter-based hash of
// (seed, element index) -- stateless, identical on the CPU oracle, so refreshes are reproducible bit for bit.
#include "rn_common.h"

#include "../../include/radnerf_fused.h"

#include <math.h>

namespace rn {

constexpr int kOccBlock = 256;

// uniform in [0,1) with 24 random bits from a 32-bit mix (public-domain "lowbias32" finaliser applied twice)
__host__ __device__ inline uint32_t mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
__host__ __device__ inline float hash_u01(uint32_t seed, uint32_t idx) {
    return (float)(mix32(mix32(idx) ^ seed) >> 8) * (1.0f / 16777216.0f);
}

struct CascadeConsts {        // per cascade, computed on the host in double exactly as the Python expressions are
    float scale[16];          // (float)(bound_c - bound_c / H)          renderer.py:427-428
    float half[16];           // (float)(bound_c / H)
};

// ---- probe points of the 3-D grid (renderer.py:421-430) ---------------------------------------------------------
__global__ void __launch_bounds__(kOccBlock)
k_occ_points(uint32_t C, uint32_t H, CascadeConsts cc, const float *__restrict__ noise, uint32_t seed, float *__restrict__ xyzs) {
    const uint32_t H3 = H * H * H;
    const uint32_t i = blockIdx.x * kOccBlock + threadIdx.x;
    if (i >= C * H3) return;
    const uint32_t cas = i / H3, mo = i - cas * H3;
    const uint32_t c[3] = {morton3D_invert(mo), morton3D_invert(mo >> 1), morton3D_invert(mo >> 2)};
    const float hm1 = (float)(H - 1);
#pragma unroll
    for (int d = 0; d < 3; d++) {
        const float base = (2.0f * (float)c[d]) / hm1 - 1.0f;           // 2 * coords.float() / (H - 1) - 1
        const float u = noise ? noise[(size_t)i * 3 + d] : hash_u01(seed, i * 3u + (uint32_t)d);
        const float jit = (u * 2.0f - 1.0f) * cc.half[cas];             // (rand * 2 - 1) * half_grid_size
        xyzs[(size_t)i * 3 + d] = base * cc.scale[cas] + jit;           // -ffp-contract=off: mul, then add, as torch does
    }
}

// ---- dilation + decayed max + mean (renderer.py:438-446, raymarching.cu:304-341) ---------------------------------
// scratch: double partial[blocks]; stats[2] = {mean_density, threshold} come from k_occ_mean
__global__ void __launch_bounds__(kOccBlock)
k_occ_update(const float *__restrict__ sigmas, float density_scale, float *__restrict__ grid, uint32_t C, uint32_t H, float decay,
             double *__restrict__ partial) {
    __shared__ double red[kOccBlock / kWave];
    const uint32_t H3 = H * H * H, total = C * H3;
    const uint32_t i = blockIdx.x * kOccBlock + threadIdx.x;
    float clamped = 0.0f;
    if (i < total) {
        const uint32_t cas = i / H3, mo = i - cas * H3;
        const uint32_t x = morton3D_invert(mo), y = morton3D_invert(mo >> 1), z = morton3D_invert(mo >> 2);
        const float *g = sigmas + (size_t)cas * H3;
        // tmp_grid = sigma * density_scale, then the 6-neighbour max of raymarching.cu:304-341
        float t = g[mo] * density_scale;
        if (x + 1 < H) t = fmaxf(t, g[morton3D(x + 1, y, z)] * density_scale);
        if (x > 0) t = fmaxf(t, g[morton3D(x - 1, y, z)] * density_scale);
        if (y + 1 < H) t = fmaxf(t, g[morton3D(x, y + 1, z)] * density_scale);
        if (y > 0) t = fmaxf(t, g[morton3D(x, y - 1, z)] * density_scale);
        if (z + 1 < H) t = fmaxf(t, g[morton3D(x, y, z + 1)] * density_scale);
        if (z > 0) t = fmaxf(t, g[morton3D(x, y, z - 1)] * density_scale);
        float v = grid[i];
        if (v >= 0.0f && t >= 0.0f) {        // valid_mask; untrained cells (-1) keep their mark
            v = fmaxf(v * decay, t);
            grid[i] = v;
        }
        clamped = v > 0.0f ? v : 0.0f;       // density_grid.clamp(min=0)
    }
    // sum in double: 2^21 floats add up exactly enough for the mean to round to the same float in any order
    double s = (double)clamped;
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        double b = 0.0;
        for (int w = 0; w < kOccBlock / kWave; w++) b += red[w];
        partial[blockIdx.x] = b;
    }
}

// The partial sums added up in index order -> mean and threshold.  A launch of its own (one workgroup): the "last workgroup to
// arrive adds them up" form needs a release fence per workgroup, and on this chip that fence is a write-back of the XCD's L2 --
// 8 192 of them made k_occ_update 243 us long; the kernel boundary orders the partials for free (k_occ_update 34 us + this).
__global__ void __launch_bounds__(kOccBlock)
k_occ_mean(const double *__restrict__ partial, uint32_t blocks, uint32_t total, float density_thresh, float *__restrict__ stats) {
    __shared__ double red[kOccBlock / kWave];
    double acc = 0.0;
    for (uint32_t b = threadIdx.x; b < blocks; b += kOccBlock) acc += partial[b];
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double sum = 0.0;
        for (int w = 0; w < kOccBlock / kWave; w++) sum += red[w];
        const float mean = (float)(sum / (double)total);
        stats[0] = mean;
        stats[1] = fminf(mean, density_thresh);                     // density_thresh = min(mean_density, self.density_thresh)
    }
}

// raymarching.cu:267-300 with the threshold in device memory
__global__ void __launch_bounds__(kOccBlock)
k_occ_pack(const float *__restrict__ grid, uint32_t N, const float *__restrict__ stats, uint8_t *__restrict__ bitfield) {
    const uint32_t n = blockIdx.x * kOccBlock + threadIdx.x;
    if (n >= N) return;
    const float thresh = stats[1];
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

// ---- untrained cells (renderer.py:318-379) -----------------------------------------------------------------------
// One lane per (cascade, cell), poses staged in LDS 64 at a time; a cell is "seen" when some camera has it in front
// (cam_z > 0) and inside the frustum widened by one cell.  The reference counts the cameras and then tests count == 0;
// the first camera that sees the cell settles it here.
constexpr int kPoseChunk = 64;

__global__ void __launch_bounds__(kOccBlock)
k_mark_untrained(const float *__restrict__ poses, uint32_t n_poses, uint32_t pose_stride, float cx_fx, float cy_fy, uint32_t C,
                 uint32_t H, CascadeConsts cc, float *__restrict__ grid) {
    __shared__ float sp[kPoseChunk][12];
    const uint32_t H3 = H * H * H, total = C * H3;
    const uint32_t i = blockIdx.x * kOccBlock + threadIdx.x;
    const bool in_range = i < total;
    const uint32_t cas = in_range ? i / H3 : 0u, mo = i - cas * H3;
    float w[3] = {0.0f, 0.0f, 0.0f};
    float margin = 0.0f;
    if (in_range) {
        const uint32_t c[3] = {morton3D_invert(mo), morton3D_invert(mo >> 1), morton3D_invert(mo >> 2)};
        const float hm1 = (float)(H - 1);
#pragma unroll
        for (int d = 0; d < 3; d++) w[d] = ((2.0f * (float)c[d]) / hm1 - 1.0f) * cc.scale[cas];   // cas_world_xyzs
        margin = cc.half[cas] * 2.0f;                                                            // half_grid_size * 2
    }
    bool seen = false;
    for (uint32_t p0 = 0; p0 < n_poses; p0 += kPoseChunk) {
        const uint32_t np = n_poses - p0 < kPoseChunk ? n_poses - p0 : kPoseChunk;
        __syncthreads();
        for (uint32_t e = threadIdx.x; e < np * 12; e += kOccBlock) {
            const uint32_t p = e / 12, k = e - p * 12;                     // rows 0..2 of the 4x4 (or 3x4) matrix
            sp[p][k] = poses[(size_t)(p0 + p) * pose_stride + k];
        }
        __syncthreads();
        if (in_range && !seen) {
            for (uint32_t p = 0; p < np && !seen; p++) {
                const float *M = sp[p];
                const float dx = w[0] - M[3], dy = w[1] - M[7], dz = w[2] - M[11];      // world - t
                // (world - t) @ R: component j = sum_k d_k R[k][j]
                const float camx = dx * M[0] + dy * M[4] + dz * M[8];
                const float camy = dx * M[1] + dy * M[5] + dz * M[9];
                const float camz = dx * M[2] + dy * M[6] + dz * M[10];
                seen = camz > 0.0f && fabsf(camx) < cx_fx * camz + margin && fabsf(camy) < cy_fy * camz + margin;
            }
        }
    }
    if (in_range && !seen) grid[i] = -1.0f;
}

// ---- torso grid (renderer.py:451-490) ----------------------------------------------------------------------------
// point i = (column x = i % H, row y = i / H) of the H x H grid -- the transposed index of renderer.py:472
__global__ void __launch_bounds__(kOccBlock)
k_torso_points(uint32_t H, float scale, float half, const float *__restrict__ noise, uint32_t seed, float *__restrict__ xys) {
    const uint32_t i = blockIdx.x * kOccBlock + threadIdx.x;
    if (i >= H * H) return;
    const uint32_t c[2] = {i % H, i / H};
    const float hm1 = (float)(H - 1);
#pragma unroll
    for (int d = 0; d < 2; d++) {
        const float base = ((2.0f * (float)c[d]) / hm1 - 1.0f) * scale;   // xys * (1 - half_grid_size)
        const float u = noise ? noise[(size_t)i * 2 + d] : hash_u01(seed, i * 2u + (uint32_t)d);
        xys[(size_t)i * 2 + d] = base + (u * 2.0f - 1.0f) * half;
    }
}

// F.max_pool2d(k = 5, s = 1, p = 2) + decayed max + mean; ONE workgroup, alphas staged in LDS (H <= 128: 64 KB)
constexpr int kTorsoUpdThreads = 1024;
__global__ void __launch_bounds__(kTorsoUpdThreads)
k_torso_update(const float *__restrict__ alphas, float *__restrict__ grid, uint32_t H, float decay, float *__restrict__ stats) {
    extern __shared__ float tile[];
    __shared__ double red[kTorsoUpdThreads / kWave];
    const uint32_t n = H * H;
    for (uint32_t i = threadIdx.x; i < n; i += kTorsoUpdThreads) tile[i] = alphas[i];
    __syncthreads();
    double s = 0.0;
    for (uint32_t i = threadIdx.x; i < n; i += kTorsoUpdThreads) {
        const int x = (int)(i % H), y = (int)(i / H);
        float m = -INFINITY;
        for (int dy = -2; dy <= 2; dy++)
            for (int dx = -2; dx <= 2; dx++) {
                const int xx = x + dx, yy = y + dy;
                if (xx >= 0 && xx < (int)H && yy >= 0 && yy < (int)H) m = fmaxf(m, tile[yy * (int)H + xx]);
            }
        const float v = fmaxf(grid[i] * decay, m);          // torch.maximum(density_grid_torso * decay, pooled)
        grid[i] = v;
        s += (double)v;
    }
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        double sum = 0.0;
        for (int w = 0; w < kTorsoUpdThreads / kWave; w++) sum += red[w];
        stats[0] = (float)(sum / (double)n);
    }
}

static CascadeConsts cascade_consts(uint32_t C, uint32_t H, double bound) {
    CascadeConsts cc{};
    for (uint32_t c = 0; c < C && c < 16; c++) {
        double b = (double)(1u << c);                 // bound = min(2 ** cas, self.bound)
        if (b > bound) b = bound;
        const double half = b / (double)H;            // half_grid_size = bound / self.grid_size
        cc.scale[c] = (float)(b - half);
        cc.half[c] = (float)half;
    }
    return cc;
}

}  // namespace rn

using namespace rn;

extern "C" {

size_t rn_occupancy_workspace(uint32_t C, uint32_t H) {
    const size_t blocks = div_up(C * H * H * H, kOccBlock);
    return blocks * sizeof(double) + 64;              // partial sums | pad
}

int rn_occupancy_points(uint32_t C, uint32_t H, float bound, const float *noise, uint32_t seed, float *xyzs, rn_stream_t stream) {
    RN_REQUIRE(xyzs && C >= 1 && C <= 16 && H >= 2 && H <= 1024, "occupancy_points: bad arguments");
    const uint32_t total = C * H * H * H;
    hipLaunchKernelGGL(k_occ_points, dim3(div_up(total, kOccBlock)), dim3(kOccBlock), 0, as_stream(stream), C, H,
                       cascade_consts(C, H, (double)bound), noise, seed, xyzs);
    return check_launch("occupancy_points");
}

int rn_occupancy_update(const float *sigmas, float density_scale, float *density_grid, uint32_t C, uint32_t H, float decay,
                        float density_thresh, uint8_t *bitfield, float *stats, void *workspace, rn_stream_t stream) {
    RN_REQUIRE(sigmas && density_grid && bitfield && stats && workspace, "occupancy_update: null pointer");
    RN_REQUIRE(C >= 1 && C <= 16 && H >= 2 && H <= 1024 && (H * H * H) % 8 == 0, "occupancy_update: bad C / H");
    RN_REQUIRE(((uintptr_t)density_grid & 15u) == 0 && ((uintptr_t)workspace & 7u) == 0, "occupancy_update: alignment");
    const uint32_t total = C * H * H * H, blocks = div_up(total, kOccBlock);
    double *partial = static_cast<double *>(workspace);
    hipStream_t s = as_stream(stream);
    hipLaunchKernelGGL(k_occ_update, dim3(blocks), dim3(kOccBlock), 0, s, sigmas, density_scale, density_grid, C, H, decay, partial);
    hipLaunchKernelGGL(k_occ_mean, dim3(1), dim3(kOccBlock), 0, s, partial, blocks, total, density_thresh, stats);
    hipLaunchKernelGGL(k_occ_pack, dim3(div_up(total / 8, kOccBlock)), dim3(kOccBlock), 0, s, density_grid, total / 8, stats, bitfield);
    return check_launch("occupancy_update");
}

int rn_mark_untrained_grid(const float *poses, uint32_t n_poses, uint32_t pose_stride, double fx, double fy, double cx, double cy,
                           uint32_t C, uint32_t H, float bound, float *density_grid, rn_stream_t stream) {
    RN_REQUIRE(poses && density_grid && n_poses >= 1 && pose_stride >= 12, "mark_untrained_grid: bad arguments");
    RN_REQUIRE(C >= 1 && C <= 16 && H >= 2 && H <= 1024, "mark_untrained_grid: bad C / H");
    const uint32_t total = C * H * H * H;
    // cx / fx and cy / fy are Python floats (double) that meet a float32 tensor: rounded to float once (renderer.py:368-369)
    hipLaunchKernelGGL(k_mark_untrained, dim3(div_up(total, kOccBlock)), dim3(kOccBlock), 0, as_stream(stream), poses, n_poses,
                       pose_stride, (float)(cx / fx), (float)(cy / fy), C, H,
                       cascade_consts(C, H, (double)bound), density_grid);
    return check_launch("mark_untrained_grid");
}

int rn_torso_grid_points(uint32_t H, const float *noise, uint32_t seed, float *xys, rn_stream_t stream) {
    RN_REQUIRE(xys && H >= 2 && H <= 1024, "torso_grid_points: bad arguments");
    const double half = 1.0 / (double)H;                                  // half_grid_size = 1 / self.grid_size
    hipLaunchKernelGGL(k_torso_points, dim3(div_up(H * H, kOccBlock)), dim3(kOccBlock), 0, as_stream(stream), H, (float)(1.0 - half),
                       (float)half, noise, seed, xys);
    return check_launch("torso_grid_points");
}

int rn_torso_grid_update(const float *alphas, float *density_grid_torso, uint32_t H, float decay, float *stats, rn_stream_t stream) {
    RN_REQUIRE(alphas && density_grid_torso && stats, "torso_grid_update: null pointer");
    RN_REQUIRE(H >= 2 && H <= 128, "torso_grid_update: H=%u out of range (2..128)", H);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_torso_update), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
        attr_set = true;
    }
    hipLaunchKernelGGL(k_torso_update, dim3(1), dim3(kTorsoUpdThreads), H * H * sizeof(float), as_stream(stream), alphas,
                       density_grid_torso, H, decay, stats);
    return check_launch("torso_grid_update");
}

uint32_t rn_hash_u01_bits(uint32_t seed, uint32_t idx) { return mix32(mix32(idx) ^ seed) >> 8; }

}  // extern "C"
