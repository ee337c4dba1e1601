// rn_rays.hip -- full-image ray generation on the device (C ABI: include/radnerf_fused.h, "ray generation").
// What is computed: get_rays, nerf/utils.py:249-333 (N = -1): the step immediately before the render path.
#include "rn_common.h"

#include "../../include/radnerf_fused.h"

namespace rn {

__global__ void __launch_bounds__(256)
k_get_rays(const float *__restrict__ pose, float fx, float fy, float cx, float cy, uint32_t H, uint32_t W,
           float *__restrict__ rays_o, float *__restrict__ rays_d) {
    const uint32_t n = blockIdx.x * 256 + threadIdx.x;
    if (n >= H * W) return;
    const uint32_t r = n / W, c = n - r * W;
    // i = col + 0.5, j = row + 0.5 (:268-270); xs = (i - cx) / fx * zs, ys = (j - cy) / fy * zs, zs = 1 (:320-322)
    const float x = ((float)c + 0.5f - cx) / fx, y = ((float)r + 0.5f - cy) / fy, z = 1.0f;
    const float norm = sqrtf(x * x + y * y + z * z);  // :324
    const float dx = x / norm, dy = y / norm, dz = z / norm;
#pragma unroll
    for (int k = 0; k < 3; k++) {  // rays_d = directions @ R^T (:325): row k of R
        rays_d[(size_t)n * 3 + k] = dx * pose[k * 4] + dy * pose[k * 4 + 1] + dz * pose[k * 4 + 2];
        rays_o[(size_t)n * 3 + k] = pose[k * 4 + 3];  // :327
    }
}

}  // namespace rn

using namespace rn;

extern "C" int rn_get_rays(const float *pose, float fx, float fy, float cx, float cy, uint32_t H, uint32_t W, float *rays_o,
                           float *rays_d, rn_stream_t stream) {
    if (H == 0 || W == 0) return RN_OK;
    RN_REQUIRE(pose && rays_o && rays_d, "get_rays: null pointer");
    RN_REQUIRE(fx != 0.0f && fy != 0.0f && (uint64_t)H * W < (1ull << 31), "get_rays: bad intrinsics / image size");
    hipLaunchKernelGGL(k_get_rays, dim3(div_up(H * W, 256)), dim3(256), 0, as_stream(stream), pose, fx, fy, cx, cy, H, W, rays_o,
                       rays_d);
    return check_launch("get_rays");
}
