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

// get_bg_coords (nerf/utils.py:240-245): [H*W, 2] in [-1, 1], component 0 along the rows -- arange / (n - 1) * 2 - 1 in fp32
__global__ void __launch_bounds__(256) k_bg_coords(uint32_t H, uint32_t W, float *__restrict__ out) {
    const uint32_t n = blockIdx.x * 256 + threadIdx.x;
    if (n >= H * W) return;
    const uint32_t r = n / W, c = n - r * W;
    out[2 * (size_t)n] = (float)r / (float)(H - 1u) * 2.0f - 1.0f;
    out[2 * (size_t)n + 1] = (float)c / (float)(W - 1u) * 2.0f - 1.0f;
}

// convert_poses (nerf/utils.py:231-237): cam2world [n, 4, 4] -> (XYZ euler angles of the rotation, translation) [n, 6];
// matrix_to_euler_angles(R, 'XYZ') (:130-169) = (atan2(-R12, R22), asin(R02), atan2(-R01, R00))
__global__ void __launch_bounds__(64) k_convert_poses(const float *__restrict__ poses, uint32_t n, float *__restrict__ out) {
    const uint32_t i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    const float *m = poses + (size_t)i * 16;
    float *o = out + (size_t)i * 6;
    o[0] = atan2f(-m[1 * 4 + 2], m[2 * 4 + 2]);
    o[1] = asinf(m[0 * 4 + 2]);
    o[2] = atan2f(-m[0 * 4 + 1], m[0 * 4 + 0]);
    o[3] = m[3];
    o[4] = m[7];
    o[5] = m[11];
}

}  // namespace rn

using namespace rn;

extern "C" int rn_get_bg_coords(uint32_t H, uint32_t W, float *bg_coords, rn_stream_t stream) {
    RN_REQUIRE(H >= 2 && W >= 2 && (uint64_t)H * W < (1ull << 31) && bg_coords, "get_bg_coords: H, W >= 2 and a destination are required");
    hipLaunchKernelGGL(k_bg_coords, dim3(div_up(H * W, 256)), dim3(256), 0, as_stream(stream), H, W, bg_coords);
    return check_launch("get_bg_coords");
}

extern "C" int rn_convert_poses(const float *poses, uint32_t n, float *poses6, rn_stream_t stream) {
    if (n == 0) return RN_OK;
    RN_REQUIRE(poses && poses6, "convert_poses: null pointer");
    hipLaunchKernelGGL(k_convert_poses, dim3(div_up(n, 64)), dim3(64), 0, as_stream(stream), poses, n, poses6);
    return check_launch("convert_poses");
}

extern "C" int rn_get_rays(const float *pose, float fx, float fy, float cx, float cy, uint32_t H, uint32_t W, float *rays_o,
                           float *rays_d, rn_stream_t stream) {
    if (H == 0 || W == 0) return RN_OK;
    RN_REQUIRE(pose && rays_o && rays_d, "get_rays: null pointer");
    RN_REQUIRE(fx != 0.0f && fy != 0.0f && (uint64_t)H * W < (1ull << 31), "get_rays: bad intrinsics / image size");
    hipLaunchKernelGGL(k_get_rays, dim3(div_up(H * W, 256)), dim3(256), 0, as_stream(stream), pose, fx, fy, cx, cy, H, W, rays_o,
                       rays_d);
    return check_launch("get_rays");
}
