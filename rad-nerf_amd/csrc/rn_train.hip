// rn_train.hip -- optimizer update of the training step (gfx950).
//
// What is computed: torch.optim.Adam as the reference configures it (main.py:204: betas (0.9, 0.99), eps 1e-15, no weight
// decay, per-group learning rates from NeRFNetwork.get_params, nerf/network.py:328-357) for ALL parameter tensors of the
// model in one launch: the 49 MB hash table, its two moments and its gradient are each read once and the three updated
// arrays written once (7 x 49 MB, the step's largest HBM stream), the ~35 small tensors ride along in the same grid.
// PyTorch's fused Adam takes one multi-tensor launch per parameter group plus the step-counter updates (~18 launches).
#include "rn_common.h"

#include "../../include/radnerf_fused.h"

namespace rn {

constexpr int kAdamBlock = 256, kAdamPerThread = 4;       // float4 per thread
constexpr int kAdamChunk = kAdamBlock * kAdamPerThread;  // elements per workgroup
constexpr int kAdamMaxTensors = 48;   // pointers travel as kernel arguments (2.1 KB block)

struct AdamArgs {
    float *p[kAdamMaxTensors];
    const float *g[kAdamMaxTensors];
    float *m[kAdamMaxTensors], *v[kAdamMaxTensors];
    uint32_t n[kAdamMaxTensors], first_block[kAdamMaxTensors + 1];
    float lr[kAdamMaxTensors];
    uint32_t count;
};

// step += 1; bias corrections in double, as Python computes them (torch/optim/adam.py: 1 - beta ** step)
__global__ void k_adam_begin(int32_t *step, float *corr, float beta1, float beta2) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const int32_t s = step[0] + 1;
    step[0] = s;
    corr[0] = (float)(1.0 - pow((double)beta1, (double)s));
    corr[1] = (float)sqrt(1.0 - pow((double)beta2, (double)s));
}

__device__ __forceinline__ void adam_one(float &p, float g, float &m, float &v, float beta1, float beta2, float eps, float step_size,
                                         float bc2_sqrt) {
    // torch/optim/adam.py (_single_tensor_adam): exp_avg.lerp_(grad, 1 - beta1); exp_avg_sq = beta2 v + (1 - beta2) g g;
    // denom = sqrt(exp_avg_sq) / sqrt(bias_correction2) + eps; param -= step_size * exp_avg / denom
    m = m + (g - m) * (1.0f - beta1);
    v = v * beta2 + (1.0f - beta2) * g * g;
    const float denom = sqrtf(v) / bc2_sqrt + eps;
    p = p - step_size * (m / denom);
}

__global__ void __launch_bounds__(kAdamBlock) k_adam(AdamArgs a, float beta1, float beta2, float eps, const float *__restrict__ corr) {
    // which tensor does this workgroup belong to?  (<= 48 entries: a scan in scalar registers)
    uint32_t t = 0;
    while (t + 1 < a.count && blockIdx.x >= a.first_block[t + 1]) t++;
    const uint32_t base = (blockIdx.x - a.first_block[t]) * kAdamChunk + threadIdx.x * kAdamPerThread;
    const uint32_t n = a.n[t];
    if (base >= n) return;
    const float bc1 = corr[0], bc2_sqrt = corr[1];
    const float step_size = a.lr[t] / bc1;
    float *p = a.p[t] + base, *m = a.m[t] + base, *v = a.v[t] + base;
    const float *g = a.g[t] + base;
    if (base + 4 <= n && ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(m) |
                           reinterpret_cast<uintptr_t>(v)) & 15u) == 0) {
        float4 P = *reinterpret_cast<float4 *>(p), M = *reinterpret_cast<float4 *>(m), V = *reinterpret_cast<float4 *>(v);
        const float4 G = *reinterpret_cast<const float4 *>(g);
        adam_one(P.x, G.x, M.x, V.x, beta1, beta2, eps, step_size, bc2_sqrt);
        adam_one(P.y, G.y, M.y, V.y, beta1, beta2, eps, step_size, bc2_sqrt);
        adam_one(P.z, G.z, M.z, V.z, beta1, beta2, eps, step_size, bc2_sqrt);
        adam_one(P.w, G.w, M.w, V.w, beta1, beta2, eps, step_size, bc2_sqrt);
        *reinterpret_cast<float4 *>(p) = P; *reinterpret_cast<float4 *>(m) = M; *reinterpret_cast<float4 *>(v) = V;
    } else {
        for (uint32_t i = 0; i < 4 && base + i < n; i++) {
            float P = p[i], M = m[i], V = v[i];
            adam_one(P, g[i], M, V, beta1, beta2, eps, step_size, bc2_sqrt);
            p[i] = P; m[i] = M; v[i] = V;
        }
    }
}

}  // namespace rn

using namespace rn;

extern "C" {

int rn_adam_step(const rn_adam_tensor_t *tensors, uint32_t count, float beta1, float beta2, float eps, int32_t *step, float *corr,
                 rn_stream_t stream) {
    RN_REQUIRE(tensors || count == 0, "adam_step: null tensor list");
    RN_REQUIRE(step && corr, "adam_step: step counter (int32) and a 2-float scratch are required");
    hipStream_t s = as_stream(stream);
    hipLaunchKernelGGL(k_adam_begin, dim3(1), dim3(64), 0, s, step, corr, beta1, beta2);
    for (uint32_t at = 0; at < count; at += kAdamMaxTensors) {
        AdamArgs a{};
        uint32_t blocks = 0, k = 0;
        for (; k < (uint32_t)kAdamMaxTensors && at + k < count; k++) {
            const rn_adam_tensor_t &t = tensors[at + k];
            RN_REQUIRE(t.param && t.grad && t.exp_avg && t.exp_avg_sq, "adam_step: null pointer in tensor %u", at + k);
            a.p[k] = t.param; a.g[k] = t.grad; a.m[k] = t.exp_avg; a.v[k] = t.exp_avg_sq;
            a.n[k] = t.numel; a.lr[k] = t.lr;
            a.first_block[k] = blocks;
            blocks += div_up(t.numel, kAdamChunk);
        }
        a.first_block[k] = blocks;
        a.count = k;
        if (blocks) hipLaunchKernelGGL(k_adam, dim3(blocks), dim3(kAdamBlock), 0, s, a, beta1, beta2, eps, corr);
    }
    return check_launch("adam_step");
}

}  // extern "C"
