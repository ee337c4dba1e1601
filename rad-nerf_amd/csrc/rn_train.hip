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

__global__ void __launch_bounds__(kAdamBlock) k_adam(AdamArgs a, float beta1, float beta2, float eps, const float *__restrict__ corr,
                                                     const float *__restrict__ lr_dev) {
    // which tensor does this workgroup belong to?  (<= 48 entries: a scan in scalar registers)
    uint32_t t = 0;
    while (t + 1 < a.count && blockIdx.x >= a.first_block[t + 1]) t++;
    const uint32_t base = (blockIdx.x - a.first_block[t]) * kAdamChunk + threadIdx.x * kAdamPerThread;
    const uint32_t n = a.n[t];
    if (base >= n) return;
    const float bc1 = corr[0], bc2_sqrt = corr[1];
    // lr_dev (nullable): the tensors' learning rates in device memory, read at run time -- a captured launch then follows a
    // learning-rate schedule (main.py:219 LambdaLR) instead of replaying the value it was captured with
    const float step_size = (lr_dev ? lr_dev[t] : a.lr[t]) / bc1;
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

// ---- elementwise glue of the training step, one kernel per direction instead of 3 - 15 PyTorch kernels --------------------
// NeRFNetwork.forward between sigma_net and color_net (nerf/network.py:266-276): sigma = trunc_exp(h[:, 0]) (activation.py:5-17:
// exp forward, g * exp(clamp(x, -15, 15)) backward), geo_feat = h[:, 1:], color input = cat[SH(d), geo_feat].
__global__ void __launch_bounds__(256) k_head_mid_fwd(const float *__restrict__ h, const float *__restrict__ enc_d, uint32_t M, uint32_t n_sh,
                                                      float *__restrict__ sigma, float *__restrict__ x_color) {
    const uint32_t ld = n_sh + 64u;
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;        // one element of x_color per thread
    if (i >= M * ld) return;
    const uint32_t b = i / ld, c = i - b * ld;
    x_color[i] = c < n_sh ? enc_d[(size_t)b * n_sh + c] : h[(size_t)b * 65u + 1u + (c - n_sh)];
    if (c == 0) sigma[b] = expf(h[(size_t)b * 65u]);
}

__global__ void __launch_bounds__(256) k_head_mid_bwd(const float *__restrict__ h, const float *__restrict__ g_sigma,
                                                      const float *__restrict__ g_xcolor, uint32_t M, uint32_t n_sh, float *__restrict__ g_h) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;        // one element of g_h [M, 65] per thread
    if (i >= M * 65u) return;
    const uint32_t b = i / 65u, c = i - b * 65u;
    float v;
    if (c == 0) {
        const float x = h[i];
        v = g_sigma[b] * expf(fminf(fmaxf(x, -15.0f), 15.0f));
    } else {
        v = g_xcolor[(size_t)b * (n_sh + 64u) + n_sh + (c - 1u)];
    }
    g_h[i] = v;
}

// ambient.abs().sum(-1) for the 2-wide ambient coordinates (nerf/renderer.py:216) and its backward (sign, 0 at 0)
__global__ void __launch_bounds__(256) k_abs_sum2(const float2 *__restrict__ a, uint32_t M, float *__restrict__ out) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i < M) { const float2 v = a[i]; out[i] = fabsf(v.x) + fabsf(v.y); }
}
__global__ void __launch_bounds__(256) k_abs_sum2_bwd(const float2 *__restrict__ a, const float *__restrict__ g, uint32_t M, float2 *__restrict__ ga) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= M) return;
    const float2 v = a[i];
    const float gi = g[i];
    auto sgn = [](float x) { return x > 0.0f ? 1.0f : (x < 0.0f ? -1.0f : 0.0f); };
    ga[i] = make_float2(gi * sgn(v.x), gi * sgn(v.y));
}

// Trainer.train_step's loss for the head (nerf/utils.py:772-803) and its gradient in one workgroup:
//   loss = mean_n mean_c (pred - target)^2 + 1e-4 mean_n H(clamp(ws, 1e-5, 1 - 1e-5)) + w_amb mean_n (ambient_n * (1 - face_n))
//   H(a) = -a log2 a - (1 - a) log2(1 - a)
constexpr int kLossThreads = 1024;
__global__ void __launch_bounds__(kLossThreads) k_train_loss(const float *__restrict__ pred, const float *__restrict__ target,
                                                             const float *__restrict__ ws, const float *__restrict__ ambient,
                                                             const float *__restrict__ face, const float *__restrict__ w_amb, uint32_t N,
                                                             float *__restrict__ loss, float *__restrict__ g_pred, float *__restrict__ g_ws,
                                                             float *__restrict__ g_amb) {
    __shared__ double red[kLossThreads / kWave];
    const float wa = w_amb[0];
    const float inv_n = 1.0f / (float)N;
    double acc = 0.0;
    for (uint32_t n = threadIdx.x; n < N; n += kLossThreads) {
        float mse = 0.0f;
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const float d = pred[n * 3 + c] - target[n * 3 + c];
            mse += d * d;
            g_pred[n * 3 + c] = 2.0f * d * (inv_n / 3.0f);
        }
        const float w = ws[n];
        const float a = fminf(fmaxf(w, 1e-5f), 1.0f - 1e-5f);
        const float la = log2f(a), lb = log2f(1.0f - a);
        const float ent = -a * la - (1.0f - a) * lb;
        const bool inside = w >= 1e-5f && w <= 1.0f - 1e-5f;        // clamp passes the gradient on [min, max]
        // d/da [-a log2 a - (1 - a) log2 (1 - a)] = -log2 a + log2 (1 - a)   (the 1/ln 2 terms cancel)
        g_ws[n] = inside ? 1e-4f * inv_n * (lb - la) : 0.0f;
        const float keep = 1.0f - face[n];
        g_amb[n] = wa * inv_n * keep;
        acc += (double)(mse / 3.0f) * inv_n + 1e-4 * (double)ent * inv_n + (double)wa * (double)(ambient[n] * keep) * inv_n;
    }
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int w = 0; w < kLossThreads / kWave; w++) t += red[w];
        loss[0] = (float)t;
    }
}

}  // namespace rn

using namespace rn;

extern "C" {

int rn_head_mid_forward(const float *h, const float *enc_d, uint32_t M, uint32_t n_sh, float *sigma, float *x_color, rn_stream_t stream) {
    if (M == 0) return RN_OK;
    RN_REQUIRE(h && enc_d && sigma && x_color && n_sh <= 64, "head_mid_forward: null pointer / n_sh > 64");
    hipLaunchKernelGGL(k_head_mid_fwd, dim3(div_up(M * (n_sh + 64u), 256)), dim3(256), 0, as_stream(stream), h, enc_d, M, n_sh, sigma, x_color);
    return check_launch("head_mid_forward");
}

int rn_head_mid_backward(const float *h, const float *grad_sigma, const float *grad_x_color, uint32_t M, uint32_t n_sh, float *grad_h,
                         rn_stream_t stream) {
    if (M == 0) return RN_OK;
    RN_REQUIRE(h && grad_sigma && grad_x_color && grad_h && n_sh <= 64, "head_mid_backward: null pointer / n_sh > 64");
    hipLaunchKernelGGL(k_head_mid_bwd, dim3(div_up(M * 65u, 256)), dim3(256), 0, as_stream(stream), h, grad_sigma, grad_x_color, M, n_sh, grad_h);
    return check_launch("head_mid_backward");
}

int rn_abs_sum2_forward(const float *a, uint32_t M, float *out, rn_stream_t stream) {
    if (M == 0) return RN_OK;
    RN_REQUIRE(a && out && ((uintptr_t)a & 7u) == 0, "abs_sum2_forward: null / unaligned pointer");
    hipLaunchKernelGGL(k_abs_sum2, dim3(div_up(M, 256)), dim3(256), 0, as_stream(stream), reinterpret_cast<const float2 *>(a), M, out);
    return check_launch("abs_sum2_forward");
}

int rn_abs_sum2_backward(const float *a, const float *grad_out, uint32_t M, float *grad_a, rn_stream_t stream) {
    if (M == 0) return RN_OK;
    RN_REQUIRE(a && grad_out && grad_a && ((uintptr_t)a & 7u) == 0 && ((uintptr_t)grad_a & 7u) == 0, "abs_sum2_backward: null / unaligned pointer");
    hipLaunchKernelGGL(k_abs_sum2_bwd, dim3(div_up(M, 256)), dim3(256), 0, as_stream(stream), reinterpret_cast<const float2 *>(a), grad_out, M,
                       reinterpret_cast<float2 *>(grad_a));
    return check_launch("abs_sum2_backward");
}

int rn_train_loss(const float *pred, const float *target, const float *weights_sum, const float *ambient, const float *face,
                  const float *w_amb, uint32_t N, float *loss, float *grad_pred, float *grad_weights_sum, float *grad_ambient,
                  rn_stream_t stream) {
    RN_REQUIRE(N > 0, "train_loss: N must be positive");
    RN_REQUIRE(pred && target && weights_sum && ambient && face && w_amb && loss && grad_pred && grad_weights_sum && grad_ambient,
               "train_loss: null pointer");
    hipLaunchKernelGGL(k_train_loss, dim3(1), dim3(kLossThreads), 0, as_stream(stream), pred, target, weights_sum, ambient, face, w_amb, N,
                       loss, grad_pred, grad_weights_sum, grad_ambient);
    return check_launch("train_loss");
}

int rn_adam_step(const rn_adam_tensor_t *tensors, uint32_t count, float beta1, float beta2, float eps, int32_t *step, float *corr,
                 rn_stream_t stream) {
    return rn_adam_step_lr(tensors, count, beta1, beta2, eps, step, corr, nullptr, stream);
}

int rn_adam_step_lr(const rn_adam_tensor_t *tensors, uint32_t count, float beta1, float beta2, float eps, int32_t *step, float *corr,
                    const float *lr_dev, rn_stream_t stream) {
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
        if (blocks) hipLaunchKernelGGL(k_adam, dim3(blocks), dim3(kAdamBlock), 0, s, a, beta1, beta2, eps, corr, lr_dev ? lr_dev + at : nullptr);
    }
    return check_launch("adam_step");
}

}  // extern "C"
