// rn_mlp.hip -- forward, backward-data and weight-gradient kernels of the path's per-sample MLPs for TRAINING (gfx950).
//
// What is computed: nerf/network.py:69-88 (`MLP`: bias-free nn.Linear stack, ReLU between layers, width 64) and its
// autograd, as the training step of nerf/utils.py:718-806 runs it on ~60 k samples per step for ambient_net (96 -> 64 ->
// 64 -> 2), sigma_net (65 -> 64 -> 64 -> 65) and color_net (84 -> 64 -> 3).  Stock PyTorch needs per layer a GEMM, a ReLU,
// a ReLU-backward, two more GEMMs (one of them a 60 k-long reduction into a 64 x 96 result) and the copies between them:
// ~150 launches per step for 10 GFLOP.  Here an MLP is three launches:
//
//   k_mlp_fwd    one wave = 32 samples; v_mfma_f32_32x32x2_f32 with the sample on the lane and the output row on the
//                register index (rn_fused.hip's scheme), so a layer's accumulators ARE the next layer's B operand; the
//                post-ReLU hidden activations are saved in that native layout (coalesced 256-B rows per register).
//   k_mlp_bwd    the same machine run backwards: dX = W^T dY is a forward layer with the transposed weight image; ReLU
//                masks come from the saved activations; the pre-activation gradients dZ are saved in the native layout.
//   k_mlp_wgrad  dW = dZ X^T for every layer of the MLP in ONE launch: a wave stages a 32-sample tile of dZ and of X
//                transposed in LDS ([feature][sample], stride 33), the sample index becomes the k of the MFMA, and the
//                accumulators (one 32 x 32 block of dW each) stay in registers over the wave's tiles; partial sums per
//                wave go to a workspace, k_mlp_wreduce adds them up into the nn.Linear [out, in] layout.
//
// The narrow output rows (ambient 2, sigma 1, rgb 3) are VALU dot products over the accumulator registers, as in the
// inference kernel.  fp32 throughout (exact products, fp32 accumulation): results equal torch's up to summation order.
#include "rn_common.h"

#include <stdlib.h>

#include "../../include/radnerf_fused.h"

namespace rn {
namespace mlp {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kThreads = 256, kWaves = kThreads / kWave;
constexpr int kStepF = 128;  // floats per MFMA step of a 64-row layer: [2 h][32 j][2 row tiles]
constexpr int kStepT = 256;  // floats per MFMA step of the input-gradient layer: [2 h][32 j][4 row tiles (3 used)]
constexpr int kTileFloats = 2048;  // native tile: [2 rt][16 r][64 lanes]

__host__ __device__ constexpr int rowmap(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }
__host__ __device__ constexpr int kmap(int s, int h) { return 32 * (s >> 4) + rowmap(s & 15, h); }

struct Acc32 {
    f32x16 v[2];
};

__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); }

__device__ __forceinline__ void acc_zero(Acc32 &a) {
#pragma unroll
    for (int rt = 0; rt < 2; rt++)
#pragma unroll
        for (int r = 0; r < 16; r++) a.v[rt][r] = 0.0f;
}

__device__ __forceinline__ void step32(Acc32 &a, const float *wl, int s, int lane_off, float b) {
    const float2 w = *reinterpret_cast<const float2 *>(wl + s * kStepF + lane_off);
    a.v[0] = mfma32(w.x, b, a.v[0]);
    a.v[1] = mfma32(w.y, b, a.v[1]);
}

__device__ __forceinline__ void layer_from_acc(Acc32 &out, const Acc32 &in, const float *wl, int lane_off) {
#pragma unroll
    for (int s = 0; s < 32; s++) step32(out, wl, s, lane_off, in.v[s >> 4][s & 15]);
}

__device__ __forceinline__ void tile_store(float *__restrict__ dst, const Acc32 &a, int lane) {
#pragma unroll
    for (int rt = 0; rt < 2; rt++)
#pragma unroll
        for (int r = 0; r < 16; r++) dst[(rt * 16 + r) * 64 + lane] = a.v[rt][r];
}

__device__ __forceinline__ void tile_load(const float *__restrict__ src, Acc32 &a, int lane) {
#pragma unroll
    for (int rt = 0; rt < 2; rt++)
#pragma unroll
        for (int r = 0; r < 16; r++) a.v[rt][r] = src[(rt * 16 + r) * 64 + lane];
}

// ---- weight images ---------------------------------------------------------------------------------------------------
// forward:    L0 [in_pad / 2 steps] | L1 [32 steps] (3 layers) | last, wide rows [32 steps] (out >= 64) | last, narrow rows
//             [NN][2 h][32]  -- MFMA layers as [step][h][j][rt] = W[32 rt + j][k(step, h)]
// transposed: last wide^T [32 steps] | L1^T [32 steps] | L0^T [32 steps][2 h][32 j][4 rt] (rows = input features)
struct Dims {
    uint32_t in_dim, in_pad, out_dim, n_layers, nn, wide;   // nn = narrow rows (out_dim % 64 or out_dim), wide = out_dim >= 64
    uint32_t ld0;                                           // row stride of w0 (>= in_dim: the MLP may read a column block of a wider nn.Linear)
    __host__ __device__ uint32_t s0() const { return in_pad / 2; }
    __host__ __device__ uint32_t off_l1() const { return s0() * kStepF; }
    __host__ __device__ uint32_t off_lw() const { return off_l1() + (n_layers == 3 ? 32u * kStepF : 0u); }
    __host__ __device__ uint32_t off_ln() const { return off_lw() + (wide ? 32u * kStepF : 0u); }
    __host__ __device__ uint32_t fwd_floats() const { return off_ln() + nn * 64u; }
    __host__ __device__ uint32_t off_tw() const { return 0; }
    __host__ __device__ uint32_t off_t1() const { return wide ? 32u * kStepF : 0u; }
    __host__ __device__ uint32_t off_t0() const { return off_t1() + (n_layers == 3 ? 32u * kStepF : 0u); }
    __host__ __device__ uint32_t bwd_floats() const { return off_t0() + 32u * kStepT + nn * 64u; }   // + a copy of the narrow rows
    __host__ __device__ uint32_t off_tn() const { return off_t0() + 32u * kStepT; }
    __host__ __device__ uint32_t rt_in() const { return (in_pad + 31u) / 32u; }
};

static bool make_dims(uint32_t in_dim, uint32_t out_dim, uint32_t n_layers, Dims &d, uint32_t ld0 = 0) {
    d.in_dim = in_dim;
    d.ld0 = ld0 ? ld0 : in_dim;
    d.in_pad = (in_dim + 3u) & ~3u;
    d.out_dim = out_dim;
    d.n_layers = n_layers;
    d.wide = out_dim >= 64 ? 1u : 0u;
    d.nn = d.wide ? out_dim - 64u : out_dim;
    return in_dim >= 1 && d.in_pad <= 96 && (n_layers == 2 || n_layers == 3) && d.nn <= 4 && out_dim >= 1 && out_dim <= 68;
}

__global__ void __launch_bounds__(256) k_mlp_pack(const float *__restrict__ w0, const float *__restrict__ w1,
                                                  const float *__restrict__ wl, Dims d, float *__restrict__ image) {
    const uint32_t e = blockIdx.x * 256 + threadIdx.x;
    const uint32_t nf = d.fwd_floats(), nb = d.bwd_floats();
    if (e >= nf + nb) return;
    float v = 0.0f;
    if (e < nf) {
        if (e < d.off_ln()) {
            const uint32_t base = e < d.off_l1() ? 0u : (e < d.off_lw() ? d.off_l1() : d.off_lw());
            const uint32_t q = e - base, s = q / kStepF, rem = q % kStepF, h = rem / 64, j = (rem % 64) / 2, rt = rem % 2;
            const uint32_t row = 32 * rt + j;
            if (e < d.off_l1()) {                                  // L0: k = 4 (s / 2) + 2 h + (s & 1)
                const uint32_t k = 4 * (s >> 1) + 2 * h + (s & 1);
                v = k < d.in_dim ? w0[row * d.ld0 + k] : 0.0f;
            } else if (e < d.off_lw()) {                           // L1: k = kmap
                v = w1[row * 64 + kmap((int)s, (int)h)];
            } else {                                               // last layer, wide rows nn .. nn + 63
                v = wl[(d.nn + row) * 64 + kmap((int)s, (int)h)];
            }
        } else {                                                   // narrow rows [o][h][q]
            const uint32_t q0 = e - d.off_ln(), o = q0 / 64, h = (q0 % 64) / 32, q = q0 % 32;
            v = wl[o * 64 + 32 * (q >> 4) + rowmap((int)(q & 15), (int)h)];
        }
    } else {
        const uint32_t t = e - nf;
        if (t < d.off_t0()) {                                      // (last wide)^T or L1^T: V[row][k] = W[k][row]
            const bool is_w = d.wide && t < d.off_t1();
            const uint32_t q = t - (is_w ? 0u : d.off_t1()), s = q / kStepF, rem = q % kStepF, h = rem / 64, j = (rem % 64) / 2, rt = rem % 2;
            const uint32_t row = 32 * rt + j, k = (uint32_t)kmap((int)s, (int)h);
            v = is_w ? wl[(d.nn + k) * 64 + row] : w1[k * 64 + row];
        } else if (t < d.off_tn()) {                               // L0^T: rows = input features (in_pad <= 96 -> 3 row tiles of 4)
            const uint32_t q = t - d.off_t0(), s = q / kStepT, rem = q % kStepT, h = rem / 128, j = (rem % 128) / 4, rt = rem % 4;
            const uint32_t row = 32 * rt + j, k = (uint32_t)kmap((int)s, (int)h);
            v = row < d.in_dim ? w0[k * d.ld0 + row] : 0.0f;
        } else {                                                   // narrow rows again (the backward kernel stages only this image)
            const uint32_t q0 = t - d.off_tn(), o = q0 / 64, h = (q0 % 64) / 32, q = q0 % 32;
            v = wl[o * 64 + 32 * (q >> 4) + rowmap((int)(q & 15), (int)h)];
        }
    }
    image[e] = v;
}

struct FwdArgs {
    const float *x;        // [M, in_pad] row-major
    uint32_t M;
    const float *image;    // forward image
    Dims d;
    float *out;            // [M, out_dim]
    float *h0, *h1;        // native tiles [n_tiles][2048]; h1 only with 3 layers
    const float *bias0;    // [64] added to the first layer's pre-activation (nullable): the per-call constant input columns
};

template <int NN, bool WIDE, bool HID2>
__global__ void __launch_bounds__(kThreads) k_mlp_fwd(FwdArgs p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const Dims d = p.d;
    const uint32_t nf = d.fwd_floats();
    for (uint32_t i = threadIdx.x; i < nf / 4; i += kThreads) reinterpret_cast<float4 *>(lds)[i] = reinterpret_cast<const float4 *>(p.image)[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, j = lane & 31, h = lane >> 5;
    const int lane_off = h * 64 + j * 2;
    const uint32_t n_tiles = (p.M + 31u) >> 5;
    for (uint32_t tile = blockIdx.x * kWaves + wave; tile < n_tiles; tile += gridDim.x * kWaves) {
        const uint32_t sample = tile * 32 + j;
        const bool live = sample < p.M;
        Acc32 a, b;
        if (p.bias0) {
#pragma unroll
            for (int rt = 0; rt < 2; rt++)
#pragma unroll
                for (int r = 0; r < 16; r++) a.v[rt][r] = p.bias0[32 * rt + rowmap(r, h)];
        } else {
            acc_zero(a);
        }
        {
            const float *row = p.x + (size_t)sample * d.in_pad + 2 * h;
            const uint32_t nq = d.in_pad / 4;
            for (uint32_t q = 0; q < nq; q++) {
                float2 f = make_float2(0.0f, 0.0f);
                if (live) f = *reinterpret_cast<const float2 *>(row + 4 * q);
                step32(a, lds, 2 * q, lane_off, f.x);
                step32(a, lds, 2 * q + 1, lane_off, f.y);
            }
        }
#pragma unroll
        for (int rt = 0; rt < 2; rt++)
#pragma unroll
            for (int r = 0; r < 16; r++) a.v[rt][r] = fmaxf(a.v[rt][r], 0.0f);
        tile_store(p.h0 + (size_t)tile * kTileFloats, a, lane);
        if constexpr (HID2) {
            acc_zero(b);
            layer_from_acc(b, a, lds + d.off_l1(), lane_off);
#pragma unroll
            for (int rt = 0; rt < 2; rt++)
#pragma unroll
                for (int r = 0; r < 16; r++) a.v[rt][r] = fmaxf(b.v[rt][r], 0.0f);
            tile_store(p.h1 + (size_t)tile * kTileFloats, a, lane);
        }
        // `a` = input of the last layer
        float *orow = p.out + (size_t)sample * d.out_dim;
        {
            const float *wn = lds + d.off_ln();
#pragma unroll
            for (int o = 0; o < NN; o++) {
                float s = 0.0f;
                const float *wo = wn + (o * 2 + h) * 32;
#pragma unroll
                for (int g = 0; g < 8; g++) {
                    const float4 w = *reinterpret_cast<const float4 *>(wo + 4 * g);
                    const int rt = g >> 2, r = (g & 3) * 4;
                    s = __builtin_fmaf(a.v[rt][r + 0], w.x, s);
                    s = __builtin_fmaf(a.v[rt][r + 1], w.y, s);
                    s = __builtin_fmaf(a.v[rt][r + 2], w.z, s);
                    s = __builtin_fmaf(a.v[rt][r + 3], w.w, s);
                }
                s += __shfl_xor(s, 32, 64);
                if (live && h == 0) orow[o] = s;
            }
        }
        if constexpr (WIDE) {
            acc_zero(b);
            layer_from_acc(b, a, lds + d.off_lw(), lane_off);
            if (live) {
#pragma unroll
                for (int rt = 0; rt < 2; rt++)
#pragma unroll
                    for (int r = 0; r < 16; r++) orow[NN + 32 * rt + rowmap(r, h)] = b.v[rt][r];
            }
        }
    }
}

struct BwdArgs {
    const float *grad_out;  // [M, out_dim]
    uint32_t M;
    const float *image;     // transposed image (image + fwd_floats)
    Dims d;
    const float *h0, *h1;
    float *grad_x;          // [M, in_pad]
    float *dz0, *dz1;       // native tiles; dz1 only with 3 layers
};

template <int NN, bool WIDE, bool HID2>
__global__ void __launch_bounds__(kThreads) k_mlp_bwd(BwdArgs p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const Dims d = p.d;
    const uint32_t nb = d.bwd_floats();
    for (uint32_t i = threadIdx.x; i < nb / 4; i += kThreads) reinterpret_cast<float4 *>(lds)[i] = reinterpret_cast<const float4 *>(p.image)[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, j = lane & 31, h = lane >> 5;
    const int lane_off = h * 64 + j * 2;
    const uint32_t n_tiles = (p.M + 31u) >> 5;
    const uint32_t rt_in = d.rt_in();
    for (uint32_t tile = blockIdx.x * kWaves + wave; tile < n_tiles; tile += gridDim.x * kWaves) {
        const uint32_t sample = tile * 32 + j;
        const bool live = sample < p.M;
        const float *grow = p.grad_out + (size_t)sample * d.out_dim;
        Acc32 g, hh, w;
        acc_zero(g);
        if constexpr (WIDE) {   // dH = Ww^T dWide
            acc_zero(w);
            if (live) {
#pragma unroll
                for (int rt = 0; rt < 2; rt++)
#pragma unroll
                    for (int r = 0; r < 16; r++) w.v[rt][r] = grow[NN + 32 * rt + rowmap(r, h)];
            }
            layer_from_acc(g, w, lds + d.off_tw(), lane_off);
        }
        {                       // + Wn^T g_narrow
            const float *wn = lds + d.off_tn();
#pragma unroll
            for (int o = 0; o < NN; o++) {
                const float go = live ? grow[o] : 0.0f;
                const float *wo = wn + (o * 2 + h) * 32;
#pragma unroll
                for (int q = 0; q < 8; q++) {
                    const float4 ww = *reinterpret_cast<const float4 *>(wo + 4 * q);
                    const int rt = q >> 2, r = (q & 3) * 4;
                    g.v[rt][r + 0] = __builtin_fmaf(ww.x, go, g.v[rt][r + 0]);
                    g.v[rt][r + 1] = __builtin_fmaf(ww.y, go, g.v[rt][r + 1]);
                    g.v[rt][r + 2] = __builtin_fmaf(ww.z, go, g.v[rt][r + 2]);
                    g.v[rt][r + 3] = __builtin_fmaf(ww.w, go, g.v[rt][r + 3]);
                }
            }
        }
        // ReLU of the last hidden layer
        tile_load((HID2 ? p.h1 : p.h0) + (size_t)tile * kTileFloats, hh, lane);
#pragma unroll
        for (int rt = 0; rt < 2; rt++)
#pragma unroll
            for (int r = 0; r < 16; r++) g.v[rt][r] = hh.v[rt][r] > 0.0f ? g.v[rt][r] : 0.0f;
        if constexpr (HID2) {
            tile_store(p.dz1 + (size_t)tile * kTileFloats, g, lane);
            acc_zero(w);
            layer_from_acc(w, g, lds + d.off_t1(), lane_off);
            tile_load(p.h0 + (size_t)tile * kTileFloats, hh, lane);
#pragma unroll
            for (int rt = 0; rt < 2; rt++)
#pragma unroll
                for (int r = 0; r < 16; r++) g.v[rt][r] = hh.v[rt][r] > 0.0f ? w.v[rt][r] : 0.0f;
        }
        tile_store(p.dz0 + (size_t)tile * kTileFloats, g, lane);
        // dX = W0^T dZ0: up to three row tiles of input features
        f32x16 x0, x1, x2;
#pragma unroll
        for (int r = 0; r < 16; r++) { x0[r] = 0.0f; x1[r] = 0.0f; x2[r] = 0.0f; }
        const float *t0 = lds + d.off_t0() + h * 128 + j * 4;
#pragma unroll
        for (int s = 0; s < 32; s++) {
            const float4 ww = *reinterpret_cast<const float4 *>(t0 + s * kStepT);
            const float bb = g.v[s >> 4][s & 15];
            x0 = mfma32(ww.x, bb, x0);
            if (rt_in > 1) x1 = mfma32(ww.y, bb, x1);
            if (rt_in > 2) x2 = mfma32(ww.z, bb, x2);
        }
        if (live) {
            float *xr = p.grad_x + (size_t)sample * d.in_pad;
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const uint32_t c = 8 * q + 4 * h;
                if (c < d.in_pad) *reinterpret_cast<float4 *>(xr + c) = make_float4(x0[4 * q], x0[4 * q + 1], x0[4 * q + 2], x0[4 * q + 3]);
                if (32 + c < d.in_pad) *reinterpret_cast<float4 *>(xr + 32 + c) = make_float4(x1[4 * q], x1[4 * q + 1], x1[4 * q + 2], x1[4 * q + 3]);
                if (64 + c < d.in_pad) *reinterpret_cast<float4 *>(xr + 64 + c) = make_float4(x2[4 * q], x2[4 * q + 1], x2[4 * q + 2], x2[4 * q + 3]);
            }
        }
    }
}

// ---- weight gradients -------------------------------------------------------------------------------------------------
// A job: dW[o][i] = sum over samples of A[o][s] * B[i][s]; an operand is either a native tile buffer (64 features) or a
// row-major [M, ld] matrix (features col0 .. col0 + rows - 1).
struct Operand {
    const float *p;
    uint32_t native, ld, rows;  // rows: real feature count (<= 96)
    uint32_t ones;              // row-major only: one more feature that is 1 for every sample (its gradient column is the bias gradient)
};
struct WJob {
    Operand a, b;
    float *partial;             // [parts][96 * 96], this job's slice of the workspace
    float *out;                 // [a.rows, out_ld] nn.Linear layout
    uint32_t out_ld, out_cols;  // columns written (= b real features)
    float *out_bias;            // [a.rows]: the column of b's `ones` feature (nullable)
};
constexpr int kMaxJobs = 3;
struct WArgs {
    WJob job[kMaxJobs];
    uint32_t n_jobs, M, parts;  // parts: workgroups per job
};
constexpr int kTS = 36;         // LDS row stride of a staged tile: [feature][sample parity][sample / 2] -> a lane's 16 k-steps are contiguous
constexpr int kStageFloats = 96 * kTS;   // one operand tile: up to 96 features x 32 samples

// Staging of one operand tile [feature][sample] by the whole workgroup, in two halves so that the global loads of the NEXT
// tile are in flight while the current one is multiplied: fetch() global -> registers, commit() registers -> LDS.
constexpr int kFetch = 12;   // 96 features x 32 samples / 256 threads
struct Fetched {
    float v[kFetch];
};

template <bool NATIVE>
__device__ __forceinline__ void fetch(Fetched &f, const Operand &op, uint32_t tile, uint32_t M) {
    if constexpr (NATIVE) {
        const float *src = op.p + (size_t)tile * kTileFloats;
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
        for (int q = 0; q < 8; q++) f.v[q] = src[(wave * 8 + q) * 64 + lane];
    } else {
        const uint32_t rows_pad = (op.rows + op.ones + 31u) & ~31u, n = rows_pad * 32u;
#pragma unroll
        for (int q = 0; q < kFetch; q++) {   // consecutive threads: consecutive features of one sample
            const uint32_t e = threadIdx.x + (uint32_t)q * kThreads, s = e / rows_pad, o = e - s * rows_pad, sample = tile * 32 + s;
            // unconditional load from a clamped address (a predicated load would be a branch with its own wait); what lies
            // outside the operand is zeroed in commit(), so nothing here waits for the load
            (void)n;
            f.v[q] = op.p[(size_t)(sample < M ? sample : M - 1u) * op.ld + (o < op.rows ? o : 0u)];
        }
    }
}

template <bool NATIVE>
__device__ __forceinline__ void commit(float *t, const Fetched &f, const Operand &op, uint32_t tile, uint32_t M) {
    if constexpr (NATIVE) {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, j = lane & 31, h = lane >> 5;
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const int idx = wave * 8 + q, rt = idx >> 4, r = idx & 15;
            t[(32 * rt + rowmap(r, h)) * kTS + (j & 1) * 16 + (j >> 1)] = f.v[q];
        }
    } else {
        const uint32_t rows_pad = (op.rows + op.ones + 31u) & ~31u, n = rows_pad * 32u;
#pragma unroll
        for (int q = 0; q < kFetch; q++) {
            const uint32_t e = threadIdx.x + (uint32_t)q * kThreads, s = e / rows_pad, o = e - s * rows_pad;
            const bool in_tile = tile * 32 + s < M;
            if (e < n) t[o * kTS + (s & 1u) * 16u + (s >> 1)] = (in_tile && o < op.rows) ? f.v[q] : ((in_tile && op.ones && o == op.rows) ? 1.0f : 0.0f);
        }
    }
}

// One workgroup = one job x one slice of the sample tiles.  The four waves share the staged tiles; the up to 3 x 3 output
// blocks of 32 x 32 are dealt round-robin to the waves (<= 3 each), whose accumulators stay in registers over all tiles.
template <bool A_NATIVE, bool B_NATIVE>
__device__ __forceinline__ void wgrad_job(const WArgs &p, const WJob &job, uint32_t part, float *lds) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i = lane & 31, h = lane >> 5;
    const uint32_t na = (job.a.rows + job.a.ones + 31u) / 32u, nb = (job.b.rows + job.b.ones + 31u) / 32u, n_blocks = na * nb;
    f32x16 acc[3];
#pragma unroll
    for (int q = 0; q < 3; q++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[q][r] = 0.0f;
    uint32_t bx[3], by[3];   // block q of this wave: index wave + 4 q -> (x, y)
#pragma unroll
    for (int q = 0; q < 3; q++) { const uint32_t b = (uint32_t)wave + 4u * q; bx[q] = b / nb; by[q] = b - bx[q] * nb; }
    const uint32_t n_tiles = (p.M + 31u) >> 5;
    float *ta = lds, *tb = lds + kStageFloats;
    // Tiles travel global -> registers two iterations ahead of their use (two register sets, alternating): the HBM latency
    // of a tile that is read exactly once (~3 us) is longer than the ~1 us of MFMA work per tile.
    Fetched fa0, fb0, fa1, fb1;
    const uint32_t stride = p.parts;
    if (part < n_tiles) {
        fetch<A_NATIVE>(fa0, job.a, part, p.M);
        fetch<B_NATIVE>(fb0, job.b, part, p.M);
    }
    if (part + stride < n_tiles) {
        fetch<A_NATIVE>(fa1, job.a, part + stride, p.M);
        fetch<B_NATIVE>(fb1, job.b, part + stride, p.M);
    }
    if (part < n_tiles) {
        commit<A_NATIVE>(ta, fa0, job.a, part, p.M);
        commit<B_NATIVE>(tb, fb0, job.b, part, p.M);
    }
    __syncthreads();
    auto multiply = [&]() {
        // k-step t of the MFMA = samples 2 t + h of the tile: 16 consecutive floats per lane and operand
#pragma unroll
        for (int q = 0; q < 3; q++)
            if ((uint32_t)wave + 4u * q < n_blocks) {
                const float4 *pa = reinterpret_cast<const float4 *>(ta + (32 * bx[q] + i) * kTS + h * 16);
                const float4 *pb = reinterpret_cast<const float4 *>(tb + (32 * by[q] + i) * kTS + h * 16);
                float4 av[4], bv[4];
#pragma unroll
                for (int u = 0; u < 4; u++) { av[u] = pa[u]; bv[u] = pb[u]; }
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    acc[q] = mfma32(av[u].x, bv[u].x, acc[q]);
                    acc[q] = mfma32(av[u].y, bv[u].y, acc[q]);
                    acc[q] = mfma32(av[u].z, bv[u].z, acc[q]);
                    acc[q] = mfma32(av[u].w, bv[u].w, acc[q]);
                }
            }
    };
    // iteration on tile T (in LDS): registers set `cur` is free (it was committed) -> fetch T + 2 strides into it; set
    // `cur ^ 1` holds T + 1 stride, committed after the multiply
    for (uint32_t tile = part; tile < n_tiles; tile += 2 * stride) {
        if (tile + 2 * stride < n_tiles) {
            fetch<A_NATIVE>(fa0, job.a, tile + 2 * stride, p.M);
            fetch<B_NATIVE>(fb0, job.b, tile + 2 * stride, p.M);
        }
        multiply();
        __syncthreads();      // everybody has read this tile
        if (tile + stride < n_tiles) {
            commit<A_NATIVE>(ta, fa1, job.a, tile + stride, p.M);
            commit<B_NATIVE>(tb, fb1, job.b, tile + stride, p.M);
        }
        __syncthreads();
        if (tile + stride >= n_tiles) break;
        if (tile + 3 * stride < n_tiles) {
            fetch<A_NATIVE>(fa1, job.a, tile + 3 * stride, p.M);
            fetch<B_NATIVE>(fb1, job.b, tile + 3 * stride, p.M);
        }
        multiply();
        __syncthreads();
        if (tile + 2 * stride < n_tiles) {
            commit<A_NATIVE>(ta, fa0, job.a, tile + 2 * stride, p.M);
            commit<B_NATIVE>(tb, fb0, job.b, tile + 2 * stride, p.M);
        }
        __syncthreads();
    }
    // partial [row][col] of this workgroup: row = 32 x + rowmap(r, h), col = 32 y + i
    float *dst = job.partial + (size_t)part * (96 * 96);
#pragma unroll
    for (int q = 0; q < 3; q++)
        if ((uint32_t)wave + 4u * q < n_blocks) {
#pragma unroll
            for (int r = 0; r < 16; r++) dst[(32 * bx[q] + rowmap(r, h)) * 96 + 32 * by[q] + i] = acc[q][r];
        }
}

// the operand kinds are compile-time inside a job (three combinations occur: dZ native x x row-major, native x native,
// grad_out row-major x native): no value of the fetch pipeline crosses a data-dependent branch
__global__ void __launch_bounds__(kThreads, 3) k_mlp_wgrad(WArgs p) {
    __shared__ __attribute__((aligned(16))) float lds[2 * kStageFloats];
    const uint32_t jb = blockIdx.x / p.parts, part = blockIdx.x % p.parts;
    if (jb >= p.n_jobs) return;
    const WJob &job = p.job[jb];
    if (job.a.native && job.b.native) wgrad_job<true, true>(p, job, part, lds);
    else if (job.a.native) wgrad_job<true, false>(p, job, part, lds);
    else wgrad_job<false, true>(p, job, part, lds);
}

__global__ void __launch_bounds__(256) k_mlp_wreduce(WArgs p) {
    const uint32_t jb = blockIdx.y;
    if (jb >= p.n_jobs) return;
    const WJob &job = p.job[jb];
    const uint32_t e = blockIdx.x * 256 + threadIdx.x;
    const uint32_t row = e / 96, col = e % 96;
    const bool bias_col = job.out_bias && job.b.ones && col == job.b.rows;
    if (row >= job.a.rows || (col >= job.out_cols && !bias_col)) return;
    const float *src = job.partial + row * 96 + col;
    float s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    uint32_t q = 0;
    for (; q + 8 <= p.parts; q += 8) {
#pragma unroll
        for (int u = 0; u < 8; u++) s[u] += src[(size_t)(q + u) * (96 * 96)];
    }
    for (; q < p.parts; q++) s[0] += src[(size_t)q * (96 * 96)];
    const float total = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
    if (bias_col) job.out_bias[row] = total;
    else job.out[row * job.out_ld + col] = total;
}

static int num_cus() {
    static int n = 0;
    if (!n) {
        int dev = 0;
        (void)hipGetDevice(&dev);
        hipDeviceProp_t prop;
        n = (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
    }
    return n;
}

template <typename K, typename A>
static void launch_with_lds(K kernel, dim3 grid, size_t shm, hipStream_t s, const A &args) {
    if (shm > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
    hipLaunchKernelGGL(kernel, grid, dim3(kThreads), shm, s, args);
}
constexpr uint32_t kWPartsMax = 512;  // workspace is sized for this many partial sums per job
static uint32_t wparts() {            // workgroups (= partial sums) per weight-gradient job: 3 jobs x 256 = 3 workgroups per CU
    static uint32_t n = 0;
    if (!n) {
        const char *e = getenv("RN_MLP_WPARTS");
        const long v = e ? atol(e) : 256;
        n = (uint32_t)(v < 1 ? 1 : (v > (long)kWPartsMax ? (long)kWPartsMax : v));
    }
    return n;
}

}  // namespace mlp
}  // namespace rn

using namespace rn;
using namespace rn::mlp;

extern "C" {

size_t rn_mlp64_image_floats(uint32_t in_dim, uint32_t out_dim, uint32_t n_layers) {
    Dims d;
    if (!make_dims(in_dim, out_dim, n_layers, d)) return 0;
    return (size_t)d.fwd_floats() + d.bwd_floats();
}

size_t rn_mlp64_tile_floats(uint32_t M) { return (size_t)((M + 31u) >> 5) * kTileFloats; }

size_t rn_mlp64_wgrad_workspace(uint32_t n_layers) { return (size_t)n_layers * kWPartsMax * 96 * 96 * sizeof(float); }

int rn_mlp64_pack(const float *w0, uint32_t ld0, const float *w1, const float *w_last, uint32_t in_dim, uint32_t out_dim, uint32_t n_layers,
                  float *image, rn_stream_t stream) {
    Dims d;
    RN_REQUIRE(ld0 >= in_dim, "mlp64_pack: ld0 (row stride of w0) must be >= in_dim");
    RN_REQUIRE(make_dims(in_dim, out_dim, n_layers, d, ld0), "mlp64: unsupported shape (hidden 64, 2 or 3 layers, in <= 96, out <= 4 or 64 .. 68)");
    RN_REQUIRE(w0 && w_last && image && (n_layers == 2 || w1) && ((uintptr_t)image & 15u) == 0, "mlp64_pack: null / unaligned pointer");
    const uint32_t n = d.fwd_floats() + d.bwd_floats();
    hipLaunchKernelGGL(k_mlp_pack, dim3(div_up(n, 256)), dim3(256), 0, as_stream(stream), w0, w1, w_last, d, image);
    return check_launch("mlp64_pack");
}

#define RN_MLP_DISPATCH(KERNEL, grid, shm, stream, args)                                                                          \
    do {                                                                                                                          \
        const bool hid2 = d.n_layers == 3;                                                                                        \
        if (d.wide && d.nn == 1 && hid2) launch_with_lds(KERNEL<1, true, true>, grid, shm, stream, args);                         \
        else if (d.wide && d.nn == 0 && hid2) launch_with_lds(KERNEL<0, true, true>, grid, shm, stream, args);                    \
        else if (!d.wide && d.nn == 2 && hid2) launch_with_lds(KERNEL<2, false, true>, grid, shm, stream, args);                  \
        else if (!d.wide && d.nn == 3 && !hid2) launch_with_lds(KERNEL<3, false, false>, grid, shm, stream, args);                \
        else if (!d.wide && d.nn == 4 && hid2) launch_with_lds(KERNEL<4, false, true>, grid, shm, stream, args);                  \
        else if (!d.wide && d.nn == 1 && hid2) launch_with_lds(KERNEL<1, false, true>, grid, shm, stream, args);                  \
        else {                                                                                                                    \
            ::rn::set_error("mlp64: this (out_dim, n_layers) combination is not instantiated");                                  \
            return RN_ERR_INVALID_ARG;                                                                                            \
        }                                                                                                                         \
    } while (0)

int rn_mlp64_forward(const float *x, uint32_t M, const float *image, const float *bias0, uint32_t in_dim, uint32_t out_dim,
                     uint32_t n_layers, float *out, float *h0, float *h1, rn_stream_t stream) {
    if (M == 0) return RN_OK;
    Dims d;
    RN_REQUIRE(make_dims(in_dim, out_dim, n_layers, d), "mlp64: unsupported shape (hidden 64, 2 or 3 layers, in <= 96, out <= 4 or 64 .. 68)");
    RN_REQUIRE(x && image && out && h0 && (n_layers == 2 || h1), "mlp64_forward: null pointer");
    RN_REQUIRE(((uintptr_t)x & 7u) == 0 && ((uintptr_t)image & 15u) == 0, "mlp64_forward: x must be 8-byte, image 16-byte aligned");
    FwdArgs p{x, M, image, d, out, h0, h1, bias0};
    const uint32_t n_tiles = (M + 31u) >> 5;
    uint32_t blocks = div_up(n_tiles, kWaves);
    const uint32_t cap = (uint32_t)num_cus() * 2u;
    if (blocks > cap) blocks = cap;
    const size_t shm = (size_t)d.fwd_floats() * sizeof(float);
    RN_MLP_DISPATCH(k_mlp_fwd, dim3(blocks), shm, as_stream(stream), p);
    return check_launch("mlp64_forward");
}

int rn_mlp64_backward(const float *grad_out, uint32_t M, const float *image, uint32_t in_dim, uint32_t out_dim, uint32_t n_layers,
                      const float *h0, const float *h1, float *grad_x, float *dz0, float *dz1, rn_stream_t stream) {
    if (M == 0) return RN_OK;
    Dims d;
    RN_REQUIRE(make_dims(in_dim, out_dim, n_layers, d), "mlp64: unsupported shape (hidden 64, 2 or 3 layers, in <= 96, out <= 4 or 64 .. 68)");
    RN_REQUIRE(grad_out && image && h0 && grad_x && dz0 && (n_layers == 2 || (h1 && dz1)), "mlp64_backward: null pointer");
    RN_REQUIRE(((uintptr_t)grad_x & 15u) == 0 && ((uintptr_t)image & 15u) == 0, "mlp64_backward: grad_x / image must be 16-byte aligned");
    BwdArgs p{grad_out, M, image + d.fwd_floats(), d, h0, h1, grad_x, dz0, dz1};
    RN_REQUIRE((d.fwd_floats() & 3u) == 0, "mlp64_backward: internal image alignment");
    const uint32_t n_tiles = (M + 31u) >> 5;
    uint32_t blocks = div_up(n_tiles, kWaves);
    const uint32_t cap = (uint32_t)num_cus() * 2u;
    if (blocks > cap) blocks = cap;
    const size_t shm = (size_t)d.bwd_floats() * sizeof(float);
    RN_MLP_DISPATCH(k_mlp_bwd, dim3(blocks), shm, as_stream(stream), p);
    return check_launch("mlp64_backward");
}

int rn_mlp64_weight_grads(const float *x, const float *grad_out, uint32_t M, uint32_t in_dim, uint32_t out_dim, uint32_t n_layers,
                          const float *h0, const float *h1, const float *dz0, const float *dz1, float *gw0, uint32_t ld0, float *gw1,
                          float *gw_last, float *grad_bias0, void *workspace, rn_stream_t stream) {
    Dims d;
    RN_REQUIRE(make_dims(in_dim, out_dim, n_layers, d), "mlp64: unsupported shape (hidden 64, 2 or 3 layers, in <= 96, out <= 4 or 64 .. 68)");
    RN_REQUIRE(x && grad_out && h0 && dz0 && gw0 && gw_last && workspace && (n_layers == 2 || (h1 && dz1 && gw1)), "mlp64_weight_grads: null pointer");
    RN_REQUIRE(M > 0, "mlp64_weight_grads: M must be positive");
    RN_REQUIRE(ld0 >= in_dim, "mlp64_weight_grads: ld0 (row stride of gw0) must be >= in_dim");
    RN_REQUIRE(!grad_bias0 || d.in_pad + 1u <= 96u, "mlp64_weight_grads: a bias gradient needs in_dim <= 92");
    WArgs p{};
    p.M = M;
    p.parts = wparts();
    float *ws = static_cast<float *>(workspace);
    const size_t per_job = (size_t)kWPartsMax * 96 * 96;
    uint32_t n = 0;
    // L0: dW0 = dZ0 x^T
    p.job[n] = WJob{Operand{dz0, 1u, 0u, 64u, 0u}, Operand{x, 0u, d.in_pad, d.in_pad, grad_bias0 ? 1u : 0u}, ws + n * per_job, gw0, ld0, d.in_dim, grad_bias0};
    n++;
    if (n_layers == 3) {
        p.job[n] = WJob{Operand{dz1, 1u, 0u, 64u, 0u}, Operand{h0, 1u, 0u, 64u, 0u}, ws + n * per_job, gw1, 64u, 64u, nullptr};
        n++;
    }
    p.job[n] = WJob{Operand{grad_out, 0u, d.out_dim, d.out_dim, 0u}, Operand{n_layers == 3 ? h1 : h0, 1u, 0u, 64u, 0u}, ws + n * per_job, gw_last, 64u, 64u, nullptr};
    n++;
    p.n_jobs = n;
    hipLaunchKernelGGL(k_mlp_wgrad, dim3(n * p.parts), dim3(kThreads), 0, as_stream(stream), p);
    hipLaunchKernelGGL(k_mlp_wreduce, dim3(div_up(96 * 96, 256), n), dim3(256), 0, as_stream(stream), p);
    return check_launch("mlp64_weight_grads");
}

}  // extern "C"
