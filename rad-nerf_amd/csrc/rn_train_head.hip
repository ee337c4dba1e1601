// rn_train_head.hip -- the per-sample network of the TRAINING step as one forward and one backward kernel (gfx950).
//
// C ABI: include/radnerf_train.h.  What is computed: NeRFNetwork.forward (nerf/network.py:222-283) and its autograd as
// Trainer.train_step runs it on the ~60 k samples of a 4096-ray batch (nerf/utils.py:718-806, nerf/renderer.py:206-223).
// How (MI355X-first; the inference kernel's machine, rn_fused.hip, made differentiable):
//
//  * k_train_fwd: one wavefront owns 32 samples; lane half h gathers level 2 r + h in round r, the two features are the B
//    operands of two v_mfma_f32_32x32x2_f32 steps of both first layers that read enc_x; a layer's accumulators (sample on
//    the lane, output row on the register index) ARE the next layer's B operand.  Everything the backward pass needs is
//    stored in that register layout -- a "native tile" is [register][64 lanes] floats, i.e. every store is a coalesced 256-B
//    row: the post-ReLU hidden activations of the five hidden layers, geo_feat, the grid features (they are operands of the
//    weight gradients) and d enc_w / d ambient of the 2-D grid.
//  * k_train_bwd: the same tile walked back.  dX = W^T dY is a forward layer with the transposed weight image, ReLU masks
//    come from the saved activations, tanh' / sigmoid' / trunc_exp' from the saved outputs.  The gradient with respect to
//    the ambient coordinates (sum over levels of g * dy_dx, gridencoder.cu:342-368) is taken inside the tile.  Feature
//    gradients of both grids leave level-major ([L, M, 2]: one coalesced 256-B row per level and lane half).
//  * k_train_wgrad: dW = dZ X^T for all eight layers in one launch.  A workgroup stages a 32-sample tile of both operands
//    transposed in LDS ([feature][sample]); the sample index is the k of the MFMA; accumulators stay in registers over the
//    workgroup's tiles; per-workgroup partial sums are folded by k_train_wreduce into the nn.Linear layout.  The columns of
//    the per-call constants (audio code, eye, individual code) ride along as one more feature that is 1 for every sample:
//    its gradient column is the bias gradient, from which k_train_const derives the constants' and their columns' gradients.
//  * k_grid_scatter: table gradient.  Float atomics run at the memory side, one request per touched 64-B line per
//    instruction (MI355X_MICROARCH.md, "Global float atomics"), so a workgroup first sums its 128 samples x 2^D corners of
//    one level in an LDS table keyed by the 64-B LINE of the gradient table (8 rows x 2 channels = 16 floats per slot; the
//    two x-neighbours of a corner pair share a line 7 times out of 8, on hashed levels too: the x prime is 1) and then
//    issues the 16 floats of a slot from 16 adjacent lanes -- one request per touched line instead of one per row.
#include "rn_fused_dev.h"

#include <stdlib.h>

#include "../../include/radnerf_train.h"

namespace rn {
namespace th {

constexpr int kStep = 128;   // floats per MFMA step of a 64-row layer: [2 h][32 j][2 row tiles]
// ---- forward image (the inference kernel's layout) ------------------------------------------------------------------
constexpr int F_A0 = 0;                    // ambient L0, enc_x part : 16 steps
constexpr int F_A1 = F_A0 + 16 * kStep;    // ambient L1            : 32 steps
constexpr int F_A2 = F_A1 + 32 * kStep;    // ambient L2 (VALU)     : [2 out][2 h][32]
constexpr int F_S0 = F_A2 + 128;           // sigma L0 (enc_x|enc_w): 32 steps
constexpr int F_S1 = F_S0 + 32 * kStep;    // sigma L1              : 32 steps
constexpr int F_S2 = F_S1 + 32 * kStep;    // sigma L2 rows 1..64   : 32 steps
constexpr int F_S2R = F_S2 + 32 * kStep;   // sigma L2 row 0 (VALU) : [2 h][32]
constexpr int F_C0 = F_S2R + 64;           // color L0 (sh | geo)   : 8 + 32 steps
constexpr int F_C1 = F_C0 + 40 * kStep;    // color L1 (VALU)       : [3 out][2 h][32]
constexpr int kFwd = F_C1 + 192;           // 23936 floats
// ---- transposed image: T[s][h][j][rt] = W[kmap(s, h)][column(32 rt + j)] ---------------------------------------------
constexpr int T_C0 = 0;                    // d geo_feat   = W_col0[:, 16:80]^T dZ_c0
constexpr int T_S2 = T_C0 + 32 * kStep;    // d h_s1       = W_sig2[1:65]^T d geo_feat
constexpr int T_S1 = T_S2 + 32 * kStep;
constexpr int T_S0 = T_S1 + 32 * kStep;    // d [enc_x | enc_w] = W_sig0[:, 0:64]^T dZ_s0, output rows in gather order (below)
constexpr int T_A1 = T_S0 + 32 * kStep;
constexpr int T_A0 = T_A1 + 32 * kStep;    // d enc_x += W_amb0[:, 0:32]^T dZ_a0 : 32 steps x [2 h][32 j]
constexpr int N_C1 = T_A0 + 32 * 64;       // narrow rows again: [3][2 h][32]
constexpr int N_S2R = N_C1 + 192;          // [2 h][32]
constexpr int N_A2 = N_S2R + 64;           // [2][2 h][32]
constexpr int kBwd = N_A2 + 128;           // 22912 floats
constexpr int kBias = 192;
constexpr int kImage = kFwd + kBwd + kBias;

// Output row j of a 32-row tile sits in register r of lane half hh with rowmap(r, hh) == j.  The grid-feature gradients want
// register 2 q + c of lane half hh to be (level 2 q + hh, channel c), the layout the forward gathers in: feature 4 q + 2 hh + c.
__host__ __device__ constexpr int gather_feature(int j) {
    const int hh = (j >> 2) & 1, r = (j & 3) + 4 * (j >> 3);
    return 4 * (r >> 1) + 2 * hh + (r & 1);
}

__global__ void __launch_bounds__(256) k_train_pack(RawW w, const float *__restrict__ enc_a, const float *__restrict__ eye,
                                                    const float *__restrict__ ind_code, const int64_t *__restrict__ ind_index,
                                                    float *__restrict__ image) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= kImage) return;
    if (ind_index) ind_code += (size_t)ind_index[0] * w.ind_dim;   // ind_code = the table individual_codes, row picked on the device
    const int ldA0 = 32 + (int)w.audio_dim, ldS0 = 64 + (int)w.has_eye, ldC0 = 80 + (int)w.ind_dim;
    float v;
    auto mfma_elem = [&](int q, const float *src, int ld, int kind) -> float {
        const int s = q / kStep, rem = q % kStep;
        const int h = rem / 64, j = (rem % 64) / 2, rt = rem % 2;
        const int row = 32 * rt + j;
        int k;
        if (kind == 3) k = 4 * (s >> 1) + 2 * h + (s & 1);       // gather rounds: half h holds level 2 (s / 2) + h
        else if (kind == 1) k = kmap(s, h);                       // previous accumulators
        else k = (s < 8) ? 2 * s + h : 16 + kmap(s - 8, h);      // color L0: sh pairs then geo accumulators
        return src[row * ld + k];
    };
    auto valu_elem = [&](int q0, const float *src) -> float {    // [out][h][q], q = rt * 16 + r
        const int o = q0 / 64, h = (q0 % 64) / 32, q = q0 % 32;
        return src[o * 64 + 32 * (q >> 4) + rowmap(q & 15, h)];
    };
    auto t_elem = [&](int q, const float *src, int ld, int col0, bool gather) -> float {   // transposed 64-row layer
        const int s = q / kStep, rem = q % kStep;
        const int h = rem / 64, j = (rem % 64) / 2, rt = rem % 2;
        const int col = gather ? 32 * rt + gather_feature(j) : 32 * rt + j;
        return src[kmap(s, h) * ld + col0 + col];
    };
    if (e < kFwd) {
        if (e < F_A1) v = mfma_elem(e - F_A0, w.amb_w0, ldA0, 3);
        else if (e < F_A2) v = mfma_elem(e - F_A1, w.amb_w1, 64, 1);
        else if (e < F_S0) v = valu_elem(e - F_A2, w.amb_w2);
        else if (e < F_S1) v = mfma_elem(e - F_S0, w.sig_w0, ldS0, 3);
        else if (e < F_S2) v = mfma_elem(e - F_S1, w.sig_w1, 64, 1);
        else if (e < F_S2R) v = mfma_elem(e - F_S2, w.sig_w2 + 64, 64, 1);   // rows 1..64 = geo_feat
        else if (e < F_C0) v = valu_elem(e - F_S2R, w.sig_w2);               // row 0 = sigma
        else if (e < F_C1) v = mfma_elem(e - F_C0, w.col_w0, ldC0, 2);
        else v = valu_elem(e - F_C1, w.col_w1);
    } else if (e < kFwd + kBwd) {
        const int t = e - kFwd;
        if (t < T_S2) v = t_elem(t - T_C0, w.col_w0, ldC0, 16, false);
        else if (t < T_S1) v = t_elem(t - T_S2, w.sig_w2 + 64, 64, 0, false);
        else if (t < T_S0) v = t_elem(t - T_S1, w.sig_w1, 64, 0, false);
        else if (t < T_A1) v = t_elem(t - T_S0, w.sig_w0, ldS0, 0, true);
        else if (t < T_A0) v = t_elem(t - T_A1, w.amb_w1, 64, 0, false);
        else if (t < N_C1) {
            const int q = t - T_A0, s = q / 64, rem = q % 64, h = rem / 32, j = rem % 32;
            v = w.amb_w0[kmap(s, h) * ldA0 + gather_feature(j)];
        } else if (t < N_S2R) v = valu_elem(t - N_C1, w.col_w1);
        else if (t < N_A2) v = valu_elem(t - N_S2R, w.sig_w2);
        else v = valu_elem(t - N_A2, w.amb_w2);
    } else {   // first-layer biases of the per-call constants (nerf/network.py:236, 262, 274)
        const int t = e - kFwd - kBwd, row = t & 63;
        float acc = 0.0f;
        if (t < 64) {
            const float *r = w.amb_w0 + row * ldA0 + 32;
            for (uint32_t a = 0; a < w.audio_dim; a++) acc += r[a] * enc_a[a];
        } else if (t < 128) {
            if (w.has_eye) acc = w.sig_w0[row * ldS0 + 64] * eye[0];
        } else {
            const float *r = w.col_w0 + row * ldC0 + 80;
            for (uint32_t c = 0; c < w.ind_dim; c++) acc += r[c] * ind_code[c];
        }
        v = acc;
    }
    image[e] = v;
}

// ---- workspace ----------------------------------------------------------------------------------------------------------
// native tiles ([registers][64 lanes] floats per 32-sample tile) and per-sample rows, one contiguous region each
struct Ws {
    float *ex, *ew;          // grid features, 16 registers: register 2 q + c of half h = level 2 q + h, channel c
    float *dw;               // d enc_w / d (normalised ambient coordinate), 32 registers: 4 q + 2 d + c of half h
    float *sh;               // SH basis, 8 registers: register q of half h = sh[2 q + h]
    float *ha0, *ha1, *hs0, *hs1, *geo, *hc0;        // 32 registers each
    float *dza0, *dza1, *dzs0, *dzs1, *dgeo, *dzc0;  // pre-activation gradients (backward), 32 registers each
    float *sraw;             // [M_pad] sigma_net output row 0
    float *daraw, *dsraw, *dprec;   // [M_pad, 2], [M_pad], [M_pad, 3]: gradients of the narrow layers' outputs
};
constexpr uint32_t kTile16 = 16 * 64, kTile32 = 32 * 64, kTile8 = 8 * 64;
constexpr uint32_t kWsPerTile = 2 * kTile16 + kTile32 + kTile8 + 12 * kTile32 + 32 * 7;

__host__ __device__ inline Ws make_ws(float *base, uint32_t M) {
    const size_t nt = (M + 31u) >> 5;
    Ws w;
    float *p = base;
    w.ex = p; p += nt * kTile16;
    w.ew = p; p += nt * kTile16;
    w.dw = p; p += nt * kTile32;
    w.sh = p; p += nt * kTile8;
    w.ha0 = p; p += nt * kTile32;
    w.ha1 = p; p += nt * kTile32;
    w.hs0 = p; p += nt * kTile32;
    w.hs1 = p; p += nt * kTile32;
    w.geo = p; p += nt * kTile32;
    w.hc0 = p; p += nt * kTile32;
    w.dza0 = p; p += nt * kTile32;
    w.dza1 = p; p += nt * kTile32;
    w.dzs0 = p; p += nt * kTile32;
    w.dzs1 = p; p += nt * kTile32;
    w.dgeo = p; p += nt * kTile32;
    w.dzc0 = p; p += nt * kTile32;
    w.sraw = p; p += nt * 32;
    w.daraw = p; p += nt * 64;
    w.dsraw = p; p += nt * 32;
    w.dprec = p; p += nt * 96;
    return w;
}

// ---- MFMA helpers (32-sample tiles) ---------------------------------------------------------------------------------------
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
__device__ __forceinline__ void acc_bias(Acc32 &a, const float *bias64, int h) {
#pragma unroll
    for (int rt = 0; rt < 2; rt++)
#pragma unroll
        for (int g = 0; g < 4; g++) {
            const float4 b = *reinterpret_cast<const float4 *>(bias64 + 32 * rt + 8 * g + 4 * h);
            a.v[rt][4 * g + 0] = b.x; a.v[rt][4 * g + 1] = b.y; a.v[rt][4 * g + 2] = b.z; a.v[rt][4 * g + 3] = b.w;
        }
}
__device__ __forceinline__ void acc_relu(Acc32 &a) {
#pragma unroll
    for (int rt = 0; rt < 2; rt++)
#pragma unroll
        for (int r = 0; r < 16; r++) a.v[rt][r] = relu_bits(a.v[rt][r]);
}
__device__ __forceinline__ void step32(Acc32 &a, const float *wl, int s, int lane_off, float b) {
    const float2 w = *reinterpret_cast<const float2 *>(wl + s * kStep + lane_off);
    a.v[0] = mfma32(w.x, b, a.v[0]);
    a.v[1] = mfma32(w.y, b, a.v[1]);
}
__device__ __forceinline__ void layer_from_acc(Acc32 &out, const Acc32 &in, const float *wl, int lane_off) {
#pragma unroll
    for (int s = 0; s < 32; s++) step32(out, wl, s, lane_off, in.v[s >> 4][s & 15]);
}
template <int NOUT>
__device__ __forceinline__ void valu_out(const Acc32 &in, const float *wl, int h, float (&out)[NOUT]) {
#pragma unroll
    for (int o = 0; o < NOUT; o++) {
        float p = 0.0f;
        const float *wo = wl + (o * 2 + h) * 32;
#pragma unroll
        for (int g = 0; g < 8; g++) {
            const float4 w = *reinterpret_cast<const float4 *>(wo + 4 * g);
            const int rt = g >> 2, r = (g & 3) * 4;
            p = __builtin_fmaf(in.v[rt][r + 0], w.x, p);
            p = __builtin_fmaf(in.v[rt][r + 1], w.y, p);
            p = __builtin_fmaf(in.v[rt][r + 2], w.z, p);
            p = __builtin_fmaf(in.v[rt][r + 3], w.w, p);
        }
        out[o] = p + __shfl_xor(p, 32, 64);
    }
}
// g[k] += sum_o W[o][k] d[o] for the k's this lane holds (the transposed narrow layer)
template <int NOUT>
__device__ __forceinline__ void valu_out_T(Acc32 &g, const float *wl, int h, const float (&d)[NOUT]) {
#pragma unroll
    for (int o = 0; o < NOUT; o++) {
        const float *wo = wl + (o * 2 + h) * 32;
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const float4 w = *reinterpret_cast<const float4 *>(wo + 4 * q);
            const int rt = q >> 2, r = (q & 3) * 4;
            g.v[rt][r + 0] = __builtin_fmaf(w.x, d[o], g.v[rt][r + 0]);
            g.v[rt][r + 1] = __builtin_fmaf(w.y, d[o], g.v[rt][r + 1]);
            g.v[rt][r + 2] = __builtin_fmaf(w.z, d[o], g.v[rt][r + 2]);
            g.v[rt][r + 3] = __builtin_fmaf(w.w, d[o], g.v[rt][r + 3]);
        }
    }
}
__device__ __forceinline__ void tile_store(float *__restrict__ dst, const Acc32 &a, int lane) {
#pragma unroll
    for (int rt = 0; rt < 2; rt++)
#pragma unroll
        for (int r = 0; r < 16; r++) dst[(rt * 16 + r) * 64 + lane] = a.v[rt][r];
}
// g = (saved activation > 0) ? g : 0, the saved tile read row by row
__device__ __forceinline__ void relu_mask(Acc32 &g, const float *__restrict__ saved, int lane) {
    float hv[32];
#pragma unroll
    for (int q = 0; q < 32; q++) hv[q] = saved[q * 64 + lane];
#pragma unroll
    for (int rt = 0; rt < 2; rt++)
#pragma unroll
        for (int r = 0; r < 16; r++) g.v[rt][r] = hv[rt * 16 + r] > 0.0f ? g.v[rt][r] : 0.0f;
}

constexpr int kThreads = 512, kWaves = kThreads / kWave;   // two waves per SIMD: a 4096-ray step is ~1 tile per wave slot

struct FwdParams {
    const float *xyzs, *dirs;
    uint32_t M;
    const int32_t *m_dev;
    GridArgs gx, gw;
    const float *image;
    float bound;
    float *sigmas, *rgbs, *ambient, *ambient_abs, *xn, *wn;
    float *ws;
};

__device__ __forceinline__ uint32_t live_count(uint32_t M, const int32_t *m_dev) {
    if (!m_dev) return M;
    const int32_t d = *m_dev;
    return d <= 0 ? 0u : ((uint32_t)d < M ? (uint32_t)d : M);
}

template <int GX, int GA>   // gather rounds in flight per wave (xyz grid / ambient grid): independent load chains hide each other's latency
__global__ void __launch_bounds__(kThreads) k_train_fwd(FwdParams p) {
    __shared__ __attribute__((aligned(16))) float lds[kFwd + kBias];
    __shared__ LevelPlan plan_x[16], plan_w[16];
    const uint32_t M = live_count(p.M, p.m_dev);
    const uint32_t n_tiles = (M + 31u) >> 5;
    if (blockIdx.x * kWaves >= n_tiles) return;
    for (int i = threadIdx.x; i < kFwd / 4; i += kThreads) reinterpret_cast<float4 *>(lds)[i] = reinterpret_cast<const float4 *>(p.image)[i];
    if (threadIdx.x < kBias) lds[kFwd + threadIdx.x] = p.image[kFwd + kBwd + threadIdx.x];
    if (threadIdx.x < 16) {
        const int t = threadIdx.x;
        const uint32_t ox = (uint32_t)p.gx.offsets[t], ow = (uint32_t)p.gw.offsets[t];
        plan_x[t] = plan_level<3>(p.gx.lc.scale[t], p.gx.lc.resolution[t], ox, (uint32_t)p.gx.offsets[t + 1] - ox, p.gx.gridtype, 8u);
        plan_w[t] = plan_level<2>(p.gw.lc.scale[t], p.gw.lc.resolution[t], ow, (uint32_t)p.gw.offsets[t + 1] - ow, p.gw.gridtype, 8u);
    }
    __syncthreads();
    const Ws ws = make_ws(p.ws, p.M);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 31, h = lane >> 5;
    const int lane_off = h * 64 + j * 2;
    const float *bias_amb = lds + kFwd, *bias_sig = lds + kFwd + 64, *bias_col = lds + kFwd + 128;
    const float *tx = static_cast<const float *>(p.gx.table), *tw = static_cast<const float *>(p.gw.table);

    for (uint32_t tile = blockIdx.x * kWaves + wave; tile < n_tiles; tile += gridDim.x * kWaves) {
        const uint32_t sample = tile * 32 + j;   // both lane halves work on the same 32 samples
        const bool live = sample < M;
        Acc32 a0, a1, a2;
        acc_bias(a0, bias_amb, h);   // ambient L0, start = W0[:, 32:] enc_a
        acc_bias(a2, bias_sig, h);   // sigma   L0, start = W0[:, 64] eye
        // ---- xyz grid (gridencoder/grid.py:145-161)
        {
            float in[3] = {0.0f, 0.0f, 0.0f};
            bool on = live;
            if (live) {
#pragma unroll
                for (int d = 0; d < 3; d++) {
                    in[d] = (p.xyzs[3 * (size_t)sample + d] + p.bound) / (2 * p.bound);
                    on = on && !(in[d] < 0 || in[d] > 1);
                }
                if (h == 0) { p.xn[3 * (size_t)sample] = in[0]; p.xn[3 * (size_t)sample + 1] = in[1]; p.xn[3 * (size_t)sample + 2] = in[2]; }
            }
            float *ex = ws.ex + (size_t)tile * kTile16;
            LevelFetch<float, 3, 2> f[GX];
#pragma unroll 1
            for (int r0 = 0; r0 < 8; r0 += GX) {
                if (on) {
#pragma unroll
                    for (int i = 0; i < GX; i++) issue_planned<float, 3, 2, false, false>(tx, plan_x[2 * (r0 + i) + h], in, f[i]);
                }
#pragma unroll
                for (int i = 0; i < GX; i++) {
                    const int r = r0 + i;
                    float f0 = 0.0f, f1 = 0.0f;
                    if (on) {
                        float res[2], dummy[1];
                        blend_level<float, 3, 2, false>(f[i], 0.0f, res, dummy);
                        f0 = res[0];
                        f1 = res[1];
                    }
                    step32(a0, lds + F_A0, 2 * r, lane_off, f0);
                    step32(a2, lds + F_S0, 2 * r, lane_off, f0);
                    step32(a0, lds + F_A0, 2 * r + 1, lane_off, f1);
                    step32(a2, lds + F_S0, 2 * r + 1, lane_off, f1);
                    ex[(2 * r) * 64 + lane] = f0;
                    ex[(2 * r + 1) * 64 + lane] = f1;
                }
            }
        }
        // ---- ambient net: [enc_x | enc_a] 96 -> 64 -> 64 -> 2, tanh
        acc_relu(a0);
        tile_store(ws.ha0 + (size_t)tile * kTile32, a0, lane);
        acc_zero(a1);
        layer_from_acc(a1, a0, lds + F_A1, lane_off);
        acc_relu(a1);
        tile_store(ws.ha1 + (size_t)tile * kTile32, a1, lane);
        float amb[2];
        valu_out<2>(a1, lds + F_A2, h, amb);
        amb[0] = tanhf(amb[0]);
        amb[1] = tanhf(amb[1]);
        // ---- ambient grid (+ d enc_w / d input): enc_w -> sigma L0 steps 16..31
        {
            float in[2] = {(amb[0] + 1.0f) / 2.0f, (amb[1] + 1.0f) / 2.0f};
            const bool on = live && !(in[0] < 0 || in[0] > 1 || in[1] < 0 || in[1] > 1);
            if (live && h == 0) {
                p.ambient[2 * (size_t)sample] = amb[0];
                p.ambient[2 * (size_t)sample + 1] = amb[1];
                if (p.ambient_abs) p.ambient_abs[sample] = fabsf(amb[0]) + fabsf(amb[1]);
                p.wn[2 * (size_t)sample] = in[0];
                p.wn[2 * (size_t)sample + 1] = in[1];
            }
            float *ew = ws.ew + (size_t)tile * kTile16, *dw = ws.dw + (size_t)tile * kTile32;
            LevelFetch<float, 2, 2> f[GA];
#pragma unroll 1
            for (int r0 = 0; r0 < 8; r0 += GA) {
                if (on) {
#pragma unroll
                    for (int i = 0; i < GA; i++) issue_planned<float, 2, 2, false, false>(tw, plan_w[2 * (r0 + i) + h], in, f[i]);
                }
#pragma unroll
                for (int i = 0; i < GA; i++) {
                    const int r = r0 + i;
                    float f0 = 0.0f, f1 = 0.0f, g[4] = {0.0f, 0.0f, 0.0f, 0.0f};
                    if (on) {
                        float res[2], grads[4];
                        blend_level<float, 2, 2, true>(f[i], plan_w[2 * r + h].scale, res, grads);
                        f0 = res[0];
                        f1 = res[1];
#pragma unroll
                        for (int q = 0; q < 4; q++) g[q] = grads[q];
                    }
                    step32(a2, lds + F_S0, 16 + 2 * r, lane_off, f0);
                    step32(a2, lds + F_S0, 16 + 2 * r + 1, lane_off, f1);
                    ew[(2 * r) * 64 + lane] = f0;
                    ew[(2 * r + 1) * 64 + lane] = f1;
#pragma unroll
                    for (int q = 0; q < 4; q++) dw[(4 * r + q) * 64 + lane] = g[q];
                }
            }
        }
        // ---- sigma net: [enc_x | enc_w | eye] 65 -> 64 -> 64 -> 1 + 64
        acc_relu(a2);
        tile_store(ws.hs0 + (size_t)tile * kTile32, a2, lane);
        acc_zero(a1);
        layer_from_acc(a1, a2, lds + F_S1, lane_off);
        acc_relu(a1);
        tile_store(ws.hs1 + (size_t)tile * kTile32, a1, lane);
        {
            float raw[1];
            valu_out<1>(a1, lds + F_S2R, h, raw);
            if (h == 0) {
                ws.sraw[sample] = raw[0];
                if (live) p.sigmas[sample] = expf(raw[0]);   // trunc_exp forward (activation.py:9-11)
            }
        }
        acc_zero(a0);
        layer_from_acc(a0, a1, lds + F_S2, lane_off);   // geo_feat (no activation)
        tile_store(ws.geo + (size_t)tile * kTile32, a0, lane);
        // ---- color net: [SH(d) | geo_feat | ind_code] 84 -> 64 -> 3, sigmoid
        acc_bias(a1, bias_col, h);
        {
            float sh[16];
            float dx = 0.0f, dy = 0.0f, dz = 0.0f;
            if (live) {
                dx = p.dirs[3 * (size_t)sample]; dy = p.dirs[3 * (size_t)sample + 1]; dz = p.dirs[3 * (size_t)sample + 2];
            }
            sh_basis<4>(dx, dy, dz, sh);
            float *st = ws.sh + (size_t)tile * kTile8;
#pragma unroll
            for (int s = 0; s < 8; s++) {
                const uint32_t m = 0u - (uint32_t)h;   // lane half h supplies k = 2 s + h (bit-select: no dynamic indexing of sh[])
                const float b = __uint_as_float((__float_as_uint(sh[2 * s]) & ~m) | (__float_as_uint(sh[2 * s + 1]) & m));
                step32(a1, lds + F_C0, s, lane_off, b);
                st[s * 64 + lane] = b;
            }
        }
#pragma unroll
        for (int s = 0; s < 32; s++) step32(a1, lds + F_C0, 8 + s, lane_off, a0.v[s >> 4][s & 15]);
        acc_relu(a1);
        tile_store(ws.hc0 + (size_t)tile * kTile32, a1, lane);
        {
            float rgb[3];
            valu_out<3>(a1, lds + F_C1, h, rgb);
            if (live && h == 0) {
#pragma unroll
                for (int c = 0; c < 3; c++) p.rgbs[3 * (size_t)sample + c] = 1.0f / (1.0f + expf(-rgb[c]));
            }
        }
    }
}

struct BwdParams {
    const float *g_sigmas, *g_rgbs, *g_ambient, *g_ambient_abs;
    const float *rgbs, *ambient;
    uint32_t M;
    const int32_t *m_dev;
    const float *image;
    float *ws;
    float *g_enc_x, *g_enc_w;   // [16, M, 2]
};

__global__ void __launch_bounds__(kThreads) k_train_bwd(BwdParams p) {
    __shared__ __attribute__((aligned(16))) float lds[kBwd];
    const uint32_t M = live_count(p.M, p.m_dev);
    const uint32_t n_tiles = (M + 31u) >> 5;
    if (blockIdx.x * kWaves >= n_tiles) return;
    for (int i = threadIdx.x; i < kBwd / 4; i += kThreads)
        reinterpret_cast<float4 *>(lds)[i] = reinterpret_cast<const float4 *>(p.image + kFwd)[i];
    __syncthreads();
    const Ws ws = make_ws(p.ws, p.M);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 31, h = lane >> 5;
    const int lane_off = h * 64 + j * 2;

    for (uint32_t tile = blockIdx.x * kWaves + wave; tile < n_tiles; tile += gridDim.x * kWaves) {
        const uint32_t sample = tile * 32 + j;
        const bool live = sample < M;
        Acc32 g, w;
        // ---- colour net: sigmoid', last layer transposed, ReLU mask
        {
            float d[3] = {0.0f, 0.0f, 0.0f};
            if (live) {
#pragma unroll
                for (int c = 0; c < 3; c++) {
                    const float y = p.rgbs[3 * (size_t)sample + c];
                    d[c] = p.g_rgbs[3 * (size_t)sample + c] * ((1.0f - y) * y);
                }
            }
            if (h == 0) {
#pragma unroll
                for (int c = 0; c < 3; c++) ws.dprec[3 * (size_t)sample + c] = d[c];
            }
            acc_zero(g);
            valu_out_T<3>(g, lds + N_C1, h, d);
        }
        relu_mask(g, ws.hc0 + (size_t)tile * kTile32, lane);
        tile_store(ws.dzc0 + (size_t)tile * kTile32, g, lane);
        acc_zero(w);
        layer_from_acc(w, g, lds + T_C0, lane_off);       // d geo_feat
        tile_store(ws.dgeo + (size_t)tile * kTile32, w, lane);
        // ---- sigma net
        {
            float d[1] = {0.0f};
            if (live) d[0] = p.g_sigmas[sample] * expf(fminf(fmaxf(ws.sraw[sample], -15.0f), 15.0f));   // activation.py:13-17
            if (h == 0) ws.dsraw[sample] = d[0];
            acc_zero(g);
            layer_from_acc(g, w, lds + T_S2, lane_off);
            valu_out_T<1>(g, lds + N_S2R, h, d);
        }
        relu_mask(g, ws.hs1 + (size_t)tile * kTile32, lane);
        tile_store(ws.dzs1 + (size_t)tile * kTile32, g, lane);
        acc_zero(w);
        layer_from_acc(w, g, lds + T_S1, lane_off);
        relu_mask(w, ws.hs0 + (size_t)tile * kTile32, lane);
        tile_store(ws.dzs0 + (size_t)tile * kTile32, w, lane);
        // d [enc_x | enc_w]: row tile 0 = enc_x, row tile 1 = enc_w, register 2 q + c of half h = (level 2 q + h, channel c)
        f32x16 x0, x1;
#pragma unroll
        for (int r = 0; r < 16; r++) { x0[r] = 0.0f; x1[r] = 0.0f; }
#pragma unroll
        for (int s = 0; s < 32; s++) {
            const float2 wv = *reinterpret_cast<const float2 *>(lds + T_S0 + s * kStep + lane_off);
            const float b = w.v[s >> 4][s & 15];
            x0 = mfma32(wv.x, b, x0);
            x1 = mfma32(wv.y, b, x1);
        }
        // ---- ambient grid: feature gradients out (level-major), input gradient = sum_l g . dy_dx (gridencoder.cu:342-368);
        // the encoder sees (ambient + 1) / 2 (gridencoder/grid.py:151, bound = 1): a factor 1/2 on the way back
        float da[2] = {0.0f, 0.0f};
        {
            const float *dw = ws.dw + (size_t)tile * kTile32;
#pragma unroll
            for (int q = 0; q < 8; q++) {
                const float g0 = x1[2 * q], g1 = x1[2 * q + 1];
                const float d00 = dw[(4 * q + 0) * 64 + lane], d01 = dw[(4 * q + 1) * 64 + lane];
                const float d10 = dw[(4 * q + 2) * 64 + lane], d11 = dw[(4 * q + 3) * 64 + lane];
                da[0] = __builtin_fmaf(g0, d00, da[0]); da[0] = __builtin_fmaf(g1, d01, da[0]);
                da[1] = __builtin_fmaf(g0, d10, da[1]); da[1] = __builtin_fmaf(g1, d11, da[1]);
                if (live) *reinterpret_cast<float2 *>(p.g_enc_w + ((size_t)(2 * q + h) * p.M + sample) * 2) = make_float2(g0, g1);
            }
            da[0] += __shfl_xor(da[0], 32, 64);
            da[1] += __shfl_xor(da[1], 32, 64);
        }
        // ---- ambient net: + the direct gradients of the ambient output, tanh'
        {
            float d[2] = {0.0f, 0.0f};
            if (live) {
                const float a0v = p.ambient[2 * (size_t)sample], a1v = p.ambient[2 * (size_t)sample + 1];
                float t0 = 0.5f * da[0], t1 = 0.5f * da[1];
                if (p.g_ambient) { t0 += p.g_ambient[2 * (size_t)sample]; t1 += p.g_ambient[2 * (size_t)sample + 1]; }
                if (p.g_ambient_abs) {   // d |a| = sign(a), 0 at 0
                    const float ga = p.g_ambient_abs[sample];
                    t0 += a0v > 0.0f ? ga : (a0v < 0.0f ? -ga : 0.0f);
                    t1 += a1v > 0.0f ? ga : (a1v < 0.0f ? -ga : 0.0f);
                }
                d[0] = t0 * (1.0f - a0v * a0v);
                d[1] = t1 * (1.0f - a1v * a1v);
            }
            if (h == 0) { ws.daraw[2 * (size_t)sample] = d[0]; ws.daraw[2 * (size_t)sample + 1] = d[1]; }
            acc_zero(g);
            valu_out_T<2>(g, lds + N_A2, h, d);
        }
        relu_mask(g, ws.ha1 + (size_t)tile * kTile32, lane);
        tile_store(ws.dza1 + (size_t)tile * kTile32, g, lane);
        acc_zero(w);
        layer_from_acc(w, g, lds + T_A1, lane_off);
        relu_mask(w, ws.ha0 + (size_t)tile * kTile32, lane);
        tile_store(ws.dza0 + (size_t)tile * kTile32, w, lane);
#pragma unroll
        for (int s = 0; s < 32; s++) x0 = mfma32(lds[T_A0 + s * 64 + h * 32 + j], w.v[s >> 4][s & 15], x0);
        if (live) {
#pragma unroll
            for (int q = 0; q < 8; q++)
                *reinterpret_cast<float2 *>(p.g_enc_x + ((size_t)(2 * q + h) * p.M + sample) * 2) = make_float2(x0[2 * q], x0[2 * q + 1]);
        }
    }
}

// ---- weight gradients ---------------------------------------------------------------------------------------------------
constexpr int kWThreads = 256;
constexpr int kTS = 36;                  // LDS row stride of a staged tile: [feature][sample parity][sample / 2]
constexpr int kStage = 96 * kTS;         // one operand tile: up to 96 features x 32 samples
constexpr int PHI_STD = 0, PHI_ENC = 1, PHI_SH = 2;
template <int PHI>
__device__ __forceinline__ int phi(int q, int h) {   // feature of register q, lane half h of a native tile
    if constexpr (PHI == PHI_STD) return 32 * (q >> 4) + rowmap(q & 15, h);
    else if constexpr (PHI == PHI_ENC) return 4 * (q >> 1) + 2 * h + (q & 1);
    else return 2 * q + h;
}
// An operand = [RM row-major columns | native segment 0 (R0 registers) | native segment 1 (R1 registers) | ones]
template <int RM_, int R0_, int PHI0_, int R1_, int PHI1_, bool ONES_>
struct OpT {
    static constexpr int RM = RM_, R0 = R0_, PHI0 = PHI0_, R1 = R1_, PHI1 = PHI1_;
    static constexpr bool ONES = ONES_;
    static constexpr int NF = RM + 2 * R0 + 2 * R1 + (ONES ? 1 : 0);   // features
    static constexpr int NB = (NF + 31) / 32;                          // 32-row blocks
    static constexpr int S0 = R0 / 4, S1 = R1 / 4;                     // registers per thread (4 waves)
};
struct OpPtr {
    const float *rm;   // [M_pad, RM]
    const float *s0, *s1;
};
template <typename Op>
struct Fetched {
    float v0[Op::S0 > 0 ? Op::S0 : 1], v1[Op::S1 > 0 ? Op::S1 : 1], rm;
};
template <typename Op>
__device__ __forceinline__ void fetch(Fetched<Op> &f, const OpPtr &o, uint32_t tile) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if constexpr (Op::S0 > 0) {
        const float *src = o.s0 + (size_t)tile * (Op::R0 * 64);
#pragma unroll
        for (int i = 0; i < Op::S0; i++) f.v0[i] = src[(wave * Op::S0 + i) * 64 + lane];
    }
    if constexpr (Op::S1 > 0) {
        const float *src = o.s1 + (size_t)tile * (Op::R1 * 64);
#pragma unroll
        for (int i = 0; i < Op::S1; i++) f.v1[i] = src[(wave * Op::S1 + i) * 64 + lane];
    }
    if constexpr (Op::RM > 0) {
        f.rm = 0.0f;
        if (threadIdx.x < 32 * Op::RM) f.rm = o.rm[(size_t)tile * (32 * Op::RM) + threadIdx.x];
    }
}
template <typename Op>
__device__ __forceinline__ void commit(float *t, const Fetched<Op> &f, uint32_t tile, uint32_t M) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, j = lane & 31, h = lane >> 5;
    const int col = (j & 1) * 16 + (j >> 1);
    if constexpr (Op::S0 > 0) {
#pragma unroll
        for (int i = 0; i < Op::S0; i++) t[(Op::RM + phi<Op::PHI0>(wave * Op::S0 + i, h)) * kTS + col] = f.v0[i];
    }
    if constexpr (Op::S1 > 0) {
#pragma unroll
        for (int i = 0; i < Op::S1; i++) t[(Op::RM + 2 * Op::R0 + phi<Op::PHI1>(wave * Op::S1 + i, h)) * kTS + col] = f.v1[i];
    }
    if constexpr (Op::RM > 0) {
        if (threadIdx.x < 32 * Op::RM) {
            const int s = threadIdx.x / Op::RM, c = threadIdx.x % Op::RM;
            t[c * kTS + (s & 1) * 16 + (s >> 1)] = f.rm;
        }
    }
    if constexpr (Op::ONES) {
        if (threadIdx.x >= 64 && threadIdx.x < 96) {
            const int s = threadIdx.x - 64;
            t[(Op::NF - 1) * kTS + (s & 1) * 16 + (s >> 1)] = (tile * 32 + s < M) ? 1.0f : 0.0f;
        }
    }
}

// One workgroup = one job x one slice of the sample tiles; the NBa x NBb output blocks of 32 x 32 are dealt round-robin to
// the 4 waves (<= 3 each), accumulators stay in registers over all tiles; tiles travel global -> registers two iterations
// ahead of their use (the latency of a once-read tile is longer than its MFMA work).
template <typename OpA, typename OpB>
__device__ __forceinline__ void wgrad_job(const OpPtr &pa, const OpPtr &pb, uint32_t n_tiles, uint32_t M, uint32_t part, uint32_t parts,
                                          float *__restrict__ partial, float *lds) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i = lane & 31, h = lane >> 5;
    constexpr int NBLK = OpA::NB * OpB::NB;
    static_assert(NBLK <= 12, "too many output blocks");
    constexpr int NQ = (NBLK + 3) / 4;
    f32x16 acc[NQ];
#pragma unroll
    for (int q = 0; q < NQ; q++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[q][r] = 0.0f;
    float *ta = lds, *tb = lds + kStage;
    for (int e = threadIdx.x; e < 2 * kStage; e += kWThreads) lds[e] = 0.0f;   // pad features stay zero
    __syncthreads();
    Fetched<OpA> fa0, fa1;
    Fetched<OpB> fb0, fb1;
    const uint32_t stride = parts;
    if (part < n_tiles) { fetch<OpA>(fa0, pa, part); fetch<OpB>(fb0, pb, part); }
    if (part + stride < n_tiles) { fetch<OpA>(fa1, pa, part + stride); fetch<OpB>(fb1, pb, part + stride); }
    if (part < n_tiles) { commit<OpA>(ta, fa0, part, M); commit<OpB>(tb, fb0, part, M); }
    __syncthreads();
    auto multiply = [&]() {
#pragma unroll
        for (int q = 0; q < NQ; q++) {
            const int b = wave + 4 * q;
            if (b < NBLK) {
                const int bx = b / OpB::NB, by = b - bx * OpB::NB;
                const float4 *qa = reinterpret_cast<const float4 *>(ta + (32 * bx + i) * kTS + h * 16);
                const float4 *qb = reinterpret_cast<const float4 *>(tb + (32 * by + i) * kTS + h * 16);
                float4 av[4], bv[4];
#pragma unroll
                for (int u = 0; u < 4; u++) { av[u] = qa[u]; bv[u] = qb[u]; }
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    acc[q] = mfma32(av[u].x, bv[u].x, acc[q]);
                    acc[q] = mfma32(av[u].y, bv[u].y, acc[q]);
                    acc[q] = mfma32(av[u].z, bv[u].z, acc[q]);
                    acc[q] = mfma32(av[u].w, bv[u].w, acc[q]);
                }
            }
        }
    };
    for (uint32_t tile = part; tile < n_tiles; tile += 2 * stride) {
        if (tile + 2 * stride < n_tiles) { fetch<OpA>(fa0, pa, tile + 2 * stride); fetch<OpB>(fb0, pb, tile + 2 * stride); }
        multiply();
        __syncthreads();
        if (tile + stride < n_tiles) { commit<OpA>(ta, fa1, tile + stride, M); commit<OpB>(tb, fb1, tile + stride, M); }
        __syncthreads();
        if (tile + stride >= n_tiles) break;
        if (tile + 3 * stride < n_tiles) { fetch<OpA>(fa1, pa, tile + 3 * stride); fetch<OpB>(fb1, pb, tile + 3 * stride); }
        multiply();
        __syncthreads();
        if (tile + 2 * stride < n_tiles) { commit<OpA>(ta, fa0, tile + 2 * stride, M); commit<OpB>(tb, fb0, tile + 2 * stride, M); }
        __syncthreads();
    }
    float *dst = partial + (size_t)part * (96 * 96);
#pragma unroll
    for (int q = 0; q < NQ; q++) {
        const int b = wave + 4 * q;
        if (b < NBLK) {
            const int bx = b / OpB::NB, by = b - bx * OpB::NB;
#pragma unroll
            for (int r = 0; r < 16; r++) dst[(32 * bx + rowmap(r, h)) * 96 + 32 * by + i] = acc[q][r];
        }
    }
}

constexpr int kJobs = 8;
enum { J_A0 = 0, J_A1, J_A2, J_S0, J_S1, J_S2, J_C0, J_C1 };
typedef OpT<0, 32, PHI_STD, 0, 0, false> OpStd;                 // a 64-feature native tile
typedef OpT<0, 16, PHI_ENC, 0, 0, true> OpEncX1;                // [enc_x | 1]
typedef OpT<0, 16, PHI_ENC, 16, PHI_ENC, true> OpEncXW1;        // [enc_x | enc_w | 1]
typedef OpT<0, 8, PHI_SH, 32, PHI_STD, true> OpShGeo1;          // [sh | geo_feat | 1]
typedef OpT<2, 0, 0, 0, 0, false> OpRm2;
typedef OpT<3, 0, 0, 0, 0, false> OpRm3;
typedef OpT<1, 32, PHI_STD, 0, 0, false> OpRm1Std;              // [d sigma_raw | d geo_feat]

struct WArgs {
    float *ws;
    uint32_t M;
    const int32_t *m_dev;
    uint32_t parts;
    float *partial;     // [kJobs][parts][96 * 96]
};

__global__ void __launch_bounds__(kWThreads, 2) k_train_wgrad(WArgs p) {
    __shared__ __attribute__((aligned(16))) float lds[2 * kStage];
    const uint32_t job = blockIdx.x / p.parts, part = blockIdx.x % p.parts;
    const uint32_t M = live_count(p.M, p.m_dev), n_tiles = (M + 31u) >> 5;
    const Ws w = make_ws(p.ws, p.M);
    float *partial = p.partial + (size_t)job * p.parts * (96 * 96);
    switch (job) {
    case J_A0: wgrad_job<OpStd, OpEncX1>(OpPtr{nullptr, w.dza0, nullptr}, OpPtr{nullptr, w.ex, nullptr}, n_tiles, M, part, p.parts, partial, lds); break;
    case J_A1: wgrad_job<OpStd, OpStd>(OpPtr{nullptr, w.dza1, nullptr}, OpPtr{nullptr, w.ha0, nullptr}, n_tiles, M, part, p.parts, partial, lds); break;
    case J_A2: wgrad_job<OpRm2, OpStd>(OpPtr{w.daraw, nullptr, nullptr}, OpPtr{nullptr, w.ha1, nullptr}, n_tiles, M, part, p.parts, partial, lds); break;
    case J_S0: wgrad_job<OpStd, OpEncXW1>(OpPtr{nullptr, w.dzs0, nullptr}, OpPtr{nullptr, w.ex, w.ew}, n_tiles, M, part, p.parts, partial, lds); break;
    case J_S1: wgrad_job<OpStd, OpStd>(OpPtr{nullptr, w.dzs1, nullptr}, OpPtr{nullptr, w.hs0, nullptr}, n_tiles, M, part, p.parts, partial, lds); break;
    case J_S2: wgrad_job<OpRm1Std, OpStd>(OpPtr{w.dsraw, w.dgeo, nullptr}, OpPtr{nullptr, w.hs1, nullptr}, n_tiles, M, part, p.parts, partial, lds); break;
    case J_C0: wgrad_job<OpStd, OpShGeo1>(OpPtr{nullptr, w.dzc0, nullptr}, OpPtr{nullptr, w.sh, w.geo}, n_tiles, M, part, p.parts, partial, lds); break;
    default: wgrad_job<OpRm3, OpStd>(OpPtr{w.dprec, nullptr, nullptr}, OpPtr{nullptr, w.hc0, nullptr}, n_tiles, M, part, p.parts, partial, lds); break;
    }
}

struct RJob {
    float *out;
    uint32_t rows, cols, ld;      // out[row * ld + col] for row < rows, col < cols
    int32_t bias_col;             // column of the partial that is the bias gradient (-1: none)
    float *bias_out;              // [rows]
};
struct RArgs {
    RJob job[kJobs];
    const float *partial;
    uint32_t parts;
};
__global__ void __launch_bounds__(256) k_train_wreduce(RArgs p) {
    const RJob &job = p.job[blockIdx.y];
    const uint32_t e = blockIdx.x * 256 + threadIdx.x;
    const uint32_t row = e / 96, col = e % 96;
    const bool bias = job.bias_col >= 0 && (int32_t)col == job.bias_col;
    if (row >= job.rows || (col >= job.cols && !bias)) return;
    const float *src = p.partial + (size_t)blockIdx.y * p.parts * (96 * 96) + row * 96 + col;
    float s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    uint32_t q = 0;
    for (; q + 8 <= p.parts; q += 8) {
#pragma unroll
        for (int u = 0; u < 8; u++) s[u] += src[(size_t)(q + u) * (96 * 96)];
    }
    for (; q < p.parts; q++) s[0] += src[(size_t)q * (96 * 96)];
    const float total = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
    if (bias) job.bias_out[row] = total;
    else job.out[row * job.ld + col] = total;
}

// The per-call constants enter the first layers as biases (k_train_pack).  With gb = the bias gradient [64]:
//   d constant[a] = sum_u W0[u][c0 + a] gb[u]        d W0[u][c0 + a] = gb[u] constant[a]
struct CArgs {
    RawW w;
    const float *enc_a, *eye, *ind_code;
    const float *gb;    // [3][64]
    float *g_a0, *g_s0, *g_c0, *g_enc_a, *g_eye, *g_ind;
    const int64_t *ind_index;   // non-NULL: ind_code / g_ind are the tables [ind_rows, ind_dim]; the row is picked here and the
    uint32_t ind_rows;          // rest of the gradient table is written as zeros (what index_select's backward builds in two launches)
};
__global__ void __launch_bounds__(256) k_train_const(CArgs p) {
    const int which = blockIdx.x;
    const float *W, *c, *gb = p.gb + 64 * which;
    float *gW, *gc;
    uint32_t n, ld, c0;
    if (which >= 3) {           // extra workgroups (row form only): zeros for the other rows of the code gradient table
        const size_t row = (size_t)p.ind_index[0], nd = p.w.ind_dim, total = (size_t)p.ind_rows * nd;
        for (size_t e = (size_t)(which - 3) * 256 + threadIdx.x; e < total; e += (size_t)(gridDim.x - 3) * 256)
            if (e / nd != row) p.g_ind[e] = 0.0f;
        return;
    }
    if (which == 0) { W = p.w.amb_w0; c = p.enc_a; gW = p.g_a0; gc = p.g_enc_a; n = p.w.audio_dim; c0 = 32; }
    else if (which == 1) { W = p.w.sig_w0; c = p.eye; gW = p.g_s0; gc = p.g_eye; n = p.w.has_eye; c0 = 64; }
    else {
        W = p.w.col_w0; c = p.ind_code; gW = p.g_c0; gc = p.g_ind; n = p.w.ind_dim; c0 = 80;
        if (p.ind_index) { c += (size_t)p.ind_index[0] * n; if (gc) gc += (size_t)p.ind_index[0] * n; }
    }
    ld = c0 + n;
    for (uint32_t e = threadIdx.x; e < 64 * n; e += 256) {
        const uint32_t u = e / n, a = e - u * n;
        gW[u * ld + c0 + a] = gb[u] * c[a];
    }
    for (uint32_t a = threadIdx.x; a < n; a += 256) {
        float s = 0.0f;
        for (uint32_t u = 0; u < 64; u++) s += W[u * ld + c0 + a] * gb[u];
        if (gc) gc[a] = s;
    }
}

// ---- table gradient -------------------------------------------------------------------------------------------------------
#ifndef RN_SC_SLOTS
#define RN_SC_SLOTS 512
#endif
// 256 threads = (256 / 2^(D-1)) samples x 2^(D-1) x-pairs of corners; 512 slots of 16 floats + key = 35 KB of LDS: four
// workgroups per CU, so one workgroup's burst of atomics (its flush) runs under the others' loads and LDS inserts
constexpr uint32_t kScThreads = 256, kScSlots = RN_SC_SLOTS, kScProbes = 24;
constexpr uint32_t kScSlotBits = kScSlots == 1024 ? 10 : (kScSlots == 512 ? 9 : 8);
static_assert((1u << kScSlotBits) == kScSlots, "slot count must be 256, 512 or 1024");
constexpr uint32_t kScEmpty = 0xffffffffu;

// Lanes hold (key, v[4]); consecutive lanes with equal keys form a run (ray-ordered samples stay in one coarse cell for many
// steps).  A segmented inclusive scan sums each run into its last lane, which alone goes on to the LDS table.
__device__ __forceinline__ bool merge_runs4(uint32_t key, uint32_t key2, float (&v)[4]) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t prev = (uint32_t)__shfl_up((int)key, 1, 64), prev2 = (uint32_t)__shfl_up((int)key2, 1, 64);
    const bool head = lane == 0 || key != prev || key2 != prev2;   // a run = both destination rows equal
    const unsigned long long heads = __ballot(head);
    if (__popcll(heads) > 40) return key != kScEmpty;
    const unsigned long long below = heads & ((2ull << lane) - 1ull);
    const uint32_t start = 63u - (uint32_t)__clzll(below);
#pragma unroll
    for (uint32_t off = 1; off < 64; off <<= 1) {
        float up[4];
#pragma unroll
        for (int c = 0; c < 4; c++) up[c] = __shfl_up(v[c], off, 64);
        if (lane >= start + off) {
#pragma unroll
            for (int c = 0; c < 4; c++) v[c] += up[c];
        }
    }
    const bool tail = lane == 63u || ((heads >> (lane + 1)) & 1ull);
    return tail && key != kScEmpty;
}

// One level of one chunk of samples through the LDS line merge (the non-binned levels).
struct ScatterJob {
    const float *grad, *inputs;
    const int32_t *offsets;
    float *grad_grid;
    LevelConsts lc;
    uint32_t gridtype, n_levels;
    uint32_t level_of[kMaxLevels];    // the levels this job covers (blockIdx.y indexes this list)
    uint32_t direct_mask;             // bit i: entry i of level_of goes straight to memory (a hashed level: nothing to merge)
    uint32_t chunks_of[kMaxLevels];   // line-merged levels: chunks of samples a workgroup sums in its LDS table before it flushes
};

template <uint32_t D>
__device__ __forceinline__ void scatter_lines(const ScatterJob &j, uint32_t Mcap, uint32_t M, uint32_t *keys, float *vals, uint32_t *occupied,
                                              uint32_t block_x) {
    constexpr uint32_t P = 1u << (D - 1);            // x-pairs of corners per sample
    constexpr uint32_t kScSamples = kScThreads / P;  // lanes 0 .. S-1: pair 0 of the S samples, lanes S .. 2S-1: pair 1, ...
    if (blockIdx.y >= j.n_levels) return;
    const uint32_t level = j.level_of[blockIdx.y];
    const bool direct = (j.direct_mask >> blockIdx.y) & 1u;   // workgroup-uniform
    // small levels (few lines in all): a workgroup sums several chunks of samples in its table before it flushes -- the more
    // samples share a table, the more of their rows coincide (the ambient coordinates of a step cluster in a few cells)
    const uint32_t chunks = direct ? 1u : j.chunks_of[blockIdx.y];
    if (block_x * chunks * kScSamples >= M) return;
    if (!direct) {
        for (uint32_t i = threadIdx.x; i < kScSlots; i += kScThreads) keys[i] = kScEmpty;
        for (uint32_t i = threadIdx.x; i < kScSlots * 16; i += kScThreads) vals[i] = 0.0f;
        if (threadIdx.x == 0) *occupied = 0u;
        __syncthreads();
    }
    const uint32_t off = (uint32_t)j.offsets[level];
    const uint32_t hashmap_size = (uint32_t)j.offsets[level + 1] - off;
    float *gg = j.grad_grid + (size_t)off * 2;
    const uint32_t q = threadIdx.x / kScSamples;      // this thread's x-pair: bits of q = the y (, z) corner
    const uint32_t resolution = j.lc.resolution[level];
  for (uint32_t chunk = 0; chunk < chunks; chunk++) {
    const uint32_t b = (block_x * chunks + chunk) * kScSamples + (threadIdx.x & (kScSamples - 1u));
    float in[D];
    bool live = b < M;
#pragma unroll
    for (uint32_t d = 0; d < D; d++) {
        in[d] = live ? j.inputs[(size_t)b * D + d] : 0.0f;
        live = live && !(in[d] < 0 || in[d] > 1);     // gridencoder.cu:275-280
    }
    float pos[D], pos_deriv[D];
    uint32_t pos_grid[D];
    lattice_pos<D>(in, j.lc.scale[level], false, 0, pos, pos_deriv, pos_grid);
    float2 g = make_float2(0.0f, 0.0f);
    if (live) g = *reinterpret_cast<const float2 *>(j.grad + ((size_t)level * Mcap + b) * 2);
    {
        uint32_t pgl[D];
        pgl[0] = pos_grid[0];
        float wyz[2] = {1.0f - pos[0], pos[0]};       // the reference multiplies the x term first (gridencoder.cu:298-308)
#pragma unroll
        for (uint32_t d = 1; d < D; d++) {
            const bool hi = (q >> (d - 1)) & 1u;
            const float wd = hi ? pos[d] : 1 - pos[d];
            wyz[0] *= wd;
            wyz[1] *= wd;
            pgl[d] = pos_grid[d] + (hi ? 1u : 0u);
        }
        uint32_t row0 = kScEmpty, row1 = kScEmpty;
        if (live) {
            row0 = grid_row<D>(j.gridtype, false, hashmap_size, resolution, pgl);
            pgl[0] += 1u;
            row1 = grid_row<D>(j.gridtype, false, hashmap_size, resolution, pgl);
        }
        float v[4] = {wyz[0] * g.x, wyz[0] * g.y, wyz[1] * g.x, wyz[1] * g.y};
        if (direct) {
            // A hashed level: the workgroup's samples never touch a line twice, so an LDS merge would spend ~3.6 clocks per lane
            // and float on LDS atomics to remove nothing.  The four floats of an x-pair (two rows that share a 64-B line 7 times out
            // of 8) leave from four ADJACENT lanes of one instruction -- one memory-side request per pair: instruction k serves the
            // pairs of lanes 16 k .. 16 k + 15, lane l carrying float (l & 3) of pair 16 k + (l >> 2).
            const uint32_t lane = threadIdx.x & 63u, f = lane & 3u;
#pragma unroll
            for (uint32_t k = 0; k < 4; k++) {
                const int src = (int)(16u * k + (lane >> 2));
                const uint32_t r0 = (uint32_t)__shfl((int)row0, src, 64), r1 = (uint32_t)__shfl((int)row1, src, 64);
                const float a0 = __shfl(v[0], src, 64), a1 = __shfl(v[1], src, 64), a2 = __shfl(v[2], src, 64), a3 = __shfl(v[3], src, 64);
                const uint32_t row = f < 2u ? r0 : r1;
                const float val = f == 0u ? a0 : (f == 1u ? a1 : (f == 2u ? a2 : a3));
                if (row != kScEmpty && val != 0.0f) atomicAdd(gg + (size_t)row * 2 + (f & 1u), val);
            }
            return;
        }
        if (merge_runs4(row0, row1, v)) {
            const uint32_t rows[2] = {row0, row1};
#pragma unroll
            for (int e = 0; e < 2; e++) {
                const uint32_t line = rows[e] >> 3, sub = rows[e] & 7u;
                uint32_t slot = (line * 2654435761u) >> (32u - kScSlotBits);
                bool placed = false;
                for (uint32_t probe = 0; probe < kScProbes; probe++) {
                    const uint32_t prev = atomicCAS(&keys[slot], kScEmpty, line);
                    if (prev == kScEmpty) atomicAdd(occupied, 1u);
                    if (prev == kScEmpty || prev == line) { placed = true; break; }
                    slot = (slot + 1u) & (kScSlots - 1u);
                }
                if (placed) {
                    atomicAdd(&vals[slot * 16 + sub * 2], v[2 * e]);
                    atomicAdd(&vals[slot * 16 + sub * 2 + 1], v[2 * e + 1]);
                } else {   // table full around this line (never with 2-D grids): straight to memory
                    atomicAdd(gg + (size_t)rows[e] * 2, v[2 * e]);
                    atomicAdd(gg + (size_t)rows[e] * 2 + 1, v[2 * e + 1]);
                }
            }
        }
    }
    // flush when the table is filling up (spread-out samples: every chunk; clustered ones: rarely) or after the last chunk.
    // 16 adjacent lanes = the 16 floats of one 64-B line of the gradient table: one memory-side request per touched line
    __syncthreads();
    const bool last = chunk + 1 == chunks || (block_x * chunks + chunk + 1) * kScSamples >= M;
    if (last || *occupied > kScSlots / 2 - kScSlots / 8) {
        for (uint32_t i = threadIdx.x; i < kScSlots * 16; i += kScThreads) {
            const uint32_t line = keys[i >> 4];
            if (line != kScEmpty) {
                const float v = vals[i];
                if (v != 0.0f) atomicAdd(gg + (size_t)line * 16 + (i & 15u), v);
                if (!last) {                                    // leave an empty table for the next chunk
                    vals[i] = 0.0f;
                    if ((i & 15u) == 15u) keys[i >> 4] = kScEmpty;   // the 16 lanes of the slot have read the key above
                }
            }
        }
        if (last) return;
        __syncthreads();
        if (threadIdx.x == 0) *occupied = 0u;
        __syncthreads();
    }
  }
}


// One launch for the levels of up to two grids that are not binned (the 3-D grid's and the 2-D grid's)
template <uint32_t D0, uint32_t D1>
__global__ void __launch_bounds__(kScThreads) k_grid_scatter(ScatterJob j0, ScatterJob j1, uint32_t n_jobs, uint32_t Mcap,
                                                             const int32_t *__restrict__ m_dev) {
    __shared__ uint32_t keys[kScSlots];
    __shared__ __attribute__((aligned(16))) float vals[kScSlots * 16];
    const uint32_t M = live_count(Mcap, m_dev);
    // two jobs: their workgroups ALTERNATE along x, so that the two grids' work is resident together -- one grid's levels are bound
    // by memory-side atomic requests, the other's by LDS atomics
    __shared__ uint32_t occupied;
    if (n_jobs == 1) scatter_lines<D0>(j0, Mcap, M, keys, vals, &occupied, blockIdx.x);
    else if ((blockIdx.x & 1u) == 0) scatter_lines<D0>(j0, Mcap, M, keys, vals, &occupied, blockIdx.x >> 1);
    else scatter_lines<D1>(j1, Mcap, M, keys, vals, &occupied, blockIdx.x >> 1);
}

// Binned levels.  A level whose gradient table is much larger than what one workgroup's samples touch (the hashed levels of the
// T = 2^19 table: 65 536 lines each, touched ~5 times per launch, never twice by the same workgroup) gains nothing from a
// per-workgroup merge: every (sample, corner) is a memory-side atomic request of its own line.  Those levels are summed by TABLE
// REGION instead: pass A (k_grid_bin) appends (row, w g0, w g1) entries to the bucket that owns the row -- a bucket = 2^shift
// consecutive rows of one level -- and pass B (k_grid_scatter_buckets) has one workgroup per bucket add the bucket's entries in
// LDS and update the region with plain coalesced loads and stores: no global float atomic at all on those levels.  A workgroup
// of pass A reserves room in the buckets with one returning atomic per bucket (LDS histogram of its 256 samples x 2^D corners).
constexpr uint32_t kMaxBucketsPerLevel = 128;
struct BinPlan {
    uint32_t n_levels, level_of[kMaxLevels];   // the binned levels
    uint32_t bucket0[kMaxLevels];     // first bucket of level_of[i]
    uint32_t n_buckets[kMaxLevels];
    uint32_t shift, cap;              // rows per bucket = 1 << shift; entries a bucket has room for
    uint32_t *cursor;                 // [total buckets] entries appended (zero before pass A; pass B leaves it zero)
    uint32_t *e_row;                  // [total buckets][cap] level-local row
    float2 *e_val;                    // [total buckets][cap]
    // Entries that find their bucket full go to ONE spill list with room for every entry of a launch (it cannot overflow); each
    // pass-B workgroup picks its own out of it.  Hashed rows load the buckets evenly, so the list stays empty unless many samples
    // coincide -- correctness does not depend on the bucket size, only speed does.
    uint32_t *spill_count;            // [2]: entries spilled | pass-B workgroups that have read it (the last one resets both)
    uint32_t *spill_key;              // [spill capacity] bucket << 16 | bucket-local row (rows per bucket <= 2^13)
    float2 *spill_val;
};
constexpr uint32_t kBinThreads = 256;

template <uint32_t D>
__global__ void __launch_bounds__(kBinThreads) k_grid_bin(ScatterJob j, uint32_t Mcap, const int32_t *__restrict__ m_dev, BinPlan bp) {
    constexpr uint32_t NC = 1u << D;
    __shared__ uint32_t hist[kMaxBucketsPerLevel], base[kMaxBucketsPerLevel];
    const uint32_t M = live_count(Mcap, m_dev);
    if (blockIdx.x * kBinThreads >= M) return;
    const uint32_t li = blockIdx.y, level = bp.level_of[li];
    if (threadIdx.x < kMaxBucketsPerLevel) hist[threadIdx.x] = 0u;
    __syncthreads();
    const uint32_t off = (uint32_t)j.offsets[level];
    const uint32_t hashmap_size = (uint32_t)j.offsets[level + 1] - off;
    const uint32_t b = blockIdx.x * kBinThreads + threadIdx.x;      // one sample per thread, all 2^D corners
    float in[D];
    bool live = b < M;
#pragma unroll
    for (uint32_t d = 0; d < D; d++) {
        in[d] = live ? j.inputs[(size_t)b * D + d] : 0.0f;
        live = live && !(in[d] < 0 || in[d] > 1);
    }
    float pos[D], pos_deriv[D];
    uint32_t pos_grid[D];
    lattice_pos<D>(in, j.lc.scale[level], false, 0, pos, pos_deriv, pos_grid);
    float2 g = make_float2(0.0f, 0.0f);
    if (live) g = *reinterpret_cast<const float2 *>(j.grad + ((size_t)level * Mcap + b) * 2);
    const uint32_t resolution = j.lc.resolution[level];
    uint32_t rows[NC], rank[NC];
    float w[NC];
#pragma unroll
    for (uint32_t idx = 0; idx < NC; idx++) {
        float wt = 1;                                 // gridencoder.cu:298-308: x term first
        uint32_t pgl[D];
#pragma unroll
        for (uint32_t d = 0; d < D; d++) {
            const bool hi = (idx >> d) & 1u;
            wt *= hi ? pos[d] : 1 - pos[d];
            pgl[d] = pos_grid[d] + (hi ? 1u : 0u);
        }
        w[idx] = wt;
        rows[idx] = live ? grid_row<D>(j.gridtype, false, hashmap_size, resolution, pgl) : 0u;
        rank[idx] = live ? atomicAdd(&hist[rows[idx] >> bp.shift], 1u) : 0u;
    }
    __syncthreads();
    if (threadIdx.x < bp.n_buckets[li]) {
        const uint32_t n = hist[threadIdx.x];
        base[threadIdx.x] = n ? atomicAdd(&bp.cursor[bp.bucket0[li] + threadIdx.x], n) : 0u;
    }
    __syncthreads();
    if (!live) return;
    uint32_t full = 0;                                  // corners whose bucket had no room left (normally none)
#pragma unroll
    for (uint32_t idx = 0; idx < NC; idx++) {
        const uint32_t b_ = rows[idx] >> bp.shift, at = base[b_] + rank[idx];
        if (at < bp.cap) {
            const size_t slot = (size_t)(bp.bucket0[li] + b_) * bp.cap + at;
            bp.e_row[slot] = rows[idx];
            bp.e_val[slot] = make_float2(w[idx] * g.x, w[idx] * g.y);
        } else {
            full |= 1u << idx;
        }
    }
    if (full) {                                         // the spill list (sized for every entry of the launch)
#pragma unroll
        for (uint32_t idx = 0; idx < NC; idx++) {
            if (full & (1u << idx)) {
                const uint32_t sp = atomicAdd(&bp.spill_count[0], 1u);
                bp.spill_key[sp] = ((bp.bucket0[li] + (rows[idx] >> bp.shift)) << 16) | (rows[idx] & ((1u << bp.shift) - 1u));
                bp.spill_val[sp] = make_float2(w[idx] * g.x, w[idx] * g.y);
            }
        }
    }
}

// Pass B: one workgroup per bucket.  acc[rows of the bucket][2] in LDS (32 KB for 4096 rows), the bucket's entries (and its share
// of the spill list, normally empty) added with LDS atomics from all lanes, then the region WRITTEN with plain 16-byte stores:
// every row of a binned level is written by exactly one workgroup, so those levels need neither a memset nor a global atomic.
constexpr uint32_t kBkThreads = 512;
__global__ void __launch_bounds__(kBkThreads) k_grid_scatter_buckets(const int32_t *__restrict__ offsets, float *__restrict__ grad_grid,
                                                                     BinPlan bp, uint32_t total_buckets) {
    extern __shared__ __attribute__((aligned(16))) float acc[];
    __shared__ uint32_t n_sh, spill_sh;
    const uint32_t b = blockIdx.x;
    if (b >= total_buckets) return;
    uint32_t li = 0;
    for (uint32_t i = 0; i < bp.n_levels; i++)
        if (b >= bp.bucket0[i] && b < bp.bucket0[i] + bp.n_buckets[i]) li = i;
    const uint32_t level = bp.level_of[li];
    const uint32_t rows_pb = 1u << bp.shift, local = b - bp.bucket0[li];
    const uint32_t off = (uint32_t)offsets[level], rows_level = (uint32_t)offsets[level + 1] - off;
    const uint32_t row_first = local << bp.shift;
    const uint32_t n_rows = rows_level - row_first < rows_pb ? rows_level - row_first : rows_pb;
    if (threadIdx.x == 0) {
        const uint32_t n = bp.cursor[b];
        n_sh = n < bp.cap ? n : bp.cap;
        bp.cursor[b] = 0u;                     // ready for the next launch of pass A
        spill_sh = __hip_atomic_load(&bp.spill_count[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    for (uint32_t i = threadIdx.x; i < rows_pb * 2; i += kBkThreads) acc[i] = 0.0f;
    __syncthreads();
    const uint32_t n = n_sh, n_spill = spill_sh;
    const uint32_t *er = bp.e_row + (size_t)b * bp.cap;
    const float2 *ev = bp.e_val + (size_t)b * bp.cap;
    // eight entries per thread in flight: the loads of a batch are issued together, then added (a load-add-load-add chain would
    // pay the memory latency once per entry)
    for (uint32_t i0 = 0; i0 < n; i0 += kBkThreads * 8) {
        uint32_t r[8];
        float2 v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const uint32_t i = i0 + u * kBkThreads + threadIdx.x;
            const uint32_t ic = i < n ? i : n - 1u;
            r[u] = er[ic] & (rows_pb - 1u);
            v[u] = ev[ic];
            if (i >= n) v[u] = make_float2(0.0f, 0.0f);
        }
#pragma unroll
        for (int u = 0; u < 8; u++) {
            if (v[u].x != 0.0f) atomicAdd(&acc[2 * r[u]], v[u].x);
            if (v[u].y != 0.0f) atomicAdd(&acc[2 * r[u] + 1], v[u].y);
        }
    }
    for (uint32_t i = threadIdx.x; i < n_spill; i += kBkThreads) {     // normally n_spill == 0
        const uint32_t key = bp.spill_key[i];
        if ((key >> 16) == b) {
            const float2 v = bp.spill_val[i];
            atomicAdd(&acc[2 * (key & 0xffffu)], v.x);
            atomicAdd(&acc[2 * (key & 0xffffu) + 1], v.y);
        }
    }
    __syncthreads();
    float4 *dst = reinterpret_cast<float4 *>(grad_grid + ((size_t)off + row_first) * 2);   // rows are 8 B, regions start on 64-B lines
    const float4 *src = reinterpret_cast<const float4 *>(acc);
    for (uint32_t i = threadIdx.x; i < n_rows / 2; i += kBkThreads) dst[i] = src[i];
    // the last workgroup to have read the spill list empties it for the next launch
    if (threadIdx.x == 0) {
        const uint32_t done = atomicAdd(&bp.spill_count[1], 1u) + 1u;
        if (done == total_buckets) {
            __hip_atomic_store(&bp.spill_count[0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&bp.spill_count[1], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// ---- loss on the composited rays -----------------------------------------------------------------------------------------
constexpr int kLossThreads = 1024;
__global__ void __launch_bounds__(kLossThreads) k_train_head_loss(const float *__restrict__ image, const float *__restrict__ ws,
                                                                  const float *__restrict__ ambient, const float *__restrict__ bg,
                                                                  uint32_t bg_stride, const float *__restrict__ target, uint32_t target_stride,
                                                                  const float *__restrict__ face, uint32_t face_stride,
                                                                  const float *__restrict__ w_amb, uint32_t N, float *__restrict__ loss,
                                                                  float *__restrict__ pred, float *__restrict__ g_image,
                                                                  float *__restrict__ g_ws, float *__restrict__ g_amb) {
    __shared__ double red[kLossThreads / kWave];
    const float wa = w_amb[0];
    const float inv_n = 1.0f / (float)N;
    double acc = 0.0;
    for (uint32_t n = threadIdx.x; n < N; n += kLossThreads) {
        const float w = ws[n];
        const float tr = 1.0f - w;                     // nerf/renderer.py:306: image + (1 - weights_sum) * bg, clamp [0, 1]
        float mse = 0.0f, gws_blend = 0.0f;
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const float b = bg[(size_t)n * bg_stride + c];
            const float raw = image[n * 3 + c] + tr * b;
            const float pr = fminf(fmaxf(raw, 0.0f), 1.0f);
            if (pred) pred[n * 3 + c] = pr;
            const float d = pr - target[(size_t)n * target_stride + c];
            mse += d * d;
            const float gp = (raw >= 0.0f && raw <= 1.0f) ? 2.0f * d * (inv_n / 3.0f) : 0.0f;   // clamp passes the gradient on [0, 1]
            g_image[n * 3 + c] = gp;
            gws_blend -= gp * b;
        }
        const float a = fminf(fmaxf(w, 1e-5f), 1.0f - 1e-5f);
        const float la = log2f(a), lb = log2f(1.0f - a);
        const float ent = -a * la - (1.0f - a) * lb;
        const bool inside = w >= 1e-5f && w <= 1.0f - 1e-5f;
        g_ws[n] = gws_blend + (inside ? 1e-4f * inv_n * (lb - la) : 0.0f);
        const float keep = 1.0f - face[(size_t)n * face_stride];
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

// ---- batch gather ---------------------------------------------------------------------------------------------------------
// A training batch = n rows picked from a per-pixel table [n_px, row_floats] whose columns are up to 8 sections (rays_o | rays_d |
// bg_coords | bg_color | target | face ...): ONE kernel writes every section as its own contiguous [n, width] array (the
// operators want contiguous rays), where stock indexing takes one gather + one strided copy per section.
struct GatherArgs {
    uint32_t width[8], col0[8], out0[8];   // section widths, first column in the table row, first float in `out`
    uint32_t sections, row_floats;
};
__global__ void __launch_bounds__(256) k_batch_gather(const float *__restrict__ table, const int64_t *__restrict__ idx, uint32_t n,
                                                      GatherArgs a, float *__restrict__ out) {
    const uint32_t t = blockIdx.x * 256 + threadIdx.x;
    if (t >= n * a.row_floats) return;
    const uint32_t r = t / a.row_floats, c = t - r * a.row_floats;
    uint32_t sec = 0;
    while (sec + 1 < a.sections && c >= a.col0[sec + 1]) sec++;
    out[a.out0[sec] + r * a.width[sec] + (c - a.col0[sec])] = table[(size_t)idx[r] * a.row_floats + c];
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
constexpr uint32_t kWPartsMax = 256;
static uint32_t wparts() {   // workgroups (= partial sums) per weight-gradient job
    static uint32_t n = 0;
    if (!n) {
        const char *e = getenv("RN_TRAIN_WPARTS");
        const long v = e ? atol(e) : 128;          // measured at 62 k samples: 64 parts 76 us, 96: 80, 128: 71, 192: 85
        n = (uint32_t)(v < 1 ? 1 : (v > (long)kWPartsMax ? (long)kWPartsMax : v));
    }
    return n;
}
static RawW raw_w(const rn_nerf_weights_t *w) {
    return RawW{w->amb_w0, w->amb_w1, w->amb_w2, w->sig_w0, w->sig_w1, w->sig_w2, w->col_w0, w->col_w1, w->audio_dim, w->has_eye, w->ind_dim};
}
static int check_w(const rn_nerf_weights_t *w) {
    RN_REQUIRE(w && w->amb_w0 && w->amb_w1 && w->amb_w2 && w->sig_w0 && w->sig_w1 && w->sig_w2 && w->col_w0 && w->col_w1,
               "train_head: null weight pointer");
    RN_REQUIRE(w->has_eye <= 1, "train_head: has_eye must be 0 or 1");
    return RN_OK;
}
static int check_grid(const rn_grid_t *g, uint32_t D, const char *name) {
    RN_REQUIRE(g && g->embeddings && g->offsets, "train_head: %s grid is null", name);
    RN_REQUIRE(g->D == D && g->L == 16 && g->dtype == RN_F32, "train_head: %s grid must be D=%u, L=16, fp32 with C=2", name, D);
    return RN_OK;
}
static GridArgs grid_args(const rn_grid_t *g) {
    return GridArgs{g->embeddings, g->offsets, make_level_consts(g->L, g->S, g->H), g->gridtype};
}

}  // namespace th
}  // namespace rn

using namespace rn;
using namespace rn::th;

extern "C" {

size_t rn_train_head_image_floats(void) { return (size_t)kImage; }
size_t rn_train_head_workspace_floats(uint32_t M) { return (size_t)((M + 31u) >> 5) * kWsPerTile; }
size_t rn_train_head_wgrad_workspace(void) { return ((size_t)kJobs * kWPartsMax * 96 * 96 + 192) * sizeof(float); }

int rn_train_head_pack(const rn_nerf_weights_t *w, const float *enc_a, const float *eye, const float *ind_code, float *image,
                       rn_stream_t stream) {
    if (int rc = check_w(w)) return rc;
    RN_REQUIRE(image && ((uintptr_t)image & 15u) == 0, "train_head_pack: image must be 16-byte aligned");
    RN_REQUIRE((enc_a || w->audio_dim == 0) && (eye || !w->has_eye) && (ind_code || w->ind_dim == 0), "train_head_pack: null constant");
    hipLaunchKernelGGL(k_train_pack, dim3(div_up(kImage, 256)), dim3(256), 0, as_stream(stream), raw_w(w), enc_a, eye, ind_code,
                       static_cast<const int64_t *>(nullptr), image);
    return check_launch("train_head_pack");
}

int rn_train_head_pack_row(const rn_nerf_weights_t *w, const float *enc_a, const float *eye, const float *ind_table,
                           const int64_t *ind_index, float *image, rn_stream_t stream) {
    if (int rc = check_w(w)) return rc;
    RN_REQUIRE(image && ((uintptr_t)image & 15u) == 0, "train_head_pack_row: image must be 16-byte aligned");
    RN_REQUIRE((enc_a || w->audio_dim == 0) && (eye || !w->has_eye) && ind_table && ind_index && w->ind_dim, "train_head_pack_row: null constant");
    hipLaunchKernelGGL(k_train_pack, dim3(div_up(kImage, 256)), dim3(256), 0, as_stream(stream), raw_w(w), enc_a, eye, ind_table, ind_index,
                       image);
    return check_launch("train_head_pack_row");
}

int rn_train_head_forward(const float *xyzs, const float *dirs, uint32_t M, const int32_t *m_dev, const rn_grid_t *grid_xyz,
                          const rn_grid_t *grid_amb, const float *image, float bound, float *sigmas, float *rgbs,
                          float *ambient, float *ambient_abs, float *xn, float *wn, float *workspace, rn_stream_t stream) {
    if (M == 0) return RN_OK;
    if (int rc = check_grid(grid_xyz, 3, "xyz")) return rc;
    if (int rc = check_grid(grid_amb, 2, "ambient")) return rc;
    RN_REQUIRE(xyzs && dirs && image && sigmas && rgbs && ambient && xn && wn && workspace, "train_head_forward: null pointer");
    RN_REQUIRE(((uintptr_t)image & 15u) == 0 && ((uintptr_t)workspace & 15u) == 0, "train_head_forward: image / workspace must be 16-byte aligned");
    FwdParams p{xyzs, dirs, M, m_dev, grid_args(grid_xyz), grid_args(grid_amb), image, bound, sigmas, rgbs, ambient, ambient_abs, xn, wn, workspace};
    const uint32_t n_tiles = (M + 31u) >> 5;
    uint32_t blocks = div_up(n_tiles, kWaves);
    const uint32_t cap = (uint32_t)num_cus();
    if (blocks > cap) blocks = cap;
    static int groups = -1;
    if (groups < 0) { const char *e = getenv("RN_TRAIN_FWD_GROUPS"); groups = e ? atoi(e) : 11; }
    // measured at 62 k samples (tools/bench_train_head.py): <1,1> 82.5 us, <1,2> 82.4, <2,2> 84.5, <2,4> 85.9 -- with one tile per
    // wave the two waves of a SIMD already hide each other's gathers; more rounds in flight only cost registers
    if (groups == 12) hipLaunchKernelGGL((k_train_fwd<1, 2>), dim3(blocks), dim3(kThreads), 0, as_stream(stream), p);
    else hipLaunchKernelGGL((k_train_fwd<1, 1>), dim3(blocks), dim3(kThreads), 0, as_stream(stream), p);
    return check_launch("train_head_forward");
}

int rn_train_head_backward(const float *grad_sigmas, const float *grad_rgbs, const float *grad_ambient,
                           const float *grad_ambient_abs, const float *rgbs, const float *ambient, uint32_t M,
                           const int32_t *m_dev, const float *image, float *workspace, float *grad_enc_x, float *grad_enc_w,
                           rn_stream_t stream) {
    if (M == 0) return RN_OK;
    RN_REQUIRE(grad_sigmas && grad_rgbs && rgbs && ambient && image && workspace && grad_enc_x && grad_enc_w, "train_head_backward: null pointer");
    RN_REQUIRE(((uintptr_t)grad_enc_x & 7u) == 0 && ((uintptr_t)grad_enc_w & 7u) == 0, "train_head_backward: feature gradients must be 8-byte aligned");
    BwdParams p{grad_sigmas, grad_rgbs, grad_ambient, grad_ambient_abs, rgbs, ambient, M, m_dev, image, workspace, grad_enc_x, grad_enc_w};
    const uint32_t n_tiles = (M + 31u) >> 5;
    uint32_t blocks = div_up(n_tiles, kWaves);
    const uint32_t cap = (uint32_t)num_cus();
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL(k_train_bwd, dim3(blocks), dim3(kThreads), 0, as_stream(stream), p);
    return check_launch("train_head_backward");
}

static int weight_grads(const rn_nerf_weights_t *w, const float *enc_a, const float *eye, const float *ind_code, const int64_t *ind_index,
                        uint32_t ind_rows, uint32_t M, const int32_t *m_dev, const float *workspace, const rn_train_head_grads_t *g,
                        void *wgrad_workspace, rn_stream_t stream) {
    if (int rc = check_w(w)) return rc;
    RN_REQUIRE(M > 0 && workspace && g && wgrad_workspace, "train_head_weight_grads: null pointer / M == 0");
    RN_REQUIRE(g->amb_w0 && g->amb_w1 && g->amb_w2 && g->sig_w0 && g->sig_w1 && g->sig_w2 && g->col_w0 && g->col_w1,
               "train_head_weight_grads: null gradient pointer");
    RN_REQUIRE((enc_a || w->audio_dim == 0) && (eye || !w->has_eye) && (ind_code || w->ind_dim == 0), "train_head_weight_grads: null constant");
    hipStream_t s = as_stream(stream);
    float *partial = static_cast<float *>(wgrad_workspace);
    float *gb = partial + (size_t)kJobs * kWPartsMax * 96 * 96;
    WArgs a{const_cast<float *>(workspace), M, m_dev, wparts(), partial};
    hipLaunchKernelGGL(k_train_wgrad, dim3(kJobs * a.parts), dim3(kWThreads), 0, s, a);
    const uint32_t ldA0 = 32 + w->audio_dim, ldS0 = 64 + w->has_eye, ldC0 = 80 + w->ind_dim;
    RArgs r{};
    r.partial = partial;
    r.parts = a.parts;
    r.job[J_A0] = RJob{g->amb_w0, 64, 32, ldA0, 32, gb};
    r.job[J_A1] = RJob{g->amb_w1, 64, 64, 64, -1, nullptr};
    r.job[J_A2] = RJob{g->amb_w2, 2, 64, 64, -1, nullptr};
    r.job[J_S0] = RJob{g->sig_w0, 64, 64, ldS0, 64, gb + 64};
    r.job[J_S1] = RJob{g->sig_w1, 64, 64, 64, -1, nullptr};
    r.job[J_S2] = RJob{g->sig_w2, 65, 64, 64, -1, nullptr};
    r.job[J_C0] = RJob{g->col_w0, 64, 80, ldC0, 80, gb + 128};
    r.job[J_C1] = RJob{g->col_w1, 3, 64, 64, -1, nullptr};
    hipLaunchKernelGGL(k_train_wreduce, dim3(div_up(96 * 96, 256), kJobs), dim3(256), 0, s, r);
    CArgs c{raw_w(w), enc_a, eye, ind_code, gb, g->amb_w0, g->sig_w0, g->col_w0, g->enc_a, g->eye, g->ind_code, ind_index, ind_rows};
    const uint32_t zero_blocks = ind_index ? (div_up(ind_rows * w->ind_dim, 1024) < 64u ? div_up(ind_rows * w->ind_dim, 1024) : 64u) : 0u;
    hipLaunchKernelGGL(k_train_const, dim3(3 + zero_blocks), dim3(256), 0, s, c);
    return check_launch("train_head_weight_grads");
}

int rn_train_head_weight_grads(const rn_nerf_weights_t *w, const float *enc_a, const float *eye, const float *ind_code,
                               uint32_t M, const int32_t *m_dev, const float *workspace, const rn_train_head_grads_t *g,
                               void *wgrad_workspace, rn_stream_t stream) {
    return weight_grads(w, enc_a, eye, ind_code, nullptr, 0, M, m_dev, workspace, g, wgrad_workspace, stream);
}

int rn_train_head_weight_grads_row(const rn_nerf_weights_t *w, const float *enc_a, const float *eye, const float *ind_table,
                                   const int64_t *ind_index, uint32_t ind_rows, uint32_t M, const int32_t *m_dev, const float *workspace,
                                   const rn_train_head_grads_t *g, void *wgrad_workspace, rn_stream_t stream) {
    RN_REQUIRE(ind_table && ind_index && ind_rows && g && g->ind_code && w && w->ind_dim,
               "train_head_weight_grads_row: the code table, its row index and the gradient table are required");
    return weight_grads(w, enc_a, eye, ind_table, ind_index, ind_rows, M, m_dev, workspace, g, wgrad_workspace, stream);
}

}  // extern "C"

namespace rn {
namespace th {

// Which levels are binned: the HASHED ones with at least kMinBuckets buckets of 2^shift rows.  A hash spreads the rows evenly
// over the buckets whatever the samples' positions, so a bucket's load is known in advance (2x the mean + slack is never
// reached) and pass B has hundreds of equal workgroups.  Dense and tiled levels keep the per-workgroup line merge: their rows
// follow the samples' positions (the ambient coordinates of a call cluster in a few cells), which is where neighbouring samples
// share lines and where a fixed bucket size would overflow.
constexpr uint32_t kMinBuckets = 16;
static uint32_t bucket_shift() {
    static uint32_t sh = 0;
    if (!sh) {
        const char *e = getenv("RN_SCATTER_BUCKET_SHIFT");
        const long v = e ? atol(e) : 12;           // 4096 rows = 32 KB of LDS per bucket: four pass-B workgroups per CU
        sh = (uint32_t)(v < 10 ? 10 : (v > 13 ? 13 : v));
    }
    return sh;
}
static uint32_t plan_bins(const rn_grid_t *grid, const int32_t *offsets_host, uint32_t M, BinPlan &bp, bool *binned /* [L] */) {
    bp = BinPlan{};
    bp.shift = bucket_shift();
    const LevelConsts lc = make_level_consts(grid->L, grid->S, grid->H);
    uint32_t total = 0, min_b = kMaxBucketsPerLevel;
    for (uint32_t l = 0; l < grid->L; l++) {
        const uint32_t rows = (uint32_t)(offsets_host[l + 1] - offsets_host[l]);
        uint64_t stride = 1;                                      // gridencoder.cu:66-84: hashed when the dense index does not fit
        for (uint32_t d = 0; d < grid->D; d++)
            if (stride <= rows) stride *= (uint64_t)lc.resolution[l] + 1u;
        const bool hashed = grid->gridtype == 0 && stride > rows;
        const uint32_t nb = (rows + (1u << bp.shift) - 1u) >> bp.shift;
        const bool bin = hashed && nb >= kMinBuckets && nb <= kMaxBucketsPerLevel;
        if (binned) binned[l] = bin;
        if (bin) {
            bp.level_of[bp.n_levels] = l;
            bp.bucket0[bp.n_levels] = total;
            bp.n_buckets[bp.n_levels] = nb;
            bp.n_levels++;
            total += nb;
            if (nb < min_b) min_b = nb;
        }
    }
    bp.cap = total ? (uint32_t)(2u * (((uint64_t)M << grid->D) / min_b) + 2048u) : 0u;   // 2 x the mean load of a bucket + slack
    return total;
}
static bool scatter_direct_enabled() {
    static int on = -1;
    if (on < 0) { const char *e = getenv("RN_SCATTER_DIRECT"); on = e ? atoi(e) : 1; }
    return on != 0;
}
static uint32_t scatter_chunks() {
    static int n = 0;
    if (!n) { const char *e = getenv("RN_SCATTER_CHUNKS"); n = e ? atoi(e) : 8; if (n < 1) n = 1; if (n > 16) n = 16; }
    return (uint32_t)n;
}
static ScatterJob make_job(const rn_scatter_job_t &j) {
    ScatterJob s{};
    s.grad = j.grad;
    s.inputs = j.inputs;
    s.offsets = j.grid->offsets;
    s.grad_grid = j.grad_table;
    s.lc = make_level_consts(j.grid->L, j.grid->S, j.grid->H);
    s.gridtype = j.grid->gridtype;
    return s;
}
template <uint32_t D0>
static void launch_lines(uint32_t D1, dim3 g, hipStream_t s, const ScatterJob &a, const ScatterJob &b, uint32_t n_jobs, uint32_t M,
                         const int32_t *m_dev) {
    if (D1 == 3) hipLaunchKernelGGL((k_grid_scatter<D0, 3>), g, dim3(kScThreads), 0, s, a, b, n_jobs, M, m_dev);
    else hipLaunchKernelGGL((k_grid_scatter<D0, 2>), g, dim3(kScThreads), 0, s, a, b, n_jobs, M, m_dev);
}

}  // namespace th
}  // namespace rn

using namespace rn;
using namespace rn::th;

extern "C" {

size_t rn_grid_scatter_workspace(uint32_t M, const rn_grid_t *grid, const int32_t *offsets_host) {
    if (!grid || !offsets_host || grid->L > kMaxLevels) return 0;
    BinPlan bp;
    const uint32_t total = plan_bins(grid, offsets_host, M, bp, nullptr);
    if (!total) return 256;
    const size_t spill = ((size_t)M << grid->D) * bp.n_levels;        // every entry of a launch fits the spill list
    return (((size_t)(total + 2) * sizeof(uint32_t) + 255u) & ~(size_t)255u) + ((size_t)total * bp.cap + spill) * (sizeof(uint32_t) + sizeof(float2)) + 256;
}

uint32_t rn_grid_scatter_binned_levels(const rn_grid_t *grid, const int32_t *offsets_host) {
    if (!grid || !offsets_host || grid->L > kMaxLevels) return 0;
    BinPlan bp;
    bool binned[kMaxLevels] = {};
    (void)plan_bins(grid, offsets_host, 1, bp, binned);
    uint32_t mask = 0;
    for (uint32_t l = 0; l < grid->L; l++) mask |= binned[l] ? (1u << l) : 0u;
    return mask;
}

int rn_grid_scatter_jobs(const rn_scatter_job_t *jobs, uint32_t n_jobs, uint32_t M, const int32_t *m_dev, void *workspace,
                         size_t workspace_bytes, rn_stream_t stream) {
    if (M == 0) return RN_OK;
    RN_REQUIRE(jobs && (n_jobs == 1 || n_jobs == 2), "grid_scatter_jobs: one or two jobs");
    for (uint32_t i = 0; i < n_jobs; i++) {
        const rn_scatter_job_t &j = jobs[i];
        RN_REQUIRE(j.grad && j.inputs && j.grid && j.grid->offsets && j.grad_table, "grid_scatter_jobs: null pointer in job %u", i);
        RN_REQUIRE((j.grid->D == 2 || j.grid->D == 3) && j.grid->L >= 1 && j.grid->L <= kMaxLevels, "grid_scatter_jobs: D must be 2 or 3, L <= 32");
        RN_REQUIRE(((uintptr_t)j.grad_table & 63u) == 0 && ((uintptr_t)j.grad & 7u) == 0, "grid_scatter_jobs: grad_table must be 64-byte, grad 8-byte aligned");
    }
    hipStream_t s = as_stream(stream);
    ScatterJob sj[2] = {make_job(jobs[0]), n_jobs == 2 ? make_job(jobs[1]) : ScatterJob{}};
    // job 0 may have binned levels (needs the host copy of its offsets and the workspace)
    bool binned[kMaxLevels] = {};
    BinPlan bp{};
    uint32_t total = 0;
    if (jobs[0].offsets_host && workspace && workspace_bytes > 256) {
        total = plan_bins(jobs[0].grid, jobs[0].offsets_host, M, bp, binned);
        if (total) {
            RN_REQUIRE(((uintptr_t)workspace & 255u) == 0 && workspace_bytes >= rn_grid_scatter_workspace(M, jobs[0].grid, jobs[0].offsets_host),
                       "grid_scatter_jobs: workspace too small / not 256-byte aligned");
            // workspace = cursors (zeroed once by the caller; pass B leaves them zero) | values | rows
            char *w = static_cast<char *>(workspace);
            const size_t spill = ((size_t)M << jobs[0].grid->D) * bp.n_levels;
            bp.cursor = reinterpret_cast<uint32_t *>(w);
            bp.spill_count = bp.cursor + total;
            size_t at = ((size_t)(total + 2) * sizeof(uint32_t) + 255u) & ~(size_t)255u;
            bp.e_val = reinterpret_cast<float2 *>(w + at);
            at += (size_t)total * bp.cap * sizeof(float2);
            bp.spill_val = reinterpret_cast<float2 *>(w + at);
            at += spill * sizeof(float2);
            bp.e_row = reinterpret_cast<uint32_t *>(w + at);
            at += (size_t)total * bp.cap * sizeof(uint32_t);
            bp.spill_key = reinterpret_cast<uint32_t *>(w + at);
        }
    }
    uint32_t max_levels = 0, max_blocks = 0;
    for (uint32_t i = 0; i < n_jobs; i++) {
        const rn_grid_t *gr = jobs[i].grid;
        const LevelConsts lc = make_level_consts(gr->L, gr->S, gr->H);
        for (uint32_t l = 0; l < gr->L; l++) {
            if (i == 0 && total && binned[l]) continue;
            // hashed (gridencoder.cu:66-84) AND large (>= 2^17 rows: a workgroup's 64 samples x 8 corners land on distinct
            // lines): straight to memory.  Needs a host view of the level sizes: jobs[i].offsets_host (else: line merge)
            bool direct = false;
            if (jobs[i].offsets_host && scatter_direct_enabled()) {
                const uint32_t rows = (uint32_t)(jobs[i].offsets_host[l + 1] - jobs[i].offsets_host[l]);
                uint64_t stride = 1;
                for (uint32_t d = 0; d < gr->D; d++)
                    if (stride <= rows) stride *= (uint64_t)lc.resolution[l] + 1u;
                direct = gr->gridtype == 0 && stride > rows && rows >= (1u << 17);
            }
            if (direct) sj[i].direct_mask |= 1u << sj[i].n_levels;
            uint32_t chunks = 1;
            if (!direct && jobs[i].offsets_host) {
                const uint32_t rows = (uint32_t)(jobs[i].offsets_host[l + 1] - jobs[i].offsets_host[l]);
                chunks = rows <= (1u << 16) ? scatter_chunks() : 1u;
            }
            sj[i].chunks_of[sj[i].n_levels] = chunks;
            sj[i].level_of[sj[i].n_levels++] = l;
        }
        if (sj[i].n_levels > max_levels) max_levels = sj[i].n_levels;
        const uint32_t blocks = div_up(M, kScThreads >> (jobs[i].grid->D - 1));
        if (blocks > max_blocks) max_blocks = blocks;
    }
    if (total) {
        const dim3 gb(div_up(M, kBinThreads), bp.n_levels);
        if (jobs[0].grid->D == 3) hipLaunchKernelGGL(k_grid_bin<3>, gb, dim3(kBinThreads), 0, s, sj[0], M, m_dev, bp);
        else hipLaunchKernelGGL(k_grid_bin<2>, gb, dim3(kBinThreads), 0, s, sj[0], M, m_dev, bp);
        const size_t shm = (size_t)2 * sizeof(float) << bp.shift;
        if (shm > 64 * 1024)
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_grid_scatter_buckets), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
        hipLaunchKernelGGL(k_grid_scatter_buckets, dim3(total), dim3(kBkThreads), shm, s, jobs[0].grid->offsets, jobs[0].grad_table, bp, total);
    }
    if (max_levels) {
        const dim3 g(n_jobs == 2 ? 2 * max_blocks : max_blocks, max_levels);
        const uint32_t D1 = n_jobs == 2 ? jobs[1].grid->D : 2u;
        if (jobs[0].grid->D == 3) launch_lines<3>(D1, g, s, sj[0], sj[1], n_jobs, M, m_dev);
        else launch_lines<2>(D1, g, s, sj[0], sj[1], n_jobs, M, m_dev);
    }
    return check_launch("grid_scatter_jobs");
}

int rn_grid_scatter_binned(const float *grad, const float *inputs, uint32_t M, const int32_t *m_dev, const rn_grid_t *grid,
                           const int32_t *offsets_host, float *grad_table, void *workspace, size_t workspace_bytes, rn_stream_t stream) {
    RN_REQUIRE(offsets_host && workspace, "grid_scatter_binned: null pointer");
    const rn_scatter_job_t j{grad, inputs, grid, offsets_host, grad_table};
    return rn_grid_scatter_jobs(&j, 1, M, m_dev, workspace, workspace_bytes, stream);
}

int rn_grid_scatter_lbc(const float *grad, const float *inputs, uint32_t M, const int32_t *m_dev, const rn_grid_t *grid,
                        float *grad_table, rn_stream_t stream) {
    const rn_scatter_job_t j{grad, inputs, grid, nullptr, grad_table};
    return rn_grid_scatter_jobs(&j, 1, M, m_dev, nullptr, 0, stream);
}

int rn_train_batch_gather(const float *table, uint32_t row_floats, const int64_t *idx, uint32_t n, const uint32_t *widths,
                          uint32_t sections, float *out, rn_stream_t stream) {
    if (n == 0) return RN_OK;
    RN_REQUIRE(table && idx && widths && out && sections >= 1 && sections <= 8, "train_batch_gather: null pointer / 1 .. 8 sections");
    GatherArgs a{};
    uint32_t col = 0;
    for (uint32_t i = 0; i < sections; i++) {
        RN_REQUIRE(widths[i] > 0, "train_batch_gather: section %u has width 0", i);
        a.width[i] = widths[i];
        a.col0[i] = col;
        a.out0[i] = col * n;      // sections follow each other in `out`: [n, w0] | [n, w1] | ...
        col += widths[i];
    }
    RN_REQUIRE(col == row_floats, "train_batch_gather: the section widths must add up to the row length");
    a.sections = sections;
    a.row_floats = row_floats;
    hipLaunchKernelGGL(k_batch_gather, dim3(div_up(n * row_floats, 256)), dim3(256), 0, as_stream(stream), table, idx, n, a, out);
    return check_launch("train_batch_gather");
}

int rn_train_head_loss(const float *image, const float *weights_sum, const float *ambient, const float *bg, uint32_t bg_stride,
                       const float *target, uint32_t target_stride, const float *face, uint32_t face_stride, const float *w_amb,
                       uint32_t N, float *loss, float *pred, float *grad_image, float *grad_weights_sum, float *grad_ambient,
                       rn_stream_t stream) {
    RN_REQUIRE(N > 0, "train_head_loss: N must be positive");
    RN_REQUIRE(image && weights_sum && ambient && bg && target && face && w_amb && loss && grad_image && grad_weights_sum && grad_ambient,
               "train_head_loss: null pointer");
    hipLaunchKernelGGL(k_train_head_loss, dim3(1), dim3(kLossThreads), 0, as_stream(stream), image, weights_sum, ambient, bg, bg_stride, target,
                       target_stride, face, face_stride, w_amb, N, loss, pred, grad_image, grad_weights_sum, grad_ambient);
    return check_launch("train_head_loss");
}

}  // extern "C"
