// rn_fused.hip -- the fused per-sample network kernel and the device-side inference loop (gfx950).
//
// C ABI: include/radnerf_fused.h.  What is computed: nerf/network.py:222-283 per sample and the inference
// branch of nerf/renderer.py:225-262 per frame.  How (MI355X-first):
//
//  * one wavefront owns a tile of 32 samples; both lane halves work on the same samples.  A gather round has lane half
//    h fetch level 2 r + h of its sample (multires grid rows with 8-byte loads, same device code as the standalone
//    encoder, so features are bit-identical); the MLP phases run on v_mfma_f32_32x32x2_f32 (exact fp32 FMA chains):
//    outputs on the 32 rows of the tile, samples on its 32 columns, k on the lane half -- so a gathered feature IS the
//    B operand of an MFMA step (no cross-lane move) and a grid costs 8 rounds, not 16 levels, of latency.
//  * a 32x32 accumulator has the sample on the lane and the output row on the register index, i.e. it already IS the
//    B operand of the next layer (k order permuted -- the weight image is packed in that order once, on the device).
//  * all weights (95.7 KB fp32) sit in LDS for the lifetime of a persistent 768-thread workgroup (one per CU, three
//    waves per SIMD: 3 x 32 accumulator VGPRs leave room for that).  fp32 MFMA and fp32 VALU work share the FMA rate of a SIMD ("where the time goes"),
//    so the instruction stream around the MFMAs is kept short: per-level plans (rn_grid_dev.h), one-instruction ReLU.
//  * the torso pass (k_torso_fused) keeps the older 64-sample form: two column tiles per wave, one
//    v_permlane32_swap per feature pair to build both B operands.
//  * inputs that are the same for every sample of a frame (audio code, eye, individual code) never enter
//    the per-sample GEMMs: they are folded into 3 x 64 bias values per frame, used as the accumulators'
//    initial value.  Outputs narrower than a tile (ambient 2, sigma 1, rgb 3) are VALU dot products over
//    the accumulator registers + one cross-half shuffle.
//  * the inference loop keeps n_alive / step / n_step in device memory (double-buffered state words), so
//    a frame is enqueued without a single host read-back; compaction is a stable ballot/mbcnt scatter.
#include "rn_fused_dev.h"

#include <float.h>

namespace rn {

#ifndef RN_FUSED_PAIR_HASHED
#define RN_FUSED_PAIR_HASHED 0
#endif
#ifndef RN_F32_XYZ_GROUP
#define RN_F32_XYZ_GROUP 1
#endif
#ifndef RN_F32_AMB_GROUP
#define RN_F32_AMB_GROUP 2
#endif
constexpr int kXyzGroup = RN_F32_XYZ_GROUP, kAmbGroup = RN_F32_AMB_GROUP;  // gather rounds in flight per wave (divide 8)
#ifndef RN_F32_WAVES
#define RN_F32_WAVES 12
#endif
constexpr int kF32Waves = RN_F32_WAVES, kF32Threads = kF32Waves * kWave;  // 3 waves per SIMD (accumulators: 3 x 32 VGPRs), one workgroup per CU
constexpr bool kPairHashed = RN_FUSED_PAIR_HASHED;  // aligned x-pair loads on hashed levels inside the fused kernels

// ---- packed weight image (floats) --------------------------------------------------------------------
// MFMA layers: [step][h][col j][row tile] -> lane (j, h) reads one float2 per step.
constexpr int kStep = 128;                           // floats per MFMA step (2 halves x 32 lanes x 2 row tiles)
constexpr int OFF_A0 = 0;                            // ambient L0, enc_x part : 16 steps
constexpr int OFF_A1 = OFF_A0 + 16 * kStep;          // ambient L1            : 32 steps
constexpr int OFF_A2 = OFF_A1 + 32 * kStep;          // ambient L2 (VALU)     : [2 out][2 h][32]
constexpr int OFF_S0 = OFF_A2 + 128;                 // sigma L0 (enc_x|enc_w): 32 steps
constexpr int OFF_S1 = OFF_S0 + 32 * kStep;          // sigma L1              : 32 steps
constexpr int OFF_S2 = OFF_S1 + 32 * kStep;          // sigma L2 rows 1..64   : 32 steps
constexpr int OFF_S2R = OFF_S2 + 32 * kStep;         // sigma L2 row 0 (VALU) : [2 h][32]
constexpr int OFF_C0 = OFF_S2R + 64;                 // color L0 (sh | geo)   : 8 + 32 steps
constexpr int OFF_C1 = OFF_C0 + 40 * kStep;          // color L1 (VALU)       : [3 out][2 h][32]
constexpr int kPacked = OFF_C1 + 192;                // 23936 floats
constexpr int kBias = 192;                           // amb | sig | col, 64 each
constexpr int kLdsFloats = kPacked + kBias;          // 96512 B of LDS

__global__ void __launch_bounds__(256) k_pack_nerf(RawW w, float *__restrict__ packed) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= kPacked) return;
    const int ldA0 = 32 + (int)w.audio_dim, ldS0 = 64 + (int)w.has_eye, ldC0 = 80 + (int)w.ind_dim;
    float v;
    auto mfma_elem = [&](int base, const float *src, int ld, int kind) -> float {
        const int q = e - base, s = q / kStep, rem = q % kStep;
        const int h = rem / 64, j = (rem % 64) / 2, rt = rem % 2;
        const int row = 32 * rt + j;
        int k;
        if (kind == 0) k = 2 * s + h;                                  // natural feature pairs
        else if (kind == 3) k = 4 * (s >> 1) + 2 * h + (s & 1);       // gather rounds: half h holds level 2 (s / 2) + h, steps = its 2 features
        else if (kind == 1) k = kmap(s, h);                            // previous accumulators
        else k = (s < 8) ? 2 * s + h : 16 + kmap(s - 8, h);            // color L0: sh pairs then geo accumulators
        return src[row * ld + k];
    };
    auto valu_elem = [&](int base, const float *src) -> float {        // [out][h][q], q = rt*16 + r
        const int q0 = e - base, o = q0 / 64, h = (q0 % 64) / 32, q = q0 % 32;
        return src[o * 64 + 32 * (q >> 4) + rowmap(q & 15, h)];
    };
    if (e < OFF_A1) v = mfma_elem(OFF_A0, w.amb_w0, ldA0, 3);
    else if (e < OFF_A2) v = mfma_elem(OFF_A1, w.amb_w1, 64, 1);
    else if (e < OFF_S0) v = valu_elem(OFF_A2, w.amb_w2);
    else if (e < OFF_S1) v = mfma_elem(OFF_S0, w.sig_w0, ldS0, 3);
    else if (e < OFF_S2) v = mfma_elem(OFF_S1, w.sig_w1, 64, 1);
    else if (e < OFF_S2R) v = mfma_elem(OFF_S2, w.sig_w2 + 64, 64, 1);  // rows 1..64 = geo_feat
    else if (e < OFF_C0) v = valu_elem(OFF_S2R, w.sig_w2);              // row 0 = sigma
    else if (e < OFF_C1) v = mfma_elem(OFF_C0, w.col_w0, ldC0, 2);
    else v = valu_elem(OFF_C1, w.col_w1);
    packed[e] = v;
}

// Per-frame bias vectors (the broadcast columns of the three first layers).
__global__ void __launch_bounds__(kBias) k_frame_bias(RawW w, const float *__restrict__ enc_a,
                                                      const float *__restrict__ eye,
                                                      const float *__restrict__ ind_code, float *__restrict__ bias) {
    const int t = threadIdx.x, row = t & 63;
    enc_a += (size_t)blockIdx.x * w.audio_dim;      // one workgroup per frame (rn_nerf_frame_bias_batch)
    bias += (size_t)blockIdx.x * kBias;
    float acc = 0.0f;
    if (t < 64) {
        const float *r = w.amb_w0 + row * (32 + w.audio_dim) + 32;
        for (uint32_t a = 0; a < w.audio_dim; a++) acc += r[a] * enc_a[a];
    } else if (t < 128) {
        if (w.has_eye) acc = w.sig_w0[row * 65 + 64] * eye[0];
    } else {
        const float *r = w.col_w0 + row * (80 + w.ind_dim) + 80;
        for (uint32_t c = 0; c < w.ind_dim; c++) acc += r[c] * ind_code[c];
    }
    bias[t] = acc;
}

// ---- MFMA helpers ---------------------------------------------------------------------------------------
__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
#ifdef RN_EXP_NO_MFMA  // experiment only: keep the data dependence, drop the matrix instruction
    c[0] += a * b;
    return c;
#else
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
#endif
}

// ---- 64-sample tiles (two column tiles per wave): the torso kernel
// one MFMA step of a 64-row layer: weights of step s from LDS, B operands b0 / b1 for the two column tiles
__device__ __forceinline__ void step64(Acc &a, const float *wl, int s, int lane_off, float b0, float b1) {
    const float2 w = *reinterpret_cast<const float2 *>(wl + s * kStep + lane_off);
    a.v[0][0] = mfma32(w.x, b0, a.v[0][0]);
    a.v[1][0] = mfma32(w.x, b1, a.v[1][0]);
    a.v[0][1] = mfma32(w.y, b0, a.v[0][1]);
    a.v[1][1] = mfma32(w.y, b1, a.v[1][1]);
}

// 64 -> 64 layer whose input is the previous layer's accumulators (32 steps)
__device__ __forceinline__ void layer_from_acc(Acc &out, const Acc &in, const float *wl, int lane_off) {
#pragma unroll
    for (int s = 0; s < 32; s++) step64(out, wl, s, lane_off, in.v[0][s >> 4][s & 15], in.v[1][s >> 4][s & 15]);
}

// "one sample per lane" feature pair (f0, f1) -> B operands of column tile 0 and 1
__device__ __forceinline__ void to_b_operands(float f0, float f1, float &b0, float &b1) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(f0), __float_as_uint(f1), false, false);
    b0 = __uint_as_float(r[0]);
    b1 = __uint_as_float(r[1]);
}

// Features (2 channels) of one sample at one planned level; zeros when `on` is false.
template <typename TT, uint32_t D>
__device__ __forceinline__ void level_features(const void *table, const LevelPlan &lp, const float (&in)[D], bool on,
                                               float &f0, float &f1) {
    f0 = 0.0f;
    f1 = 0.0f;
    if (on) {
        LevelFetch<TT, D, 2> f;
        issue_planned<TT, D, 2, false>(static_cast<const TT *>(table), lp, in, f);
        TT res[2];
        TT dummy[1];
        blend_level<TT, D, 2, false>(f, 0.0f, res, dummy);
        f0 = to_f<TT>(res[0]);
        f1 = to_f<TT>(res[1]);
    }
}

// ---- 32-sample tiles: the head kernel
// Accumulators of one 64-row layer for the 32 samples of a tile: [row tile], row on the register index, sample on the lane.
struct Acc32 {
    f32x16 v[2];
};

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

// one MFMA step of a 64-row layer: weights of step s from LDS (one float2 = both row tiles), B operand b
__device__ __forceinline__ void step32(Acc32 &a, const float *wl, int s, int lane_off, float b) {
    const float2 w = *reinterpret_cast<const float2 *>(wl + s * kStep + lane_off);
    a.v[0] = mfma32(w.x, b, a.v[0]);
    a.v[1] = mfma32(w.y, b, a.v[1]);
}

// 64 -> 64 layer whose input is the previous layer's accumulators (32 steps)
__device__ __forceinline__ void layer_from_acc(Acc32 &out, const Acc32 &in, const float *wl, int lane_off) {
#pragma unroll
    for (int s = 0; s < 32; s++) step32(out, wl, s, lane_off, in.v[s >> 4][s & 15]);
}

// out[o] = sum_k in[k] * W[o][k]: each lane half sums the k's it holds, one cross-half shuffle adds the other half's
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

// -DRN_PHASE_CLOCK (tools/gpu_phase_clock.sh only): lanes 0..1 of every tile overwrite their `ambient` outputs with the
// 100 MHz wall-clock ticks spent in the phases of the tile loop (xyz gather, ambient net, ambient gather, rest).
#ifdef RN_PHASE_CLOCK
#define RN_PHASE_MARK(i) const uint64_t phase_t##i = wall_clock64()
#define RN_PHASE_STORE()                                                                                        \
    if (p.ambient && lane < 2 && entry + 1 < M) {                                                               \
        p.ambient[2 * (size_t)sample] = (float)(lane ? phase_t3 - phase_t2 : phase_t1 - phase_t0);                \
        p.ambient[2 * (size_t)sample + 1] = (float)(lane ? phase_t4 - phase_t3 : phase_t2 - phase_t1);            \
    }
#else
#define RN_PHASE_MARK(i)
#define RN_PHASE_STORE()
#endif

template <typename TX, typename TW>
__global__ void __launch_bounds__(kF32Threads, kF32Waves / 4) k_nerf_fused(FusedParams p) {
    __shared__ __attribute__((aligned(16))) float lds[kLdsFloats];
    __shared__ LevelPlan plan_x[16], plan_w[16];

    uint32_t M = p.M;
    if (p.m_dev) { const uint32_t d = (uint32_t)*p.m_dev; M = d < M ? d : M; }
    const uint32_t n_tiles = (M + 31u) >> 5;
    {
        const TileSchedule w0(n_tiles, kF32Waves, 0u);
        if (w0.first >= w0.end) return;  // nothing for this workgroup (uniform)
    }

    for (int i = threadIdx.x; i < kPacked / 4; i += kF32Threads)
        reinterpret_cast<float4 *>(lds)[i] = reinterpret_cast<const float4 *>(p.packed)[i];
    if (threadIdx.x < kBias) lds[kPacked + threadIdx.x] = p.bias[threadIdx.x];
    if (threadIdx.x < 16) {
        const int t = threadIdx.x;
        const uint32_t ox = (uint32_t)p.gx.offsets[t], ow = (uint32_t)p.gw.offsets[t];
        plan_x[t] = plan_level<3>(p.gx.lc.scale[t], p.gx.lc.resolution[t], ox, (uint32_t)p.gx.offsets[t + 1] - ox,
                                  p.gx.gridtype, (uint32_t)sizeof(TX) * 2u);
        plan_w[t] = plan_level<2>(p.gw.lc.scale[t], p.gw.lc.resolution[t], ow, (uint32_t)p.gw.offsets[t + 1] - ow,
                                  p.gw.gridtype, (uint32_t)sizeof(TW) * 2u);
    }
    __syncthreads();

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 31, h = lane >> 5;
    const int lane_off = h * 64 + j * 2;
    const float *bias_amb = lds + kPacked, *bias_sig = lds + kPacked + 64, *bias_col = lds + kPacked + 128;

    const TileSchedule sched(n_tiles, kF32Waves, (uint32_t)wave);
    for (uint32_t tile = sched.first; tile < sched.end; tile += sched.stride) {
        const uint32_t entry = tile * 32 + j;  // both lane halves work on the same 32 samples
        bool live = entry < M;
        uint32_t sample = entry;  // the slot this lane's sample lives in
        if (p.slots) {
            if (live) sample = (uint32_t)p.slots[entry];
        } else if (live && p.deltas) {
            live = p.deltas[2 * (size_t)sample] != 0.0f;
        }
        if (__ballot(live) == 0ull) continue;  // whole tile dead (wave-uniform)

        // ---- xyz grid (gridencoder/grid.py:145-161: (x + bound) / (2 bound)).  Round r: lane half h gathers level
        // 2 r + h of its sample; the two features are the B operands of two MFMA steps of BOTH first layers that
        // consume enc_x (ambient L0 and sigma L0), so enc_x is never kept in registers.
        Acc32 a0, a1, a2;
        RN_PHASE_MARK(0);
        acc_bias(a0, bias_amb, h);  // ambient L0 accumulators, start = W0[:, 32:] enc_a
        acc_bias(a2, bias_sig, h);  // sigma   L0 accumulators, start = W0[:, 64] eye
        {
            float in[3] = {0.0f, 0.0f, 0.0f};
            bool on = live;
            if (live) {
#pragma unroll
                for (int d = 0; d < 3; d++) {
                    in[d] = (p.xyzs[3 * (size_t)sample + d] + p.bound) / (2 * p.bound);
                    on = on && !(in[d] < 0 || in[d] > 1);
                }
            }
            // kXyzGroup rounds in flight: independent chains (index arithmetic, loads, blends) that hide each other's latency
            LevelFetch<TX, 3, 2> f[kXyzGroup];
#pragma unroll 1
            for (int r = 0; r < 8; r += kXyzGroup) {
                if (on) {
#pragma unroll
                    for (int i = 0; i < kXyzGroup; i++)
                        issue_planned<TX, 3, 2, kPairHashed, false>(static_cast<const TX *>(p.gx.table), plan_x[2 * (r + i) + h], in, f[i]);
                }
#pragma unroll
                for (int i = 0; i < kXyzGroup; i++) {
                    float f0 = 0.0f, f1 = 0.0f;
                    if (on) {
                        TX res[2];
                        TX dummy[1];
                        blend_level<TX, 3, 2, false>(f[i], 0.0f, res, dummy);
                        f0 = to_f<TX>(res[0]);
                        f1 = to_f<TX>(res[1]);
                    }
                    step32(a0, lds + OFF_A0, 2 * (r + i), lane_off, f0);
                    step32(a2, lds + OFF_S0, 2 * (r + i), lane_off, f0);
                    step32(a0, lds + OFF_A0, 2 * (r + i) + 1, lane_off, f1);
                    step32(a2, lds + OFF_S0, 2 * (r + i) + 1, lane_off, f1);
                }
            }
        }

        // ---- ambient net: [enc_x | enc_a] 96 -> 64 -> 64 -> 2, tanh
        RN_PHASE_MARK(1);
        acc_relu(a0);
        acc_zero(a1);
        layer_from_acc(a1, a0, lds + OFF_A1, lane_off);
        acc_relu(a1);
        float amb[2];
        valu_out<2>(a1, lds + OFF_A2, h, amb);
        amb[0] = tanhf(amb[0]);
        amb[1] = tanhf(amb[1]);
        if (p.ambient && live && h == 0) {
            p.ambient[2 * (size_t)sample] = amb[0];
            p.ambient[2 * (size_t)sample + 1] = amb[1];
        }

        // ---- ambient grid: enc_w = encoder_ambient(ambient, bound=1) -> sigma L0 steps 16..31
        RN_PHASE_MARK(2);
        {
            float in[2] = {(amb[0] + 1.0f) / 2.0f, (amb[1] + 1.0f) / 2.0f};
            const bool on = live && !(in[0] < 0 || in[0] > 1 || in[1] < 0 || in[1] > 1);
            LevelFetch<TW, 2, 2> f[kAmbGroup];
#pragma unroll 1
            for (int r = 0; r < 8; r += kAmbGroup) {
                if (on) {
#pragma unroll
                    for (int i = 0; i < kAmbGroup; i++)
                        issue_planned<TW, 2, 2, kPairHashed, false>(static_cast<const TW *>(p.gw.table), plan_w[2 * (r + i) + h], in, f[i]);
                }
#pragma unroll
                for (int i = 0; i < kAmbGroup; i++) {
                    float f0 = 0.0f, f1 = 0.0f;
                    if (on) {
                        TW res[2];
                        TW dummy[1];
                        blend_level<TW, 2, 2, false>(f[i], 0.0f, res, dummy);
                        f0 = to_f<TW>(res[0]);
                        f1 = to_f<TW>(res[1]);
                    }
                    step32(a2, lds + OFF_S0, 16 + 2 * (r + i), lane_off, f0);
                    step32(a2, lds + OFF_S0, 16 + 2 * (r + i) + 1, lane_off, f1);
                }
            }
        }

        // ---- sigma net: [enc_x | enc_w | eye] 65 -> 64 -> 64 -> 1 + 64
        RN_PHASE_MARK(3);
        acc_relu(a2);
        acc_zero(a1);
        layer_from_acc(a1, a2, lds + OFF_S1, lane_off);
        acc_relu(a1);
        float sigma;
        {
            float raw[1];
            valu_out<1>(a1, lds + OFF_S2R, h, raw);
            sigma = expf(raw[0]);  // trunc_exp forward (activation.py:9-11)
        }
        if (!p.rgbs) {  // density query (NeRFNetwork.density, nerf/network.py:286-325): no geo_feat layer, no SH, no colour net
            if (live && h == 0) p.sigmas[sample] = sigma;
            continue;
        }
        acc_zero(a0);
        layer_from_acc(a0, a1, lds + OFF_S2, lane_off);  // geo_feat (no activation)

        // ---- color net: [SH(d) | geo_feat | ind_code] 84 -> 64 -> 3, sigmoid
        acc_bias(a1, bias_col, h);
        {
            float sh[16];
            float dx = 0.0f, dy = 0.0f, dz = 0.0f;
            if (live) {
                dx = p.dirs[3 * (size_t)sample]; dy = p.dirs[3 * (size_t)sample + 1]; dz = p.dirs[3 * (size_t)sample + 2];
            }
            sh_basis<4>(dx, dy, dz, sh);
#pragma unroll
            for (int s = 0; s < 8; s++) {
                // lane half h supplies k = 2 s + h; a bit-select, because ?: on two array elements makes the compiler
                // index sh[] dynamically and move it to LDS
                const uint32_t m = 0u - (uint32_t)h;
                const uint32_t b = (__float_as_uint(sh[2 * s]) & ~m) | (__float_as_uint(sh[2 * s + 1]) & m);
                step32(a1, lds + OFF_C0, s, lane_off, __uint_as_float(b));
            }
        }
#pragma unroll
        for (int s = 0; s < 32; s++) step32(a1, lds + OFF_C0, 8 + s, lane_off, a0.v[s >> 4][s & 15]);
        acc_relu(a1);
        {
            float rgb[3];
            valu_out<3>(a1, lds + OFF_C1, h, rgb);
            if (live && h == 0) {
                p.sigmas[sample] = sigma;
#pragma unroll
                for (int c = 0; c < 3; c++) p.rgbs[3 * (size_t)sample + c] = 1.0f / (1.0f + expf(-rgb[c]));
            }
        }
        RN_PHASE_MARK(4);
        RN_PHASE_STORE();
    }
}

// ==========================================================================================================
// Device-side inference loop (nerf/renderer.py:225-262)
//
// state words (int32), two banks of 8 selected by (iteration & 1):
//   [0] n_alive  [1] step  [2] n_step  [3] M = n_alive * n_step  [4] active  [5] live-partial workgroups (0: default)
//   [6] live samples listed by the marchers of this iteration (entries of rn_head_t.live_slots); zeroed by the previous
//       iteration's compositor (by rn_head_begin for iteration 0), never by next_state()
// plus stats at [16..]: iterations that did work, live samples, sample slots.
constexpr int kLoopBlock = 256;

__device__ __forceinline__ uint32_t policy_n_step(uint32_t N, uint32_t n_alive) {
    uint32_t n_step = n_alive ? N / n_alive : 1u;   // max(min(N // n_alive, 8), 1)  (renderer.py:249)
    n_step = n_step > 8u ? 8u : n_step;
    return n_step < 1u ? 1u : n_step;
}

__device__ __forceinline__ void next_state(int32_t *st, uint32_t N, uint32_t n_alive, uint32_t step, uint32_t max_steps) {
    const uint32_t n_step = policy_n_step(N, n_alive);
    const bool active = step < max_steps && n_alive > 0;
    st[0] = (int32_t)n_alive;
    st[1] = (int32_t)step;
    st[2] = (int32_t)n_step;
    st[3] = active ? (int32_t)(n_alive * n_step) : 0;  // sample slots of the coming iteration (0: loop is over)
    st[4] = active ? 1 : 0;
    st[5] = 0;  // workgroups that hold live-sample partial sums of the coming iteration; 0 = ceil(n_alive / 256)
}

// The marchers' epilogue: every lane holds `emitted` live samples in slots base .. base + emitted - 1.  One atomicAdd per
// workgroup reserves a run of the iteration's live list (its order is arrival order -- irrelevant, every sample is
// independent), a block-wide scan places each lane's entries.  Returns the workgroup's live-sample count.
__device__ __forceinline__ uint32_t list_live_slots(uint32_t emitted, uint32_t base, int32_t *live_count,
                                                    int32_t *__restrict__ live_slots, uint32_t *sh /* [kLoopBlock / kWave + 1] */) {
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t incl = emitted;
#pragma unroll
    for (uint32_t d = 1; d < 64; d <<= 1) {
        const uint32_t t = __shfl_up(incl, d, 64);
        if (lane >= d) incl += t;
    }
    __syncthreads();  // sh may still be read by the caller's previous phase
    if (lane == 63) sh[wave] = incl;
    __syncthreads();
    uint32_t total = 0, before = 0;
#pragma unroll
    for (uint32_t w = 0; w < kLoopBlock / kWave; w++) {
        total += sh[w];
        before += w < wave ? sh[w] : 0u;
    }
    if (live_slots) {
        if (threadIdx.x == 0) sh[kLoopBlock / kWave] = total ? (uint32_t)atomicAdd(live_count, (int32_t)total) : 0u;
        __syncthreads();
        const uint32_t at = sh[kLoopBlock / kWave] + before + incl - emitted;
        for (uint32_t k = 0; k < emitted; k++) live_slots[at + k] = (int32_t)(base + k);
    }
    return total;
}

// near/far (raymarching.cu:91-145) + loop initialisation (renderer.py:229-237)
__global__ void __launch_bounds__(kLoopBlock)
k_head_begin(const float *__restrict__ rays_o, const float *__restrict__ rays_d, const float *__restrict__ aabb,
             uint32_t N, float min_near, uint32_t max_steps, float *__restrict__ nears, float *__restrict__ fars,
             float *__restrict__ weights_sum, float *__restrict__ depth, float *__restrict__ image,
             int32_t *__restrict__ rays_alive, float *__restrict__ rays_t, int32_t *__restrict__ state,
             uint32_t order_w) {
    const uint32_t n = blockIdx.x * kLoopBlock + threadIdx.x;
    if (n == 0) {
        next_state(state, N, N, 0, max_steps);
        state[6] = 0;
        for (int i = 8; i < 16; i++) state[i] = 0;  // statistics words [16..] accumulate across frames (caller-owned)
        state[RN_HEAD_ST_HIST] = (int32_t)N;
    }
    if (n >= N) return;
    const float ox = rays_o[n * 3], oy = rays_o[n * 3 + 1], oz = rays_o[n * 3 + 2];
    const float dx = rays_d[n * 3], dy = rays_d[n * 3 + 1], dz = rays_d[n * 3 + 2];
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
    near = miss ? FLT_MAX : near;
    far = miss ? FLT_MAX : far;
    nears[n] = near; fars[n] = far;
    rays_t[n] = near;
    // Slot n of the alive list: ray n, or (order_w = image width) the rays of 8 x 8 pixel blocks together, so that the 64
    // samples of a wave and the tiles of a CU cover a compact patch of the image instead of a one-pixel-high strip --
    // more of their grid rows coincide.  Rays are independent, so the order changes no pixel.
    uint32_t ray = n;
    if (order_w) {
        const uint32_t t = n >> 6, within = n & 63u, tiles_x = order_w >> 3;
        ray = ((t / tiles_x) * 8u + (within >> 3)) * order_w + (t % tiles_x) * 8u + (within & 7u);
    }
    rays_alive[n] = (int32_t)ray;
    weights_sum[n] = 0.0f; depth[n] = 0.0f;
    image[n * 3] = 0.0f; image[n * 3 + 1] = 0.0f; image[n * 3 + 2] = 0.0f;
}

// raymarching.cu:827-929 with device-resident n_alive / n_step; every slot of a live ray is written
// (unused slots get deltas = 0), so the sample buffers never need a memset.
__global__ void __launch_bounds__(kLoopBlock)
k_head_march(const int32_t *__restrict__ st, const int32_t *__restrict__ rays_alive, const float *__restrict__ rays_t,
             const float *__restrict__ rays_o, const float *__restrict__ rays_d, float bound, float dt_gamma,
             uint32_t max_steps, uint32_t C, uint32_t H, const uint8_t *__restrict__ grid,
             const float *__restrict__ fars, float *__restrict__ xyzs, float *__restrict__ dirs, float *__restrict__ deltas, int32_t *__restrict__ stats,
             uint32_t *__restrict__ block_live, int32_t *__restrict__ live_count, int32_t *__restrict__ live_slots) {
    if (!st[4]) return;
    const uint32_t n_alive = (uint32_t)st[0], n_step = (uint32_t)st[2];
    const uint32_t n = blockIdx.x * kLoopBlock + threadIdx.x;
    uint32_t emitted = 0;
    const uint32_t base = n * n_step;
    if (n < n_alive) {
        const int index = rays_alive[n];
        Dda s;
        s.init(rays_o + (size_t)index * 3, rays_d + (size_t)index * 3, bound, dt_gamma, max_steps, C, H, grid, fars[index]);
        float t = rays_t[index];  // perturb is off at inference: no noise term (renderer.py:251)
        emitted = s.walk<true>(t, n_step, xyzs + (size_t)base * 3, dirs + (size_t)base * 3, deltas + (size_t)base * 2);
        for (uint32_t k = emitted; k < n_step; k++) { deltas[((size_t)base + k) * 2] = 0.0f; deltas[((size_t)base + k) * 2 + 1] = 0.0f; }
    }
    // live samples of this iteration: listed for the network kernel, and counted -- one partial sum per workgroup, added up by
    // the compaction kernel (a same-address atomic per wavefront for the statistic cost 38 us per frame)
    __shared__ uint32_t sh[kLoopBlock / kWave + 1];
    const uint32_t total = list_live_slots(emitted, base, live_count, live_slots, sh);
    if (threadIdx.x == 0) block_live[blockIdx.x] = total;
    if (n == 0) { atomicAdd(&stats[RN_HEAD_ST_ITERS], 1); atomicAdd(&stats[RN_HEAD_ST_SLOTS], (int32_t)(n_alive * n_step)); }
}

// Frame prologue in ONE launch: [ray generation (nerf/utils.py:249-333)] + near/far + loop initialisation + the march of
// iteration 0.  Slot n of the alive list is handled by lane n from start to end: it builds (or loads) the ray that the list
// order puts there, intersects it with the box, resets its accumulators and walks it for the first iteration's
// n_step = max(min(N // N, 8), 1) = 1 sample.  Nothing here depends on another lane's ray, so what used to be three
// launches (k_get_rays, k_head_begin, k_head_march) and two [N,3] round trips is one pass.
// state[6] (live-sample count of even iterations) must be zero on entry: the loop's last compaction leaves it zero.
struct RaySource {
    const float *pose;    // [3,4] / [4,4] row-major cam2world, or NULL: rays are given
    float fx, fy, cx, cy;
    uint32_t W;
};

__global__ void __launch_bounds__(kLoopBlock)
k_frame_begin(RaySource rs, float *__restrict__ rays_o, float *__restrict__ rays_d, const float *__restrict__ aabb, uint32_t N,
              float min_near, uint32_t max_steps, float bound, float dt_gamma, uint32_t C, uint32_t H,
              const uint8_t *__restrict__ grid, float *__restrict__ nears, float *__restrict__ fars, float *__restrict__ weights_sum,
              float *__restrict__ depth, float *__restrict__ image, int32_t *__restrict__ rays_alive, float *__restrict__ rays_t,
              int32_t *__restrict__ state, uint32_t order_w, float *__restrict__ xyzs, float *__restrict__ dirs,
              float *__restrict__ deltas, uint32_t *__restrict__ block_live, int32_t *__restrict__ live_slots) {
    const uint32_t n = blockIdx.x * kLoopBlock + threadIdx.x;
    if (n == 0) {
        next_state(state, N, N, 0, max_steps);
        for (int i = 8; i < 16; i++) state[i] = 0;
        state[RN_HEAD_ST_HIST] = (int32_t)N;
        atomicAdd(&state[RN_HEAD_ST_ITERS], 1);
        atomicAdd(&state[RN_HEAD_ST_SLOTS], (int32_t)N);          // n_alive * n_step = N * 1
    }
    uint32_t emitted = 0;
    if (n < N) {
        uint32_t ray = n;                                         // alive-list order (see k_head_begin)
        if (order_w) {
            const uint32_t t = n >> 6, within = n & 63u, tiles_x = order_w >> 3;
            ray = ((t / tiles_x) * 8u + (within >> 3)) * order_w + (t % tiles_x) * 8u + (within & 7u);
        }
        float o[3], d[3];
        if (rs.pose) {                                            // same expressions as k_get_rays
            const uint32_t r = ray / rs.W, c = ray - r * rs.W;
            const float x = ((float)c + 0.5f - rs.cx) / rs.fx, y = ((float)r + 0.5f - rs.cy) / rs.fy, z = 1.0f;
            const float norm = sqrtf(x * x + y * y + z * z);
            const float ux = x / norm, uy = y / norm, uz = z / norm;
#pragma unroll
            for (int k = 0; k < 3; k++) {
                d[k] = ux * rs.pose[k * 4] + uy * rs.pose[k * 4 + 1] + uz * rs.pose[k * 4 + 2];
                o[k] = rs.pose[k * 4 + 3];
                rays_d[(size_t)ray * 3 + k] = d[k];
                rays_o[(size_t)ray * 3 + k] = o[k];
            }
        } else {
#pragma unroll
            for (int k = 0; k < 3; k++) { o[k] = rays_o[(size_t)ray * 3 + k]; d[k] = rays_d[(size_t)ray * 3 + k]; }
        }
        // raymarching.cu:91-145
        const float rdx = 1 / d[0], rdy = 1 / d[1], rdz = 1 / d[2];
        float near = (aabb[0] - o[0]) * rdx, far = (aabb[3] - o[0]) * rdx;
        if (near > far) { float c = near; near = far; far = c; }
        float near_y = (aabb[1] - o[1]) * rdy, far_y = (aabb[4] - o[1]) * rdy;
        if (near_y > far_y) { float c = near_y; near_y = far_y; far_y = c; }
        bool miss = (near > far_y || near_y > far);
        if (!miss) {
            if (near_y > near) near = near_y;
            if (far_y < far) far = far_y;
            float near_z = (aabb[2] - o[2]) * rdz, far_z = (aabb[5] - o[2]) * rdz;
            if (near_z > far_z) { float c = near_z; near_z = far_z; far_z = c; }
            miss = (near > far_z || near_z > far);
            if (!miss) {
                if (near_z > near) near = near_z;
                if (far_z < far) far = far_z;
                if (near < min_near) near = min_near;
            }
        }
        near = miss ? FLT_MAX : near;
        far = miss ? FLT_MAX : far;
        nears[ray] = near; fars[ray] = far;
        rays_t[ray] = near;
        rays_alive[n] = (int32_t)ray;
        weights_sum[ray] = 0.0f; depth[ray] = 0.0f;
        image[(size_t)ray * 3] = 0.0f; image[(size_t)ray * 3 + 1] = 0.0f; image[(size_t)ray * 3 + 2] = 0.0f;
        // iteration 0 (k_head_march with n_alive = N, n_step = 1): slot n
        Dda s;
        s.init(o, d, bound, dt_gamma, max_steps, C, H, grid, far);
        float t = near;
        emitted = s.walk<true>(t, 1u, xyzs + (size_t)n * 3, dirs + (size_t)n * 3, deltas + (size_t)n * 2);
        if (!emitted) { deltas[(size_t)n * 2] = 0.0f; deltas[(size_t)n * 2 + 1] = 0.0f; }
    }
    __shared__ uint32_t sh[kLoopBlock / kWave + 1];
    const uint32_t total = list_live_slots(emitted, n, state + 6, live_slots, sh);
    if (threadIdx.x == 0) block_live[blockIdx.x] = total;
}

// raymarching.cu:942-1029 + per-block survivor counts for the compaction that follows.  One chunk = kLoopBlock consecutive
// entries of the alive list; COOP: the counts cross workgroups INSIDE a launch (k_head_step), so they are written with
// agent-scope atomic stores (the per-XCD L2s are not coherent with each other for plain stores).
constexpr uint32_t kTagShift = 10;               // a chunk has <= kLoopBlock = 256 survivors
constexpr uint32_t kTagMask = (1u << 22) - 1u;
constexpr uint32_t kBarrierPolls = 1u << 20;     // ~1 s of polling before a workgroup gives up waiting for a chunk's count

template <bool COOP>
__device__ __forceinline__ void composite_chunk(uint32_t c, uint32_t n_alive, uint32_t n_step, float T_thresh,
                                                int32_t *__restrict__ rays_alive, float *__restrict__ rays_t,
                                                const float *__restrict__ sigmas, const float *__restrict__ rgbs,
                                                const float *__restrict__ deltas, float *__restrict__ weights_sum,
                                                float *__restrict__ depth, float *__restrict__ image,
                                                uint32_t *__restrict__ block_counts, uint32_t *wave_cnt /* LDS [kLoopBlock / kWave] */,
                                                uint32_t tag = 0) {
    const uint32_t n = c * kLoopBlock + threadIdx.x;
    bool survive = false;
    if (n < n_alive) {
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
        survive = !(step < n_step);
        if (survive) rays_t[index] = t;
        else rays_alive[n] = -1;
        weights_sum[index] = weight_sum;
        depth[index] = d;
        image[index * 3] = r; image[index * 3 + 1] = g; image[index * 3 + 2] = b;
    }
    const unsigned long long mask = __ballot(survive);
    if ((threadIdx.x & 63) == 0) wave_cnt[threadIdx.x >> 6] = (uint32_t)__popcll(mask);
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t s = 0;
        for (int w = 0; w < kLoopBlock / kWave; w++) s += wave_cnt[w];
        if constexpr (COOP) {
            // count + launch tag in one word: the word IS the chunk's arrival flag (k_head_step).  The first store of workgroup
            // 0 is a release: its reset of the next live-sample counter must be visible before anybody passes the barrier.
            if (c == 0) __hip_atomic_store(&block_counts[c], (tag << kTagShift) | s, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            else __hip_atomic_store(&block_counts[c], (tag << kTagShift) | s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            block_counts[c] = s;
        }
    }
}

__global__ void __launch_bounds__(kLoopBlock)
k_head_composite(const int32_t *__restrict__ st, float T_thresh, int32_t *__restrict__ rays_alive,
                 float *__restrict__ rays_t, const float *__restrict__ sigmas, const float *__restrict__ rgbs,
                 const float *__restrict__ deltas, float *__restrict__ weights_sum, float *__restrict__ depth,
                 float *__restrict__ image, uint32_t *__restrict__ block_counts, int32_t *__restrict__ st_next) {
    __shared__ uint32_t wave_cnt[kLoopBlock / kWave];
    if (!st[4]) return;
    if (blockIdx.x == 0 && threadIdx.x == 0) st_next[6] = 0;  // the marchers of the next iteration count their live samples here
    const uint32_t n_alive = (uint32_t)st[0], n_step = (uint32_t)st[2];
    if (blockIdx.x * kLoopBlock >= n_alive) return;
    composite_chunk<false>(blockIdx.x, n_alive, n_step, T_thresh, rays_alive, rays_t, sigmas, rgbs, deltas, weights_sum, depth, image,
                           block_counts, wave_cnt);
}

// stable compaction (renderer.py:258) + loop control (renderer.py:242-249, 262) for the next iteration and, with MARCH,
// the next iteration's march as well: every workgroup adds up all survivor counts (<= 1024 words), so it knows the new
// n_alive and n_step, and a surviving ray is marched by the lane that has just computed its slot in the new list.  One
// launch (and one pass over the ray list) less per iteration; the lanes of dead rays idle, which costs nothing here:
// these launches are bound by the length of one ray's walk, not by lane throughput.
struct MarchArgs {
    const float *rays_t, *rays_o, *rays_d, *fars;
    float bound, dt_gamma;
    uint32_t cascade, grid_size;
    const uint8_t *grid;
    float *xyzs, *dirs, *deltas;
    uint32_t *block_live_next;
    int32_t *live_slots;
};

// What a launch does when the loop is already over (st[4] == 0): carry the state over, close the frame's counters.
__device__ __forceinline__ void loop_idle(const int32_t *__restrict__ st, int32_t *__restrict__ st_next, int32_t *__restrict__ stats,
                                          uint32_t close_frame, uint32_t iter) {
    if (blockIdx.x == 0 && threadIdx.x < 8) st_next[threadIdx.x] = st[threadIdx.x];
    // close_frame (last compaction of a frame's loop): both live-sample counters back to zero for the next frame's
    // prologue; the loop is over, so rn_head_check_done has nothing to flag
    if (close_frame && blockIdx.x == 0 && threadIdx.x == 0) { stats[6] = 0; stats[8 + 6] = 0; }
    if (blockIdx.x == 0 && threadIdx.x == 0 && iter + 1 < 32) stats[RN_HEAD_ST_HIST + iter + 1] = 0;
}

// Chunk c of the n_blocks chunks of the alive list (see k_head_compact).  COOP: the survivor counts were written by other
// workgroups of THIS launch -> agent-scope atomic loads.
template <bool MARCH, bool COOP>
__device__ __forceinline__ void compact_chunk(uint32_t c, uint32_t n_blocks, const int32_t *__restrict__ st, int32_t *__restrict__ st_next,
                                              uint32_t N, uint32_t max_steps, const int32_t *__restrict__ rays_in,
                                              int32_t *__restrict__ rays_out, const uint32_t *__restrict__ block_counts,
                                              const uint32_t *__restrict__ block_live, int32_t *__restrict__ stats, const MarchArgs &m,
                                              uint32_t close_frame, uint32_t iter, uint32_t tag = 0) {
    __shared__ uint32_t red[kLoopBlock / kWave];
    __shared__ uint32_t red_live[kLoopBlock / kWave];
    __shared__ uint32_t red_all[kLoopBlock / kWave];
    __shared__ uint32_t wave_off[kLoopBlock / kWave];
    const uint32_t n_alive = (uint32_t)st[0];
    const bool last = c == n_blocks - 1;

    uint32_t part = 0, live = 0, all = 0;
    for (uint32_t b = threadIdx.x; b < n_blocks; b += kLoopBlock) {
        uint32_t cnt;
        if constexpr (COOP) {   // the grid-wide barrier: wait until chunk b's count of THIS launch has arrived
            uint32_t polls = 0;
            cnt = __hip_atomic_load(&block_counts[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            while ((cnt >> kTagShift) != tag) {
                __builtin_amdgcn_s_sleep(1);
                if (++polls > kBarrierPolls) {   // never expected with a cooperative launch; the frame is then not to be used:
                    atomicAdd(&stats[RN_HEAD_ST_STALLED], 1);      // counted, and flagged like a frame whose loop was cut short, so the
                    atomicAdd(&stats[RN_HEAD_ST_UNFINISHED], 1);   // host renders it again instead of consuming pixels built on stale counts
                    break;
                }
                cnt = __hip_atomic_load(&block_counts[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            cnt &= (1u << kTagShift) - 1u;
        } else {
            cnt = block_counts[b];
        }
        all += cnt;
        part += b < c ? cnt : 0u;
    }
    if (last) {  // the last workgroup also adds up the live-sample partial sums of this iteration's march
        const uint32_t n_live = st[5] ? (uint32_t)st[5] : n_blocks;
        for (uint32_t b = threadIdx.x; b < n_live; b += kLoopBlock) live += block_live[b];
    }
    for (int off = 32; off > 0; off >>= 1) {
        part += __shfl_down(part, off, 64); live += __shfl_down(live, off, 64); all += __shfl_down(all, off, 64);
    }
    if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6] = part; red_live[threadIdx.x >> 6] = live; red_all[threadIdx.x >> 6] = all; }
    __syncthreads();
    uint32_t offset = 0, n_next = 0;
    for (int w = 0; w < kLoopBlock / kWave; w++) { offset += red[w]; n_next += red_all[w]; }

    const uint32_t n = c * kLoopBlock + threadIdx.x;
    const int32_t v = (n < n_alive) ? rays_in[n] : -1;
    const bool keep = v >= 0;
    const unsigned long long mask = __ballot(keep);
    const uint32_t within = ballot_prefix(mask);
    if ((threadIdx.x & 63) == 0) wave_off[threadIdx.x >> 6] = (uint32_t)__popcll(mask);
    __syncthreads();
    uint32_t before = 0;
    for (uint32_t w = 0; w < (threadIdx.x >> 6); w++) before += wave_off[w];
    const uint32_t slot = offset + before + within;
    if (keep) rays_out[slot] = v;

    const uint32_t step_next = (uint32_t)st[1] + (uint32_t)st[2];
    const uint32_t n_step_next = policy_n_step(N, n_next);
    const bool active_next = step_next < max_steps && n_next > 0;
    if (last && threadIdx.x == 0) {
        next_state(st_next, N, n_next, step_next, max_steps);
        if (iter + 1 < 32) stats[RN_HEAD_ST_HIST + iter + 1] = active_next ? (int32_t)n_next : 0;   // live rays entering iteration iter + 1
        uint32_t sum = 0;
        for (int w = 0; w < kLoopBlock / kWave; w++) sum += red_live[w];
        if (sum) atomicAdd(&stats[RN_HEAD_ST_LIVE], (int32_t)sum);
        if (close_frame) {   // what rn_head_check_done does, folded in: was the loop really over after this iteration?
            if (active_next) atomicAdd(&stats[RN_HEAD_ST_UNFINISHED], 1);
            stats[6] = 0; stats[8 + 6] = 0;   // no marcher runs after the frame's last compaction
        }
        if (MARCH && active_next) {
            st_next[5] = (int32_t)n_blocks;  // the partial sums written below are indexed by THIS launch's chunks
            atomicAdd(&stats[RN_HEAD_ST_ITERS], 1);
            atomicAdd(&stats[RN_HEAD_ST_SLOTS], (int32_t)(n_next * n_step_next));
        }
    }
    if constexpr (MARCH) {
        if (!active_next) return;  // uniform
        uint32_t emitted = 0;
        const uint32_t base = slot * n_step_next;
        if (keep) {
            Dda s;
            s.init(m.rays_o + (size_t)v * 3, m.rays_d + (size_t)v * 3, m.bound, m.dt_gamma, max_steps, m.cascade, m.grid_size, m.grid,
                   m.fars[v]);
            float t = m.rays_t[v];
            emitted = s.walk<true>(t, n_step_next, m.xyzs + (size_t)base * 3, m.dirs + (size_t)base * 3, m.deltas + (size_t)base * 2);
            for (uint32_t k = emitted; k < n_step_next; k++) { m.deltas[((size_t)base + k) * 2] = 0.0f; m.deltas[((size_t)base + k) * 2 + 1] = 0.0f; }
        }
        __shared__ uint32_t sh[kLoopBlock / kWave + 1];
        const uint32_t total = list_live_slots(emitted, base, st_next + 6, m.live_slots, sh);
        if (threadIdx.x == 0) m.block_live_next[c] = total;
    }
}

template <bool MARCH>
__global__ void __launch_bounds__(kLoopBlock)
k_head_compact(const int32_t *__restrict__ st, int32_t *__restrict__ st_next, uint32_t N, uint32_t max_steps,
               const int32_t *__restrict__ rays_in, int32_t *__restrict__ rays_out,
               const uint32_t *__restrict__ block_counts, const uint32_t *__restrict__ block_live, int32_t *__restrict__ stats,
               MarchArgs m, uint32_t close_frame, uint32_t iter) {
    if (!st[4]) { loop_idle(st, st_next, stats, close_frame, iter); return; }
    const uint32_t n_blocks = ((uint32_t)st[0] + kLoopBlock - 1) / kLoopBlock;
    if (blockIdx.x >= n_blocks) return;
    compact_chunk<MARCH, false>(blockIdx.x, n_blocks, st, st_next, N, max_steps, rays_in, rays_out, block_counts, block_live, stats, m,
                                close_frame, iter);
}

// Compositor + compaction (+ next march) of one loop iteration in ONE launch: what k_head_composite and k_head_compact do,
// with a grid-wide barrier where the kernel boundary was (the compaction needs every chunk's survivor count: the new live
// count decides the next n_step).  At most kStepGrid workgroups take part, each walking chunks b, b + G, ...; with <= 80
// VGPRs and 256 threads six of them fit on a CU, so kStepGrid workgroups are co-resident three times over on this chip --
// launches of up to three streams may overlap (the host falls back to the two-kernel form beyond that).
// The barrier has no counter (512 same-address atomics cost ~25 us here: they execute one after the other at the memory
// side): a chunk's survivor count is stored together with a per-launch tag (state[RN_HEAD_ST_BARRIER], bumped by every
// launch, never reset), and the summation every workgroup does anyway waits for each word to carry this launch's tag.
// Exit condition every wave reaches: the wait is bounded; running into the bound counts in state[RN_HEAD_ST_STALLED]
// (the host treats such a frame as not rendered) and the workgroup carries on.
constexpr uint32_t kStepGrid = 512;

template <bool MARCH>
__global__ void __launch_bounds__(kLoopBlock, 6)
k_head_step(const int32_t *__restrict__ st, int32_t *__restrict__ st_next, uint32_t N, uint32_t max_steps, float T_thresh,
            int32_t *rays_in, int32_t *__restrict__ rays_out, float *rays_t /* = m.rays_t */,
            const float *__restrict__ sigmas, const float *__restrict__ rgbs, const float *deltas /* = m.deltas */,
            float *__restrict__ weights_sum, float *__restrict__ depth, float *__restrict__ image,
            uint32_t *block_counts, const uint32_t *__restrict__ block_live, int32_t *__restrict__ stats,
            MarchArgs m, uint32_t close_frame, uint32_t iter) {
    __shared__ uint32_t wave_cnt[kLoopBlock / kWave];
    if (!st[4]) { loop_idle(st, st_next, stats, close_frame, iter); return; }
    const uint32_t n_alive = (uint32_t)st[0], n_step = (uint32_t)st[2];
    const uint32_t n_chunks = (n_alive + kLoopBlock - 1) / kLoopBlock;
    const uint32_t G = n_chunks < gridDim.x ? n_chunks : gridDim.x;
    if (blockIdx.x >= G) return;
    const uint32_t epoch = (uint32_t)stats[RN_HEAD_ST_BARRIER];   // written by the previous launch of this state's stream
    const uint32_t tag = epoch % kTagMask + 1u;                   // 1 .. 2^22 - 1; never 0: a zeroed scratch block carries no valid tag
    if (blockIdx.x == 0 && threadIdx.x == 0)
        __hip_atomic_store(&st_next[6], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // next iteration's live-sample counter
    for (uint32_t c = blockIdx.x; c < n_chunks; c += G) {
        if (c != blockIdx.x) __syncthreads();  // wave_cnt is read by thread 0 of the previous round
        composite_chunk<true>(c, n_alive, n_step, T_thresh, rays_in, rays_t, sigmas, rgbs, deltas, weights_sum, depth, image, block_counts,
                              wave_cnt, tag);
    }
    for (uint32_t c = blockIdx.x; c < n_chunks; c += G) {
        __syncthreads();  // the reduction arrays of compact_chunk are reused; all of this workgroup's counts are on their way
        compact_chunk<MARCH, true>(c, n_chunks, st, st_next, N, max_steps, rays_in, rays_out, block_counts, block_live, stats, m,
                                   close_frame, iter, tag);
    }
    // every workgroup has read the epoch before it stored its first count, and nobody gets here before all counts are in
    if (blockIdx.x == 0 && threadIdx.x == 0) stats[RN_HEAD_ST_BARRIER] = (int32_t)(epoch + 1u);
}

// Was the loop over after the iterations the caller enqueued?  (`st` = the state bank the NEXT iteration would read.)
__global__ void k_head_check_done(const int32_t *__restrict__ st, int32_t *__restrict__ unfinished) {
    if (threadIdx.x == 0 && st[4]) atomicAdd(unfinished, 1);
}

// Whole-frame step schedule for a shard of the frame (see radnerf_fused.h): same policy as next_state, fed with the
// frame-wide ray and live counts.
__global__ void k_head_reschedule(int32_t *__restrict__ st, uint32_t schedule_N, const int32_t *__restrict__ alive_total) {
    if (threadIdx.x != 0 || !st[4]) return;
    const uint32_t total = (uint32_t)alive_total[0];
    uint32_t n_step = total ? schedule_N / total : 1u;
    n_step = n_step > 8u ? 8u : n_step;
    n_step = n_step < 1u ? 1u : n_step;
    st[2] = (int32_t)n_step;
    st[3] = (int32_t)((uint32_t)st[0] * n_step);
}

// ==========================================================================================================
// Torso pass (nerf/renderer.py:269-299, nerf/network.py:188-219) and final blend (renderer.py:306-311)
//
// Packed torso image: deform L0 (21 steps, freq(x) part) | deform L1 (32 steps) | deform L2 VALU [2][2][32]
//                     | torso L0 (37 steps x 64: grid 16 + freq 21, 32 rows) | torso L1 (16 steps x 64)
//                     | torso L2 VALU [4][2][16] | raw broadcast columns for the bias: def [64][54+ind], tor [32][54+ind]
#ifndef RN_TORSO_GROUP
#define RN_TORSO_GROUP 2
#endif
constexpr int kTorsoGroup = RN_TORSO_GROUP;  // torso-grid levels gathered together (divides 16)
constexpr int kTStep32 = 64;  // floats per MFMA step of a 32-row layer ([2 h][32 j])
constexpr int TOFF_D0 = 0;
constexpr int TOFF_D1 = TOFF_D0 + 21 * kStep;
constexpr int TOFF_D2 = TOFF_D1 + 32 * kStep;
constexpr int TOFF_T0 = TOFF_D2 + 128;
constexpr int TOFF_T1 = TOFF_T0 + 37 * kTStep32;
constexpr int TOFF_T2 = TOFF_T1 + 16 * kTStep32;
constexpr int kTorsoPacked = TOFF_T2 + 128;  // 10320 floats
constexpr int kTorsoBias = 96;               // deform 64 | torso 32

struct RawT {
    const float *def_w0, *def_w1, *def_w2, *tor_w0, *tor_w1, *tor_w2;
    uint32_t ind_dim;
};

// k index of a 32-wide hidden vector held in one accumulator row tile
__host__ __device__ constexpr int kmap32(int s, int h) { return rowmap(s & 15, h); }

__global__ void __launch_bounds__(256) k_pack_torso(RawT w, float *__restrict__ packed) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= kTorsoPacked) return;
    const int ldD0 = 96 + (int)w.ind_dim, ldT0 = 128 + (int)w.ind_dim;
    float v;
    if (e < TOFF_D1) {  // deform L0: k = 2s + h over freq(x) (42)
        const int q = e - TOFF_D0, s = q / kStep, rem = q % kStep, h = rem / 64, j = (rem % 64) / 2, rt = rem % 2;
        v = w.def_w0[(32 * rt + j) * ldD0 + 2 * s + h];
    } else if (e < TOFF_D2) {
        const int q = e - TOFF_D1, s = q / kStep, rem = q % kStep, h = rem / 64, j = (rem % 64) / 2, rt = rem % 2;
        v = w.def_w1[(32 * rt + j) * 64 + kmap(s, h)];
    } else if (e < TOFF_T0) {
        const int q0 = e - TOFF_D2, o = q0 / 64, h = (q0 % 64) / 32, q = q0 % 32;
        v = w.def_w2[o * 64 + 32 * (q >> 4) + rowmap(q & 15, h)];
    } else if (e < TOFF_T1) {  // torso L0: steps 0..15 grid (cols 0..31), 16..36 freq(x) (cols 32..73)
        const int q = e - TOFF_T0, s = q / kTStep32, rem = q % kTStep32, h = rem / 32, j = rem % 32;
        v = w.tor_w0[j * ldT0 + 2 * s + h];
    } else if (e < TOFF_T2) {
        const int q = e - TOFF_T1, s = q / kTStep32, rem = q % kTStep32, h = rem / 32, j = rem % 32;
        v = w.tor_w1[j * 32 + kmap32(s, h)];
    } else {
        const int q0 = e - TOFF_T2, o = q0 / 32, h = (q0 % 32) / 16, r = q0 % 16;
        v = w.tor_w2[o * 32 + rowmap(r, h)];
    }
    packed[e] = v;
}

struct BlendArgs {   // final blend folded into the torso pass (renderer.py:306-311; rn_torso_blend_frame)
    float *image;
    const float *weights_sum;
    float *depth;
    const float *nears, *fars;
    uint8_t *u8;
};

struct TorsoParams {
    const float *bg_coords;
    uint32_t N;
    const float *density_grid;
    uint32_t G;
    float thresh;
    const float *poses6, *ind_code;
    float shrink;
    RawT w;
    const float *packed;
    GridArgs gt;
    const float *bg_in;
    float *bg_out, *alpha_out, *deform_out;
    BlendArgs blend;
};

// image = clamp(image + (1 - weights_sum) * bg, 0, 1); depth = max(depth - near, 0) / (far - near)  [, uint8 frame]
__device__ __forceinline__ void blend_pixel(const BlendArgs &b, size_t px, const float (&bg)[3]) {
    const float w = 1 - b.weights_sum[px];
#pragma unroll
    for (int c = 0; c < 3; c++) {
        float v = b.image[3 * px + c] + w * bg[c];
        v = fminf(fmaxf(v, 0.0f), 1.0f);
        b.image[3 * px + c] = v;
        if (b.u8) b.u8[3 * px + c] = (uint8_t)(v * 255.0f);
    }
    const float dd = b.depth[px] - b.nears[px];
    b.depth[px] = fmaxf(dd, 0.0f) / (b.fars[px] - b.nears[px]);
}

// F.grid_sample(bilinear, zeros, align_corners=True) of the [G,G] torso grid at (gx, gy) (renderer.py:282)
__device__ __forceinline__ float sample_torso_grid(const float *__restrict__ img, uint32_t G, float gx, float gy) {
    const float ix = ((gx + 1.f) / 2) * (float)(G - 1);
    const float iy = ((gy + 1.f) / 2) * (float)(G - 1);
    const float ix_nw = floorf(ix), iy_nw = floorf(iy);
    const float ix_se = ix_nw + 1, iy_se = iy_nw + 1;
    const float nw = (ix_se - ix) * (iy_se - iy), ne = (ix - ix_nw) * (iy_se - iy);
    const float sw = (ix_se - ix) * (iy - iy_nw), se = (ix - ix_nw) * (iy - iy_nw);
    const int x0 = (int)ix_nw, y0 = (int)iy_nw, x1 = x0 + 1, y1 = y0 + 1;
    const int g = (int)G;
    float out = 0.0f;
    if (x0 >= 0 && x0 < g && y0 >= 0 && y0 < g) out += img[y0 * g + x0] * nw;
    if (x1 >= 0 && x1 < g && y0 >= 0 && y0 < g) out += img[y0 * g + x1] * ne;
    if (x0 >= 0 && x0 < g && y1 >= 0 && y1 < g) out += img[y1 * g + x0] * sw;
    if (x1 >= 0 && x1 < g && y1 >= 0 && y1 < g) out += img[y1 * g + x1] * se;
    return out;
}

template <typename TT, bool BLEND>
__global__ void __launch_bounds__(kFusedThreads, 2) k_torso_fused(TorsoParams p) {
    __shared__ __attribute__((aligned(16))) float lds[kTorsoPacked + kTorsoBias + 64];
    __shared__ LevelPlan plan_t[16];
    float *bias_def = lds + kTorsoPacked, *bias_tor = bias_def + 64, *enc_pose = bias_tor + 32;
    if (threadIdx.x >= 64 && threadIdx.x < 80) {
        const int t = threadIdx.x - 64;
        const uint32_t o = (uint32_t)p.gt.offsets[t];
        plan_t[t] = plan_level<2>(p.gt.lc.scale[t], p.gt.lc.resolution[t], o, (uint32_t)p.gt.offsets[t + 1] - o, p.gt.gridtype,
                                  (uint32_t)sizeof(TT) * 2u);
    }

    for (int i = threadIdx.x; i < kTorsoPacked / 4; i += kFusedThreads)
        reinterpret_cast<float4 *>(lds)[i] = reinterpret_cast<const float4 *>(p.packed)[i];
    // enc_pose = freq(poses6, deg 4) -> 54 values (network.py:197), same layout as k_freq_forward
    if (threadIdx.x < 54) {
        const int c = threadIdx.x;
        float v;
        if (c < 6) v = p.poses6[c];
        else {
            const int col = c / 6 - 1, d = c % 6, f = col / 2;
            const float a = scalbnf(p.poses6[d], f);
            v = (col & 1) ? sinf(a + 3.141592653589793f / 2) : sinf(a);
        }
        enc_pose[c] = v;
    }
    __syncthreads();
    // broadcast columns: deform [42 .. 96+ind), torso [74 .. 128+ind)  (network.py:201, 212)
    if (threadIdx.x < 96) {
        const int t = threadIdx.x;
        const bool is_def = t < 64;
        const int row = is_def ? t : t - 64;
        const int ld = (is_def ? 96 : 128) + (int)p.w.ind_dim;
        const float *r = (is_def ? p.w.def_w0 : p.w.tor_w0) + row * ld + (is_def ? 42 : 74);
        float acc = 0.0f;
        for (int k = 0; k < 54; k++) acc += r[k] * enc_pose[k];
        for (uint32_t c = 0; c < p.w.ind_dim; c++) acc += r[54 + c] * p.ind_code[c];
        (is_def ? bias_def : bias_tor)[row] = acc;
    }
    __syncthreads();

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 31, h = lane >> 5;
    const int lane_off = h * 64 + j * 2, lane_off32 = h * 32 + j;
    const uint32_t n_tiles = (p.N + 63u) >> 6;

    for (uint32_t tile = blockIdx.x * kWavesPerBlock + wave; tile < n_tiles; tile += gridDim.x * kWavesPerBlock) {
        const uint32_t px = tile * 64 + lane;
        const bool in_range = px < p.N;
        float cx = 0.0f, cy = 0.0f;
        bool on = false;
        if (in_range) {
            cx = p.bg_coords[2 * (size_t)px]; cy = p.bg_coords[2 * (size_t)px + 1];
            on = sample_torso_grid(p.density_grid, p.G, cx, cy) > p.thresh;
        }
        float bgc[3] = {1.0f, 1.0f, 1.0f};
        if (in_range && p.bg_in) { bgc[0] = p.bg_in[3 * (size_t)px]; bgc[1] = p.bg_in[3 * (size_t)px + 1]; bgc[2] = p.bg_in[3 * (size_t)px + 2]; }
        if (__ballot(on) == 0ull) {  // no torso pixel in this tile: background passes through
            if (in_range) {
                if (p.bg_out) { p.bg_out[3 * (size_t)px] = bgc[0]; p.bg_out[3 * (size_t)px + 1] = bgc[1]; p.bg_out[3 * (size_t)px + 2] = bgc[2]; }
                if constexpr (BLEND) blend_pixel(p.blend, px, bgc);
                if (p.alpha_out) p.alpha_out[px] = 0.0f;
                if (p.deform_out) { p.deform_out[2 * (size_t)px] = 0.0f; p.deform_out[2 * (size_t)px + 1] = 0.0f; }
            }
            continue;
        }
        // x = x * torso_shrink; enc_x = freq(x, deg 10) (network.py:194,198): [x, sin(2^f x), cos(2^f x)]_f
        const float x0 = cx * p.shrink, x1 = cy * p.shrink;
        float bq[2][21];
        {
            float fq[42];
            fq[0] = x0; fq[1] = x1;
#pragma unroll
            for (int f = 0; f < 10; f++) {
                const float a0 = scalbnf(x0, f), a1 = scalbnf(x1, f);
                fq[2 + 4 * f + 0] = on ? sinf(a0) : 0.0f;
                fq[2 + 4 * f + 1] = on ? sinf(a1) : 0.0f;
                fq[2 + 4 * f + 2] = on ? sinf(a0 + 3.141592653589793f / 2) : 0.0f;
                fq[2 + 4 * f + 3] = on ? sinf(a1 + 3.141592653589793f / 2) : 0.0f;
            }
            if (!on) { fq[0] = 0.0f; fq[1] = 0.0f; }
#pragma unroll
            for (int s = 0; s < 21; s++) to_b_operands(fq[2 * s], fq[2 * s + 1], bq[0][s], bq[1][s]);
        }
        // deform net 104 -> 64 -> 64 -> 2
        Acc a0, a1;
        acc_bias(a0, bias_def, h);
#pragma unroll
        for (int s = 0; s < 21; s++) step64(a0, lds + TOFF_D0, s, lane_off, bq[0][s], bq[1][s]);
        acc_relu(a0);
        acc_zero(a1);
        layer_from_acc(a1, a0, lds + TOFF_D1, lane_off);
        acc_relu(a1);
        float dxy[2];
        {
            float part[2][2];
            valu_out<2>(a1, lds + TOFF_D2, h, part);
            dxy[0] = h ? part[1][0] : part[0][0];
            dxy[1] = h ? part[1][1] : part[0][1];
        }
        // x = clamp(x + dx, -1, 1); torso grid (bound = 1)
        float bg_[2][16];
        {
            float in[2] = {(fminf(fmaxf(x0 + dxy[0], -1.0f), 1.0f) + 1.0f) / 2.0f,
                           (fminf(fmaxf(x1 + dxy[1], -1.0f), 1.0f) + 1.0f) / 2.0f};
            const bool ok = on && !(in[0] < 0 || in[0] > 1 || in[1] < 0 || in[1] > 1);
            // kTorsoGroup levels in flight (the deform net's accumulators are dead by now): the 16 gathers are a latency chain
            // of one tile, and a torso launch is one tile per wave
            LevelFetch<TT, 2, 2> f[kTorsoGroup];
#pragma unroll
            for (int g = 0; g < 16; g += kTorsoGroup) {
                if (ok) {
#pragma unroll
                    for (int i = 0; i < kTorsoGroup; i++)
                        issue_planned<TT, 2, 2, false>(static_cast<const TT *>(p.gt.table), plan_t[g + i], in, f[i]);
                }
#pragma unroll
                for (int i = 0; i < kTorsoGroup; i++) {
                    float f0 = 0.0f, f1 = 0.0f;
                    if (ok) {
                        TT res[2];
                        TT dummy[1];
                        blend_level<TT, 2, 2, false>(f[i], 0.0f, res, dummy);
                        f0 = to_f<TT>(res[0]);
                        f1 = to_f<TT>(res[1]);
                    }
                    to_b_operands(f0, f1, bg_[0][g + i], bg_[1][g + i]);
                }
            }
        }
        // torso net 136 -> 32 -> 32 -> 4 : a single 32-row tile
        f32x16 t0[2], t1[2];
#pragma unroll
        for (int g = 0; g < 4; g++) {
            const float4 b = *reinterpret_cast<const float4 *>(bias_tor + 8 * g + 4 * h);
            t0[0][4 * g] = b.x; t0[0][4 * g + 1] = b.y; t0[0][4 * g + 2] = b.z; t0[0][4 * g + 3] = b.w;
            t0[1][4 * g] = b.x; t0[1][4 * g + 1] = b.y; t0[1][4 * g + 2] = b.z; t0[1][4 * g + 3] = b.w;
        }
#pragma unroll
        for (int s = 0; s < 16; s++) {
            const float wv = lds[TOFF_T0 + s * kTStep32 + lane_off32];
            t0[0] = mfma32(wv, bg_[0][s], t0[0]); t0[1] = mfma32(wv, bg_[1][s], t0[1]);
        }
#pragma unroll
        for (int s = 0; s < 21; s++) {
            const float wv = lds[TOFF_T0 + (16 + s) * kTStep32 + lane_off32];
            t0[0] = mfma32(wv, bq[0][s], t0[0]); t0[1] = mfma32(wv, bq[1][s], t0[1]);
        }
#pragma unroll
        for (int r = 0; r < 16; r++) { t0[0][r] = fmaxf(t0[0][r], 0.0f); t0[1][r] = fmaxf(t0[1][r], 0.0f); t1[0][r] = 0.0f; t1[1][r] = 0.0f; }
#pragma unroll
        for (int s = 0; s < 16; s++) {
            const float wv = lds[TOFF_T1 + s * kTStep32 + lane_off32];
            t1[0] = mfma32(wv, t0[0][s], t1[0]); t1[1] = mfma32(wv, t0[1][s], t1[1]);
        }
        float o4[4];
#pragma unroll
        for (int o = 0; o < 4; o++) {
            float p0 = 0.0f, p1 = 0.0f;
            const float *wo = lds + TOFF_T2 + (o * 2 + h) * 16;
#pragma unroll
            for (int r = 0; r < 16; r++) {
                p0 = __builtin_fmaf(fmaxf(t1[0][r], 0.0f), wo[r], p0);
                p1 = __builtin_fmaf(fmaxf(t1[1][r], 0.0f), wo[r], p1);
            }
            p0 += __shfl_xor(p0, 32, 64);
            p1 += __shfl_xor(p1, 32, 64);
            o4[o] = h ? p1 : p0;
        }
        if (in_range) {
            float alpha = 0.0f, col[3] = {0.0f, 0.0f, 0.0f};
            if (on) {
                alpha = 1.0f / (1.0f + expf(-o4[0]));
#pragma unroll
                for (int c = 0; c < 3; c++) col[c] = 1.0f / (1.0f + expf(-o4[1 + c]));
            }
            // bg = torso_color * alpha + bg * (1 - alpha)  (renderer.py:299)
            float bgf[3];
#pragma unroll
            for (int c = 0; c < 3; c++) bgf[c] = col[c] * alpha + bgc[c] * (1 - alpha);
            if (p.bg_out) {
#pragma unroll
                for (int c = 0; c < 3; c++) p.bg_out[3 * (size_t)px + c] = bgf[c];
            }
            if constexpr (BLEND) blend_pixel(p.blend, px, bgf);
            if (p.alpha_out) p.alpha_out[px] = alpha;
            if (p.deform_out) { p.deform_out[2 * (size_t)px] = on ? dxy[0] : 0.0f; p.deform_out[2 * (size_t)px + 1] = on ? dxy[1] : 0.0f; }
        }
    }
}

// Pixels the torso layer covers: bilinear occupancy of the 2-D torso grid above the threshold (renderer.py:281-283).  Used by
// the differentiable (training) formulation, which gathers those pixels for the PyTorch layers; inference goes through
// k_torso_fused, which tests the same expression per pixel.
__global__ void __launch_bounds__(256)
k_torso_mask(const float *__restrict__ bg_coords, uint32_t N, const float *__restrict__ grid, uint32_t G, float thresh,
             uint8_t *__restrict__ mask) {
    const uint32_t n = blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    mask[n] = sample_torso_grid(grid, G, bg_coords[2 * (size_t)n], bg_coords[2 * (size_t)n + 1]) > thresh ? 1 : 0;
}

// renderer.py:306-311
__global__ void __launch_bounds__(256)
k_blend(float *__restrict__ image, const float *__restrict__ weights_sum, const float *__restrict__ bg,
        float *__restrict__ depth, const float *__restrict__ nears, const float *__restrict__ fars, uint32_t N,
        uint8_t *__restrict__ u8) {
    const uint32_t n = blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    const float w = 1 - weights_sum[n];
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const float b = bg ? bg[3 * (size_t)n + c] : 1.0f;
        float v = image[3 * (size_t)n + c] + w * b;
        v = fminf(fmaxf(v, 0.0f), 1.0f);
        image[3 * (size_t)n + c] = v;
        if (u8) u8[3 * (size_t)n + c] = (uint8_t)(v * 255.0f);
    }
    const float dd = depth[n] - nears[n];
    depth[n] = fmaxf(dd, 0.0f) / (fars[n] - nears[n]);
}

// ---- host helpers ---------------------------------------------------------------------------------------
static int check_grid(const rn_grid_t *g, uint32_t D, const char *name) {
    RN_REQUIRE(g && g->embeddings && g->offsets, "%s: null grid", name);
    RN_REQUIRE(g->D == D && g->L == 16, "%s: fused path needs D=%u, L=16 (got D=%u L=%u)", name, D, g->D, g->L);
    RN_REQUIRE(g->gridtype <= 1 && (g->dtype == RN_F32 || g->dtype == RN_F16), "%s: bad gridtype/dtype", name);
    RN_REQUIRE(((uintptr_t)g->embeddings & 7u) == 0, "%s: table must be 8-byte aligned", name);
    return RN_OK;
}
static GridArgs grid_args(const rn_grid_t *g) {
    return GridArgs{g->embeddings, g->offsets, make_level_consts(g->L, g->S, g->H), g->gridtype};
}
static RawW raw_w(const rn_nerf_weights_t *w) {
    return RawW{w->amb_w0, w->amb_w1, w->amb_w2, w->sig_w0, w->sig_w1, w->sig_w2, w->col_w0, w->col_w1,
                w->audio_dim, w->has_eye, w->ind_dim};
}
static int check_w(const rn_nerf_weights_t *w) {
    RN_REQUIRE(w && w->amb_w0 && w->amb_w1 && w->amb_w2 && w->sig_w0 && w->sig_w1 && w->sig_w2 && w->col_w0 && w->col_w1,
               "nerf weights: null pointer");
    RN_REQUIRE(w->has_eye <= 1 && w->audio_dim <= 1024 && w->ind_dim <= 1024, "nerf weights: bad dims");
    return RN_OK;
}

static int num_cus() {
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
        if (cus <= 0) cus = 256;
    }
    return cus;
}

template <typename TX, typename TW>
static void launch_fused(const FusedParams &p, hipStream_t s) {
    const uint32_t n_tiles = (p.M + 31u) >> 5;
    uint32_t blocks = div_up(n_tiles, kF32Waves);
    const uint32_t cap = (uint32_t)num_cus();  // one persistent workgroup per CU (96.5 KB of LDS each)
    if (blocks > cap) blocks = cap;
    RN_LAUNCH_TIMED((k_nerf_fused<TX, TW>), dim3(blocks), dim3(kF32Threads), s, p);
}

static int run_fused(const float *xyzs, const float *dirs, const float *deltas, uint32_t M, const int32_t *m_dev,
                     const rn_grid_t *gx, const rn_grid_t *gw, const float *packed, const float *bias, float bound,
                     float *sigmas, float *rgbs, float *ambient, int mlp_dtype, hipStream_t s, const int32_t *slots = nullptr) {
    FusedParams p{xyzs, dirs, deltas, M, m_dev, grid_args(gx), grid_args(gw), packed, bias, bound, sigmas, rgbs, ambient, slots};
    if (mlp_dtype == RN_F32_SPLIT) {
        launch_fused_x2(p, gx->dtype, gw->dtype, (uint32_t)num_cus(), s);
    } else if (mlp_dtype == RN_F16) {
        uint32_t blocks = div_up((M + 63u) >> 6, kWavesPerBlock);
        const uint32_t cap = (uint32_t)num_cus();
        launch_fused_h16(p, gx->dtype, gw->dtype, blocks > cap ? cap : blocks, s);
    } else if (gx->dtype == RN_F32 && gw->dtype == RN_F32) launch_fused<float, float>(p, s);
    else if (gx->dtype == RN_F16 && gw->dtype == RN_F16) launch_fused<__half, __half>(p, s);
    else if (gx->dtype == RN_F32) launch_fused<float, __half>(p, s);
    else launch_fused<__half, float>(p, s);
    return RN_OK;
}

}  // namespace rn

using namespace rn;

extern "C" {

size_t rn_nerf_packed_floats(void) { return (size_t)kPacked; }
size_t rn_nerf_packed_floats_h16(void) { return packed_floats_h16(); }
size_t rn_nerf_packed_floats_split(void) { return packed_floats_x2(); }

int rn_nerf_pack_weights_split(const rn_nerf_weights_t *w, float *packed, rn_stream_t stream) {
    if (int rc = check_w(w)) return rc;
    RN_REQUIRE(packed && ((uintptr_t)packed & 15u) == 0, "nerf_pack_weights_split: packed must be 16-byte aligned");
    launch_pack_nerf_x2(raw_w(w), packed, as_stream(stream));
    return check_launch("nerf_pack_weights_split");
}

int rn_nerf_pack_weights_h16(const rn_nerf_weights_t *w, float *packed, rn_stream_t stream) {
    if (int rc = check_w(w)) return rc;
    RN_REQUIRE(packed && ((uintptr_t)packed & 15u) == 0, "nerf_pack_weights_h16: packed must be 16-byte aligned");
    launch_pack_nerf_h16(raw_w(w), packed, as_stream(stream));
    return check_launch("nerf_pack_weights_h16");
}
size_t rn_nerf_bias_floats(void) { return (size_t)kBias; }

int rn_nerf_pack_weights(const rn_nerf_weights_t *w, float *packed, rn_stream_t stream) {
    if (int rc = check_w(w)) return rc;
    RN_REQUIRE(packed && ((uintptr_t)packed & 15u) == 0, "nerf_pack_weights: packed must be 16-byte aligned");
    hipLaunchKernelGGL(k_pack_nerf, dim3(div_up(kPacked, 256)), dim3(256), 0, as_stream(stream), raw_w(w), packed);
    return check_launch("nerf_pack_weights");
}

int rn_nerf_frame_bias(const rn_nerf_weights_t *w, const float *enc_a, const float *eye, const float *ind_code,
                       float *bias, rn_stream_t stream) {
    if (int rc = check_w(w)) return rc;
    RN_REQUIRE(bias, "nerf_frame_bias: null output");
    RN_REQUIRE(enc_a || w->audio_dim == 0, "nerf_frame_bias: enc_a is required");
    RN_REQUIRE(eye || !w->has_eye, "nerf_frame_bias: eye is required when has_eye");
    RN_REQUIRE(ind_code || w->ind_dim == 0, "nerf_frame_bias: ind_code is required when ind_dim > 0");
    hipLaunchKernelGGL(k_frame_bias, dim3(1), dim3(kBias), 0, as_stream(stream), raw_w(w), enc_a, eye, ind_code, bias);
    return check_launch("nerf_frame_bias");
}

int rn_nerf_frame_bias_batch(const rn_nerf_weights_t *w, const float *enc_a, uint32_t n, const float *eye, const float *ind_code,
                             float *bias, rn_stream_t stream) {
    if (n == 0) return RN_OK;
    if (int rc = check_w(w)) return rc;
    RN_REQUIRE(bias && (enc_a || w->audio_dim == 0) && (eye || !w->has_eye) && (ind_code || w->ind_dim == 0),
               "nerf_frame_bias_batch: null pointer");
    hipLaunchKernelGGL(k_frame_bias, dim3(n), dim3(kBias), 0, as_stream(stream), raw_w(w), enc_a, eye, ind_code, bias);
    return check_launch("nerf_frame_bias_batch");
}

int rn_nerf_fused_forward(const float *xyzs, const float *dirs, const float *deltas, uint32_t M, const int32_t *m_dev,
                          const rn_grid_t *grid_xyz, const rn_grid_t *grid_amb, const float *packed, const float *bias,
                          float bound, float *sigmas, float *rgbs, float *ambient, int mlp_dtype, rn_stream_t stream) {
    if (M == 0) return RN_OK;
    RN_REQUIRE(xyzs && packed && bias && sigmas && (dirs || !rgbs), "nerf_fused_forward: null pointer");
    RN_REQUIRE(((uintptr_t)packed & 15u) == 0, "nerf_fused_forward: packed must be 16-byte aligned");
    if (int rc = check_grid(grid_xyz, 3, "nerf_fused_forward(xyz grid)")) return rc;
    if (int rc = check_grid(grid_amb, 2, "nerf_fused_forward(ambient grid)")) return rc;
    RN_REQUIRE(mlp_dtype == RN_F32 || mlp_dtype == RN_F16 || mlp_dtype == RN_F32_SPLIT,
               "nerf_fused_forward: mlp_dtype must be RN_F32, RN_F16 or RN_F32_SPLIT");
    run_fused(xyzs, dirs, deltas, M, m_dev, grid_xyz, grid_amb, packed, bias, bound, sigmas, rgbs, ambient, mlp_dtype,
              as_stream(stream));
    return check_launch("nerf_fused_forward");
}

static int check_head(const rn_head_t *h) {
    RN_REQUIRE(h, "head: null descriptor");
    RN_REQUIRE(h->rays_o && h->rays_d && h->aabb && h->bitfield && h->nears && h->fars && h->weights_sum && h->depth &&
                   h->image && h->rays_alive_a && h->rays_alive_b && h->rays_t && h->xyzs && h->dirs && h->deltas &&
                   h->sigmas && h->rgbs && h->state && h->block_counts,
               "head: null pointer");
    RN_REQUIRE(h->N >= 1 && h->max_steps >= 1 && h->cascade >= 1 && h->cascade <= 16 && h->grid_size >= 1, "head: bad sizes");
    return RN_OK;
}

int rn_head_begin(const rn_head_t *h, rn_stream_t stream) {
    if (int rc = check_head(h)) return rc;
    uint32_t order_w = h->order_w;
    if (order_w && (order_w % 8u || h->N % order_w || (h->N / order_w) % 8u)) order_w = 0;
    hipLaunchKernelGGL(k_head_begin, dim3(div_up(h->N, kLoopBlock)), dim3(kLoopBlock), 0, as_stream(stream), h->rays_o,
                       h->rays_d, h->aabb, h->N, h->min_near, h->max_steps, h->nears, h->fars, h->weights_sum, h->depth,
                       h->image, h->rays_alive_a, h->rays_t, h->state, order_w);
    return check_launch("head_begin");
}

int rn_head_iterate_ex(const rn_head_t *h, const rn_grid_t *grid_xyz, const rn_grid_t *grid_amb, const float *packed,
                       const float *bias, uint32_t first_iter, uint32_t n_iters, int mlp_dtype, uint32_t flags, rn_stream_t stream) {
    if (int rc = check_head(h)) return rc;
    RN_REQUIRE(mlp_dtype == RN_F32 || mlp_dtype == RN_F16 || mlp_dtype == RN_F32_SPLIT,
               "head_iterate: mlp_dtype must be RN_F32, RN_F16 or RN_F32_SPLIT");
    RN_REQUIRE(packed && bias && ((uintptr_t)packed & 15u) == 0, "head_iterate: packed/bias");
    if (int rc = check_grid(grid_xyz, 3, "head_iterate(xyz grid)")) return rc;
    if (int rc = check_grid(grid_amb, 2, "head_iterate(ambient grid)")) return rc;
    hipStream_t s = as_stream(stream);
    const dim3 rgrid(div_up(h->N, kLoopBlock)), rblock(kLoopBlock);
    // caller's scratch: survivor counts | live-sample partial sums of even iterations | ... of odd iterations
    const uint32_t nb = div_up(h->N, kLoopBlock) + 1;
    uint32_t *block_live[2] = {h->block_counts + nb, h->block_counts + 2 * nb};
    for (uint32_t it = first_iter; it < first_iter + n_iters; it++) {
        int32_t *st = h->state + (it & 1u) * 8, *st_next = h->state + ((it + 1) & 1u) * 8;
        int32_t *alive = (it & 1u) ? h->rays_alive_b : h->rays_alive_a;
        int32_t *alive_next = (it & 1u) ? h->rays_alive_a : h->rays_alive_b;
        // A call marches its own first iteration; after that the compaction kernel marches the next iteration itself.
        // The last compaction of a call does not, so that a caller may adjust the schedule between calls
        // (rn_head_reschedule) -- enqueueing the loop one iteration per call reproduces the four-kernel sequence.
        if (it == first_iter && !(flags & RN_LOOP_FIRST_MARCHED))
            hipLaunchKernelGGL(k_head_march, rgrid, rblock, 0, s, st, alive, h->rays_t, h->rays_o, h->rays_d, h->bound,
                               h->dt_gamma, h->max_steps, h->cascade, h->grid_size, h->bitfield, h->fars, h->xyzs, h->dirs,
                               h->deltas, h->state, block_live[it & 1u], st + 6, h->live_slots);
        // the network runs over the iteration's live list when the caller gave room for one (st[6] entries), else over all
        // st[3] slots, skipping the dead ones by their deltas
        run_fused(h->xyzs, h->dirs, h->deltas, h->N, h->live_slots ? st + 6 : st + 3, grid_xyz, grid_amb, packed, bias, h->bound,
                  h->sigmas, h->rgbs, nullptr, mlp_dtype, s, h->live_slots);
        const MarchArgs m{h->rays_t, h->rays_o, h->rays_d, h->fars, h->bound, h->dt_gamma, h->cascade, h->grid_size, h->bitfield,
                          h->xyzs, h->dirs, h->deltas, block_live[(it + 1) & 1u], h->live_slots};
        const bool march_next = it + 1 < first_iter + n_iters;
        const uint32_t close = (!march_next && (flags & RN_LOOP_CLOSE_FRAME)) ? 1u : 0u;
        if (flags & RN_LOOP_COOP) {  // compositor + compaction (+ next march) behind one launch (k_head_step)
            // a COOPERATIVE launch: the runtime places all workgroups of the grid on the device together or refuses the launch
            // (the in-kernel barrier polls counts other workgroups of the same launch publish, so they must be resident)
            const dim3 cgrid(rgrid.x < kStepGrid ? rgrid.x : kStepGrid);
            uint32_t a_N = h->N, a_max = h->max_steps, a_close = march_next ? 0u : close, a_it = it;
            float a_T = h->T_thresh;
            const int32_t *a_st = st;
            int32_t *a_st_next = st_next, *a_alive = alive, *a_alive_next = alive_next, *a_state = h->state;
            float *a_rays_t = h->rays_t, *a_ws = h->weights_sum, *a_depth = h->depth, *a_image = h->image;
            const float *a_sig = h->sigmas, *a_rgb = h->rgbs, *a_deltas = h->deltas;
            uint32_t *a_counts = h->block_counts;
            const uint32_t *a_live = block_live[it & 1u];
            MarchArgs a_m = m;
            void *args[] = {&a_st, &a_st_next, &a_N, &a_max, &a_T, &a_alive, &a_alive_next, &a_rays_t, &a_sig, &a_rgb, &a_deltas, &a_ws,
                            &a_depth, &a_image, &a_counts, &a_live, &a_state, &a_m, &a_close, &a_it};
            const void *fn = march_next ? reinterpret_cast<const void *>(&k_head_step<true>) : reinterpret_cast<const void *>(&k_head_step<false>);
            const hipError_t e = hipLaunchCooperativeKernel(fn, cgrid, rblock, args, 0, s);
            if (e != hipSuccess) {
                (void)hipGetLastError();
                set_error("head_iterate: cooperative launch of the one-launch loop step refused (%s); use the split loop (no RN_LOOP_COOP)",
                          hipGetErrorString(e));
                return RN_ERR_INVALID_ARG;
            }
            continue;
        }
        hipLaunchKernelGGL(k_head_composite, rgrid, rblock, 0, s, st, h->T_thresh, alive, h->rays_t, h->sigmas, h->rgbs,
                           h->deltas, h->weights_sum, h->depth, h->image, h->block_counts, st_next);
        if (march_next)
            hipLaunchKernelGGL(k_head_compact<true>, rgrid, rblock, 0, s, st, st_next, h->N, h->max_steps, alive, alive_next,
                               h->block_counts, block_live[it & 1u], h->state, m, 0u, it);
        else
            hipLaunchKernelGGL(k_head_compact<false>, rgrid, rblock, 0, s, st, st_next, h->N, h->max_steps, alive, alive_next,
                               h->block_counts, block_live[it & 1u], h->state, m, close, it);
    }
    return check_launch("head_iterate");
}

int rn_head_iterate(const rn_head_t *h, const rn_grid_t *grid_xyz, const rn_grid_t *grid_amb, const float *packed,
                    const float *bias, uint32_t first_iter, uint32_t n_iters, int mlp_dtype, rn_stream_t stream) {
    return rn_head_iterate_ex(h, grid_xyz, grid_amb, packed, bias, first_iter, n_iters, mlp_dtype, 0u, stream);
}

int rn_frame_begin(const rn_head_t *h, const float *pose, float fx, float fy, float cx, float cy, uint32_t W, rn_stream_t stream) {
    if (int rc = check_head(h)) return rc;
    RN_REQUIRE(!pose || (fx != 0.0f && fy != 0.0f && W >= 1 && h->N % W == 0), "frame_begin: bad intrinsics / image width");
    uint32_t order_w = h->order_w;
    if (order_w && (order_w % 8u || h->N % order_w || (h->N / order_w) % 8u)) order_w = 0;
    const uint32_t nb = div_up(h->N, kLoopBlock) + 1;
    const RaySource rs{pose, fx, fy, cx, cy, W ? W : 1u};
    hipLaunchKernelGGL(k_frame_begin, dim3(div_up(h->N, kLoopBlock)), dim3(kLoopBlock), 0, as_stream(stream), rs,
                       const_cast<float *>(h->rays_o), const_cast<float *>(h->rays_d), h->aabb, h->N, h->min_near, h->max_steps, h->bound,
                       h->dt_gamma, h->cascade, h->grid_size, h->bitfield, h->nears, h->fars, h->weights_sum, h->depth, h->image,
                       h->rays_alive_a, h->rays_t, h->state, order_w, h->xyzs, h->dirs, h->deltas, h->block_counts + nb, h->live_slots);
    return check_launch("frame_begin");
}

int rn_head_check_done(const rn_head_t *h, uint32_t iters_done, rn_stream_t stream) {
    if (int rc = check_head(h)) return rc;
    hipLaunchKernelGGL(k_head_check_done, dim3(1), dim3(64), 0, as_stream(stream), h->state + (iters_done & 1u) * 8,
                       h->state + RN_HEAD_ST_UNFINISHED);
    return check_launch("head_check_done");
}

int rn_head_reschedule(const rn_head_t *h, uint32_t iter_done, uint32_t schedule_N, const int32_t *alive_total,
                       rn_stream_t stream) {
    if (int rc = check_head(h)) return rc;
    RN_REQUIRE(alive_total && schedule_N >= h->N, "head_reschedule: alive_total is null or schedule_N < N");
    hipLaunchKernelGGL(k_head_reschedule, dim3(1), dim3(64), 0, as_stream(stream), h->state + ((iter_done + 1) & 1u) * 8,
                       schedule_N, alive_total);
    return check_launch("head_reschedule");
}

size_t rn_torso_packed_floats(void) { return (size_t)kTorsoPacked; }

int rn_torso_pack_weights(const rn_torso_weights_t *w, float *packed, rn_stream_t stream) {
    RN_REQUIRE(w && w->def_w0 && w->def_w1 && w->def_w2 && w->tor_w0 && w->tor_w1 && w->tor_w2 && packed,
               "torso_pack_weights: null pointer");
    RN_REQUIRE(((uintptr_t)packed & 15u) == 0, "torso_pack_weights: packed must be 16-byte aligned");
    RawT r{w->def_w0, w->def_w1, w->def_w2, w->tor_w0, w->tor_w1, w->tor_w2, w->ind_dim};
    hipLaunchKernelGGL(k_pack_torso, dim3(div_up(kTorsoPacked, 256)), dim3(256), 0, as_stream(stream), r, packed);
    return check_launch("torso_pack_weights");
}

int rn_torso_fused(const float *bg_coords, uint32_t N, const float *density_grid_torso, uint32_t grid_size, float thresh,
                   const float *poses6, const float *ind_code, float torso_shrink, const rn_torso_weights_t *w,
                   const float *packed, const rn_grid_t *grid_torso, const float *bg_in, float *bg_out,
                   float *torso_alpha, float *deform, rn_stream_t stream) {
    if (N == 0) return RN_OK;
    RN_REQUIRE(bg_coords && density_grid_torso && poses6 && w && packed && (bg_out || torso_alpha), "torso_fused: null pointer");
    RN_REQUIRE(ind_code || w->ind_dim == 0, "torso_fused: ind_code required when ind_dim > 0");
    RN_REQUIRE(((uintptr_t)packed & 15u) == 0, "torso_fused: packed must be 16-byte aligned");
    if (int rc = check_grid(grid_torso, 2, "torso_fused(torso grid)")) return rc;
    RawT r{w->def_w0, w->def_w1, w->def_w2, w->tor_w0, w->tor_w1, w->tor_w2, w->ind_dim};
    TorsoParams p{bg_coords, N, density_grid_torso, grid_size, thresh, poses6, ind_code, torso_shrink, r, packed,
                  grid_args(grid_torso), bg_in, bg_out, torso_alpha, deform, BlendArgs{}};
    uint32_t blocks = div_up((N + 63u) >> 6, kWavesPerBlock);
    const uint32_t cap = (uint32_t)num_cus() * 2;
    if (blocks > cap) blocks = cap;
    if (grid_torso->dtype == RN_F32) hipLaunchKernelGGL((k_torso_fused<float, false>), dim3(blocks), dim3(kFusedThreads), 0, as_stream(stream), p);
    else hipLaunchKernelGGL((k_torso_fused<__half, false>), dim3(blocks), dim3(kFusedThreads), 0, as_stream(stream), p);
    return check_launch("torso_fused");
}

int rn_torso_blend_frame(const float *bg_coords, uint32_t N, const float *density_grid_torso, uint32_t grid_size, float thresh,
                         const float *poses6, const float *ind_code, float torso_shrink, const rn_torso_weights_t *w,
                         const float *packed, const rn_grid_t *grid_torso, const float *bg_in, float *bg_out, float *torso_alpha,
                         float *image, const float *weights_sum, float *depth, const float *nears, const float *fars,
                         uint8_t *image_u8, rn_stream_t stream) {
    if (N == 0) return RN_OK;
    RN_REQUIRE(bg_coords && density_grid_torso && poses6 && w && packed, "torso_blend_frame: null pointer");
    RN_REQUIRE(image && weights_sum && depth && nears && fars, "torso_blend_frame: null frame buffers");
    RN_REQUIRE(ind_code || w->ind_dim == 0, "torso_blend_frame: ind_code required when ind_dim > 0");
    RN_REQUIRE(((uintptr_t)packed & 15u) == 0, "torso_blend_frame: packed must be 16-byte aligned");
    if (int rc = check_grid(grid_torso, 2, "torso_blend_frame(torso grid)")) return rc;
    RawT r{w->def_w0, w->def_w1, w->def_w2, w->tor_w0, w->tor_w1, w->tor_w2, w->ind_dim};
    TorsoParams p{bg_coords, N, density_grid_torso, grid_size, thresh, poses6, ind_code, torso_shrink, r, packed,
                  grid_args(grid_torso), bg_in, bg_out, torso_alpha, nullptr, BlendArgs{image, weights_sum, depth, nears, fars, image_u8}};
    uint32_t blocks = div_up((N + 63u) >> 6, kWavesPerBlock);
    const uint32_t cap = (uint32_t)num_cus() * 2;
    if (blocks > cap) blocks = cap;
    if (grid_torso->dtype == RN_F32) hipLaunchKernelGGL((k_torso_fused<float, true>), dim3(blocks), dim3(kFusedThreads), 0, as_stream(stream), p);
    else hipLaunchKernelGGL((k_torso_fused<__half, true>), dim3(blocks), dim3(kFusedThreads), 0, as_stream(stream), p);
    return check_launch("torso_blend_frame");
}

int rn_torso_mask(const float *bg_coords, uint32_t N, const float *density_grid_torso, uint32_t grid_size, float thresh,
                  uint8_t *mask, rn_stream_t stream) {
    if (N == 0) return RN_OK;
    RN_REQUIRE(bg_coords && density_grid_torso && mask && grid_size >= 2, "torso_mask: bad arguments");
    hipLaunchKernelGGL(k_torso_mask, dim3(div_up(N, 256)), dim3(256), 0, as_stream(stream), bg_coords, N, density_grid_torso, grid_size,
                       thresh, mask);
    return check_launch("torso_mask");
}

int rn_blend_frame(float *image, const float *weights_sum, const float *bg, float *depth, const float *nears,
                   const float *fars, uint32_t N, uint8_t *image_u8, rn_stream_t stream) {
    if (N == 0) return RN_OK;
    RN_REQUIRE(image && weights_sum && depth && nears && fars, "blend_frame: null pointer");
    hipLaunchKernelGGL(k_blend, dim3(div_up(N, 256)), dim3(256), 0, as_stream(stream), image, weights_sum, bg, depth, nears,
                       fars, N, image_u8);
    return check_launch("blend_frame");
}

}  // extern "C"
