// rn_fused_x2.hip -- the fused per-sample network kernel with fp32-grade contractions on the 16-bit matrix cores
// (mlp_dtype = RN_F32_SPLIT).
//
// Same computation and tile ownership as k_nerf_fused / k_nerf_fused_h16 (nerf/network.py:222-283).  Every operand of a
// contraction is split into two fp16 numbers, v = hi + lo with hi = fp16(v), lo = fp16(v - hi) (22 significant bits), and
// a product a * b is evaluated as a_hi b_hi + a_hi b_lo + a_lo b_hi on the f16 matrix instruction (mfma16 below) with fp32 accumulation;
// the dropped a_lo b_lo term is below 2^-22 |a b|.  Per-product error is ~3e-7 |a b| (fp32: 6e-8), so the kernel meets
// the SAME tolerances against the fp32 oracle as the fp32-MFMA kernel (sigma rel 2e-4, rgb / ambient abs 2e-5), at 3
// MFMA k-steps of 32 cycles per 16 k instead of 8 fp32 MFMAs of 64 cycles: 5.3x less matrix-core time.
//
// Layout differences from the f16 kernel: both halves of the weights live in LDS (94 KB), the narrow fp32 output layers
// and the per-frame bias are read from global memory (L1/L2 hits), the staging tile holds hi and lo of enc_x only
// (32 words per sample, XOR-swizzled 16-byte chunks instead of padding) while the ambient-grid features go from registers
// to fragments with v_permlane32_swap -- 65.5 + 94.2 KB of LDS for one 512-thread workgroup per CU, two waves per SIMD.
#include "rn_fused_dev.h"


namespace rn {

#ifndef RN_FUSED_PAIR_HASHED
#define RN_FUSED_PAIR_HASHED 0
#endif
constexpr bool kPairHashedX2 = RN_FUSED_PAIR_HASHED;

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// ---- packed weight image ----------------------------------------------------------------------------------
// MFMA section, fp16: [k-step][row tile][lane half h][row i][8 k] -> lane (i, h) reads 16 B per row tile and k-step.
constexpr int kHStep = 2 * 2 * 32 * 8;        // halves per k-step (both row tiles)
constexpr int KS_A0 = 0;                      // ambient L0, enc_x        : 2 k-steps
constexpr int KS_A1 = KS_A0 + 2;              // ambient L1               : 4
constexpr int KS_S0 = KS_A1 + 4;              // sigma L0, enc_x | enc_w  : 2 + 2
constexpr int KS_S1 = KS_S0 + 4;              // sigma L1                 : 4
constexpr int KS_S2 = KS_S1 + 4;              // sigma L2 rows 1..64      : 4
constexpr int KS_C0 = KS_S2 + 4;              // color L0, sh | geo       : 1 + 4
constexpr int kHSteps = KS_C0 + 5;            // 23
// after both fp16 sections: the fp32 narrow layers (same [out][h][q] layout as the fp32 kernel): ambient L2 | sigma L2 row 0
// | color L1
#ifndef RN_X2_THREADS
#define RN_X2_THREADS 512
#endif
constexpr int kX2Threads = RN_X2_THREADS;               // one workgroup per CU, two waves per SIMD
constexpr int kX2Waves = kX2Threads / kWave;
constexpr int kLoOff = kHSteps * kHStep;      // halves: the lo section follows the hi section
constexpr int kX2MfmaFloats = kHSteps * kHStep;          // both fp16 sections, counted in 4-byte units
constexpr int XOFF_A2 = kX2MfmaFloats, XOFF_S2R = XOFF_A2 + 128, XOFF_C1 = XOFF_S2R + 64;
constexpr int kX2Packed = XOFF_C1 + 192;      // 23936 four-byte units (as large as the fp32 image)
constexpr int kStageRow = 32;                 // words per sample: hi of 16 feature pairs | lo of them (enc_x, later SH)
constexpr int kStageWords = 64 * kStageRow;   // per wave

// k index that element j of lane half h feeds at k-step (2 rt + g) when the B fragment is registers 8g..8g+7 of row
// tile rt of the previous layer's accumulators
__host__ __device__ constexpr int hmap(int rt, int g, int h, int j) { return 32 * rt + 16 * g + 8 * (j >> 2) + 4 * h + (j & 3); }

__global__ void __launch_bounds__(256) k_pack_nerf_x2(RawW w, float *__restrict__ packed) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    const int ldA0 = 32 + (int)w.audio_dim, ldS0 = 64 + (int)w.has_eye, ldC0 = 80 + (int)w.ind_dim;
    if (e < kHSteps * kHStep) {  // one fp16 element
        const int ks = e / kHStep, rem = e % kHStep;
        const int rt_out = rem / 512, h = (rem % 512) / 256, i = (rem % 256) / 8, j = rem % 8;
        const int row = 32 * rt_out + i;
        const int nat = 8 * h + j;  // natural k inside a k-step
        float v;
        if (ks < KS_A1) v = w.amb_w0[row * ldA0 + 16 * (ks - KS_A0) + nat];
        else if (ks < KS_S0) { const int q = ks - KS_A1; v = w.amb_w1[row * 64 + hmap(q >> 1, q & 1, h, j)]; }
        else if (ks < KS_S1) v = w.sig_w0[row * ldS0 + 16 * (ks - KS_S0) + nat];
        else if (ks < KS_S2) { const int q = ks - KS_S1; v = w.sig_w1[row * 64 + hmap(q >> 1, q & 1, h, j)]; }
        else if (ks < KS_C0) { const int q = ks - KS_S2; v = w.sig_w2[(1 + row) * 64 + hmap(q >> 1, q & 1, h, j)]; }
        else if (ks == KS_C0) v = w.col_w0[row * ldC0 + nat];
        else { const int q = ks - KS_C0 - 1; v = w.col_w0[row * ldC0 + 16 + hmap(q >> 1, q & 1, h, j)]; }
        const _Float16 hi = (_Float16)v;
        reinterpret_cast<_Float16 *>(packed)[e] = hi;
        reinterpret_cast<_Float16 *>(packed)[kLoOff + e] = (_Float16)(v - (float)hi);
        return;
    }
    const int f = e - kHSteps * kHStep + kX2MfmaFloats;  // fp32 section
    if (f >= kX2Packed) return;
    auto valu_elem = [&](int base, const float *src) -> float {  // [out][h][q], q = rt*16 + r
        const int q0 = f - base, o = q0 / 64, h = (q0 % 64) / 32, q = q0 % 32;
        return src[o * 64 + 32 * (q >> 4) + rowmap(q & 15, h)];
    };
    float v;
    if (f < XOFF_S2R) v = valu_elem(XOFF_A2, w.amb_w2);
    else if (f < XOFF_C1) v = valu_elem(XOFF_S2R, w.sig_w2);
    else v = valu_elem(XOFF_C1, w.col_w1);
    packed[f] = v;
}

// The contraction instruction.  gfx950's double-rate v_mfma_f32_32x32x16_f16 is NOT used: with two waves resident per
// SIMD this kernel then returned, in ~1 of 300 tiles and differently from launch to launch, results computed with stale
// operand data in lanes 48..63 of one fragment (reproduced with identical inputs in every lane; gone with one wave per
// SIMD, with every mix of s_nop / s_waitcnt around the instruction still present) -- see DESIGN.md section 3.  The
// K = 8 form below is bit-stable under the same conditions.  One 16-k step = two K = 8 instructions on elements
// 0..3 and 4..7 of both fragments (any pairing that takes the same elements from A and B sums the same products);
// the matrix pipe is far from binding in these kernels, so the 2x instruction count is not measurable.
__device__ __forceinline__ f32x16 mfma16(f16x8 a, f16x8 b, f32x16 c) {
    typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
    const f16x4 a0 = {a[0], a[1], a[2], a[3]}, a1 = {a[4], a[5], a[6], a[7]};
    const f16x4 b0 = {b[0], b[1], b[2], b[3]}, b1 = {b[4], b[5], b[6], b[7]};
    return __builtin_amdgcn_mfma_f32_32x32x8f16(a1, b1, __builtin_amdgcn_mfma_f32_32x32x8f16(a0, b0, c, 0, 0, 0), 0, 0, 0);
}

struct Frag2 {
    f16x8 hi, lo;
};

// one k-step (16 k) of a 64-row layer in split precision: W X ~ Wh Xh + Wh Xl + Wl Xh for both row tiles x both column tiles
__device__ __forceinline__ void xstep(Acc &a, const _Float16 *wl, int ks, int lane_off8, const Frag2 &b0, const Frag2 &b1) {
#pragma unroll
    for (int rt = 0; rt < 2; rt++) {
        const f16x8 wh = *reinterpret_cast<const f16x8 *>(wl + ks * kHStep + rt * 512 + lane_off8);
        const f16x8 wo = *reinterpret_cast<const f16x8 *>(wl + kLoOff + ks * kHStep + rt * 512 + lane_off8);
        a.v[0][rt] = mfma16(wh, b0.lo, a.v[0][rt]);
        a.v[1][rt] = mfma16(wh, b1.lo, a.v[1][rt]);
        a.v[0][rt] = mfma16(wo, b0.hi, a.v[0][rt]);
        a.v[1][rt] = mfma16(wo, b1.hi, a.v[1][rt]);
        a.v[0][rt] = mfma16(wh, b0.hi, a.v[0][rt]);
        a.v[1][rt] = mfma16(wh, b1.hi, a.v[1][rt]);
    }
}

// registers 8g..8g+7 of one accumulator tile, split into fp16 hi + lo: the B fragments of k-step (2 rt + g) of the next layer
__device__ __forceinline__ Frag2 acc_frag2(const Acc &in, int nt, int rt, int g) {
    Frag2 r;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const float v = in.v[nt][rt][8 * g + j];
        const _Float16 hi = (_Float16)v;
        r.hi[j] = hi;
        r.lo[j] = (_Float16)(v - (float)hi);
    }
    return r;
}

__device__ __forceinline__ void xlayer_from_acc(Acc &out, const Acc &in, const _Float16 *wl, int ks0, int lane_off8) {
#pragma unroll
    for (int rt = 0; rt < 2; rt++)
#pragma unroll
        for (int g = 0; g < 2; g++)
            xstep(out, wl, ks0 + 2 * rt + g, lane_off8, acc_frag2(in, 0, rt, g), acc_frag2(in, 1, rt, g));
}

// staging tile: row = sample, 8 chunks of 4 words; chunk c holds hi of pairs 4c..4c+3 (c < 4) or lo of pairs 4(c-4)..
// (c >= 4); the chunk index is XOR-ed with (row & 7) so that the 16-byte fragment reads of consecutive rows spread over
// the banks without padding (the tile must stay at 8 KB per wave to fit next to 94 KB of weights)
__device__ __forceinline__ uint32_t stage_word(int row, int chunk, int within) {
    return (uint32_t)(row * kStageRow + ((chunk ^ (row & 7)) << 2) + within);
}
__device__ __forceinline__ void split2(float f0, float f1, uint32_t &hi, uint32_t &lo) {
    const _Float16 h0 = (_Float16)f0, h1 = (_Float16)f1;
    f16x2 vh, vl;
    vh[0] = h0; vh[1] = h1;
    vl[0] = (_Float16)(f0 - (float)h0); vl[1] = (_Float16)(f1 - (float)h1);
    hi = __builtin_bit_cast(uint32_t, vh);
    lo = __builtin_bit_cast(uint32_t, vl);
}
__device__ __forceinline__ void stage_pair(uint32_t *stage, int row, int q, float f0, float f1) {
    uint32_t hi, lo;
    split2(f0, f1, hi, lo);
    stage[stage_word(row, q >> 2, q & 3)] = hi;
    stage[stage_word(row, 4 + (q >> 2), q & 3)] = lo;
}
// B fragments (hi, lo) of k-step s (feature pairs 8 s .. 8 s + 7, s = 0, 1) for both column tiles
__device__ __forceinline__ void stage_frags2(const uint32_t *stage, int s, int j, int h, Frag2 &b0, Frag2 &b1) {
    const int c = 2 * s + h;
    b0.hi = __builtin_bit_cast(f16x8, *reinterpret_cast<const u32x4 *>(stage + stage_word(j, c, 0)));
    b0.lo = __builtin_bit_cast(f16x8, *reinterpret_cast<const u32x4 *>(stage + stage_word(j, 4 + c, 0)));
    b1.hi = __builtin_bit_cast(f16x8, *reinterpret_cast<const u32x4 *>(stage + stage_word(32 + j, c, 0)));
    b1.lo = __builtin_bit_cast(f16x8, *reinterpret_cast<const u32x4 *>(stage + stage_word(32 + j, 4 + c, 0)));
}
// Eight "one sample per lane" packed pairs (features 16 s .. 16 s + 15 of k-step s) -> the B fragments of both column
// tiles: v_permlane32_swap(w[r], w[4 + r]) leaves [own low-half word | partner's] = tile 0's register r and tile 1's.
__device__ __forceinline__ void lanes_to_frags(const uint32_t (&w)[8], f16x8 &t0, f16x8 &t1) {
    u32x4 a, b;
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const auto sw = __builtin_amdgcn_permlane32_swap(w[r], w[4 + r], false, false);
        a[r] = sw[0];
        b[r] = sw[1];
    }
    t0 = __builtin_bit_cast(f16x8, a);
    t1 = __builtin_bit_cast(f16x8, b);
}

// all lanes of the wave have written their staging rows; make them visible to the wave's reads (wave-private tile)
__device__ __forceinline__ void stage_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

#ifndef RN_X2_XYZ_GROUP
#define RN_X2_XYZ_GROUP 1
#endif
constexpr int kXyzGroup = RN_X2_XYZ_GROUP;  // xyz levels fetched together (each: 16 row words + 4 in flight)

template <typename TX, typename TW>
__global__ void __launch_bounds__(kX2Threads, 2) k_nerf_fused_x2(FusedParams p) {
    __shared__ __attribute__((aligned(16))) float lds[kX2MfmaFloats];
    __shared__ __attribute__((aligned(16))) uint32_t stage_all[kX2Waves * kStageWords];
    __shared__ LevelPlan plan_x[16], plan_w[16];

    uint32_t M = p.M;
    if (p.m_dev) { const uint32_t d = (uint32_t)*p.m_dev; M = d < M ? d : M; }
    const uint32_t n_tiles = (M + 63u) >> 6;
    {
        const TileSchedule w0(n_tiles, kX2Waves, 0u);
        if (w0.first >= w0.end) return;  // nothing for this workgroup (uniform)
    }

    for (int i = threadIdx.x; i < kX2MfmaFloats / 4; i += kX2Threads)
        reinterpret_cast<float4 *>(lds)[i] = reinterpret_cast<const float4 *>(p.packed)[i];
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
    const int lane_off8 = (h * 32 + j) * 8;
    const _Float16 *wl = reinterpret_cast<const _Float16 *>(lds);
    const float *bias_amb = p.bias, *bias_sig = p.bias + 64, *bias_col = p.bias + 128;  // global: L1-resident
    const float *valu_w = p.packed;                                                        // narrow fp32 layers, global
    uint32_t *stage = stage_all + wave * kStageWords;

    const TileSchedule sched(n_tiles, kX2Waves, (uint32_t)wave);
    for (uint32_t tile = sched.first; tile < sched.end; tile += sched.stride) {
        const uint32_t entry = tile * 64 + lane;
        bool live = entry < M;
        uint32_t sample = entry;  // the slot this lane's sample lives in
        if (p.slots) {
            if (live) sample = (uint32_t)p.slots[entry];
        } else if (live && p.deltas) {
            live = p.deltas[2 * (size_t)sample] != 0.0f;
        }
        if (__ballot(live) == 0ull) continue;  // whole tile dead (wave-uniform)

        // ---- xyz grid, one sample per lane -> 16 fp16 feature pairs in the staging tile
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
            LevelFetch<TX, 3, 2> f[kXyzGroup];
#pragma unroll 1
            for (int g = 0; g < 16; g += kXyzGroup) {
                if (on) {
#pragma unroll
                    for (int i = 0; i < kXyzGroup; i++) {
                        issue_planned<TX, 3, 2, kPairHashedX2>(static_cast<const TX *>(p.gx.table), plan_x[g + i], in, f[i]);
                    }
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
                    stage_pair(stage, lane, g + i, f0, f1);
                }
            }
        }
        stage_sync();

        // ---- ambient net: [enc_x | enc_a] 96 -> 64 -> 64 -> 2, tanh (bias = W0[:, 32:] enc_a)
        Acc a0, a1;
        acc_bias(a0, bias_amb, h);
#pragma unroll
        for (int s = 0; s < 2; s++) {
            Frag2 b0, b1;
            stage_frags2(stage, s, j, h, b0, b1);
            xstep(a0, wl, KS_A0 + s, lane_off8, b0, b1);
        }
        acc_relu(a0);
        acc_zero(a1);
        xlayer_from_acc(a1, a0, wl, KS_A1, lane_off8);
        acc_relu(a1);
        float amb[2];
        {
            float part[2][2];
            valu_out<2>(a1, valu_w + XOFF_A2, h, part);
            amb[0] = tanhf(h ? part[1][0] : part[0][0]);
            amb[1] = tanhf(h ? part[1][1] : part[0][1]);
        }
        if (p.ambient && live) {
            p.ambient[2 * (size_t)sample] = amb[0];
            p.ambient[2 * (size_t)sample + 1] = amb[1];
        }

        // ---- sigma net, first layer: [enc_x | enc_w | eye] (bias = W0[:, 64] eye).  The enc_x half comes from the
        // staging tile now (it is overwritten later); the enc_w half is fed from registers as the ambient grid is gathered.
        acc_bias(a0, bias_sig, h);
#pragma unroll
        for (int s = 0; s < 2; s++) {
            Frag2 b0, b1;
            stage_frags2(stage, s, j, h, b0, b1);
            xstep(a0, wl, KS_S0 + s, lane_off8, b0, b1);
        }

        // ---- ambient grid: enc_w = encoder_ambient(ambient, bound=1), 8 levels (= one k-step) at a time
        {
            float in[2] = {(amb[0] + 1.0f) / 2.0f, (amb[1] + 1.0f) / 2.0f};
            const bool on = live && !(in[0] < 0 || in[0] > 1 || in[1] < 0 || in[1] > 1);
            LevelFetch<TW, 2, 2> f[4];
#pragma unroll 1
            for (int g = 0; g < 16; g += 8) {
                uint32_t whi[8], wlo[8];
#pragma unroll
                for (int half = 0; half < 2; half++) {  // four levels in flight at a time (register budget)
                    if (on) {
#pragma unroll
                        for (int i = 0; i < 4; i++) {
                            issue_planned<TW, 2, 2, kPairHashedX2>(static_cast<const TW *>(p.gw.table), plan_w[g + 4 * half + i], in, f[i]);
                        }
                    }
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        float f0 = 0.0f, f1 = 0.0f;
                        if (on) {
                            TW res[2];
                            TW dummy[1];
                            blend_level<TW, 2, 2, false>(f[i], 0.0f, res, dummy);
                            f0 = to_f<TW>(res[0]);
                            f1 = to_f<TW>(res[1]);
                        }
                        split2(f0, f1, whi[4 * half + i], wlo[4 * half + i]);
                    }
                }
                Frag2 b0, b1;
                lanes_to_frags(whi, b0.hi, b1.hi);
                lanes_to_frags(wlo, b0.lo, b1.lo);
                xstep(a0, wl, KS_S0 + 2 + (g >> 3), lane_off8, b0, b1);
            }
        }

        // ---- sigma net: 65 -> 64 -> 64 -> 1 + 64
        acc_relu(a0);
        acc_zero(a1);
        xlayer_from_acc(a1, a0, wl, KS_S1, lane_off8);
        acc_relu(a1);
        float sigma;
        {
            float part[2][1];
            valu_out<1>(a1, valu_w + XOFF_S2R, h, part);
            sigma = expf(h ? part[1][0] : part[0][0]);  // trunc_exp forward (activation.py:9-11)
        }
        acc_zero(a0);
        xlayer_from_acc(a0, a1, wl, KS_S2, lane_off8);  // geo_feat (no activation)

        // ---- color net: [SH(d) | geo_feat | ind_code] 84 -> 64 -> 3, sigmoid
        stage_sync();
        {
            float sh[16];
            float dx = 0.0f, dy = 0.0f, dz = 0.0f;
            if (live && p.dirs) {
                dx = p.dirs[3 * (size_t)sample]; dy = p.dirs[3 * (size_t)sample + 1]; dz = p.dirs[3 * (size_t)sample + 2];
            }
            sh_basis<4>(dx, dy, dz, sh);
#pragma unroll
            for (int s = 0; s < 8; s++) stage_pair(stage, lane, s, sh[2 * s], sh[2 * s + 1]);
        }
        stage_sync();
        acc_bias(a1, bias_col, h);
        {
            Frag2 b0, b1;
            stage_frags2(stage, 0, j, h, b0, b1);
            xstep(a1, wl, KS_C0, lane_off8, b0, b1);
        }
        xlayer_from_acc(a1, a0, wl, KS_C0 + 1, lane_off8);
        acc_relu(a1);
        {
            float part[2][3];
            valu_out<3>(a1, valu_w + XOFF_C1, h, part);
            if (live) {
                p.sigmas[sample] = sigma;
#pragma unroll
                for (int c = 0; c < 3; c++) {
                    const float x = h ? part[1][c] : part[0][c];
                    if (p.rgbs) p.rgbs[3 * (size_t)sample + c] = 1.0f / (1.0f + expf(-x));   // NULL: density query
                }
            }
        }
        stage_sync();  // the next tile's gathers overwrite the staging rows
    }
}

void launch_fused_x2(const FusedParams &p, int gx_dtype, int gw_dtype, uint32_t n_cus, hipStream_t s) {
    uint32_t blocks = div_up((p.M + 63u) >> 6, kX2Waves);
    if (blocks > n_cus) blocks = n_cus;
    const dim3 g(blocks), b(kX2Threads);
    if (gx_dtype == RN_F32 && gw_dtype == RN_F32) RN_LAUNCH_TIMED((k_nerf_fused_x2<float, float>), g, b, s, p);
    else if (gx_dtype == RN_F16 && gw_dtype == RN_F16) RN_LAUNCH_TIMED((k_nerf_fused_x2<__half, __half>), g, b, s, p);
    else if (gx_dtype == RN_F32) RN_LAUNCH_TIMED((k_nerf_fused_x2<float, __half>), g, b, s, p);
    else RN_LAUNCH_TIMED((k_nerf_fused_x2<__half, float>), g, b, s, p);
}

void launch_pack_nerf_x2(const RawW &w, float *packed, hipStream_t s) {
    const int n = kHSteps * kHStep + (kX2Packed - kX2MfmaFloats);
    hipLaunchKernelGGL(k_pack_nerf_x2, dim3(div_up(n, 256)), dim3(256), 0, s, w, packed);
}

size_t packed_floats_x2() { return (size_t)kX2Packed; }

}  // namespace rn
