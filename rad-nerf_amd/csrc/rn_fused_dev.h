// rn_fused_dev.h -- pieces shared by the fp32-MFMA (rn_fused.hip) and the f16-MFMA (rn_fused_h16.hip) variants of
// the fused per-sample network kernel: accumulator tiles, the VALU output layers, kernel parameter blocks.
#pragma once

#include "rn_dda_dev.h"
#include "rn_grid_dev.h"
#include "rn_sh_dev.h"

#include "../../include/radnerf_fused.h"

#ifndef RN_XCD_TILES
#define RN_XCD_TILES 1
#endif
#ifndef RN_TILE_CHUNK
#define RN_TILE_CHUNK 2
#endif

namespace rn {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kFusedThreads = 512;
constexpr int kWavesPerBlock = kFusedThreads / kWave;


__host__ __device__ constexpr int rowmap(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }
// k index (within a 64-wide hidden vector) that lane-half h feeds at MFMA step s when the B operand is
// register (s & 15) of row tile (s >> 4) of the previous layer's accumulators.
__host__ __device__ constexpr int kmap(int s, int h) { return 32 * (s >> 4) + rowmap(s & 15, h); }

struct RawW {
    const float *amb_w0, *amb_w1, *amb_w2, *sig_w0, *sig_w1, *sig_w2, *col_w0, *col_w1;
    uint32_t audio_dim, has_eye, ind_dim;
};


// acc[column tile][row tile]
struct Acc {
    f32x16 v[2][2];
};

__device__ __forceinline__ void acc_zero(Acc &a) {
#pragma unroll
    for (int nt = 0; nt < 2; nt++)
#pragma unroll
        for (int rt = 0; rt < 2; rt++)
#pragma unroll
            for (int r = 0; r < 16; r++) a.v[nt][rt][r] = 0.0f;
}

// accumulator rows of lane half h: 32 rt + (r & 3) + 8 (r >> 2) + 4 h -> four consecutive floats per r >> 2
__device__ __forceinline__ void acc_bias(Acc &a, const float *bias64, int h) {
#pragma unroll
    for (int rt = 0; rt < 2; rt++)
#pragma unroll
        for (int g = 0; g < 4; g++) {
            const float4 b = *reinterpret_cast<const float4 *>(bias64 + 32 * rt + 8 * g + 4 * h);
            a.v[0][rt][4 * g + 0] = b.x; a.v[0][rt][4 * g + 1] = b.y; a.v[0][rt][4 * g + 2] = b.z; a.v[0][rt][4 * g + 3] = b.w;
            a.v[1][rt][4 * g + 0] = b.x; a.v[1][rt][4 * g + 1] = b.y; a.v[1][rt][4 * g + 2] = b.z; a.v[1][rt][4 * g + 3] = b.w;
        }
}

// max(x, 0) as ONE v_max_i32 on the bit pattern (a non-negative float is a non-negative integer, a negative one a negative
// integer); fmaxf(x, 0) costs two VALU instructions because IEEE mode first quiets a possible signalling NaN.
__device__ __forceinline__ float relu_bits(float x) {
    const int b = __float_as_int(x);
    return __int_as_float(b > 0 ? b : 0);
}

__device__ __forceinline__ void acc_relu(Acc &a) {
#pragma unroll
    for (int nt = 0; nt < 2; nt++)
#pragma unroll
        for (int rt = 0; rt < 2; rt++)
#pragma unroll
            for (int r = 0; r < 16; r++) a.v[nt][rt][r] = relu_bits(a.v[nt][rt][r]);
}


// out[o] (both column tiles) = sum_k in[k] * W[o][k] with the k's this lane holds; caller adds the other half
template <int NOUT>
__device__ __forceinline__ void valu_out(const Acc &in, const float *wl, int h, float (&part)[2][NOUT]) {
#pragma unroll
    for (int o = 0; o < NOUT; o++) {
        float p0 = 0.0f, p1 = 0.0f;
        const float *wo = wl + (o * 2 + h) * 32;
#pragma unroll
        for (int g = 0; g < 8; g++) {
            const float4 w = *reinterpret_cast<const float4 *>(wo + 4 * g);
            const int rt = g >> 2, r = (g & 3) * 4;
            p0 = __builtin_fmaf(in.v[0][rt][r + 0], w.x, p0); p1 = __builtin_fmaf(in.v[1][rt][r + 0], w.x, p1);
            p0 = __builtin_fmaf(in.v[0][rt][r + 1], w.y, p0); p1 = __builtin_fmaf(in.v[1][rt][r + 1], w.y, p1);
            p0 = __builtin_fmaf(in.v[0][rt][r + 2], w.z, p0); p1 = __builtin_fmaf(in.v[1][rt][r + 2], w.z, p1);
            p0 = __builtin_fmaf(in.v[0][rt][r + 3], w.w, p0); p1 = __builtin_fmaf(in.v[1][rt][r + 3], w.w, p1);
        }
        part[0][o] = p0 + __shfl_xor(p0, 32, 64);
        part[1][o] = p1 + __shfl_xor(p1, 32, 64);
    }
}


struct GridArgs {
    const void *table;
    const int32_t *offsets;
    LevelConsts lc;
    uint32_t gridtype;
};

struct FusedParams {
    const float *xyzs, *dirs, *deltas;
    uint32_t M;
    const int32_t *m_dev;
    GridArgs gx, gw;
    const float *packed, *bias;
    float bound;
    float *sigmas, *rgbs, *ambient;
    // optional list of live sample slots: entry j names the slot (row of xyzs / dirs / sigmas / rgbs / ambient) that the j-th
    // sample of the launch works on, every entry is live and M counts entries.  NULL: sample j is slot j (dead where
    // deltas[2 j] == 0).  Inside the frame loop the marchers write it, so the network skips the dead slots of rays that
    // ended in the middle of their n_step samples (16 % of the slots of the benchmark stream).
    const int32_t *slots;
};

// XCD-aware tile schedule.  Workgroups are dealt round-robin over the 8 XCDs (workgroup b lands on XCD b % 8; used for
// speed only -- any placement gives the same results), and every XCD has its own 4 MB L2.  Samples arrive ray-ordered,
// i.e. consecutive tiles are neighbouring pixels whose samples share most of their grid rows on the coarse and middle
// levels.  Giving each XCD one CONTIGUOUS eighth of the tiles (a band of the image) instead of every eighth workgroup's
// tiles keeps those shared rows in one L2 instead of fetching them into all eight.
struct TileSchedule {
    uint32_t first, end, stride;
    __device__ __forceinline__ TileSchedule(uint32_t n_tiles, uint32_t waves_per_block, uint32_t wave) {
        const uint32_t G = gridDim.x, b = blockIdx.x;
#if RN_XCD_TILES
        if (G >= 8 && (G & 7u) == 0) {
            const uint32_t xcd = b & 7u, local = b >> 3, per_xcd = (n_tiles + 7u) >> 3;
            const uint32_t lo = xcd * per_xcd, hi = lo + per_xcd < n_tiles ? lo + per_xcd : n_tiles;
            // Within the band, RN_TILE_CHUNK consecutive tiles go to one workgroup and the next chunk to the next
            // workgroup; a workgroup's second chunk lands on its next RN_TILE_CHUNK waves.  A launch of the frame loop
            // holds 1.0 - 2.0 tiles per wave slot of the chip, so what matters is that the tiles beyond one full round
            // are spread over all CUs (chunk < waves per workgroup) instead of doubling up a few of them.
            constexpr uint32_t C = RN_TILE_CHUNK;
            const uint32_t chunk = (C < waves_per_block && waves_per_block % C == 0) ? C : waves_per_block, B = G >> 3;
            first = lo + ((wave / chunk) * B + local) * chunk + wave % chunk;
            end = lo < hi ? hi : lo;
            stride = B * waves_per_block;
            return;
        }
#endif
        first = b * waves_per_block + wave;
        end = n_tiles;
        stride = G * waves_per_block;
    }
};

// Launch of the f16-MFMA variant (rn_fused_h16.hip); `gx_dtype` / `gw_dtype` are the grid table dtypes.
void launch_fused_h16(const FusedParams &p, int gx_dtype, int gw_dtype, uint32_t blocks, hipStream_t s);
void launch_pack_nerf_h16(const RawW &w, float *packed, hipStream_t s);
size_t packed_floats_h16();
// Launch of the split-precision variant (rn_fused_x2.hip): picks its own grid (one 256-thread workgroup per CU).
void launch_fused_x2(const FusedParams &p, int gx_dtype, int gw_dtype, uint32_t n_cus, hipStream_t s);
void launch_pack_nerf_x2(const RawW &w, float *packed, hipStream_t s);
size_t packed_floats_x2();

}  // namespace rn
