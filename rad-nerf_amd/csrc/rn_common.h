// rn_common.h -- shared host/device helpers for libradnerf_hip.so (gfx950 only).
//
// Built with -ffp-contract=off: float expressions that feed integer results
// (DDA cell index, lattice position) round exactly as written, which makes
// them bit-identical to the CPU oracle.  Fused multiply-adds are spelled out
// with __builtin_fmaf where a kernel wants them.
#pragma once

#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <hip/hip_fp16.h>
#include <stdint.h>

#include "../../include/radnerf_hip.h"

namespace rn {

constexpr int kWave = 64;  // CDNA wavefront width

void set_error(const char *fmt, ...);
int check_launch(const char *what);
// HIP-event timing of the dominant kernel (rn_prof_enable / rn_prof_collect)
bool prof_enabled();
void prof_pair(hipEvent_t *start, hipEvent_t *stop);
// launch with the timing events of prof_pair() attached to the dispatch (plain launch when timing is off)
#define RN_LAUNCH_TIMED(kernel, grid, block, stream, ...)                                      \
    do {                                                                                       \
        hipEvent_t rn_e0_, rn_e1_;                                                             \
        ::rn::prof_pair(&rn_e0_, &rn_e1_);                                                     \
        hipExtLaunchKernelGGL(kernel, grid, block, 0, stream, rn_e0_, rn_e1_, 0, __VA_ARGS__); \
    } while (0)

static inline uint32_t div_up(uint32_t a, uint32_t b) { return (a + b - 1) / b; }
static inline hipStream_t as_stream(rn_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

#define RN_REQUIRE(cond, ...)                 \
    do {                                      \
        if (!(cond)) {                        \
            ::rn::set_error(__VA_ARGS__);     \
            return RN_ERR_INVALID_ARG;        \
        }                                     \
    } while (0)

// ---- device helpers ------------------------------------------------------------------

__device__ __forceinline__ float clampf(float x, float lo, float hi) { return fminf(hi, fmaxf(lo, x)); }

// Morton code of a 10-bit-per-axis cell (raymarching.cu:56-71).
__device__ __forceinline__ uint32_t expand_bits(uint32_t v) {
    v = (v * 0x00010001u) & 0xFF0000FFu;
    v = (v * 0x00000101u) & 0x0F00F00Fu;
    v = (v * 0x00000011u) & 0xC30C30C3u;
    v = (v * 0x00000005u) & 0x49249249u;
    return v;
}
__device__ __forceinline__ uint32_t morton3D(uint32_t x, uint32_t y, uint32_t z) {
    return expand_bits(x) | (expand_bits(y) << 1) | (expand_bits(z) << 2);
}
// The same for coordinates < 256 (every occupancy grid of this repo: H = 128): expand_bits()' first step only moves bits 8 and 9.
__device__ __forceinline__ uint32_t expand_bits8(uint32_t v) {
    v = (v * 0x00000101u) & 0x0F00F00Fu;
    v = (v * 0x00000011u) & 0xC30C30C3u;
    v = (v * 0x00000005u) & 0x49249249u;
    return v;
}
__device__ __forceinline__ uint32_t morton3D_8(uint32_t x, uint32_t y, uint32_t z) {
    return expand_bits8(x) | (expand_bits8(y) << 1) | (expand_bits8(z) << 2);
}
// raymarching.cu:73-81
__device__ __forceinline__ uint32_t morton3D_invert(uint32_t x) {
    x = x & 0x49249249u;
    x = (x | (x >> 2)) & 0xc30c30c3u;
    x = (x | (x >> 4)) & 0x0f00f00fu;
    x = (x | (x >> 8)) & 0xff0000ffu;
    x = (x | (x >> 16)) & 0x0000ffffu;
    return x;
}

// Lane index inside the wavefront and wave-level exclusive prefix of a ballot.
__device__ __forceinline__ uint32_t lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }
__device__ __forceinline__ uint32_t ballot_prefix(unsigned long long mask) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

}  // namespace rn
