// rn_grid.hip -- multiresolution hash / tiled grid encoder for gfx950.
//
// Behaviour follows gridencoder/src/gridencoder.cu of the reference (cited per kernel).
// MI355X-first choices:
//   * per-level constants (scale, resolution) are computed once on the HOST with the same libm
//     exp2f the CPU oracle uses and travel as kernel arguments, so the lattice position -- and with
//     it every integer table index -- is bit-identical to the oracle (-ffp-contract=off keeps
//     x*scale+0.5 unfused);
//   * a feature row (C scalars) is fetched with ONE load of C*sizeof(T) bytes;
//   * two work decompositions: level-major (one (sample, level) per lane, reference layout
//     [L,B,C], only one level's table is live in an XCD's L2 at a time) and sample-major (one
//     sample per lane walks all levels and stores its whole [L*C] row, layout [B, L*C], which
//     removes the reference's permute copy).
#include "rn_grid_dev.h"

#include <math.h>
#include <stdlib.h>

#ifndef RN_GRID_PAIR_HASHED
#define RN_GRID_PAIR_HASHED 1
#endif

namespace rn {

// GridEncoder.forward's `inputs = (inputs + bound) / (2 * bound)` (gridencoder/grid.py:149) folded into the lookup's coordinate
// load: x -> (x + add) * mul with mul = 1 / (2 * bound) in fp32 -- what the division by a host scalar computes on the device
// in PyTorch -- so the module needs no pass of its own over the coordinates.  on = 0: coordinates are used as given.
struct InXform { float add, mul; uint32_t on; };

template <uint32_t D>
__device__ __forceinline__ bool load_input(const float *__restrict__ inputs, uint32_t b, float (&in)[D], const InXform xf = InXform{0, 1, 0}) {
    bool oob = false;
#pragma unroll
    for (uint32_t d = 0; d < D; d++) {
        float x = inputs[(size_t)b * D + d];
        if (xf.on) x = (x + xf.add) * xf.mul;
        in[d] = x;
        oob |= (in[d] < 0 || in[d] > 1);  // gridencoder.cu:113-117
    }
    return oob;
}

// One level of one sample.  The common case (align_corners off, linear interpolation, D = 2 / 3) goes through the per-level
// plan of rn_grid_dev.h -- the level's facts are wave-uniform here, so plan_level() runs on the scalar unit -- and costs
// ~115 VALU instructions instead of ~500: the level-major lookup was bound by its instruction stream, not by memory.
template <typename T, uint32_t D, uint32_t C, bool DYDX>
__device__ __forceinline__ void encode_one(const T *__restrict__ table, uint32_t off, const float (&in)[D], float scale,
                                           uint32_t resolution, uint32_t hashmap_size, uint32_t gridtype, bool align_corners,
                                           uint32_t interp, T (&results)[C], T (&grads)[DYDX ? D * C : 1]) {
    if constexpr (D == 2 || D == 3) {
        if (!align_corners && interp == 0) {
            const LevelPlan lp = plan_level<D>(scale, resolution, off, hashmap_size, gridtype, (uint32_t)(sizeof(T) * C));
            LevelFetch<T, D, C> f;
            issue_planned<T, D, C, RN_GRID_PAIR_HASHED != 0, true>(table, lp, in, f);
            blend_level<T, D, C, DYDX>(f, scale, results, grads);
            return;
        }
    }
    encode_level<T, D, C, DYDX>(table, off, in, scale, resolution, hashmap_size, gridtype, align_corners, interp, results, grads);
}

// ------------------------------------------------------------------------------------------------
// Level-major forward (gridencoder.cu:87-244): grid = (ceil(B/256), L).
template <typename T, uint32_t D, uint32_t C, bool DYDX, int LAYOUT>
__global__ void __launch_bounds__(256)
k_grid_fwd_level(const float *__restrict__ inputs, const T *__restrict__ table, const int32_t *__restrict__ offsets,
                 T *__restrict__ outputs, uint32_t B, uint32_t L, LevelConsts lc, T *__restrict__ dy_dx,
                 uint32_t gridtype, bool align_corners, uint32_t interp, uint32_t level_base, uint32_t level_skip, InXform xf) {
    const uint32_t b = blockIdx.x * 256 + threadIdx.x;
    if (b >= B) return;
    uint32_t level = blockIdx.y + level_base;         // level_base > 0: the coarse levels were done by k_grid_fwd_coarse
    level += level >= level_skip ? 1u : 0u;           // level_skip < L: that level is another launch's (k_grid_fwd_level_rows)

    float in[D];
    const bool oob = load_input<D>(inputs, b, in, xf);

    T results[C];
    T grads[DYDX ? D * C : 1];
    if (oob) {
#pragma unroll
        for (uint32_t ch = 0; ch < C; ch++) results[ch] = from_f<T>(0.0f);
        if constexpr (DYDX) {
#pragma unroll
            for (uint32_t i = 0; i < D * C; i++) grads[i] = from_f<T>(0.0f);
        }
    } else {
        const uint32_t off = (uint32_t)offsets[level];
        const uint32_t hashmap_size = (uint32_t)offsets[level + 1] - off;
        encode_one<T, D, C, DYDX>(table, off, in, lc.scale[level], lc.resolution[level], hashmap_size,
                                  gridtype, align_corners, interp, results, grads);
    }
    T *o = (LAYOUT == RN_LAYOUT_LBC) ? outputs + ((size_t)level * B + b) * C : outputs + ((size_t)b * L + level) * C;
    store_row_nt<T, C>(o, results);  // +3 % on the hash table; non-temporal LOADS of the coordinates were -9 % (measured)
    if constexpr (DYDX) {
        T *g = dy_dx + ((size_t)b * L + level) * D * C;  // [B, L, D, C]
#pragma unroll
        for (uint32_t i = 0; i < D * C; i++) g[i] = grads[i];
    }
}

// Sample-major forward: one sample per lane walks every level, [B, L*C] rows.
template <typename T, uint32_t D, uint32_t C, bool DYDX, int LAYOUT>
__global__ void __launch_bounds__(256)
k_grid_fwd_sample(const float *__restrict__ inputs, const T *__restrict__ table, const int32_t *__restrict__ offsets,
                  T *__restrict__ outputs, uint32_t B, uint32_t L, LevelConsts lc, T *__restrict__ dy_dx,
                  uint32_t gridtype, bool align_corners, uint32_t interp) {
    const uint32_t b = blockIdx.x * 256 + threadIdx.x;
    if (b >= B) return;
    float in[D];
    const bool oob = load_input<D>(inputs, b, in);
    uint32_t off = (uint32_t)offsets[0];
    for (uint32_t level = 0; level < L; level++) {
        const uint32_t next = (uint32_t)offsets[level + 1];
        T results[C];
        T grads[DYDX ? D * C : 1];
        if (oob) {
#pragma unroll
            for (uint32_t ch = 0; ch < C; ch++) results[ch] = from_f<T>(0.0f);
            if constexpr (DYDX) {
#pragma unroll
                for (uint32_t i = 0; i < D * C; i++) grads[i] = from_f<T>(0.0f);
            }
        } else {
            encode_one<T, D, C, DYDX>(table, off, in, lc.scale[level], lc.resolution[level], next - off,
                                      gridtype, align_corners, interp, results, grads);
        }
        T *o = (LAYOUT == RN_LAYOUT_LBC) ? outputs + ((size_t)level * B + b) * C : outputs + ((size_t)b * L + level) * C;
        store_row<T, C>(o, results);
        if constexpr (DYDX) {
            T *g = dy_dx + ((size_t)b * L + level) * D * C;
#pragma unroll
            for (uint32_t i = 0; i < D * C; i++) g[i] = grads[i];
        }
        off = next;
    }
}

// ------------------------------------------------------------------------------------------------
// Coarse levels in ONE pass with the dense ones staged in LDS (level-major layout, [L,B,C]).
//
// Why: the level-major lookup spends one grid row (blockIdx.y) per level, and each re-reads the 12-byte coordinates and runs
// its own wave of workgroups.  For the hashed / capped levels that is right -- one 4 MB level is exactly what an XCD's L2
// holds, and those levels run at the chip's line-gather rate -- but the first levels are small, dense and served from L1/L2
// anyway: there the time goes into streaming the coordinates in and the features out, level after level.  Here a
// persistent 1024-thread workgroup per CU copies the dense levels that fit (hash T=2^19 / tiled T=2^16, fp32: level 0 =
// 4 920 rows + level 1 = 13 824 rows = 150 KB of the CU's 160 KB; fp16: levels 0..2) into LDS once, then walks sample
// tiles: coordinates loaded once, the LDS levels gathered with ds_read (no L1 tag, no L2 request), the remaining coarse
// levels (dense or not, up to kCoarseGlobal of them: their tables share an XCD's L2 comfortably) gathered from global
// memory with all of them in flight, features stored level by level (coalesced 8-byte rows, streaming).
// Bit-identical to the per-level kernel: same plans, same blend.
constexpr uint32_t kCoarseThreads = 1024;
constexpr uint32_t kCoarseGlobal = 4;        // coarse levels gathered from global memory per sample, all in flight
constexpr uint32_t kCoarseMaxLevels = 12;    // LDS-staged + global coarse levels of one pass
constexpr uint32_t kCoarseLdsBudget = 160u * 1024u - 2048u;   // dynamic LDS for staged rows (plans + slack stay below 160 KiB)

template <typename T, uint32_t D, uint32_t C>
__global__ void __launch_bounds__(kCoarseThreads)
k_grid_fwd_coarse(const float *__restrict__ inputs, const T *__restrict__ table, const int32_t *__restrict__ offsets,
                  T *__restrict__ outputs, uint32_t B, LevelConsts lc, uint32_t gridtype, uint32_t n_lds, uint32_t n_group,
                  uint32_t lds_bytes, InXform xf) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_rows[];
    __shared__ LevelPlan plans[kCoarseMaxLevels];
    for (uint32_t i = threadIdx.x * 16u; i < lds_bytes; i += kCoarseThreads * 16u)   // level sizes are multiples of 8 rows: 16-B chunks
        *reinterpret_cast<uint4 *>(lds_rows + i) = *reinterpret_cast<const uint4 *>(reinterpret_cast<const unsigned char *>(table) + i);
    if (threadIdx.x < n_group) {
        const uint32_t t = threadIdx.x, o = (uint32_t)offsets[t];
        plans[t] = plan_level<D>(lc.scale[t], lc.resolution[t], o, (uint32_t)offsets[t + 1] - o, gridtype, (uint32_t)(sizeof(T) * C));
    }
    __syncthreads();
    const uint32_t n_tiles = (B + kCoarseThreads - 1u) / kCoarseThreads;
    for (uint32_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const uint32_t b = tile * kCoarseThreads + threadIdx.x;
        if (b >= B) continue;
        float in[D];
        const bool oob = load_input<D>(inputs, b, in, xf);
        T dummy[1];
        if (oob) {
            T zero[C];
#pragma unroll
            for (uint32_t ch = 0; ch < C; ch++) zero[ch] = from_f<T>(0.0f);
            for (uint32_t l = 0; l < n_group; l++) store_row_nt<T, C>(outputs + ((size_t)l * B + b) * C, zero);
            continue;
        }
        // global coarse levels first (their loads stay in flight under the LDS levels)
        LevelFetch<T, D, C> fg[kCoarseGlobal];
#pragma unroll
        for (uint32_t i = 0; i < kCoarseGlobal; i++)
            if (n_lds + i < n_group) issue_planned<T, D, C, RN_GRID_PAIR_HASHED != 0, true>(table, plans[n_lds + i], in, fg[i]);
        for (uint32_t l = 0; l < n_lds; l++) {
            LevelFetch<T, D, C> f;
            issue_dense_lds<T, D, C>(lds_rows, plans[l], in, f);
            T res[C];
            blend_level<T, D, C, false>(f, 0.0f, res, dummy);
            store_row_nt<T, C>(outputs + ((size_t)l * B + b) * C, res);
        }
#pragma unroll
        for (uint32_t i = 0; i < kCoarseGlobal; i++)
            if (n_lds + i < n_group) {
                T res[C];
                blend_level<T, D, C, false>(fg[i], 0.0f, res, dummy);
                store_row_nt<T, C>(outputs + ((size_t)(n_lds + i) * B + b) * C, res);
            }
    }
}

// [L, Bc, C] (one chunk of the level-major lookup, in the caller's workspace) -> rows b0 .. b0 + Bc - 1 of [B, L*C].
// A workgroup moves a tile of samples through LDS: level slabs come in with coalesced 16-byte loads (every load of a
// thread's 16 levels is in flight before the first LDS write: 256 B per lane), whole feature rows go out as 16-byte stores,
// 64 lanes = 1 KiB contiguous.  Row stride in LDS is padded by 16 B (rows stay 16-byte aligned, banks spread).
constexpr uint32_t kTrThreads = 256;

template <uint32_t WORDS /* 32-bit words per (sample, level) row: 1, 2 or 4 */, uint32_t LMAX>
__global__ void __launch_bounds__(kTrThreads)
k_grid_lbc_to_blc(const uint32_t *__restrict__ src, uint32_t *__restrict__ dst, uint32_t Bc, uint32_t L, uint32_t tile_samples) {
    extern __shared__ __attribute__((aligned(16))) uint32_t tile[];
    constexpr uint32_t SPQ = 4u / WORDS;                                // samples per 16-byte quad of a level slab
    const uint32_t row_words = L * WORDS, stride = row_words + 4u;
    const uint32_t s0 = blockIdx.x * tile_samples;
    const uint32_t n = Bc - s0 < tile_samples ? Bc - s0 : tile_samples;
    // 16-byte loads need every level slab to start 16-byte aligned: Bc a multiple of SPQ (then n is one too, tiles being multiples
    // of 4 samples).  Otherwise (an odd tail chunk) the slabs are read word by word.
    const bool vec = (Bc % SPQ) == 0u;
    if (!vec) {
        for (uint32_t l = 0; l < L; l++) {
            const uint32_t *slab = src + ((size_t)l * Bc + s0) * WORDS;
            for (uint32_t e = threadIdx.x; e < n * WORDS; e += kTrThreads) tile[(e / WORDS) * stride + l * WORDS + (e % WORDS)] = slab[e];
        }
    }
    const uint32_t quads = vec ? n / SPQ : 0u;
    for (uint32_t q = threadIdx.x; q < quads; q += kTrThreads) {
        uint4 v[LMAX];
#pragma unroll
        for (uint32_t l = 0; l < LMAX; l++)
            if (l < L) v[l] = *reinterpret_cast<const uint4 *>(src + ((size_t)l * Bc + s0) * WORDS + (size_t)q * 4u);
#pragma unroll
        for (uint32_t l = 0; l < LMAX; l++)
            if (l < L) {
                const uint32_t w[4] = {v[l].x, v[l].y, v[l].z, v[l].w};
#pragma unroll
                for (uint32_t i = 0; i < 4; i++) tile[(q * SPQ + i / WORDS) * stride + l * WORDS + (i % WORDS)] = w[i];
            }
    }
    __syncthreads();
    uint32_t *out = dst + (size_t)s0 * row_words;
    if ((row_words & 3u) == 0u) {
        const uint32_t q_per_row = row_words >> 2, total = n * q_per_row;
        for (uint32_t q = threadIdx.x; q < total; q += kTrThreads) {
            const uint32_t s = q / q_per_row, k = q - s * q_per_row;
            const uint4 v = *reinterpret_cast<const uint4 *>(tile + s * stride + 4u * k);
            __builtin_nontemporal_store(v.x, out + (size_t)q * 4u);     // written once, read by another kernel
            __builtin_nontemporal_store(v.y, out + (size_t)q * 4u + 1);
            __builtin_nontemporal_store(v.z, out + (size_t)q * 4u + 2);
            __builtin_nontemporal_store(v.w, out + (size_t)q * 4u + 3);
        }
    } else {
        const uint32_t total = n * row_words;
        for (uint32_t q = threadIdx.x; q < total; q += kTrThreads) {
            const uint32_t s = q / row_words, k = q - s * row_words;
            out[q] = tile[s * stride + k];
        }
    }
}

// ------------------------------------------------------------------------------------------------
// A fine level's pass that ALSO moves NT finished levels of the chunk to their place in the [B, L*C] rows, so the module's
// layout needs no transposition pass of its own.  Why here: a hashed level's kernel waits for its 8 gathers per sample and
// leaves most of the memory pipes idle, so the coalesced reads of the other levels' slabs and the row stores ride along --
// the 2 x 0.5 GB that `k_grid_lbc_to_blc` streams in 205 us at B = 2^22 cost 140 us more than the level alone when they
// travel with level 5 (the first fine level of the T = 2^19 table: the cheapest gathers; 180 us with level 15).
// A launch has a duty of NT levels [t0, t0 + NT), all complete when it runs except possibly its own (slot `own`, taken from
// registers); per sample it writes one aligned segment of NT * C * sizeof(T) bytes: the whole row (<= 128 B) or a 128-byte
// line of it.  Segments smaller than a line were measured and are slower than the transposition pass (two 64-byte halves
// written 100 us apart: 1.03 ms against 0.95), so the LDS tile holds 128 samples x whole segments at a time (two passes,
// <= 20 KB: 7-8 workgroups per CU) rather than 256 samples x half segments.
// Same plan / issue / blend as k_grid_fwd_level: bit-identical features.
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <uint32_t W>
__device__ __forceinline__ void load_words_nt(const uint32_t *p, uint32_t (&v)[W]) {
    if constexpr (W == 1) v[0] = __builtin_nontemporal_load(p);
    else if constexpr (W == 2) { const u32x2 t = __builtin_nontemporal_load(reinterpret_cast<const u32x2 *>(p)); v[0] = t.x; v[1] = t.y; }
    else { static_assert(W == 4, "rows of 1, 2 or 4 words");
           const u32x4 t = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(p)); v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w; }
}

template <typename T, uint32_t D, uint32_t C, uint32_t NT>
__global__ void __launch_bounds__(256)
k_grid_fwd_level_rows(const float *__restrict__ inputs, const T *__restrict__ table, const int32_t *__restrict__ offsets,
                      T *__restrict__ ws /* [L, B, C] */, uint32_t *__restrict__ rows /* [B, L * C] as words */, uint32_t B, uint32_t L,
                      LevelConsts lc, uint32_t gridtype, uint32_t level, uint32_t t0, uint32_t own /* level - t0, or >= NT */, InXform xf) {
    constexpr uint32_t W = (uint32_t)(sizeof(T) * C / 4);     // words per (sample, level)
    constexpr uint32_t SEG = NT * W, STRIDE = SEG + 4u;       // words per sample moved here; padded LDS row (stays 16-byte aligned)
    constexpr uint32_t SP = SEG > 16u ? 128u : 256u;          // samples per LDS pass: <= 20 KB per workgroup keeps 8 of them on a CU
    static_assert(SEG % 4 == 0, "a sample's segment is whole 16-byte quads");
    __shared__ __attribute__((aligned(16))) uint32_t tile[SP * STRIDE];
    const uint32_t s0 = blockIdx.x * 256u, b = s0 + threadIdx.x;
    const bool have = b < B;
    float in[D];
    bool oob = true;
    if (have) oob = load_input<D>(inputs, b, in, xf);
    const uint32_t off = (uint32_t)offsets[level];
    const LevelPlan lp = plan_level<D>(lc.scale[level], lc.resolution[level], off, (uint32_t)offsets[level + 1] - off, gridtype,
                                       (uint32_t)(sizeof(T) * C));
    LevelFetch<T, D, C> f;
    if (!oob) issue_planned<T, D, C, RN_GRID_PAIR_HASHED != 0, true>(table, lp, in, f);   // this level's gathers: in flight first
    uint32_t v[NT][W];
    const uint32_t *slab = reinterpret_cast<const uint32_t *>(ws);
#pragma unroll
    for (uint32_t j = 0; j < NT; j++) {
        if (have) load_words_nt<W>(slab + ((size_t)(t0 + j) * B + b) * W, v[j]);   // (the own level's slot too: stale words, replaced below)
        else {
#pragma unroll
            for (uint32_t i = 0; i < W; i++) v[j][i] = 0u;
        }
    }
    T res[C], dummy[1];
    if (oob) {
#pragma unroll
        for (uint32_t ch = 0; ch < C; ch++) res[ch] = from_f<T>(0.0f);
    } else {
        blend_level<T, D, C, false>(f, 0.0f, res, dummy);
    }
    uint32_t own_words[W];
    __builtin_memcpy(own_words, res, sizeof(T) * C);
    if (own >= NT && have) store_row_nt<T, C>(ws + ((size_t)level * B + b) * C, res);   // a later launch's duty
    constexpr uint32_t QS = SEG / 4u;                         // 16-byte quads per sample
    const uint32_t n_all = B - s0 < 256u ? B - s0 : 256u;
    const size_t row_words = (size_t)L * W;
#pragma unroll
    for (uint32_t pass = 0; pass < 256u / SP; pass++) {
        if (pass) __syncthreads();                            // the previous pass's rows have left the tile
        if (threadIdx.x / SP == pass) {
            uint32_t *mine = tile + (threadIdx.x - pass * SP) * STRIDE;
#pragma unroll
            for (uint32_t j = 0; j < NT; j++) {
#pragma unroll
                for (uint32_t i = 0; i < W; i++) mine[j * W + i] = v[j][i];
            }
            if (own < NT) {                                       // this launch's own level sits in its duty: from registers
#pragma unroll
                for (uint32_t i = 0; i < W; i++) mine[own * W + i] = own_words[i];
            }
        }
        __syncthreads();
        const uint32_t first = pass * SP;
        const uint32_t n = n_all > first ? (n_all - first < SP ? n_all - first : SP) : 0u;
        for (uint32_t q = threadIdx.x; q < n * QS; q += 256u) {
            const uint32_t s = q / QS, k = q - s * QS;
            const u32x4 val = *reinterpret_cast<const u32x4 *>(tile + s * STRIDE + 4u * k);
            __builtin_nontemporal_store(val, reinterpret_cast<u32x4 *>(rows + (size_t)(s0 + first + s) * row_words + (size_t)t0 * W + 4u * k));
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Backward to the table (gridencoder.cu:247-339): one (sample, level, channel pair) per lane.
__device__ __forceinline__ void atomic_add_row(float *p, const float (&v)[1]) { atomicAdd(p, v[0]); }
__device__ __forceinline__ void atomic_add_row(float *p, const float (&v)[2]) { atomicAdd(p, v[0]); atomicAdd(p + 1, v[1]); }
__device__ __forceinline__ void atomic_add_row(__half *p, const float (&v)[1]) {
    // C == 1 in half: emulate with a 32-bit CAS on the containing word (the reference never takes this path fast either)
    uint32_t *w = reinterpret_cast<uint32_t *>(reinterpret_cast<uintptr_t>(p) & ~uintptr_t(3));
    const bool hi = reinterpret_cast<uintptr_t>(p) & 2;
    uint32_t old = *w, assumed;
    do {
        assumed = old;
        const uint16_t cur = hi ? (uint16_t)(assumed >> 16) : (uint16_t)(assumed & 0xffffu);
        const __half sum = __float2half_rn(__half2float(__ushort_as_half(cur)) + __half2float(__float2half_rn(v[0])));
        const uint32_t nw = hi ? ((assumed & 0x0000ffffu) | ((uint32_t)__half_as_ushort(sum) << 16))
                               : ((assumed & 0xffff0000u) | (uint32_t)__half_as_ushort(sum));
        old = atomicCAS(w, assumed, nw);
    } while (old != assumed);
}
__device__ __forceinline__ void atomic_add_row(__half *p, const float (&v)[2]) {
    // packed half2 atomic (gridencoder.cu:324-330) -> global_atomic_pk_add_f16
    typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
    half2_t x;
    x[0] = (_Float16)v[0];
    x[1] = (_Float16)v[1];
    typedef __attribute__((address_space(1))) half2_t *gptr_t;
    __builtin_amdgcn_global_atomic_fadd_v2f16((gptr_t)(p), x);
}

// Scatter-add with wave-level pre-reduction.  Samples arrive ray-ordered, so neighbouring lanes often hit the same
// table row (always on the coarse levels; on every level of the ambient grid, whose coordinates cluster around 0) and
// plain atomics then serialise on one address in L2.  Lanes with equal keys in consecutive lanes form a run; a
// segmented inclusive scan (6 shuffle steps, the run heads taken from one ballot) sums each run and only its last lane
// issues the atomic.  Waves with few repeats skip the scan.  `key` must be ~0u on lanes that contribute nothing.
template <typename T, uint32_t N_C>
__device__ __forceinline__ void scatter_add_runs(T *__restrict__ table, uint32_t key, float (&v)[N_C]) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t prev = (uint32_t)__shfl_up((int)key, 1, 64);
    const bool head = lane == 0 || key != prev;
    const unsigned long long heads = __ballot(head);
    const bool valid = key != ~0u;
    if (__popcll(heads) > 40) {  // mostly distinct rows: the scan would not pay
        if (valid) atomic_add_row(table + key, v);
        return;
    }
    // first lane of my run: highest head bit at or below my lane
    const unsigned long long below = heads & ((2ull << lane) - 1ull);
    const uint32_t start = 63u - (uint32_t)__clzll(below);
#pragma unroll
    for (uint32_t off = 1; off < 64; off <<= 1) {
        float up[N_C];
#pragma unroll
        for (uint32_t c = 0; c < N_C; c++) up[c] = __shfl_up(v[c], off, 64);
        if (lane >= start + off) {
#pragma unroll
            for (uint32_t c = 0; c < N_C; c++) v[c] += up[c];
        }
    }
    const bool tail = lane == 63u || ((heads >> (lane + 1)) & 1ull);
    if (tail && valid) atomic_add_row(table + key, v);
}

template <typename T, uint32_t D, uint32_t C, uint32_t N_C, int LAYOUT>
__global__ void __launch_bounds__(256)
k_grid_bwd_table(const T *__restrict__ grad, const float *__restrict__ inputs, const int32_t *__restrict__ offsets,
                 T *__restrict__ grad_grid, uint32_t B, uint32_t L, LevelConsts lc, uint32_t gridtype,
                 bool align_corners, uint32_t interp) {
    const uint32_t tid = blockIdx.x * 256 + threadIdx.x;
    const uint32_t b_raw = tid * N_C / C;
    const bool in_range = b_raw < B;
    const uint32_t b = in_range ? b_raw : 0u;   // lanes past the end stay in the wave (shuffles below), contributing nothing
    const uint32_t level = blockIdx.y;
    const uint32_t ch = tid * N_C - b_raw * C;

    float in[D];
    const bool live = !load_input<D>(inputs, b, in) && in_range;  // grad table is zero-initialised (gridencoder.cu:275-280)

    const uint32_t off = (uint32_t)offsets[level];
    const uint32_t hashmap_size = (uint32_t)offsets[level + 1] - off;
    const float scale = lc.scale[level];
    const uint32_t resolution = lc.resolution[level];

    float pos[D], pos_deriv[D];
    uint32_t pos_grid[D];
    lattice_pos<D>(in, scale, align_corners, interp, pos, pos_deriv, pos_grid);

    float grad_cur[N_C];
#pragma unroll
    for (uint32_t c = 0; c < N_C; c++) grad_cur[c] = 0.0f;
    if (live) {
        const T *g = (LAYOUT == RN_LAYOUT_LBC) ? grad + ((size_t)level * B + b) * C + ch : grad + ((size_t)b * L + level) * C + ch;
        T gc[N_C];
        load_row<T, N_C>(g, gc);
#pragma unroll
        for (uint32_t c = 0; c < N_C; c++) grad_cur[c] = to_f<T>(gc[c]);
    }

    T *gg = grad_grid + (size_t)off * C;
#pragma unroll
    for (uint32_t idx = 0; idx < (1u << D); idx++) {
        float w = 1;
        uint32_t pgl[D];
#pragma unroll
        for (uint32_t d = 0; d < D; d++) {
            const bool hi = (idx >> d) & 1u;
            w *= hi ? pos[d] : 1 - pos[d];
            pgl[d] = pos_grid[d] + (hi ? 1u : 0u);
        }
        const uint32_t row = live ? grid_row<D>(gridtype, align_corners, hashmap_size, resolution, pgl) : 0u;
        float v[N_C];
#pragma unroll
        for (uint32_t c = 0; c < N_C; c++) v[c] = w * grad_cur[c];
        scatter_add_runs<T, N_C>(gg, live ? row * C + ch : ~0u, v);
    }
}

// Wave-level pre-reduction for the LDS merge below: lanes hold (key, v); consecutive lanes with equal keys form a run
// (ray-ordered samples sit in the same coarse cell for many steps, so on the coarse levels a run is most of a ray).  A
// segmented inclusive scan sums each run into its last lane, which alone goes on to the LDS table: without it the coarse
// levels spend their time in same-address LDS atomics (thousands of inserts into a few dozen slots: 600 us per training
// step on the xyz grid, measured).  Returns true on the lanes that must insert.  Waves with few repeats skip the scan.
template <uint32_t N_C>
__device__ __forceinline__ bool merge_runs(uint32_t key, float (&v)[N_C]) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t prev = (uint32_t)__shfl_up((int)key, 1, 64);
    const bool head = lane == 0 || key != prev;
    const unsigned long long heads = __ballot(head);
    if (__popcll(heads) > 40) return key != ~0u;
    const unsigned long long below = heads & ((2ull << lane) - 1ull);
    const uint32_t start = 63u - (uint32_t)__clzll(below);
#pragma unroll
    for (uint32_t off = 1; off < 64; off <<= 1) {
        float up[N_C];
#pragma unroll
        for (uint32_t c = 0; c < N_C; c++) up[c] = __shfl_up(v[c], off, 64);
        if (lane >= start + off) {
#pragma unroll
            for (uint32_t c = 0; c < N_C; c++) v[c] += up[c];
        }
    }
    const bool tail = lane == 63u || ((heads >> (lane + 1)) & 1ull);
    return tail && key != ~0u;
}

// Scatter-add with block-level pre-reduction (C <= 2, D <= 3: one lane per sample and level).  The 256 samples of a
// block touch 256 * 2^D rows of one level; ray-ordered samples share most of them on the coarse levels, and the
// corners of neighbouring samples coincide, so the block first sums them in an LDS hash table (open addressing,
// load factor <= 0.5, ds_cmpst + ds_add_f32) and then issues ONE global atomic per distinct row.  Device-scope
// float atomics on this part resolve behind the per-XCD L2s, so their count -- not their bytes -- is the cost.
template <typename T, uint32_t D, uint32_t C, int LAYOUT>
__global__ void __launch_bounds__(256)
k_grid_bwd_table_merge(const T *__restrict__ grad, const float *__restrict__ inputs, const int32_t *__restrict__ offsets,
                       T *__restrict__ grad_grid, uint32_t B, uint32_t L, LevelConsts lc, uint32_t gridtype,
                       bool align_corners, uint32_t interp) {
    constexpr uint32_t kSlots = 256u * (1u << D) * 2u;
    constexpr uint32_t kEmpty = 0xffffffffu;
    __shared__ uint32_t keys[kSlots];
    __shared__ float vals[kSlots * C];
    for (uint32_t i = threadIdx.x; i < kSlots; i += 256) keys[i] = kEmpty;
    for (uint32_t i = threadIdx.x; i < kSlots * C; i += 256) vals[i] = 0.0f;
    __syncthreads();

    const uint32_t b = blockIdx.x * 256 + threadIdx.x;
    const uint32_t level = blockIdx.y;
    const uint32_t off = (uint32_t)offsets[level];
    float in[D];
    const bool live = b < B && !load_input<D>(inputs, b < B ? b : 0u, in);       // every lane stays for the wave shuffles
    {
        const uint32_t hashmap_size = (uint32_t)offsets[level + 1] - off;
        const float scale = lc.scale[level];
        const uint32_t resolution = lc.resolution[level];
        float pos[D], pos_deriv[D];
        uint32_t pos_grid[D];
        float g0[C];
#pragma unroll
        for (uint32_t c = 0; c < C; c++) g0[c] = 0.0f;
        if (live) {
            lattice_pos<D>(in, scale, align_corners, interp, pos, pos_deriv, pos_grid);
            const T *g = (LAYOUT == RN_LAYOUT_LBC) ? grad + ((size_t)level * B + b) * C : grad + ((size_t)b * L + level) * C;
            T gc[C];
            load_row<T, C>(g, gc);
#pragma unroll
            for (uint32_t c = 0; c < C; c++) g0[c] = to_f<T>(gc[c]);
        }
#pragma unroll
        for (uint32_t idx = 0; idx < (1u << D); idx++) {
            float w = 1;
            uint32_t pgl[D];
            uint32_t row = ~0u;
            if (live) {
#pragma unroll
                for (uint32_t d = 0; d < D; d++) {
                    const bool hi = (idx >> d) & 1u;
                    w *= hi ? pos[d] : 1 - pos[d];
                    pgl[d] = pos_grid[d] + (hi ? 1u : 0u);
                }
                row = grid_row<D>(gridtype, align_corners, hashmap_size, resolution, pgl);
            }
            float v[C];
#pragma unroll
            for (uint32_t c = 0; c < C; c++) v[c] = live ? w * g0[c] : 0.0f;
            if (merge_runs<C>(row, v)) {
                uint32_t slot = (row * 2654435761u) & (kSlots - 1u);
                while (true) {
                    const uint32_t prev = atomicCAS(&keys[slot], kEmpty, row);
                    if (prev == kEmpty || prev == row) break;
                    slot = (slot + 1u) & (kSlots - 1u);
                }
#pragma unroll
                for (uint32_t c = 0; c < C; c++) atomicAdd(&vals[slot * C + c], v[c]);
            }
        }
    }
    __syncthreads();
    T *gg = grad_grid + (size_t)off * C;
    if constexpr (C == 2 && sizeof(T) == 4) {
        // Two lanes per slot, one channel each: the two float atomics of a row leave in ONE wave-instruction from adjacent
        // lanes and share their 64-byte request (global float atomics are executed at the memory side, one request per
        // touched 64 B per instruction: a scattered instruction costs its lane count, MI355X_MICROARCH.md "Global float
        // atomics") -- half the requests of one-row-per-lane with two instructions.
        for (uint32_t i = threadIdx.x; i < kSlots * 2u; i += 256) {
            const uint32_t slot = i >> 1, c = i & 1u;
            const uint32_t row = keys[slot];
            if (row != kEmpty) atomicAdd(reinterpret_cast<float *>(gg) + (size_t)row * 2u + c, vals[slot * 2u + c]);
        }
    } else {
        for (uint32_t i = threadIdx.x; i < kSlots; i += 256) {
            const uint32_t row = keys[i];
            if (row != kEmpty) {
                float v[C];
#pragma unroll
                for (uint32_t c = 0; c < C; c++) v[c] = vals[i * C + c];
                atomic_add_row(gg + (size_t)row * C, v);
            }
        }
    }
}

// Backward to the inputs (gridencoder.cu:342-368): grad_inputs[b,d] = sum_{l,ch} grad * dy_dx.
template <typename T, uint32_t D, uint32_t C, int LAYOUT>
__global__ void __launch_bounds__(256)
k_grid_bwd_input(const T *__restrict__ grad, const T *__restrict__ dy_dx, T *__restrict__ grad_inputs, uint32_t B,
                 uint32_t L) {
    const uint32_t t = blockIdx.x * 256 + threadIdx.x;
    if (t >= B * D) return;
    const uint32_t b = t / D, d = t - b * D;
    const T *dd = dy_dx + (size_t)b * L * D * C;
    T result = from_f<T>(0.0f);
    for (uint32_t l = 0; l < L; l++) {
#pragma unroll
        for (uint32_t ch = 0; ch < C; ch++) {
            const T gv = (LAYOUT == RN_LAYOUT_LBC) ? grad[((size_t)l * B + b) * C + ch] : grad[((size_t)b * L + l) * C + ch];
            const T prod = from_f<T>(to_f<T>(gv) * to_f<T>(dd[(size_t)l * D * C + d * C + ch]));
            result = from_f<T>(to_f<T>(result) + to_f<T>(prod));
        }
    }
    grad_inputs[t] = result;
}

// Total-variation gradient (gridencoder.cu:505-609), float32.
template <uint32_t D, uint32_t C>
__global__ void __launch_bounds__(256)
k_grad_tv(const float *__restrict__ inputs, const float *__restrict__ table, float *__restrict__ grad,
          const int32_t *__restrict__ offsets, float weight, uint32_t B, LevelConsts lc, uint32_t gridtype,
          bool align_corners) {
    const uint32_t b = blockIdx.x * 256 + threadIdx.x;
    if (b >= B) return;
    const uint32_t level = blockIdx.y;
    float in[D];
    if (load_input<D>(inputs, b, in)) return;

    const uint32_t off = (uint32_t)offsets[level];
    const uint32_t hashmap_size = (uint32_t)offsets[level + 1] - off;
    const float scale = lc.scale[level];
    const uint32_t resolution = lc.resolution[level];
    const float *grid = table + (size_t)off * C;
    float *gr = grad + (size_t)off * C;

    uint32_t pos_grid[D];
#pragma unroll
    for (uint32_t d = 0; d < D; d++) pos_grid[d] = (uint32_t)floorf(in[d] * scale + (align_corners ? 0.0f : 0.5f));

    float results[C], idelta[C], center[C];
#pragma unroll
    for (uint32_t ch = 0; ch < C; ch++) { results[ch] = 0; idelta[ch] = 0; }
    const uint32_t index = grid_row<D>(gridtype, align_corners, hashmap_size, resolution, pos_grid) * C;
    load_row<float, C>(grid + index, center);
    const float w = weight / (float)(2 * D);
#pragma unroll
    for (uint32_t d = 0; d < D; d++) {
        const uint32_t cur_d = pos_grid[d];
        if (cur_d < resolution) {
            pos_grid[d] = cur_d + 1;
            float nb[C];
            load_row<float, C>(grid + grid_row<D>(gridtype, align_corners, hashmap_size, resolution, pos_grid) * C, nb);
#pragma unroll
            for (uint32_t ch = 0; ch < C; ch++) { const float gv = center[ch] - nb[ch]; results[ch] += gv; idelta[ch] += gv * gv; }
        }
        if (cur_d > 0) {
            pos_grid[d] = cur_d - 1;
            float nb[C];
            load_row<float, C>(grid + grid_row<D>(gridtype, align_corners, hashmap_size, resolution, pos_grid) * C, nb);
#pragma unroll
            for (uint32_t ch = 0; ch < C; ch++) { const float gv = center[ch] - nb[ch]; results[ch] += gv; idelta[ch] += gv * gv; }
        }
        pos_grid[d] = cur_d;
    }
#pragma unroll
    for (uint32_t ch = 0; ch < C; ch++) atomicAdd(&gr[index + ch], w * results[ch] * (1.0f / sqrtf(idelta[ch] + 1e-9f)));
}

// ------------------------------------------------------------------------------------------------
// host-side dispatch
struct FwdArgs {
    const float *inputs; const void *table; const int32_t *offsets; void *outputs; uint32_t B, L; LevelConsts lc;
    void *dy_dx; uint32_t gridtype; bool align_corners; uint32_t interp; int layout; hipStream_t stream;
};

template <typename T, uint32_t D, uint32_t C>
static void launch_fwd(const FwdArgs &a) {
    const T *table = static_cast<const T *>(a.table);
    T *out = static_cast<T *>(a.outputs);
    T *dy = static_cast<T *>(a.dy_dx);
    const dim3 block(256);
#define RN_FWD(KERNEL, GRID, DY, LAY, ...)                                                                         \
    hipLaunchKernelGGL((KERNEL<T, D, C, DY, LAY>), GRID, block, 0, a.stream, a.inputs, table, a.offsets, out, a.B, \
                       a.L, a.lc, dy, a.gridtype, a.align_corners, a.interp, ##__VA_ARGS__)
    if (a.layout == RN_LAYOUT_LBC) {
        const dim3 grid(div_up(a.B, 256), a.L);
        if (dy) RN_FWD(k_grid_fwd_level, grid, true, RN_LAYOUT_LBC, 0u, ~0u, InXform{0, 1, 0}); else RN_FWD(k_grid_fwd_level, grid, false, RN_LAYOUT_LBC, 0u, ~0u, InXform{0, 1, 0});
    } else if (a.layout == RN_LAYOUT_BLC_LEVELMAJOR) {
        const dim3 grid(div_up(a.B, 256), a.L);
        if (dy) RN_FWD(k_grid_fwd_level, grid, true, RN_LAYOUT_BLC, 0u, ~0u, InXform{0, 1, 0}); else RN_FWD(k_grid_fwd_level, grid, false, RN_LAYOUT_BLC, 0u, ~0u, InXform{0, 1, 0});
    } else {
        const dim3 grid(div_up(a.B, 256));
        if (dy) RN_FWD(k_grid_fwd_sample, grid, true, RN_LAYOUT_BLC); else RN_FWD(k_grid_fwd_sample, grid, false, RN_LAYOUT_BLC);
    }
#undef RN_FWD
}

template <typename T, uint32_t D>
static int dispatch_fwd_c(uint32_t C, const FwdArgs &a) {
    switch (C) {
        case 1: launch_fwd<T, D, 1>(a); return RN_OK;
        case 2: launch_fwd<T, D, 2>(a); return RN_OK;
        case 4: launch_fwd<T, D, 4>(a); return RN_OK;
        case 8: launch_fwd<T, D, 8>(a); return RN_OK;
    }
    set_error("GridEncoding: C must be 1, 2, 4, or 8.");  // gridencoder.cu:380
    return RN_ERR_INVALID_ARG;
}
template <typename T>
static int dispatch_fwd_d(uint32_t D, uint32_t C, const FwdArgs &a) {
    switch (D) {
        case 2: return dispatch_fwd_c<T, 2>(C, a);
        case 3: return dispatch_fwd_c<T, 3>(C, a);
        case 4: return dispatch_fwd_c<T, 4>(C, a);
        case 5: return dispatch_fwd_c<T, 5>(C, a);
    }
    set_error("GridEncoding: D must be 2, 3, 4 or 5.");  // gridencoder.cu:397
    return RN_ERR_INVALID_ARG;
}


// ------------------------------------------------------------------------------------------------
// Planned forward: coarse pass (LDS-staged dense levels) + level-major pass for the remaining levels, and for [B, L*C] the
// transposition of each chunk out of the caller's workspace.  Needs a host copy of the offsets (level sizes decide what is
// staged); D = 2 / 3, align_corners off, linear interpolation, no dy_dx -- everything else takes the per-level kernels.
constexpr size_t kCoarseTableBytes = 3u << 20;   // coarse levels gathered together: their tables share an XCD's 4 MB L2

template <uint32_t D>
static bool host_level_is_dense(uint32_t resolution, uint32_t size, uint32_t gridtype) {
    uint64_t stride = 1;                             // gridencoder.cu:66-84, as plan_level() evaluates it
    for (uint32_t d = 0; d < D; d++) {
        if (stride > size) return false;             // a dimension dropped (tiled) or the level hashed
        stride *= (uint64_t)resolution + 1u;
    }
    (void)gridtype;
    return stride <= size;
}

static int grid_cus() {
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
        if (cus <= 0) cus = 256;
    }
    return cus;
}

struct CoarseSplit { uint32_t n_lds, n_group, lds_bytes; };

template <typename T, uint32_t D, uint32_t C>
static CoarseSplit coarse_split(const int32_t *oh, uint32_t L, const LevelConsts &lc, uint32_t gridtype, uint32_t B) {
    CoarseSplit cs{0, 0, 0};
    static const char *no_lds = getenv("RN_GRID_NO_LDS");          // experiments only
    static const char *no_coarse = getenv("RN_GRID_NO_COARSE");
    if (no_coarse || B < 4096) return cs;
    while (!no_lds && cs.n_lds < L && cs.n_lds < kCoarseMaxLevels) {
        const uint32_t size = (uint32_t)(oh[cs.n_lds + 1] - oh[cs.n_lds]);
        const size_t bytes = (size_t)oh[cs.n_lds + 1] * C * sizeof(T);
        if (oh[0] != 0 || !host_level_is_dense<D>(lc.resolution[cs.n_lds], size, gridtype) || bytes > kCoarseLdsBudget) break;
        cs.lds_bytes = (uint32_t)bytes;
        cs.n_lds++;
    }
    cs.n_group = cs.n_lds;
    while (cs.n_group < L && cs.n_group - cs.n_lds < kCoarseGlobal && cs.n_group < kCoarseMaxLevels &&
           (size_t)oh[cs.n_group + 1] * C * sizeof(T) <= kCoarseTableBytes)
        cs.n_group++;
    if (cs.n_group < 2) cs = CoarseSplit{0, 0, 0};                  // a single level gains nothing over the per-level kernel
    return cs;
}

// [L, Bc, C] for samples of one chunk
template <typename T, uint32_t D, uint32_t C>
static void launch_planned_lbc(const float *inputs, const T *table, const int32_t *offsets, const int32_t *oh, T *out, uint32_t Bc,
                               uint32_t L, const LevelConsts &lc, uint32_t gridtype, const InXform &xf, hipStream_t s, uint32_t skip_last = 0,
                               uint32_t skip_level = ~0u) {
    const CoarseSplit cs = coarse_split<T, D, C>(oh, L, lc, gridtype, Bc);
    if (cs.n_group) {
        static bool attr_set = false;
        if (!attr_set) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_grid_fwd_coarse<T, D, C>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)kCoarseLdsBudget);
            attr_set = true;
        }
        uint32_t blocks = div_up(Bc, kCoarseThreads);
        const uint32_t cap = (uint32_t)grid_cus();       // one persistent workgroup per CU (its LDS holds the staged levels)
        if (blocks > cap) blocks = cap;
        hipLaunchKernelGGL((k_grid_fwd_coarse<T, D, C>), dim3(blocks), dim3(kCoarseThreads), cs.lds_bytes, s, inputs, table, offsets,
                           out, Bc, lc, gridtype, cs.n_lds, cs.n_group, cs.lds_bytes, xf);
    }
    // skip_last / skip_level: levels the caller launches itself (launch_planned_rows)
    uint32_t n_fine = L - skip_last - cs.n_group;
    if (skip_level >= cs.n_group && skip_level < L - skip_last) n_fine--;
    else skip_level = ~0u;
    if (cs.n_group < L - skip_last && n_fine) {
        const dim3 grid(div_up(Bc, 256), n_fine);
        hipLaunchKernelGGL((k_grid_fwd_level<T, D, C, false, RN_LAYOUT_LBC>), grid, dim3(256), 0, s, inputs, table, offsets, out, Bc, L,
                           lc, static_cast<T *>(nullptr), gridtype, false, 0u, cs.n_group, skip_level, xf);
    }
}

// [B, L*C] without a transposition pass (k_grid_fwd_level_rows): which launches carry a duty, and how many levels each.
// n_seg = 0: not this shape -> transposition kernel.  One segment (rows of <= 128 bytes: every L = 16, C = 2 grid): the duty
// goes to the FIRST fine level, launched last -- the cheapest gathers leave the most room for the streams (measured at
// B = 2^22, hash T = 2^19, same box: transposition pass 0.968 ms, duty on level 15: 0.926, on level 5: 0.884).  Rows wider
// than a line are split into 128-byte segments over the last n_seg levels, in level order.
struct RowsPlan { uint32_t n_seg, nt, own_level; };

template <typename T, uint32_t D, uint32_t C>
static RowsPlan rows_plan(const int32_t *oh, uint32_t L, const LevelConsts &lc, uint32_t gridtype, uint32_t Bc) {
    const char *off = getenv("RN_GRID_ROWS");                        // "0": keep the transposition pass (tests toggle it: read per call)
    static const char *segs = getenv("RN_GRID_ROWS_SEGS");           // experiments only
    static const char *own_env = getenv("RN_GRID_ROWS_OWN");         // experiments only: the level that carries a one-segment duty
    const RowsPlan none{0, 0, 0};
    if (off && off[0] == '0') return none;
    constexpr uint32_t W = (uint32_t)(sizeof(T) * C / 4);
    const uint32_t row_bytes = L * W * 4u;
    uint32_t n_seg = (row_bytes + 127u) / 128u;                      // a launch writes whole 128-byte lines (or the whole row)
    if (segs) n_seg = (uint32_t)atoi(segs);
    if (n_seg < 1u || L % n_seg) return none;
    const uint32_t nt = L / n_seg;
    if ((nt != 4u && nt != 8u && nt != 16u) || (nt * W) % 4u || nt * W > 32u) return none;
    const CoarseSplit cs = coarse_split<T, D, C>(oh, L, lc, gridtype, Bc);
    if (L - cs.n_group < n_seg) return none;                         // the duty levels must come after everything they move
    uint32_t own = n_seg == 1u ? cs.n_group : 0u;
    if (own_env && n_seg == 1u && (uint32_t)atoi(own_env) >= cs.n_group && (uint32_t)atoi(own_env) < L) own = (uint32_t)atoi(own_env);
    return RowsPlan{n_seg, nt, own};
}

template <typename T, uint32_t D, uint32_t C, uint32_t NT>
static void launch_rows_nt(const float *inputs, const T *table, const int32_t *offsets, T *ws, void *rows, uint32_t Bc, uint32_t L,
                           const LevelConsts &lc, uint32_t gridtype, uint32_t level, uint32_t t0, const InXform &xf, hipStream_t s) {
    const uint32_t own = (level >= t0 && level - t0 < NT) ? level - t0 : NT;
    hipLaunchKernelGGL((k_grid_fwd_level_rows<T, D, C, NT>), dim3(div_up(Bc, 256)), dim3(256), 0, s, inputs, table, offsets, ws,
                       static_cast<uint32_t *>(rows), Bc, L, lc, gridtype, level, t0, own, xf);
}

template <typename T, uint32_t D, uint32_t C>
static void launch_planned_rows(const RowsPlan &rp, const float *inputs, const T *table, const int32_t *offsets, const int32_t *oh, T *ws,
                                void *rows, uint32_t Bc, uint32_t L, const LevelConsts &lc, uint32_t gridtype, const InXform &xf,
                                hipStream_t s) {
    if (rp.n_seg == 1) launch_planned_lbc<T, D, C>(inputs, table, offsets, oh, ws, Bc, L, lc, gridtype, xf, s, 0, rp.own_level);
    else launch_planned_lbc<T, D, C>(inputs, table, offsets, oh, ws, Bc, L, lc, gridtype, xf, s, rp.n_seg);
    for (uint32_t i = 0; i < rp.n_seg; i++) {
        const uint32_t level = rp.n_seg == 1 ? rp.own_level : L - rp.n_seg + i, t0 = i * rp.nt;
        if (rp.nt == 4) launch_rows_nt<T, D, C, 4>(inputs, table, offsets, ws, rows, Bc, L, lc, gridtype, level, t0, xf, s);
        else if (rp.nt == 8) launch_rows_nt<T, D, C, 8>(inputs, table, offsets, ws, rows, Bc, L, lc, gridtype, level, t0, xf, s);
        else if constexpr (16u * sizeof(T) * C / 4u <= 32u)
            launch_rows_nt<T, D, C, 16>(inputs, table, offsets, ws, rows, Bc, L, lc, gridtype, level, t0, xf, s);
    }
}

template <uint32_t WORDS>
static void launch_transpose(const void *src, void *dst, uint32_t Bc, uint32_t L, hipStream_t s) {
    const uint32_t stride_bytes = (L * WORDS + 4u) * 4u;
    uint32_t tile = (72u * 1024u) / stride_bytes;                       // two workgroups per CU
    tile = tile >= 512u ? 512u : (tile >= 64u ? (tile & ~63u) : (tile >= 4u ? (tile & ~3u) : 4u));
    const dim3 grid(div_up(Bc, tile)), block(kTrThreads);
    const uint32_t *sp = static_cast<const uint32_t *>(src);
    uint32_t *dp = static_cast<uint32_t *>(dst);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_grid_lbc_to_blc<WORDS, 16>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_grid_lbc_to_blc<WORDS, 32>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
        attr_set = true;
    }
    if (L <= 16) hipLaunchKernelGGL((k_grid_lbc_to_blc<WORDS, 16>), grid, block, tile * stride_bytes, s, sp, dp, Bc, L, tile);
    else hipLaunchKernelGGL((k_grid_lbc_to_blc<WORDS, 32>), grid, block, tile * stride_bytes, s, sp, dp, Bc, L, tile);
}

struct PlannedArgs {
    const float *inputs; const void *table; const int32_t *offsets, *offsets_host; void *outputs; uint32_t B, L; LevelConsts lc;
    uint32_t gridtype; int layout; void *ws; size_t ws_bytes; hipStream_t stream; InXform xf;
};

static uint32_t planned_chunk_default() {
    static uint32_t chunk = 0;
    if (!chunk) {
        const char *e = getenv("RN_GRID_CHUNK_LOG2");             // experiments only
        const int lg = e ? atoi(e) : 22;   // measured: L2 residency of a level wants long level passes (2^18: -20 %)
        chunk = 1u << (lg < 12 ? 12 : (lg > 26 ? 26 : lg));
    }
    return chunk;
}

template <typename T, uint32_t D, uint32_t C>
static int run_planned(const PlannedArgs &a) {
    const T *table = static_cast<const T *>(a.table);
    if (a.layout == RN_LAYOUT_LBC) {
        launch_planned_lbc<T, D, C>(a.inputs, table, a.offsets, a.offsets_host, static_cast<T *>(a.outputs), a.B, a.L, a.lc, a.gridtype,
                                    a.xf, a.stream);
        return RN_OK;
    }
    constexpr uint32_t kRowBytes = sizeof(T) * C;
    static_assert(kRowBytes % 4 == 0, "planned [B,L*C] path: rows of whole 32-bit words");
    const size_t per_sample = (size_t)a.L * kRowBytes;
    size_t chunk = a.ws_bytes / per_sample;
    if (chunk > planned_chunk_default()) chunk = planned_chunk_default();
    if (chunk >= a.B) chunk = a.B;
    else chunk &= ~(size_t)255;                                   // whole transposition tiles
    RN_REQUIRE(chunk >= 256 || chunk == a.B, "grid_encode_forward_ws: workspace of %zu bytes is too small (%zu bytes per sample)", a.ws_bytes,
               per_sample);
    for (size_t b0 = 0; b0 < a.B; b0 += chunk) {
        const uint32_t Bc = (uint32_t)((a.B - b0) < chunk ? (a.B - b0) : chunk);
        void *rows = static_cast<char *>(a.outputs) + b0 * per_sample;
        const RowsPlan rp = rows_plan<T, D, C>(a.offsets_host, a.L, a.lc, a.gridtype, Bc);
        if (rp.n_seg) {
            launch_planned_rows<T, D, C>(rp, a.inputs + b0 * D, table, a.offsets, a.offsets_host, static_cast<T *>(a.ws), rows, Bc, a.L,
                                         a.lc, a.gridtype, a.xf, a.stream);
            continue;
        }
        launch_planned_lbc<T, D, C>(a.inputs + b0 * D, table, a.offsets, a.offsets_host, static_cast<T *>(a.ws), Bc, a.L, a.lc, a.gridtype,
                                    a.xf, a.stream);
        launch_transpose<kRowBytes / 4>(a.ws, rows, Bc, a.L, a.stream);
    }
    return RN_OK;
}

template <typename T>
static int dispatch_planned(uint32_t D, uint32_t C, const PlannedArgs &a) {
    if (D == 3 && C == 2) return run_planned<T, 3, 2>(a);
    if (D == 2 && C == 2) return run_planned<T, 2, 2>(a);
    if (D == 3 && C == 4) return run_planned<T, 3, 4>(a);
    if (D == 2 && C == 4) return run_planned<T, 2, 4>(a);
    return 1;   // not a planned shape: caller falls back
}

struct BwdArgs {
    const void *grad; const float *inputs; const int32_t *offsets; void *grad_table; uint32_t B, L; LevelConsts lc;
    const void *dy_dx; void *grad_inputs; uint32_t gridtype; bool align_corners; uint32_t interp; int layout;
    hipStream_t stream;
};

template <typename T, uint32_t D, uint32_t C>
static void launch_bwd(const BwdArgs &a) {
    constexpr uint32_t N_C = C < 2 ? C : 2;  // gridencoder.cu:404
    const T *grad = static_cast<const T *>(a.grad);
    T *gt = static_cast<T *>(a.grad_table);
    const dim3 block(256), grid(div_up(a.B * C / N_C, 256), a.L);
    if constexpr (C <= 2 && D <= 3) {
        if (a.layout == RN_LAYOUT_LBC)
            hipLaunchKernelGGL((k_grid_bwd_table_merge<T, D, C, RN_LAYOUT_LBC>), grid, block, 0, a.stream, grad, a.inputs,
                               a.offsets, gt, a.B, a.L, a.lc, a.gridtype, a.align_corners, a.interp);
        else
            hipLaunchKernelGGL((k_grid_bwd_table_merge<T, D, C, RN_LAYOUT_BLC>), grid, block, 0, a.stream, grad, a.inputs,
                               a.offsets, gt, a.B, a.L, a.lc, a.gridtype, a.align_corners, a.interp);
    } else if (a.layout == RN_LAYOUT_LBC)
        hipLaunchKernelGGL((k_grid_bwd_table<T, D, C, N_C, RN_LAYOUT_LBC>), grid, block, 0, a.stream, grad, a.inputs,
                           a.offsets, gt, a.B, a.L, a.lc, a.gridtype, a.align_corners, a.interp);
    else
        hipLaunchKernelGGL((k_grid_bwd_table<T, D, C, N_C, RN_LAYOUT_BLC>), grid, block, 0, a.stream, grad, a.inputs,
                           a.offsets, gt, a.B, a.L, a.lc, a.gridtype, a.align_corners, a.interp);
    if (a.dy_dx && a.grad_inputs) {
        const T *dy = static_cast<const T *>(a.dy_dx);
        T *gi = static_cast<T *>(a.grad_inputs);
        const dim3 g2(div_up(a.B * D, 256));
        if (a.layout == RN_LAYOUT_LBC)
            hipLaunchKernelGGL((k_grid_bwd_input<T, D, C, RN_LAYOUT_LBC>), g2, block, 0, a.stream, grad, dy, gi, a.B, a.L);
        else
            hipLaunchKernelGGL((k_grid_bwd_input<T, D, C, RN_LAYOUT_BLC>), g2, block, 0, a.stream, grad, dy, gi, a.B, a.L);
    }
}
template <typename T, uint32_t D>
static int dispatch_bwd_c(uint32_t C, const BwdArgs &a) {
    switch (C) {
        case 1: launch_bwd<T, D, 1>(a); return RN_OK;
        case 2: launch_bwd<T, D, 2>(a); return RN_OK;
        case 4: launch_bwd<T, D, 4>(a); return RN_OK;
        case 8: launch_bwd<T, D, 8>(a); return RN_OK;
    }
    set_error("GridEncoding: C must be 1, 2, 4, or 8.");
    return RN_ERR_INVALID_ARG;
}
template <typename T>
static int dispatch_bwd_d(uint32_t D, uint32_t C, const BwdArgs &a) {
    switch (D) {
        case 2: return dispatch_bwd_c<T, 2>(C, a);
        case 3: return dispatch_bwd_c<T, 3>(C, a);
        case 4: return dispatch_bwd_c<T, 4>(C, a);
        case 5: return dispatch_bwd_c<T, 5>(C, a);
    }
    set_error("GridEncoding: D must be 2, 3, 4 or 5.");
    return RN_ERR_INVALID_ARG;
}

template <uint32_t D>
static int dispatch_tv_c(uint32_t C, const float *inputs, const float *table, float *grad, const int32_t *offsets,
                         float weight, uint32_t B, uint32_t L, const LevelConsts &lc, uint32_t gridtype, bool ac,
                         hipStream_t s) {
    const dim3 block(256), grid(div_up(B, 256), L);
    switch (C) {
        case 1: hipLaunchKernelGGL((k_grad_tv<D, 1>), grid, block, 0, s, inputs, table, grad, offsets, weight, B, lc, gridtype, ac); return RN_OK;
        case 2: hipLaunchKernelGGL((k_grad_tv<D, 2>), grid, block, 0, s, inputs, table, grad, offsets, weight, B, lc, gridtype, ac); return RN_OK;
        case 4: hipLaunchKernelGGL((k_grad_tv<D, 4>), grid, block, 0, s, inputs, table, grad, offsets, weight, B, lc, gridtype, ac); return RN_OK;
        case 8: hipLaunchKernelGGL((k_grad_tv<D, 8>), grid, block, 0, s, inputs, table, grad, offsets, weight, B, lc, gridtype, ac); return RN_OK;
    }
    set_error("GridEncoding: C must be 1, 2, 4, or 8.");
    return RN_ERR_INVALID_ARG;
}

}  // namespace rn

using namespace rn;

extern "C" {

int rn_grid_encode_forward(const float *inputs, const void *embeddings, const int32_t *offsets, void *outputs,
                           uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H, void *dy_dx,
                           uint32_t gridtype, int align_corners, uint32_t interp, int dtype, int layout,
                           rn_stream_t stream) {
    if (B == 0) return RN_OK;
    RN_REQUIRE(inputs && embeddings && offsets && outputs, "grid_encode_forward: null pointer");
    RN_REQUIRE(L >= 1 && L <= kMaxLevels, "grid_encode_forward: L=%u out of range (1..%u)", L, kMaxLevels);
    RN_REQUIRE(dtype == RN_F32 || dtype == RN_F16, "grid_encode_forward: dtype must be RN_F32 or RN_F16");
    RN_REQUIRE(layout == RN_LAYOUT_LBC || layout == RN_LAYOUT_BLC || layout == RN_LAYOUT_BLC_LEVELMAJOR,
               "grid_encode_forward: bad layout");
    RN_REQUIRE(gridtype <= 1 && interp <= 1, "grid_encode_forward: bad gridtype / interpolation id");
    FwdArgs a{inputs, embeddings, offsets, outputs, B, L, make_level_consts(L, S, H), dy_dx, gridtype,
              align_corners != 0, interp, layout, as_stream(stream)};
    const int rc = (dtype == RN_F32) ? dispatch_fwd_d<float>(D, C, a) : dispatch_fwd_d<__half>(D, C, a);
    if (rc != RN_OK) return rc;
    return check_launch("grid_encode_forward");
}

size_t rn_grid_encode_forward_workspace(uint32_t B, uint32_t L, uint32_t C, int dtype) {
    const size_t per_sample = (size_t)L * C * (dtype == RN_F16 ? 2 : 4);
    const size_t chunk = B < planned_chunk_default() ? B : planned_chunk_default();
    return chunk * per_sample;
}

static int grid_forward_ws(const float *inputs, const void *embeddings, const int32_t *offsets, const int32_t *offsets_host,
                           void *outputs, uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H, void *dy_dx,
                           uint32_t gridtype, int align_corners, uint32_t interp, int dtype, int layout, void *workspace,
                           size_t workspace_bytes, const InXform &xf, rn_stream_t stream, const char *who) {
    if (B == 0) return RN_OK;
    RN_REQUIRE(inputs && embeddings && offsets && outputs, "%s: null pointer", who);
    RN_REQUIRE(L >= 1 && L <= kMaxLevels, "%s: L=%u out of range (1..%u)", who, L, kMaxLevels);
    RN_REQUIRE(dtype == RN_F32 || dtype == RN_F16, "%s: dtype must be RN_F32 or RN_F16", who);
    RN_REQUIRE(layout == RN_LAYOUT_LBC || layout == RN_LAYOUT_BLC, "%s: layout must be RN_LAYOUT_LBC or RN_LAYOUT_BLC", who);
    const bool shape_ok = offsets_host && !dy_dx && !align_corners && interp == 0 && (D == 2 || D == 3) && (C == 2 || C == 4) &&
                          ((uintptr_t)embeddings & 15u) == 0 && ((uintptr_t)outputs & 15u) == 0 &&
                          (layout == RN_LAYOUT_LBC || (workspace && ((uintptr_t)workspace & 15u) == 0));
    if (shape_ok) {
        PlannedArgs a{inputs, embeddings, offsets, offsets_host, outputs, B, L, make_level_consts(L, S, H), gridtype, layout, workspace,
                      workspace_bytes, as_stream(stream), xf};
        const int rc = (dtype == RN_F32) ? dispatch_planned<float>(D, C, a) : dispatch_planned<__half>(D, C, a);
        if (rc <= 0) return rc != RN_OK ? rc : check_launch(who);
    }
    RN_REQUIRE(!xf.on, "%s: only the planned shapes (offsets_host given, D in {2,3}, C in {2,4}, 16-byte aligned buffers); normalise the "
               "coordinates yourself and call rn_grid_encode_forward_ws", who);
    return rn_grid_encode_forward(inputs, embeddings, offsets, outputs, B, D, C, L, S, H, dy_dx, gridtype, align_corners, interp, dtype,
                                  layout, stream);
}

int rn_grid_encode_forward_ws(const float *inputs, const void *embeddings, const int32_t *offsets, const int32_t *offsets_host,
                              void *outputs, uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H, void *dy_dx,
                              uint32_t gridtype, int align_corners, uint32_t interp, int dtype, int layout, void *workspace,
                              size_t workspace_bytes, rn_stream_t stream) {
    return grid_forward_ws(inputs, embeddings, offsets, offsets_host, outputs, B, D, C, L, S, H, dy_dx, gridtype, align_corners, interp,
                           dtype, layout, workspace, workspace_bytes, InXform{0.0f, 1.0f, 0u}, stream, "grid_encode_forward_ws");
}

int rn_grid_encode_forward_bound(const float *inputs, float bound, const void *embeddings, const int32_t *offsets,
                                 const int32_t *offsets_host, void *outputs, uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S,
                                 uint32_t H, uint32_t gridtype, int dtype, int layout, void *workspace, size_t workspace_bytes,
                                 rn_stream_t stream) {
    RN_REQUIRE(bound > 0.0f, "grid_encode_forward_bound: bound must be positive");
    const float denom = 2.0f * bound;
    return grid_forward_ws(inputs, embeddings, offsets, offsets_host, outputs, B, D, C, L, S, H, nullptr, gridtype, 0, 0u, dtype, layout,
                           workspace, workspace_bytes, InXform{bound, 1.0f / denom, 1u}, stream, "grid_encode_forward_bound");
}

int rn_grid_encode_backward(const void *grad, const float *inputs, const void *embeddings, const int32_t *offsets,
                            void *grad_embeddings, uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S,
                            uint32_t H, const void *dy_dx, void *grad_inputs, uint32_t gridtype, int align_corners,
                            uint32_t interp, int dtype, int layout, rn_stream_t stream) {
    if (B == 0) return RN_OK;
    (void)embeddings;
    RN_REQUIRE(grad && inputs && offsets && grad_embeddings, "grid_encode_backward: null pointer");
    RN_REQUIRE(L >= 1 && L <= kMaxLevels, "grid_encode_backward: L=%u out of range (1..%u)", L, kMaxLevels);
    RN_REQUIRE(dtype == RN_F32 || dtype == RN_F16, "grid_encode_backward: dtype must be RN_F32 or RN_F16");
    RN_REQUIRE(layout == RN_LAYOUT_LBC || layout == RN_LAYOUT_BLC || layout == RN_LAYOUT_BLC_LEVELMAJOR,
               "grid_encode_backward: bad layout");
    if (layout == RN_LAYOUT_BLC_LEVELMAJOR) layout = RN_LAYOUT_BLC;
    RN_REQUIRE(gridtype <= 1 && interp <= 1, "grid_encode_backward: bad gridtype / interpolation id");
    BwdArgs a{grad, inputs, offsets, grad_embeddings, B, L, make_level_consts(L, S, H), dy_dx, grad_inputs, gridtype,
              align_corners != 0, interp, layout, as_stream(stream)};
    const int rc = (dtype == RN_F32) ? dispatch_bwd_d<float>(D, C, a) : dispatch_bwd_d<__half>(D, C, a);
    if (rc != RN_OK) return rc;
    return check_launch("grid_encode_backward");
}

int rn_grad_total_variation(const float *inputs, const float *embeddings, float *grad, const int32_t *offsets,
                            float weight, uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H,
                            uint32_t gridtype, int align_corners, rn_stream_t stream) {
    if (B == 0) return RN_OK;
    RN_REQUIRE(inputs && embeddings && grad && offsets, "grad_total_variation: null pointer");
    RN_REQUIRE(L >= 1 && L <= kMaxLevels, "grad_total_variation: L=%u out of range (1..%u)", L, kMaxLevels);
    const LevelConsts lc = make_level_consts(L, S, H);
    int rc;
    switch (D) {
        case 2: rc = dispatch_tv_c<2>(C, inputs, embeddings, grad, offsets, weight, B, L, lc, gridtype, align_corners != 0, as_stream(stream)); break;
        case 3: rc = dispatch_tv_c<3>(C, inputs, embeddings, grad, offsets, weight, B, L, lc, gridtype, align_corners != 0, as_stream(stream)); break;
        case 4: rc = dispatch_tv_c<4>(C, inputs, embeddings, grad, offsets, weight, B, L, lc, gridtype, align_corners != 0, as_stream(stream)); break;
        case 5: rc = dispatch_tv_c<5>(C, inputs, embeddings, grad, offsets, weight, B, L, lc, gridtype, align_corners != 0, as_stream(stream)); break;
        default: set_error("GridEncoding: D must be 2, 3, 4 or 5."); return RN_ERR_INVALID_ARG;
    }
    if (rc != RN_OK) return rc;
    return check_launch("grad_total_variation");
}

}  // extern "C"
