// rn_grid_dev.h -- device-side building blocks of the grid encoder, shared by the standalone
// operator (rn_grid.hip) and the fused per-sample network kernel (rn_fused.hip).  Both therefore
// produce bit-identical features for the same inputs.
#pragma once

#include "rn_common.h"

namespace rn {

constexpr uint32_t kMaxLevels = 32;

// Per-level constants computed on the HOST with libm (gridencoder.cu:138-139), passed by value.
struct LevelConsts {
    float scale[kMaxLevels];
    uint32_t resolution[kMaxLevels];
};

static inline LevelConsts make_level_consts(uint32_t L, float S, uint32_t H) {
    LevelConsts lc{};
    for (uint32_t l = 0; l < L; l++) {
        const float scale = exp2f((float)l * S) * (float)H - 1.0f;
        lc.scale[l] = scale;
        lc.resolution[l] = (uint32_t)ceilf(scale) + 1;
    }
    return lc;
}

// ---- scalar helpers ---------------------------------------------------------------------------
template <typename T> __device__ __forceinline__ float to_f(T v);
template <> __device__ __forceinline__ float to_f<float>(float v) { return v; }
template <> __device__ __forceinline__ float to_f<__half>(__half v) { return __half2float(v); }
template <typename T> __device__ __forceinline__ T from_f(float v);
template <> __device__ __forceinline__ float from_f<float>(float v) { return v; }
template <> __device__ __forceinline__ __half from_f<__half>(float v) { return __float2half_rn(v); }

// One aligned load / store of a C-wide feature row.
template <typename T, uint32_t C>
__device__ __forceinline__ void load_row(const T *p, T (&v)[C]) {
    constexpr uint32_t bytes = sizeof(T) * C;
    if constexpr (bytes == 2) { *reinterpret_cast<uint16_t *>(v) = *reinterpret_cast<const uint16_t *>(p); }
    else if constexpr (bytes == 4) { *reinterpret_cast<uint32_t *>(v) = *reinterpret_cast<const uint32_t *>(p); }
    else if constexpr (bytes == 8) { *reinterpret_cast<uint2 *>(v) = *reinterpret_cast<const uint2 *>(p); }
    else if constexpr (bytes == 16) { *reinterpret_cast<uint4 *>(v) = *reinterpret_cast<const uint4 *>(p); }
    else {
        static_assert(bytes == 32, "unsupported row width");
        reinterpret_cast<uint4 *>(v)[0] = reinterpret_cast<const uint4 *>(p)[0];
        reinterpret_cast<uint4 *>(v)[1] = reinterpret_cast<const uint4 *>(p)[1];
    }
}
// Streaming store of a feature row: the outputs are written once and never re-read by the kernel, so they should not
// displace table lines from L2 (a hashed level of the T = 2^19 table is exactly one XCD's L2).
template <typename T, uint32_t C>
__device__ __forceinline__ void store_row_nt(T *p, const T (&v)[C]) {
    constexpr uint32_t bytes = sizeof(T) * C;
    if constexpr (bytes % 4 == 0) {
        uint32_t w[bytes / 4];
        __builtin_memcpy(w, v, bytes);
#pragma unroll
        for (uint32_t i = 0; i < bytes / 4; i++) __builtin_nontemporal_store(w[i], reinterpret_cast<uint32_t *>(p) + i);
    } else {  // a single fp16 channel
        uint16_t h;
        __builtin_memcpy(&h, v, 2);
        __builtin_nontemporal_store(h, reinterpret_cast<uint16_t *>(p));
    }
}
template <typename T, uint32_t C>
__device__ __forceinline__ void store_row(T *p, const T (&v)[C]) {
    constexpr uint32_t bytes = sizeof(T) * C;
    if constexpr (bytes == 2) { *reinterpret_cast<uint16_t *>(p) = *reinterpret_cast<const uint16_t *>(v); }
    else if constexpr (bytes == 4) { *reinterpret_cast<uint32_t *>(p) = *reinterpret_cast<const uint32_t *>(v); }
    else if constexpr (bytes == 8) { *reinterpret_cast<uint2 *>(p) = *reinterpret_cast<const uint2 *>(v); }
    else if constexpr (bytes == 16) { *reinterpret_cast<uint4 *>(p) = *reinterpret_cast<const uint4 *>(v); }
    else {
        static_assert(bytes == 32, "unsupported row width");
        reinterpret_cast<uint4 *>(p)[0] = reinterpret_cast<const uint4 *>(v)[0];
        reinterpret_cast<uint4 *>(p)[1] = reinterpret_cast<const uint4 *>(v)[1];
    }
}

// A gathered row kept as raw 32-bit words until it is blended: the loaders below hand their result over in plain
// registers (no pointer punning on arrays), so a fetch in flight never has to live in scratch memory.
template <typename T, uint32_t C>
struct RowWords {
    static constexpr uint32_t bytes = sizeof(T) * C;
    static constexpr uint32_t W = (bytes + 3) / 4;
    static_assert(bytes == 2 || bytes == 4 || bytes == 8 || bytes == 16 || bytes == 32, "unsupported row width");
};

// The loaders address `table` (a wave-uniform pointer: kernel argument) + a 32-bit byte offset, which the compiler
// turns into the scalar-base form of global_load (one offset VGPR per load instead of a 64-bit address pair).
template <typename T, uint32_t C>
__device__ __forceinline__ void load_row_words(const T *table, uint32_t byte_off, uint32_t (&w)[RowWords<T, C>::W]) {
#ifdef RN_EXP_NO_LOADS  // experiment only (tools/gpu_phase_clock.sh): index arithmetic without the memory access
#pragma unroll
    for (uint32_t i = 0; i < RowWords<T, C>::W; i++) w[i] = byte_off + i;
    return;
#endif
    const char *p = reinterpret_cast<const char *>(table) + byte_off;
    constexpr uint32_t bytes = RowWords<T, C>::bytes;
    if constexpr (bytes == 2) {
        w[0] = *reinterpret_cast<const uint16_t *>(p);
    } else if constexpr (bytes == 4) {
        w[0] = *reinterpret_cast<const uint32_t *>(p);
    } else if constexpr (bytes == 8) {
        const uint2 v = *reinterpret_cast<const uint2 *>(p);
        w[0] = v.x, w[1] = v.y;
    } else {
#pragma unroll
        for (uint32_t q = 0; q < bytes / 16; q++) {
            const uint4 v = reinterpret_cast<const uint4 *>(p)[q];
            w[4 * q] = v.x, w[4 * q + 1] = v.y, w[4 * q + 2] = v.z, w[4 * q + 3] = v.w;
        }
    }
}

// Two adjacent rows with one load of 2*C scalars; the address is only guaranteed to be row-aligned
// (C * sizeof(T)), which global loads on gfx950 accept (dword alignment is all the hardware needs).
template <typename T, uint32_t C>
__device__ __forceinline__ void load_pair_words(const T *table, uint32_t byte_off, uint32_t (&a)[RowWords<T, C>::W],
                                                uint32_t (&b)[RowWords<T, C>::W]) {
#ifdef RN_EXP_NO_LOADS
#pragma unroll
    for (uint32_t i = 0; i < RowWords<T, C>::W; i++) a[i] = byte_off + i, b[i] = byte_off - i;
    return;
#endif
    const char *p = reinterpret_cast<const char *>(table) + byte_off;
    constexpr uint32_t bytes = RowWords<T, C>::bytes * 2;
    constexpr uint32_t words = bytes / 4;
    if constexpr (bytes == 4) {  // two fp16 scalars
        const uint32_t v = *reinterpret_cast<const uint32_t *>(p);
        a[0] = v & 0xffffu;
        b[0] = v >> 16;
    } else {
        typedef uint32_t vec_t __attribute__((ext_vector_type(words), aligned(4)));
        const vec_t v = *reinterpret_cast<const vec_t *>(p);
#pragma unroll
        for (uint32_t i = 0; i < words / 2; i++) {
            a[i] = v[i];
            b[i] = v[words / 2 + i];
        }
    }
}

template <typename T> __device__ __forceinline__ T word_to(uint32_t w, uint32_t ch);
template <> __device__ __forceinline__ float word_to<float>(uint32_t w, uint32_t) { return __uint_as_float(w); }
template <> __device__ __forceinline__ __half word_to<__half>(uint32_t w, uint32_t ch) {
    return __ushort_as_half((unsigned short)((ch & 1u) ? (w >> 16) : (w & 0xffffu)));
}
template <typename T, uint32_t C>
__device__ __forceinline__ void unpack_row(const uint32_t (&w)[RowWords<T, C>::W], T (&v)[C]) {
#pragma unroll
    for (uint32_t ch = 0; ch < C; ch++) v[ch] = word_to<T>(w[ch * sizeof(T) / 4], ch);
}

// gridencoder.cu:50-63
template <uint32_t D>
__device__ __forceinline__ uint32_t fast_hash(const uint32_t (&pos_grid)[D]) {
    constexpr uint32_t primes[7] = {1u, 2654435761u, 805459861u, 3674653429u, 2097192037u, 1434869437u, 2165219737u};
    uint32_t result = 0;
#pragma unroll
    for (uint32_t i = 0; i < D; ++i) result ^= pos_grid[i] * primes[i];
    return result;
}

__device__ __forceinline__ uint32_t fast_mod(uint32_t index, uint32_t hashmap_size) {
    // index % hashmap_size without the integer division on the two cases that occur in practice:
    // capped levels have a power-of-two row count, dense levels have index < row count.
    if ((hashmap_size & (hashmap_size - 1u)) == 0u) return index & (hashmap_size - 1u);
    if (index < hashmap_size) return index;
    return index % hashmap_size;
}

// Row index of a lattice corner (gridencoder.cu:66-84 without the "* C + ch").
template <uint32_t D>
__device__ __forceinline__ uint32_t grid_row(uint32_t gridtype, bool align_corners, uint32_t hashmap_size,
                                             uint32_t resolution, const uint32_t (&pos_grid)[D]) {
    uint32_t stride = 1, index = 0;
#pragma unroll
    for (uint32_t d = 0; d < D; d++) {
        if (stride <= hashmap_size) {
            index += pos_grid[d] * stride;
            stride *= align_corners ? resolution : (resolution + 1);
        }
    }
    if (gridtype == 0 && stride > hashmap_size) index = fast_hash<D>(pos_grid);
    return fast_mod(index, hashmap_size);
}

__device__ __forceinline__ float smoothstep(float v) { return v * v * (3.0f - 2.0f * v); }
__device__ __forceinline__ float smoothstep_derivative(float v) { return 6 * v * (1.0f - v); }

// Lattice position of one sample at one level (gridencoder.cu:146-158). Returns false when out of [0,1].
template <uint32_t D>
__device__ __forceinline__ void lattice_pos(const float (&in)[D], float scale, bool align_corners, uint32_t interp,
                                            float (&pos)[D], float (&pos_deriv)[D], uint32_t (&pos_grid)[D]) {
#pragma unroll
    for (uint32_t d = 0; d < D; d++) {
        pos[d] = in[d] * scale + (align_corners ? 0.0f : 0.5f);
        const float fl = floorf(pos[d]);
        pos_grid[d] = (uint32_t)fl;
        pos[d] -= (float)pos_grid[d];
        if (interp == 1) {
            pos_deriv[d] = smoothstep_derivative(pos[d]);
            pos[d] = smoothstep(pos[d]);
        } else {
            pos_deriv[d] = 1.0f;
        }
    }
}

// One level's gather in flight: the 2^D corner rows plus the interpolation fractions that will blend them.
// Splitting "issue" from "blend" lets a caller overlap the loads of level l+1 with the arithmetic of level l.
template <typename T, uint32_t D, uint32_t C>
struct LevelFetch {
    uint32_t rows[1 << D][RowWords<T, C>::W];
    float pos[D], pos_deriv[D];
    uint32_t swapped;  // bit (idx >> 1): rows[idx] and rows[idx + 1] arrived exchanged (hashed x-pairs, see issue_level)
};


// Compute the lattice position and the 2^D row indices of one sample at one level and ISSUE the row loads.
// Index arithmetic (gridencoder.cu:66-84) is hoisted per dimension: (p + 1) * m == p * m + m (mod 2^32), so a
// level costs D multiplies instead of D * 2^D.  On levels that are not hashed the two corners that differ only
// in x sit in adjacent rows (unless the modulo wraps): one load of 2*C scalars fetches both.  On hashed levels the x
// prime is 1 (gridencoder.cu:52), so for an even lattice x the two x-neighbours hash to rows r and r ^ 1 -- the two
// halves of one aligned 2-row block (every level size is a multiple of 8, so the modulo keeps them together): one
// aligned load fetches both, in table order; `swapped` tells blend_level which half is which.  Odd x: two loads
// (PAIR_HASHED = false keeps hashed levels to plain row loads: no divergent branch, for callers that are not load-bound).
template <typename T, uint32_t D, uint32_t C, bool PAIR_HASHED = true>
__device__ __forceinline__ void issue_level(const T *__restrict__ table, uint32_t level_row, const float (&in)[D],
                                            float scale, uint32_t resolution, uint32_t hashmap_size, uint32_t gridtype,
                                            bool align_corners, uint32_t interp, LevelFetch<T, D, C> &f) {
    constexpr uint32_t kRowBytes = sizeof(T) * C;   // tables stay far below 4 GB: byte offsets fit 32 bits
    constexpr uint32_t primes[7] = {1u, 2654435761u, 805459861u, 3674653429u, 2097192037u, 1434869437u, 2165219737u};
    uint32_t pos_grid[D];
    lattice_pos<D>(in, scale, align_corners, interp, f.pos, f.pos_deriv, pos_grid);

    // wave-uniform: which dimensions take part in the dense index, and is the level hashed
    uint32_t mult[D];
    uint32_t stride = 1;
#pragma unroll
    for (uint32_t d = 0; d < D; d++) {
        const bool act = stride <= hashmap_size;
        mult[d] = act ? stride : 0u;
        if (act) stride *= align_corners ? resolution : (resolution + 1);
    }
    const bool hashed = gridtype == 0 && stride > hashmap_size;
    if (hashed) {
#pragma unroll
        for (uint32_t d = 0; d < D; d++) mult[d] = primes[d];
    }
    uint32_t t0[D], t1[D];
#pragma unroll
    for (uint32_t d = 0; d < D; d++) {
        t0[d] = pos_grid[d] * mult[d];
        t1[d] = t0[d] + mult[d];
    }

    constexpr bool kCanPair = sizeof(T) * C * 2 <= 32;
    constexpr uint32_t P = 1u << (D - 1);  // x-pairs of corners
    f.swapped = 0;
    if (!hashed) {
        uint32_t r0[P], r1[P];
        bool adjacent = kCanPair;
#pragma unroll
        for (uint32_t q = 0; q < P; q++) {
            uint32_t base = 0;
#pragma unroll
            for (uint32_t d = 1; d < D; d++) base += ((q >> (d - 1)) & 1u) ? t1[d] : t0[d];
            r0[q] = fast_mod(base + t0[0], hashmap_size);
            r1[q] = fast_mod(base + t1[0], hashmap_size);
            adjacent = adjacent && r1[q] == r0[q] + 1;
        }
        // one wave-uniform decision for the level (the modulo wraps on a handful of samples only), so the loads of a
        // level sit in one basic block and the scheduler can batch them
        if (kCanPair && __builtin_amdgcn_ballot_w64(!adjacent) == 0ull) {
#pragma unroll
            for (uint32_t q = 0; q < P; q++)
                load_pair_words<T, C>(table, (level_row + r0[q]) * kRowBytes, f.rows[2 * q], f.rows[2 * q + 1]);
        } else {
#pragma unroll
            for (uint32_t q = 0; q < P; q++) {
                load_row_words<T, C>(table, (level_row + r0[q]) * kRowBytes, f.rows[2 * q]);
                load_row_words<T, C>(table, (level_row + r1[q]) * kRowBytes, f.rows[2 * q + 1]);
            }
        }
    } else {
#pragma unroll
        for (uint32_t idx = 0; idx < (1u << D); idx += 2) {
            uint32_t base = 0;
#pragma unroll
            for (uint32_t d = 1; d < D; d++) base ^= ((idx >> d) & 1u) ? t1[d] : t0[d];
            const uint32_t row0 = fast_mod(base ^ t0[0], hashmap_size);
            const uint32_t row1 = fast_mod(base ^ t1[0], hashmap_size);
            if (PAIR_HASHED && kCanPair && (row0 ^ row1) == 1u) {
                load_pair_words<T, C>(table, (level_row + (row0 & ~1u)) * kRowBytes, f.rows[idx], f.rows[idx + 1]);
                f.swapped |= (row0 & 1u) << (idx >> 1);
            } else {
                load_row_words<T, C>(table, (level_row + row0) * kRowBytes, f.rows[idx]);
                load_row_words<T, C>(table, (level_row + row1) * kRowBytes, f.rows[idx + 1]);
            }
        }
    }
}

// ---- planned levels ----------------------------------------------------------------------------------------------
// The fused network kernels walk 32 levels per sample tile and are bound by instruction issue, not by memory, as long
// as every level re-derives its wave-uniform facts (which dimensions are dense, hashed or not, power-of-two row count)
// and pays a chain of branches per corner for the modulo.  A LevelPlan holds those facts, computed once per workgroup:
//   row(corner) = combine(x, y * mult1, z * mult2) & mask           combine = + (dense / tiled)  or  ^ (hashed)
// `mask` is size - 1 for a power-of-two row count (every capped level of the shipped configurations) and all-ones for a
// dense level, whose index is always in range; any other level is "generic" and takes a real modulo.  Same integer
// arithmetic as grid_row() / issue_level(), so the rows -- and with blend_level() the features -- are bit-identical.
struct LevelPlan {
    float scale;
    uint32_t mult1, mult2;  // index multipliers of dimensions 1 and 2 (dimension 0: stride 1 and prime 1)
    uint32_t mask;
    uint32_t byte_base;     // first row of the level, in bytes
    uint32_t size;          // row count (generic levels)
    uint32_t mode;          // kPlanHashed | kPlanGeneric
    uint32_t pad_;
};
constexpr uint32_t kPlanHashed = 1u, kPlanGeneric = 2u;

template <uint32_t D>
__device__ __forceinline__ LevelPlan plan_level(float scale, uint32_t resolution, uint32_t level_row, uint32_t size,
                                                uint32_t gridtype, uint32_t row_bytes) {
    static_assert(D == 2 || D == 3, "planned levels: 2-D and 3-D grids");
    constexpr uint32_t primes[3] = {1u, 2654435761u, 805459861u};
    uint32_t mult[3] = {0u, 0u, 0u};
    uint32_t stride = 1;
    for (uint32_t d = 0; d < D; d++) {           // gridencoder.cu:66-84 with align_corners = false
        if (stride <= size) {
            mult[d] = stride;
            stride *= resolution + 1;
        }
    }
    const bool hashed = gridtype == 0 && stride > size;
    if (hashed)
        for (uint32_t d = 0; d < D; d++) mult[d] = primes[d];
    const bool pow2 = (size & (size - 1u)) == 0u;
    const bool in_range = !hashed && stride <= size;  // every dimension dense: index < (resolution + 1)^D <= size
    LevelPlan lp;
    lp.scale = scale;
    lp.mult1 = mult[1];
    lp.mult2 = mult[2];
    lp.mask = pow2 ? size - 1u : 0xffffffffu;
    lp.byte_base = level_row * row_bytes;
    lp.size = size;
    lp.mode = (hashed ? kPlanHashed : 0u) | ((pow2 || in_range) ? 0u : kPlanGeneric);
    lp.pad_ = 0;
    return lp;
}

// issue_level() for a planned level (align_corners = false, linear interpolation).  One wave-uniform branch picks the
// level kind; inside, the 2^D loads sit in straight-line code.
// UNIFORM: every lane of the wave passes the same plan (one scalar branch); otherwise the level kind is a per-lane branch
// (the 32-sample-tile kernel gives its two lane halves two different levels).
template <typename T, uint32_t D, uint32_t C, bool PAIR_HASHED = true, bool UNIFORM = true>
__device__ __forceinline__ void issue_planned(const T *__restrict__ table, const LevelPlan &lp, const float (&in)[D],
                                              LevelFetch<T, D, C> &f) {
    constexpr uint32_t kRowBytes = sizeof(T) * C;
    constexpr bool kCanPair = kRowBytes * 2 <= 32;
    constexpr uint32_t P = 1u << (D - 1);  // x-pairs of corners
    uint32_t pg[D];
    lattice_pos<D>(in, lp.scale, false, 0, f.pos, f.pos_deriv, pg);
    uint32_t t0[D], t1[D];
    t0[0] = pg[0];
    t1[0] = pg[0] + 1u;
    t0[1] = pg[1] * lp.mult1;
    t1[1] = t0[1] + lp.mult1;
    if constexpr (D == 3) {
        t0[2] = pg[2] * lp.mult2;
        t1[2] = t0[2] + lp.mult2;
    }
    f.swapped = 0;
    const uint32_t mode = UNIFORM ? __builtin_amdgcn_readfirstlane(lp.mode) : lp.mode;
    const uint32_t mask = lp.mask, base_bytes = lp.byte_base;
    if (mode == 0u) {
        uint32_t r0[P];
        bool wraps = false;
#pragma unroll
        for (uint32_t q = 0; q < P; q++) {
            uint32_t base = t0[0];
#pragma unroll
            for (uint32_t d = 1; d < D; d++) base += ((q >> (d - 1)) & 1u) ? t1[d] : t0[d];
            r0[q] = base & mask;
            wraps = wraps || r0[q] == mask;   // then the x + 1 neighbour is row 0, not r0 + 1
        }
        if (kCanPair && __builtin_amdgcn_ballot_w64(wraps) == 0ull) {
#pragma unroll
            for (uint32_t q = 0; q < P; q++)
                load_pair_words<T, C>(table, base_bytes + r0[q] * kRowBytes, f.rows[2 * q], f.rows[2 * q + 1]);
        } else {
#pragma unroll
            for (uint32_t q = 0; q < P; q++) {
                load_row_words<T, C>(table, base_bytes + r0[q] * kRowBytes, f.rows[2 * q]);
                load_row_words<T, C>(table, base_bytes + ((r0[q] + 1u) & mask) * kRowBytes, f.rows[2 * q + 1]);
            }
        }
    } else if (mode == kPlanHashed) {
        uint32_t h[P];
#pragma unroll
        for (uint32_t q = 0; q < P; q++) {
            h[q] = 0;
#pragma unroll
            for (uint32_t d = 1; d < D; d++) h[q] ^= ((q >> (d - 1)) & 1u) ? t1[d] : t0[d];
        }
        // x even: x ^ (x + 1) == 1, so the two x-neighbours of every pair are the halves of one aligned 2-row block
        if (PAIR_HASHED && kCanPair && !(pg[0] & 1u)) {
#pragma unroll
            for (uint32_t q = 0; q < P; q++) {
                const uint32_t row0 = (h[q] ^ t0[0]) & mask;
                load_pair_words<T, C>(table, base_bytes + (row0 & ~1u) * kRowBytes, f.rows[2 * q], f.rows[2 * q + 1]);
                f.swapped |= (row0 & 1u) << q;
            }
        } else {
#pragma unroll
            for (uint32_t q = 0; q < P; q++) {
                load_row_words<T, C>(table, base_bytes + ((h[q] ^ t0[0]) & mask) * kRowBytes, f.rows[2 * q]);
                load_row_words<T, C>(table, base_bytes + ((h[q] ^ t1[0]) & mask) * kRowBytes, f.rows[2 * q + 1]);
            }
        }
    } else {
        const uint32_t size = lp.size;
        const bool hashed = mode & kPlanHashed;
#pragma unroll
        for (uint32_t idx = 0; idx < (1u << D); idx++) {
            uint32_t index = (idx & 1u) ? t1[0] : t0[0];
#pragma unroll
            for (uint32_t d = 1; d < D; d++) {
                const uint32_t t = ((idx >> d) & 1u) ? t1[d] : t0[d];
                index = hashed ? index ^ t : index + t;
            }
            load_row_words<T, C>(table, base_bytes + (index % size) * kRowBytes, f.rows[idx]);
        }
    }
}

// issue_planned() for a DENSE level whose rows have been staged in LDS (`lds` = image of the table from its first byte, so
// lp.byte_base is the level's offset in it).  A dense level's index is always in range (mask all ones) and the two
// x-neighbours are adjacent rows, so each of the 2^(D-1) pairs is two row reads 8 bytes apart (the compiler pairs them into
// one ds_read2); no vector-memory instruction, no L1 / L2 request.  Same integer arithmetic as issue_planned(), hence the
// same rows and -- through blend_level() -- bit-identical features.
template <typename T, uint32_t D, uint32_t C>
__device__ __forceinline__ void issue_dense_lds(const unsigned char *lds, const LevelPlan &lp, const float (&in)[D],
                                                LevelFetch<T, D, C> &f) {
    constexpr uint32_t kRowBytes = sizeof(T) * C;
    constexpr uint32_t W = RowWords<T, C>::W;
    constexpr uint32_t P = 1u << (D - 1);
    uint32_t pg[D];
    lattice_pos<D>(in, lp.scale, false, 0, f.pos, f.pos_deriv, pg);
    uint32_t t0[D], t1[D];
    t0[0] = pg[0];
    t0[1] = pg[1] * lp.mult1;
    t1[1] = t0[1] + lp.mult1;
    if constexpr (D == 3) {
        t0[2] = pg[2] * lp.mult2;
        t1[2] = t0[2] + lp.mult2;
    }
    f.swapped = 0;
#pragma unroll
    for (uint32_t q = 0; q < P; q++) {
        uint32_t row = t0[0];
#pragma unroll
        for (uint32_t d = 1; d < D; d++) row += ((q >> (d - 1)) & 1u) ? t1[d] : t0[d];
        const unsigned char *p = lds + lp.byte_base + row * kRowBytes;
        if constexpr (kRowBytes % 4 == 0) {
#pragma unroll
            for (uint32_t i = 0; i < W; i++) {
                f.rows[2 * q][i] = reinterpret_cast<const uint32_t *>(p)[i];
                f.rows[2 * q + 1][i] = reinterpret_cast<const uint32_t *>(p + kRowBytes)[i];
            }
        } else {  // one fp16 channel
            f.rows[2 * q][0] = *reinterpret_cast<const uint16_t *>(p);
            f.rows[2 * q + 1][0] = *reinterpret_cast<const uint16_t *>(p + kRowBytes);
        }
    }
}

// Interpolated features (and optionally d/dx) from a completed LevelFetch.  Accumulation follows the
// reference's scalar_t semantics: results live in T and every += rounds to T (gridencoder.cu:163,186,234).
template <typename T, uint32_t D, uint32_t C, bool DYDX>
__device__ __forceinline__ void blend_level(const LevelFetch<T, D, C> &f, float scale, T (&results)[C],
                                            T (&grads)[DYDX ? D * C : 1]) {
    const float (&pos)[D] = f.pos;
    const float (&pos_deriv)[D] = f.pos_deriv;
    // undo the exchange of hashed x-pairs so the corner order (and with it the summation order) is the reference's;
    // the wave-uniform test keeps the selects off the levels that have none (every tiled / dense level)
    constexpr uint32_t W = RowWords<T, C>::W;
    T rows[1 << D][C];
    if (__builtin_amdgcn_ballot_w64(f.swapped != 0u) != 0ull) {
#pragma unroll
        for (uint32_t idx = 0; idx < (1u << D); idx += 2) {
            // bit-select (v_bfi_b32) instead of ?: -- a select between two array elements invites the compiler to
            // index the array dynamically, which would push the whole fetch into scratch
            const uint32_t m = 0u - ((f.swapped >> (idx >> 1)) & 1u);
            uint32_t lo[W], hi[W];
#pragma unroll
            for (uint32_t i = 0; i < W; i++) {
                const uint32_t a = f.rows[idx][i], b = f.rows[idx + 1][i];
                lo[i] = (a & ~m) | (b & m);
                hi[i] = (b & ~m) | (a & m);
            }
            unpack_row<T, C>(lo, rows[idx]);
            unpack_row<T, C>(hi, rows[idx + 1]);
        }
    } else {
#pragma unroll
        for (uint32_t idx = 0; idx < (1u << D); idx++) unpack_row<T, C>(f.rows[idx], rows[idx]);
    }
#pragma unroll
    for (uint32_t ch = 0; ch < C; ch++) results[ch] = from_f<T>(0.0f);
#pragma unroll
    for (uint32_t idx = 0; idx < (1u << D); idx++) {
        float w = 1;
#pragma unroll
        for (uint32_t d = 0; d < D; d++) w *= ((idx >> d) & 1u) ? pos[d] : 1 - pos[d];
#pragma unroll
        for (uint32_t ch = 0; ch < C; ch++) results[ch] = from_f<T>(to_f<T>(results[ch]) + w * to_f<T>(rows[idx][ch]));
    }

    if constexpr (DYDX) {
        // gridencoder.cu:200-243; corner `idx` of the (D-1)-face with bit gd cleared / set
#pragma unroll
        for (uint32_t gd = 0; gd < D; gd++) {
            T rg[C];
#pragma unroll
            for (uint32_t ch = 0; ch < C; ch++) rg[ch] = from_f<T>(0.0f);
#pragma unroll
            for (uint32_t idx = 0; idx < (1u << (D - 1)); idx++) {
                float w = scale;
                uint32_t corner = 0;  // index into rows[] of the "left" corner
#pragma unroll
                for (uint32_t nd = 0; nd < D - 1; nd++) {
                    const uint32_t d = (nd >= gd) ? (nd + 1) : nd;
                    const bool hi = (idx >> nd) & 1u;
                    w *= hi ? pos[d] : 1 - pos[d];
                    corner |= hi ? (1u << d) : 0u;
                }
                const uint32_t left = corner, right = corner | (1u << gd);
#pragma unroll
                for (uint32_t ch = 0; ch < C; ch++) {
                    const float diff = to_f<T>(from_f<T>(to_f<T>(rows[right][ch]) - to_f<T>(rows[left][ch])));
                    rg[ch] = from_f<T>(to_f<T>(rg[ch]) + w * diff * pos_deriv[gd]);
                }
            }
#pragma unroll
            for (uint32_t ch = 0; ch < C; ch++) grads[gd * C + ch] = rg[ch];
        }
    }
}

// issue + blend back to back (callers that do not pipeline)
template <typename T, uint32_t D, uint32_t C, bool DYDX>
__device__ __forceinline__ void encode_level(const T *__restrict__ table, uint32_t level_row, const float (&in)[D], float scale,
                                             uint32_t resolution, uint32_t hashmap_size, uint32_t gridtype,
                                             bool align_corners, uint32_t interp, T (&results)[C],
                                             T (&grads)[DYDX ? D * C : 1]) {
    LevelFetch<T, D, C> f;
    issue_level<T, D, C>(table, level_row, in, scale, resolution, hashmap_size, gridtype, align_corners, interp, f);
    blend_level<T, D, C, DYDX>(f, scale, results, grads);
}

}  // namespace rn
