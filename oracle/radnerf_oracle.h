/*
 * radnerf_oracle.h -- CPU ORACLE for the RAD-NeRF render hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This library is a plain-C restatement of the
 * reference's CUDA kernels and of the PyTorch arithmetic around them.  Only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it,
 * and there only as the checker.  The product path (rad-nerf_amd/) never links,
 * imports or falls back to anything in oracle/.
 *
 * PARITY STATUS: "parity unpinned" by the reference's own tests -- the
 * reference ships no tests, goldens or fixtures, and its kernels are CUDA-only
 * (no nvcc / NVIDIA GPU here), so they cannot be run.  The oracle is pinned
 * instead by (1) independent numpy/scipy derivations in tests/test_oracle_*.py
 * (bit-interleave morton, numpy.packbits, dense 6-neighbour max, scipy real SH,
 * numpy trilinear, big-int grid index, cumprod compositing, slab test) and
 * (2) golden vectors produced by running the reference's UNMODIFIED Python
 * control flow (nerf/network.py, nerf/renderer.py) on top of these functions
 * (tests/golden/make_golden.py).
 *
 * Every function cites the reference file:line it follows (paths relative to
 * the reference repository root).
 */
#ifndef RADNERF_ORACLE_H
#define RADNERF_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---------------------------------------------------------------- raymarching
 * raymarching/src/raymarching.cu */

/* :91-156 */
void orc_near_far_from_aabb(const float *rays_o, const float *rays_d, const float *aabb,
                            uint32_t N, float min_near, float *nears, float *fars);
/* :162-209 */
void orc_sph_from_ray(const float *rays_o, const float *rays_d, float radius, uint32_t N,
                      float *coords);
/* :56-71, 214-232 */
void orc_morton3D(const int32_t *coords, uint32_t N, int32_t *indices);
/* :73-81, 237-260 */
void orc_morton3D_invert(const int32_t *indices, uint32_t N, int32_t *coords);
/* :267-300  (N = C*H^3/8 output bytes) */
void orc_packbits(const float *grid, uint32_t N, float density_thresh, uint8_t *bitfield);
/* :304-341 */
void orc_morton3D_dilation(const float *grid, uint32_t C, uint32_t H, float *grid_dilation);
/* :352-528.  The two atomicAdd()s are executed in ray order (a legal
 * serialisation of the reference's unordered atomics). */
void orc_march_rays_train(const float *rays_o, const float *rays_d, const uint8_t *grid,
                          float bound, float dt_gamma, uint32_t max_steps, uint32_t N,
                          uint32_t C, uint32_t H, uint32_t M, const float *nears,
                          const float *fars, float *xyzs, float *dirs, float *deltas,
                          int32_t *rays, int32_t *counter, const float *noises);
/* :535-593 */
void orc_march_rays_train_backward(const float *grad_xyzs, const float *grad_dirs,
                                   const int32_t *rays, const float *deltas, uint32_t N,
                                   uint32_t M, float *grad_rays_o, float *grad_rays_d);
/* :603-698 */
void orc_composite_rays_train_forward(const float *sigmas, const float *rgbs,
                                      const float *ambient, const float *deltas,
                                      const int32_t *rays, uint32_t M, uint32_t N,
                                      float T_thresh, float *weights_sum, float *ambient_sum,
                                      float *depth, float *image);
/* :711-820 */
void orc_composite_rays_train_backward(const float *grad_weights_sum,
                                       const float *grad_ambient_sum, const float *grad_image,
                                       const float *sigmas, const float *rgbs,
                                       const float *ambient, const float *deltas,
                                       const int32_t *rays, const float *weights_sum,
                                       const float *ambient_sum, const float *image, uint32_t M,
                                       uint32_t N, float T_thresh, float *grad_sigmas,
                                       float *grad_rgbs, float *grad_ambient);
/* :827-939 */
void orc_march_rays(uint32_t n_alive, uint32_t n_step, const int32_t *rays_alive,
                    const float *rays_t, const float *rays_o, const float *rays_d, float bound,
                    float dt_gamma, uint32_t max_steps, uint32_t C, uint32_t H,
                    const uint8_t *grid, const float *nears, const float *fars, float *xyzs,
                    float *dirs, float *deltas, const float *noises);
/* :942-1038 */
void orc_composite_rays(uint32_t n_alive, uint32_t n_step, float T_thresh, int32_t *rays_alive,
                        float *rays_t, const float *sigmas, const float *rgbs,
                        const float *deltas, float *weights_sum, float *depth, float *image);

/* ---------------------------------------------------------------- gridencoder
 * gridencoder/src/gridencoder.cu.  `is_half` selects the table/output scalar:
 * 0 = float32, 1 = IEEE binary16 stored as uint16_t (c10::Half arithmetic
 * rules: every Half op computes in float and rounds back to half). */

/* :50-244, 372-399 */
void orc_grid_encode_forward(const float *inputs, const void *embeddings, const int32_t *offsets,
                             void *outputs, uint32_t B, uint32_t D, uint32_t C, uint32_t L,
                             float S, uint32_t H, void *dy_dx, uint32_t gridtype,
                             int align_corners, uint32_t interp, int is_half);
/* :247-368, 401-443.  Scatter-adds run in (level, b, channel) order. */
void orc_grid_encode_backward(const void *grad, const float *inputs, const void *embeddings,
                              const int32_t *offsets, void *grad_embeddings, uint32_t B,
                              uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H,
                              const void *dy_dx, void *grad_inputs, uint32_t gridtype,
                              int align_corners, uint32_t interp, int is_half);
/* :505-644 (fp32 only) */
void orc_grad_total_variation(const float *inputs, const float *embeddings, float *grad,
                              const int32_t *offsets, float weight, uint32_t B, uint32_t D,
                              uint32_t C, uint32_t L, float S, uint32_t H, uint32_t gridtype,
                              int align_corners);
/* The integer index of one lattice corner (gridencoder.cu:50-84); exported so
 * tests can pin it against a Python big-int restatement. */
uint32_t orc_grid_index(uint32_t D, uint32_t C, uint32_t gridtype, int align_corners,
                        uint32_t ch, uint32_t hashmap_size, uint32_t resolution,
                        const uint32_t *pos_grid);

/* ---------------------------------------------------------------- shencoder
 * shencoder/src/shencoder.cu:28-382 (fp32; degree C in 1..8) */
void orc_sh_encode_forward(const float *inputs, float *outputs, uint32_t B, uint32_t D,
                           uint32_t C, float *dy_dx);
void orc_sh_encode_backward(const float *grad, const float *inputs, uint32_t B, uint32_t D,
                            uint32_t C, const float *dy_dx, float *grad_inputs);

/* ---------------------------------------------------------------- freqencoder
 * freqencoder/src/freqencoder.cu:30-94 */
void orc_freq_encode_forward(const float *inputs, uint32_t B, uint32_t D, uint32_t deg,
                             uint32_t C, float *outputs);
void orc_freq_encode_backward(const float *grad, const float *outputs, uint32_t B, uint32_t D,
                              uint32_t deg, uint32_t C, float *grad_inputs);

/* ---------------------------------------------------------------- network
 * nerf/network.py (PyTorch arithmetic restated as plain fp32 loops) */

/* One multires grid as GridEncoder holds it (gridencoder/grid.py:96-161). */
typedef struct {
    const float *embeddings; /* [rows, C] */
    const int32_t *offsets;  /* [L+1] */
    uint32_t D, C, L, H;
    float S;                 /* log2(per_level_scale) as float (grid.py:39) */
    uint32_t gridtype;       /* 0 hash, 1 tiled */
    int align_corners;
    uint32_t interp;
} orc_grid_t;

/* Bias-free MLP (nerf/network.py:69-88): weights[l] is [out_l, in_l] row-major
 * exactly as nn.Linear stores it. */
typedef struct {
    uint32_t num_layers;
    uint32_t dim_in, dim_hidden, dim_out;
    const float *weights[4];
} orc_mlp_t;

typedef struct {
    orc_grid_t enc_xyz;     /* network.py:133 */
    orc_grid_t enc_ambient; /* network.py:134 */
    orc_mlp_t ambient_net;  /* :140  96 -> 64 -> 64 -> 2   */
    orc_mlp_t sigma_net;    /* :149  65 -> 64 -> 64 -> 65  */
    orc_mlp_t color_net;    /* :156  84 -> 64 -> 3         */
    uint32_t audio_dim;     /* 64 */
    uint32_t ind_dim;       /* 4 (0 = no individual code) */
    uint32_t sh_degree;     /* 4 */
    int has_eye;            /* exp_eye */
    float bound;
    /* torso branch (network.py:158-167, 188-219) */
    orc_grid_t enc_torso;
    orc_mlp_t torso_deform_net; /* 104 -> 64 -> 64 -> 2 */
    orc_mlp_t torso_net;        /* 136 -> 32 -> 32 -> 4 */
    uint32_t ind_dim_torso;     /* 8 */
    float torso_shrink;         /* 0.8 */
} orc_model_t;

void orc_mlp_forward(const orc_mlp_t *mlp, const float *x, uint32_t B, float *out);

/* NeRFNetwork.forward (nerf/network.py:222-283): sigma [M], color [M,3], ambient [M,2]. */
void orc_nerf_forward(const orc_model_t *m, const float *xyzs, const float *dirs, uint32_t M,
                      const float *enc_a, const float *ind_code, const float *eye,
                      float *sigma, float *color, float *ambient);
/* The same forward under the arithmetic of the opt-in 16-bit matrix-core kernel (fp16 operands where they enter a
 * matrix instruction, fp32 accumulation; see mlp_rows_mp in orc_nerf.c) -- the checker for mlp_dtype = RN_F16. */
void orc_nerf_forward_mp16(const orc_model_t *m, const float *xyzs, const float *dirs, uint32_t M,
                           const float *enc_a, const float *ind_code, const float *eye,
                           float *sigma, float *color, float *ambient);
/* NeRFNetwork.density (nerf/network.py:286-325): sigma only. */
void orc_nerf_density(const orc_model_t *m, const float *xyzs, uint32_t M, const float *enc_a,
                      const float *eye, float *sigma);
/* NeRFNetwork.forward_torso (nerf/network.py:188-219): alpha [P,1], color [P,3], dx [P,2]. */
void orc_torso_forward(const orc_model_t *m, const float *x, uint32_t P, const float *poses6,
                       const float *ind_code_torso, float *alpha, float *color, float *dx);

/* ---------------------------------------------------------------- renderer
 * NeRFRenderer.run_cuda, inference branch + torso + blend
 * (nerf/renderer.py:158-204, 225-316).  All inputs already flattened to N rays.
 * stats (optional, 4 x uint64): {loop iterations, live samples, padded sample
 * slots, torso pixels}. */
typedef struct {
    const uint8_t *density_bitfield;
    uint32_t cascade, grid_size;
    float bound, min_near;
    float aabb_infer[6];
    float dt_gamma;
    uint32_t max_steps;
    float T_thresh;
    int torso;
    const float *density_grid_torso; /* [grid_size^2] */
    float density_thresh_torso, mean_density_torso;
} orc_render_cfg_t;

void orc_render_frame(const orc_model_t *m, const orc_render_cfg_t *cfg, const float *rays_o,
                      const float *rays_d, uint32_t N, const float *enc_a,
                      const float *ind_code, const float *eye, const float *bg_coords,
                      const float *poses6, const float *ind_code_torso, const float *bg_color,
                      float *image, float *depth, uint64_t *stats);

/* ---------------------------------------------------------------- occupancy-grid maintenance
 * nerf/renderer.py:318-499 (orc_occupancy.c).  Cells in morton order: element i of cascade c = the cell with morton code i. */
/* :421-430  probe points; noise [C*H^3, 3] uniform [0,1) (torch.rand_like) or NULL = built-in hash of (seed, index) */
void orc_occupancy_points(uint32_t C, uint32_t H, float bound, const float *noise, uint32_t seed, float *xyzs);
/* :437-449  dilation, decayed max on valid cells, mean (double sum), threshold, packbits; stats = {mean, threshold} */
void orc_occupancy_update(const float *sigmas, float density_scale, float *grid, uint32_t C, uint32_t H, float decay,
                          float density_thresh, uint8_t *bitfield, float *stats);
/* :318-379 */
void orc_mark_untrained_grid(const float *poses, uint32_t n_poses, uint32_t pose_stride, double fx, double fy, double cx,
                             double cy, uint32_t C, uint32_t H, float bound, float *grid);
/* :464-476, :482-490 */
void orc_torso_grid_points(uint32_t H, const float *noise, uint32_t seed, float *xys);
void orc_torso_grid_update(const float *alphas, float *grid, uint32_t H, float decay, float *stats);
uint32_t orc_hash_u01_bits(uint32_t seed, uint32_t idx);

/* half <-> float helpers (IEEE round-to-nearest-even), exported for tests */
uint16_t orc_float_to_half(float f);
float orc_half_to_float(uint16_t h);

int orc_num_threads(void);

#ifdef __cplusplus
}
#endif
#endif
