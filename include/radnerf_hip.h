/*
 * radnerf_hip.h -- C ABI of libradnerf_hip.so, the MI355X (gfx950 / CDNA4)
 * implementation of the RAD-NeRF render hot path.
 *
 * This is the drop-in boundary.  Every entry point below replaces one pybind11
 * function of the reference's four CUDA extensions (cited per function as
 * <reference file>:<line>); the reference functions take at::Tensor and launch
 * on the legacy default stream, these take raw device pointers + sizes and an
 * explicit HIP stream.  No torch types cross this boundary.
 *
 * Conventions
 *   - All pointers are DEVICE pointers (hipMalloc / torch CUDA tensors on ROCm)
 *     to contiguous buffers; the CALLER allocates every output.  Buffers the
 *     reference zero-initialises in Python (xyzs/dirs/deltas of the marchers,
 *     grad_embeddings, SH grad_inputs) must be zeroed by the caller here too.
 *   - `stream` is a hipStream_t passed as void* (0 = the null stream).  Work is
 *     only enqueued; nothing here synchronises with the host.
 *   - Return value: RN_OK (0) or a negative RN_ERR_* code.  rn_last_error()
 *     returns a thread-local human-readable message for the last failure.
 *   - Grid scalar type selector `dtype`: RN_F32 or RN_F16 (IEEE binary16), the
 *     type of the embedding table, of `outputs` and of `dy_dx` -- mirrors
 *     AT_DISPATCH_FLOATING_TYPES_AND_HALF on embeddings.scalar_type()
 *     (gridencoder/src/gridencoder.cu:466).  Ray-marching / SH / frequency
 *     entry points are float32 only, which is what their Python wrappers force
 *     (@custom_fwd(cast_inputs=torch.float32), raymarching/raymarching.py:21 etc.).
 */
#ifndef RADNERF_HIP_H
#define RADNERF_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void *rn_stream_t; /* hipStream_t */

#define RN_OK 0
#define RN_ERR_INVALID_ARG (-1) /* unsupported D / C / degree, null pointer, bad size */
#define RN_ERR_LAUNCH (-2)      /* HIP reported a launch/runtime error */
#define RN_ERR_NO_DEVICE (-3)   /* no usable gfx950 device */

#define RN_F32 0
#define RN_F16 1

/* Output layout selector of the grid encoder.
 * RN_LAYOUT_LBC is the reference's [L, B, C] (gridencoder.cu:387);
 * RN_LAYOUT_BLC writes/reads [B, L*C] directly, which removes the
 * permute(1,0,2).reshape copy of gridencoder/grid.py:57 and :75. */
#define RN_LAYOUT_LBC 0
#define RN_LAYOUT_BLC 1
/* Same memory layout as RN_LAYOUT_BLC; forward computed level-major (one (sample, level) per lane, only one
 * level's table live in an XCD's L2 at a time) instead of one sample per lane walking all levels. */
#define RN_LAYOUT_BLC_LEVELMAJOR 2

const char *rn_last_error(void);
int rn_version(void);
/* Number of visible HIP devices (>=0), or RN_ERR_NO_DEVICE. Does not create a context. */
int rn_device_count(void);
/* HIP-event timing of the fused per-sample kernel on its launch stream: enable(1) resets the record,
 * collect() waits for the recorded events and returns the number of launches and their summed duration. */
int rn_prof_enable(int on);
/* Suspend (1) / resume (0) the timing without resetting the record: a timed dispatch carries two barrier packets
 * (~10 us of idle GPU around it), so a caller that also measures whole-job throughput times a sample of its steps. */
int rn_prof_pause(int paused);
int rn_prof_collect(uint32_t *launches, float *total_ms);
/* Per-launch durations (ms) in launch order; returns how many were written (<= capacity) or a negative error. */
int rn_prof_durations(float *out_ms, uint32_t capacity);

/* ===================================================================== raymarching
 * reference: raymarching/src/raymarching.h:7-20, kernels in raymarching/src/raymarching.cu */

/* raymarching.h:7   near_far_from_aabb      (raymarching.cu:91-156) */
int rn_near_far_from_aabb(const float *rays_o, const float *rays_d, const float *aabb, uint32_t N,
                          float min_near, float *nears, float *fars, rn_stream_t stream);
/* raymarching.h:8   sph_from_ray            (raymarching.cu:162-209) */
int rn_sph_from_ray(const float *rays_o, const float *rays_d, float radius, uint32_t N,
                    float *coords, rn_stream_t stream);
/* raymarching.h:9   morton3D                (raymarching.cu:214-232) */
int rn_morton3D(const int32_t *coords, uint32_t N, int32_t *indices, rn_stream_t stream);
/* raymarching.h:10  morton3D_invert         (raymarching.cu:237-260) */
int rn_morton3D_invert(const int32_t *indices, uint32_t N, int32_t *coords, rn_stream_t stream);
/* raymarching.h:11  packbits  (N = output bytes = C*H^3/8)  (raymarching.cu:267-300) */
int rn_packbits(const float *grid, uint32_t N, float density_thresh, uint8_t *bitfield,
                rn_stream_t stream);
/* raymarching.h:12  morton3D_dilation       (raymarching.cu:304-341) */
int rn_morton3D_dilation(const float *grid, uint32_t C, uint32_t H, float *grid_dilation,
                         rn_stream_t stream);

/* raymarching.h:14  march_rays_train        (raymarching.cu:352-528)
 * Sample slices are reserved with a deterministic exclusive scan in ray order
 * instead of the reference's unordered atomicAdd (one of its legal orders), so
 * rays[i] = (i, offset_i, count_i).  counter[0] += total samples, counter[1] += N.
 * `workspace` needs rn_march_rays_train_workspace(N) bytes. */
size_t rn_march_rays_train_workspace(uint32_t N);
int rn_march_rays_train(const float *rays_o, const float *rays_d, const uint8_t *grid, float bound,
                        float dt_gamma, uint32_t max_steps, uint32_t N, uint32_t C, uint32_t H,
                        uint32_t M, const float *nears, const float *fars, float *xyzs, float *dirs,
                        float *deltas, int32_t *rays, int32_t *counter, const float *noises,
                        void *workspace, rn_stream_t stream);
/* The same with the sample budget on the DEVICE: M is the capacity of xyzs / dirs / deltas (rows), *M_dev (<= M) the
 * budget the drop rule `offset + n > M` uses (raymarching.cu:446-457).  A captured training step (hipGraph) can then
 * follow the running-average budget of raymarching.py:226-229 without new shapes.  Rays dropped by the budget get
 * rays[i].count = 0 (the compositors only know the capacity); all outputs equal those of rn_march_rays_train with
 * M = *M_dev followed by the compositors with the same M. */
int rn_march_rays_train_budget(const float *rays_o, const float *rays_d, const uint8_t *grid, float bound,
                               float dt_gamma, uint32_t max_steps, uint32_t N, uint32_t C, uint32_t H,
                               uint32_t M, const int32_t *M_dev, const float *nears, const float *fars,
                               float *xyzs, float *dirs, float *deltas, int32_t *rays, int32_t *counter,
                               const float *noises, void *workspace, rn_stream_t stream);
/* A training step's marcher in ONE launch (no counterpart in the reference, which calls near_far_from_aabb, zeroes its
 * counters and calls march_rays_train: raymarching/raymarching.py:19-49, 187-262, nerf/renderer.py:183, 209-213):
 * near / far against `aabb` (written to nears / fars), count pass, ordered slices, write pass, and the counters SET to
 * (samples of this step, N) -- not added to.  Same rays / xyzs / dirs / deltas as rn_near_far_from_aabb +
 * rn_march_rays_train_budget on zeroed counters, except that the buffers need NOT be zero on entry: the slice of a ray
 * the budget cuts is cleared by the launch, so rows [0, min(counter[0], M)) are all defined; rows beyond are not touched.
 * The workgroups exchange their counts inside the launch, so all ceil(N / 256) of them must be resident together:
 * N <= 256 * (number of CUs), else RN_ERR_INVALID_ARG.  `state`: rn_march_rays_train_step_state(N) bytes, 8-byte aligned,
 * zeroed ONCE by the caller and then left to these launches (word 0 = launch epoch, word 1 = launches whose exchange
 * timed out -- must stay 0; such a step reports counter[0] = 0 and marks every ray empty).  Jitter of the first sample
 * (raymarching.cu:392): noises [N] uniform [0,1) as the reference draws them, or noises == NULL and jitter_seed != 0: a
 * counter-based hash of (jitter_seed, launch epoch, ray) -- a new draw per launch without a launch of its own; both NULL / 0:
 * no jitter. */
size_t rn_march_rays_train_step_state(uint32_t N);
int rn_march_rays_train_step(const float *rays_o, const float *rays_d, const uint8_t *grid, const float *aabb,
                             float min_near, float bound, float dt_gamma, uint32_t max_steps, uint32_t N, uint32_t C,
                             uint32_t H, uint32_t M, const int32_t *M_dev, const float *noises, float *nears,
                             float *fars, float *xyzs, float *dirs, float *deltas, int32_t *rays, int32_t *counter,
                             void *state, uint32_t jitter_seed, rn_stream_t stream);
/* raymarching.h:15  march_rays_train_backward   (raymarching.cu:535-593) */
int rn_march_rays_train_backward(const float *grad_xyzs, const float *grad_dirs, const int32_t *rays,
                                 const float *deltas, uint32_t N, uint32_t M, float *grad_rays_o,
                                 float *grad_rays_d, rn_stream_t stream);
/* raymarching.h:16  composite_rays_train_forward  (raymarching.cu:603-698) */
int rn_composite_rays_train_forward(const float *sigmas, const float *rgbs, const float *ambient,
                                    const float *deltas, const int32_t *rays, uint32_t M, uint32_t N,
                                    float T_thresh, float *weights_sum, float *ambient_sum,
                                    float *depth, float *image, rn_stream_t stream);
/* raymarching.h:17  composite_rays_train_backward (raymarching.cu:711-820) */
int rn_composite_rays_train_backward(const float *grad_weights_sum, const float *grad_ambient_sum,
                                     const float *grad_image, const float *sigmas, const float *rgbs,
                                     const float *ambient, const float *deltas, const int32_t *rays,
                                     const float *weights_sum, const float *ambient_sum,
                                     const float *image, uint32_t M, uint32_t N, float T_thresh,
                                     float *grad_sigmas, float *grad_rgbs, float *grad_ambient,
                                     rn_stream_t stream);
/* raymarching.h:19  march_rays              (raymarching.cu:827-939)
 * n_alive_dev (optional, may be NULL): device int32 holding the live-ray count;
 * when given it overrides `n_alive` inside the kernel (`n_alive` is then only
 * the launch bound), so the inference loop needs no host read-back. */
int rn_march_rays(uint32_t n_alive, uint32_t n_step, const int32_t *rays_alive, const float *rays_t,
                  const float *rays_o, const float *rays_d, float bound, float dt_gamma,
                  uint32_t max_steps, uint32_t C, uint32_t H, const uint8_t *grid,
                  const float *nears, const float *fars, float *xyzs, float *dirs, float *deltas,
                  const float *noises, const int32_t *n_alive_dev, rn_stream_t stream);
/* raymarching.h:20  composite_rays          (raymarching.cu:942-1038) */
int rn_composite_rays(uint32_t n_alive, uint32_t n_step, float T_thresh, int32_t *rays_alive,
                      float *rays_t, const float *sigmas, const float *rgbs, const float *deltas,
                      float *weights_sum, float *depth, float *image, const int32_t *n_alive_dev,
                      rn_stream_t stream);

/* Stable compaction of the live-ray list: out = in[in >= 0], order preserved,
 * *n_out = number kept.  Replaces the boolean-mask indexing + implicit host sync
 * of nerf/renderer.py:258 with a wavefront-ballot / prefix-sum kernel.
 * `workspace` needs rn_compact_rays_workspace(n) bytes. */
size_t rn_compact_rays_workspace(uint32_t n);
int rn_compact_rays(const int32_t *rays_alive_in, uint32_t n, const int32_t *n_dev,
                    int32_t *rays_alive_out, int32_t *n_out, void *workspace, rn_stream_t stream);

/* ===================================================================== gridencoder
 * reference: gridencoder/src/gridencoder.h:12-15, kernels in gridencoder/src/gridencoder.cu */

/* gridencoder.h:12  grid_encode_forward     (gridencoder.cu:87-244, 372-399, 447-470)
 * inputs float32 [B,D] in [0,1]; embeddings [rows,C]; offsets int32 [L+1];
 * outputs [L,B,C] (RN_LAYOUT_LBC) or [B,L*C] (RN_LAYOUT_BLC); dy_dx [B,L,D,C] or NULL.
 * D in {2,3,4,5}; C in {1,2,4,8}; L <= 32.  S = log2(per_level_scale). */
int rn_grid_encode_forward(const float *inputs, const void *embeddings, const int32_t *offsets,
                           void *outputs, uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S,
                           uint32_t H, void *dy_dx, uint32_t gridtype, int align_corners,
                           uint32_t interp, int dtype, int layout, rn_stream_t stream);
/* The same lookup with room to work in (MI355X-native planning; no counterpart in the reference, whose kernel writes [L,B,C]
 * and leaves the permute to PyTorch, grid.py:57).  With offsets_host (host copy of `offsets`) and no dy_dx, D in {2,3},
 * C in {2,4}, align_corners off, linear interpolation:
 *   - the first levels -- small, dense -- are done in ONE pass by persistent workgroups that stage the dense ones in LDS
 *     (150 KB of a CU's 160 KB hold levels 0 and 1 of the standard L=16 grids) and gather the next few from L2;
 *   - the remaining levels run level-major (one 4 MB hashed level at a time is what an XCD's L2 holds);
 *   - RN_LAYOUT_BLC: each chunk of samples is computed level-major into `workspace` and its [B, L*C] rows are written by
 *     the launch of the first of those levels, which runs last and moves the other levels' slabs along with its own gathers
 *     (rows of <= 128 bytes; wider rows: one 128-byte segment per launch of the last levels; other shapes: a transposition
 *     pass).  16-byte coalesced stores (workspace: rn_grid_encode_forward_workspace() bytes; smaller is allowed, >= 256
 *     samples).
 * Any other shape (or offsets_host == NULL) takes rn_grid_encode_forward.  Results are bit-identical either way. */
size_t rn_grid_encode_forward_workspace(uint32_t B, uint32_t L, uint32_t C, int dtype);
int rn_grid_encode_forward_ws(const float *inputs, const void *embeddings, const int32_t *offsets,
                              const int32_t *offsets_host, void *outputs, uint32_t B, uint32_t D, uint32_t C,
                              uint32_t L, float S, uint32_t H, void *dy_dx, uint32_t gridtype, int align_corners,
                              uint32_t interp, int dtype, int layout, void *workspace, size_t workspace_bytes,
                              rn_stream_t stream);
/* GridEncoder.forward without its pass over the coordinates (gridencoder/grid.py:145-163): takes world coordinates in
 * [-bound, bound] and applies `inputs = (inputs + bound) / (2 * bound)` (grid.py:149) inside the lookup's coordinate load --
 * as (x + bound) * (1 / (2 bound)), both in fp32: what that line computes on the device, where PyTorch turns the division by
 * a host scalar into a multiplication by its fp32 reciprocal (identical to the division whenever 2 * bound is a power of
 * two, e.g. the reference's bound = 1).  Planned shapes only (see rn_grid_encode_forward_ws: offsets_host given, D in {2,3},
 * C in {2,4}, no dy_dx, align_corners off, linear interpolation); anything else returns RN_ERR_INVALID_ARG.
 * With RN_LAYOUT_BLC the [B, L*C] rows are written by the last level launch itself (no transposition pass). */
int rn_grid_encode_forward_bound(const float *inputs, float bound, const void *embeddings, const int32_t *offsets,
                                 const int32_t *offsets_host, void *outputs, uint32_t B, uint32_t D, uint32_t C, uint32_t L,
                                 float S, uint32_t H, uint32_t gridtype, int dtype, int layout, void *workspace,
                                 size_t workspace_bytes, rn_stream_t stream);
/* gridencoder.h:13  grid_encode_backward    (gridencoder.cu:247-368, 401-443, 472-502)
 * grad_embeddings must be zero-initialised (grid.py:77); grad_inputs may be NULL. */
int rn_grid_encode_backward(const void *grad, const float *inputs, const void *embeddings,
                            const int32_t *offsets, void *grad_embeddings, uint32_t B, uint32_t D,
                            uint32_t C, uint32_t L, float S, uint32_t H, const void *dy_dx,
                            void *grad_inputs, uint32_t gridtype, int align_corners, uint32_t interp,
                            int dtype, int layout, rn_stream_t stream);
/* gridencoder.h:15  grad_total_variation    (gridencoder.cu:505-644)  float32 */
int rn_grad_total_variation(const float *inputs, const float *embeddings, float *grad,
                            const int32_t *offsets, float weight, uint32_t B, uint32_t D, uint32_t C,
                            uint32_t L, float S, uint32_t H, uint32_t gridtype, int align_corners,
                            rn_stream_t stream);

/* ===================================================================== shencoder
 * reference: shencoder/src/shencoder.h:9-10, kernels shencoder/src/shencoder.cu:28-382 */
int rn_sh_encode_forward(const float *inputs, float *outputs, uint32_t B, uint32_t D, uint32_t C,
                         float *dy_dx, rn_stream_t stream);
int rn_sh_encode_backward(const float *grad, const float *inputs, uint32_t B, uint32_t D, uint32_t C,
                          const float *dy_dx, float *grad_inputs, rn_stream_t stream);

/* ===================================================================== freqencoder
 * reference: freqencoder/src/freqencoder.h:7-10, kernels freqencoder/src/freqencoder.cu:30-94 */
int rn_freq_encode_forward(const float *inputs, uint32_t B, uint32_t D, uint32_t deg, uint32_t C,
                           float *outputs, rn_stream_t stream);
int rn_freq_encode_backward(const float *grad, const float *outputs, uint32_t B, uint32_t D,
                            uint32_t deg, uint32_t C, float *grad_inputs, rn_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* RADNERF_HIP_H */
