/*
 * radnerf_fused.h -- C ABI of the MI355X-native fused render path of libradnerf_hip.so.
 *
 * These entry points have no single counterpart among the reference's pybind functions: they replace
 * whole stretches of its Python hot loop with a handful of gfx950 kernels, while computing exactly what
 * that loop computes:
 *
 *   rn_nerf_fused_forward   NeRFNetwork.forward (nerf/network.py:222-283): xyz grid -> [enc_x | enc_a] ->
 *                           ambient MLP -> tanh -> ambient grid -> [enc_x | enc_w | eye] -> sigma MLP ->
 *                           exp, SH(d) -> [enc_d | geo | ind] -> color MLP -> sigmoid, per sample, as ONE
 *                           kernel: hash/tiled-grid gathers + fp32 MFMA (v_mfma_f32_32x32x2_f32) tiles with the
 *                           weights resident in LDS; the broadcast inputs (audio code, eye, individual code)
 *                           are folded into per-frame bias vectors (rn_nerf_frame_bias).
 *   rn_head_begin /         the inference branch of NeRFRenderer.run_cuda (nerf/renderer.py:183, 225-262):
 *   rn_head_iterate         near/far, then the <= max_steps loop {march, network, composite, compact}.  The
 *                           live-ray count, the n_step policy max(min(N // n_alive, 8), 1) and the loop
 *                           condition live in device memory, so the host enqueues the loop without reading
 *                           anything back; compaction is a stable wavefront ballot/prefix-sum scatter.
 *   rn_torso_fused          the torso pass (nerf/renderer.py:269-299 + nerf/network.py:188-219): bilinear
 *                           occupancy test, deformation MLP, 2-D grid, torso MLP, blend over the background.
 *   rn_blend_frame          image + (1 - weights_sum) * bg, clamp, depth normalisation (renderer.py:306-311).
 *
 * Conventions are those of radnerf_hip.h (device pointers, caller allocates, explicit stream, int status).
 * Supported network shape (validated; anything else returns RN_ERR_INVALID_ARG and the caller must use the
 * per-operator path): grids L=16, C=2 (xyz D=3, ambient/torso D=2), hidden width 64, geo_feat 64, SH degree
 * 4, ambient_dim 2; audio_dim, eye and individual-code widths are free (they only enter the bias vectors).
 */
#ifndef RADNERF_FUSED_H
#define RADNERF_FUSED_H

#include "radnerf_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* One multiresolution grid as GridEncoder holds it (gridencoder/grid.py:96-161). */
typedef struct {
    const void *embeddings;  /* [rows, 2], float32 or float16 */
    const int32_t *offsets;  /* [L+1] */
    uint32_t D, L, H;        /* input dim, levels (16), base resolution */
    float S;                 /* log2(per_level_scale) */
    uint32_t gridtype;       /* 0 hash, 1 tiled */
    int dtype;               /* RN_F32 / RN_F16 */
} rn_grid_t;

/* Raw nn.Linear weights ([out, in] row-major, bias-free) of the per-sample MLPs (nerf/network.py:140-167). */
typedef struct {
    const float *amb_w0, *amb_w1, *amb_w2; /* ambient_net: [64, 32+audio_dim] [64,64] [2,64]      */
    const float *sig_w0, *sig_w1, *sig_w2; /* sigma_net:   [64, 64+has_eye]   [64,64] [65,64]     */
    const float *col_w0, *col_w1;          /* color_net:   [64, 80+ind_dim]   [3,64]              */
    uint32_t audio_dim, has_eye, ind_dim;
} rn_nerf_weights_t;

/* Number of floats of the packed (MFMA-ordered) weight image / of the per-frame bias block. */
size_t rn_nerf_packed_floats(void);
/* Opt-in 16-bit matrix-core variant (mlp_dtype = RN_F16 below): weights and per-sample activations are rounded to fp16
 * where they enter the f16 matrix instruction (v_mfma_f32_32x32x8f16), accumulation stays fp32 -- the arithmetic of the reference's `-O`
 * (autocast) mode, nerf/utils.py:944.  Grid interpolation, per-frame bias vectors, the narrow output layers and the
 * activations stay fp32.  Its weight image has its own size and packer; `packed` passed to rn_nerf_fused_forward /
 * rn_head_iterate must be the image that matches `mlp_dtype` (RN_F32: rn_nerf_pack_weights). */
size_t rn_nerf_packed_floats_h16(void);
int rn_nerf_pack_weights_h16(const rn_nerf_weights_t *w, float *packed, rn_stream_t stream);
/* Split-precision variant (mlp_dtype = RN_F32_SPLIT): fp32-grade contractions on the 16-bit matrix cores.  Every operand
 * is carried as two fp16 numbers (hi = fp16(v), lo = fp16(v - hi): 22 significant bits) and a product is evaluated as
 * hi*hi + hi*lo + lo*hi with fp32 accumulation on v_mfma_f32_32x32x8f16 (~3e-7 relative per product).  It meets the
 * tolerances of the fp32-MFMA kernel against the fp32 oracle; it is NOT bit-identical to it. */
#define RN_F32_SPLIT 2
size_t rn_nerf_packed_floats_split(void);
int rn_nerf_pack_weights_split(const rn_nerf_weights_t *w, float *packed, rn_stream_t stream);
size_t rn_nerf_bias_floats(void);
/* Re-order the raw weights into the image the fused kernel stages into LDS (call when weights change). */
int rn_nerf_pack_weights(const rn_nerf_weights_t *w, float *packed, rn_stream_t stream);
/* Per-frame bias vectors: W0_amb[:,32:] enc_a | W0_sig[:,64] eye | W0_col[:,80:] ind_code  (3 x 64). */
int rn_nerf_frame_bias(const rn_nerf_weights_t *w, const float *enc_a, const float *eye, const float *ind_code,
                       float *bias, rn_stream_t stream);
/* The same for n frames at once: enc_a [n, audio_dim] -> bias [n, rn_nerf_bias_floats()] (eye / ind_code shared). */
int rn_nerf_frame_bias_batch(const rn_nerf_weights_t *w, const float *enc_a, uint32_t n, const float *eye, const float *ind_code,
                             float *bias, rn_stream_t stream);

/* NeRFNetwork.forward for M sample slots.  deltas (nullable): when given, slots with deltas[2*i] == 0 are
 * dead (raymarching.cu:982) and skipped -- their outputs are left untouched.  m_dev (nullable): device
 * count of slots, overrides M (M is then only the launch bound).  ambient (nullable): [M,2] output.
 * rgbs == NULL (dirs may then be NULL too): density query -- NeRFNetwork.density (nerf/network.py:286-325), sigma only; the
 * fp32 kernel then skips the geo_feat layer, the SH basis and the colour network (a third of its matrix instructions). */
int rn_nerf_fused_forward(const float *xyzs, const float *dirs, const float *deltas, uint32_t M,
                          const int32_t *m_dev, const rn_grid_t *grid_xyz, const rn_grid_t *grid_amb,
                          const float *packed, const float *bias, float bound, float *sigmas, float *rgbs,
                          float *ambient, int mlp_dtype, rn_stream_t stream);

/* ---- device-side inference loop ------------------------------------------------------------------ */
#define RN_HEAD_STATE_INTS 64 /* int32 words of loop state the caller provides (zeroed once before first use) */

typedef struct {
    /* inputs */
    const float *rays_o, *rays_d; /* [N,3] */
    uint32_t N;
    const float *aabb;            /* [6] */
    float min_near;
    const uint8_t *bitfield;
    float bound, dt_gamma;
    uint32_t max_steps, cascade, grid_size;
    float T_thresh;
    /* outputs */
    float *nears, *fars;                  /* [N] */
    float *weights_sum, *depth, *image;   /* [N] [N] [N,3] (zeroed by rn_head_begin) */
    /* scratch, caller-allocated: N slots are always enough (n_alive * n_step <= N) */
    int32_t *rays_alive_a, *rays_alive_b; /* [N] each */
    float *rays_t;                        /* [N] */
    float *xyzs, *dirs, *deltas;          /* [N,3] [N,3] [N,2] */
    float *sigmas, *rgbs;                 /* [N] [N,3] */
    int32_t *state;                       /* [RN_HEAD_STATE_INTS] */
    uint32_t *block_counts;               /* [3 * (ceil(N/256) + 1)]: survivor counts | live-sample partial sums x 2; zeroed once
                                             before first use when RN_LOOP_COOP is used */
    int32_t *live_slots;                  /* [N] or NULL.  Scratch for the list of live sample slots of an iteration: the
                                             marchers write it and the network kernel then runs over st[6] live samples
                                             instead of all n_alive * n_step slots (rays that end in the middle of their
                                             n_step samples leave dead slots: 16 % of the benchmark stream).  NULL: the
                                             network visits every slot and skips the dead ones.  Same pixels either way. */
    uint32_t order_w;                     /* 0: alive list starts in ray order.  Image width W (rays are row-major pixels,
                                             W % 8 == 0, N % W == 0, (N / W) % 8 == 0; otherwise treated as 0): the list
                                             starts in 8 x 8 pixel blocks, so neighbouring samples share more grid rows.
                                             Speed only -- rays are independent, every pixel is unchanged. */
} rn_head_t;

/* near/far + loop initialisation (rays_alive = arange(N), rays_t = nears, accumulators = 0, step = 0). */
int rn_head_begin(const rn_head_t *h, rn_stream_t stream);
/* Enqueue loop iterations first_iter .. first_iter + n_iters - 1.  Iterations past the end of the loop
 * (step >= max_steps or no ray alive) are no-ops decided on the device.  Kernels per call: march(first_iter), then per
 * iteration {fused network, composite, compaction}; inside a call the compaction kernel also marches the next iteration
 * (3 launches per iteration), the last compaction of a call does not -- so a loop enqueued one iteration per call is the
 * plain four-kernel sequence and its schedule may be adjusted between calls (rn_head_reschedule).  mlp_dtype selects the
 * arithmetic variant of the network kernel (RN_F32 / RN_F32_SPLIT / RN_F16) and must match the packed image. */
int rn_head_iterate(const rn_head_t *h, const rn_grid_t *grid_xyz, const rn_grid_t *grid_amb, const float *packed,
                    const float *bias, uint32_t first_iter, uint32_t n_iters, int mlp_dtype, rn_stream_t stream);
/* The same loop with options (flags):
 *   RN_LOOP_FIRST_MARCHED  iteration first_iter has been marched already (rn_frame_begin marches iteration 0 inside the prologue
 *                          kernel), so the call starts with the network launch;
 *   RN_LOOP_CLOSE_FRAME    the call's last compaction is the frame's last: it also does what rn_head_check_done does (flags the
 *                          frame in state[RN_HEAD_ST_UNFINISHED] when the loop was still active) and leaves both live-sample
 *                          counters (state[6], state[14]) at zero, which rn_frame_begin relies on;
 *   RN_LOOP_COOP           compositor + compaction (+ next march) of an iteration run as ONE launch with a grid-wide barrier inside
 *                          (2 launches per iteration instead of 3; same results; slower than the split form: 2 % with a plain launch, 15 % with the
 *                          cooperative launch it now uses for the residency guarantee).  The launch
 *                          needs its <= 512 workgroups of 256 threads resident together: it is a COOPERATIVE launch
 *                          (hipLaunchCooperativeKernel), which the runtime places whole or refuses -- a refusal comes back as
 *                          RN_ERR_INVALID_ARG and the caller falls back to the split loop.  Should a workgroup ever give up waiting
 *                          at the barrier, state[RN_HEAD_ST_STALLED] counts it AND the frame is flagged in
 *                          state[RN_HEAD_ST_UNFINISHED] like a frame whose loop was cut short: it must be rendered again. */
#define RN_LOOP_FIRST_MARCHED 1u
#define RN_LOOP_CLOSE_FRAME 2u
#define RN_LOOP_COOP 4u
int rn_head_iterate_ex(const rn_head_t *h, const rn_grid_t *grid_xyz, const rn_grid_t *grid_amb, const float *packed,
                       const float *bias, uint32_t first_iter, uint32_t n_iters, int mlp_dtype, uint32_t flags,
                       rn_stream_t stream);
/* Frame prologue in ONE launch: ray generation (pose != NULL: get_rays with N = -1, nerf/utils.py:249-333, written to
 * h->rays_o / h->rays_d, which must then be writable; pose: device pointer to a row-major [3,4] / [4,4] cam2world matrix, W =
 * image width) + near/far + loop initialisation (= rn_head_begin) + the march of iteration 0 (n_step = 1).  Follow with
 * rn_head_iterate_ex(..., first_iter = 0, flags = RN_LOOP_FIRST_MARCHED | RN_LOOP_CLOSE_FRAME).  h->state[6] must be zero on
 * entry (a zeroed state block on first use; afterwards RN_LOOP_CLOSE_FRAME keeps it so). */
int rn_frame_begin(const rn_head_t *h, const float *pose, float fx, float fy, float cx, float cy, uint32_t W,
                   rn_stream_t stream);
/* Step schedule of a SHARD of a frame (tile-parallel rendering, BASELINE config 4).  The reference's policy
 * n_step = max(min(N // n_alive, 8), 1) (nerf/renderer.py:249) uses the ray count and the live count of the whole
 * call; a rank that renders only a band of the image reproduces the whole-frame schedule (and so the whole-frame
 * pixels: `step += n_step` may overshoot max_steps, so the schedule is visible in the image) by calling this after
 * iteration `iter_done` with schedule_N = rays of the whole frame and *alive_total = live rays of the whole frame
 * for the coming iteration (the all-reduced sum of each rank's state[((iter_done + 1) & 1) * 8 + 0]).  Only
 * n_step and the slot count of the coming iteration are rewritten; everything stays on the device. */
int rn_head_reschedule(const rn_head_t *h, uint32_t iter_done, uint32_t schedule_N, const int32_t *alive_total,
                       rn_stream_t stream);
/* Layout of h->state (int32 words).  Words 0..15 are the loop state, reset by rn_head_begin; words 16.. are
 * statistics that ACCUMULATE across frames (wrapping int32; the caller zeroes them when it wants to). */
#define RN_HEAD_ST_ACTIVE 4      /* state[(iter & 1) * 8 + 4]: 1 while the loop wants another iteration */
#define RN_HEAD_ST_ITERS 16      /* iterations that did work */
#define RN_HEAD_ST_LIVE 17       /* live samples evaluated */
#define RN_HEAD_ST_SLOTS 18      /* sample slots n_alive * n_step summed over iterations */
#define RN_HEAD_ST_UNFINISHED 19 /* frames whose loop was still active when rn_head_check_done looked (see below) */
#define RN_HEAD_ST_BARRIER 20    /* launch counter of the RN_LOOP_COOP kernels (tags the survivor counts of a launch; never reset) */
#define RN_HEAD_ST_STALLED 22    /* RN_LOOP_COOP: workgroups that gave up waiting at the in-launch barrier (must stay 0; a frame
                                    rendered while it moved is invalid and has to be rendered again without RN_LOOP_COOP) */
#define RN_HEAD_ST_HIST 32       /* state[32 + i], i = 0 .. max_steps (<= 31): live rays entering loop iteration i of the frame in
                                    flight (0 once the loop is over).  A shard of a frame publishes these so the ranks can verify,
                                    after the fact, that their band-local step schedules were the whole frame's (radnerf/parallel.py) */
/* Speculative loop length.  A caller that knows how many iterations frames of this stream need (device counters of
 * earlier frames) may enqueue fewer than max_steps iterations and skip the no-op launches behind them; this entry point
 * then records, on the device, whether the loop really was over after iteration iters_done - 1: if it was not,
 * state[RN_HEAD_ST_UNFINISHED] is incremented and the caller must render that frame again with more iterations (the
 * count is checked whenever the caller synchronises anyway).  Never needed when all max_steps iterations are enqueued. */
int rn_head_check_done(const rn_head_t *h, uint32_t iters_done, rn_stream_t stream);

/* ---- torso + blend ----------------------------------------------------------------------------------- */
typedef struct {
    const float *def_w0, *def_w1, *def_w2; /* torso_deform_net: [64, 96+ind] [64,64] [2,64]  */
    const float *tor_w0, *tor_w1, *tor_w2; /* torso_net:        [32, 128+ind] [32,32] [4,32] */
    uint32_t ind_dim;
} rn_torso_weights_t;

size_t rn_torso_packed_floats(void);
int rn_torso_pack_weights(const rn_torso_weights_t *w, float *packed, rn_stream_t stream);
/* bg_out[i] = torso_color * alpha + bg_in[i] * (1 - alpha) for pixels whose bilinear torso occupancy exceeds
 * `thresh`, else bg_in[i].  bg_in may be NULL (= white, the reference's `bg_color = 1`).
 * torso_alpha (nullable) [N], deform (nullable) [N,2] receive the per-pixel values (0 where masked out). */
int rn_torso_fused(const float *bg_coords, uint32_t N, const float *density_grid_torso, uint32_t grid_size,
                   float thresh, const float *poses6, const float *ind_code, float torso_shrink,
                   const rn_torso_weights_t *w, const float *packed, const rn_grid_t *grid_torso,
                   const float *bg_in, float *bg_out, float *torso_alpha, float *deform, rn_stream_t stream);

/* rn_torso_fused + rn_blend_frame in one pass over the pixels (frame epilogue): the blended background never goes to
 * memory unless bg_out is given (nullable, like torso_alpha). */
int rn_torso_blend_frame(const float *bg_coords, uint32_t N, const float *density_grid_torso, uint32_t grid_size,
                         float thresh, const float *poses6, const float *ind_code, float torso_shrink,
                         const rn_torso_weights_t *w, const float *packed, const rn_grid_t *grid_torso, const float *bg_in,
                         float *bg_out, float *torso_alpha, float *image, const float *weights_sum, float *depth,
                         const float *nears, const float *fars, uint8_t *image_u8, rn_stream_t stream);

/* mask[i] = 1 where the bilinear torso occupancy at bg_coords[i] exceeds `thresh` (F.grid_sample(bilinear, zeros,
 * align_corners=True) > thresh, nerf/renderer.py:281-283) -- the test rn_torso_fused applies per pixel, exposed for the
 * differentiable training formulation that gathers those pixels. */
int rn_torso_mask(const float *bg_coords, uint32_t N, const float *density_grid_torso, uint32_t grid_size, float thresh,
                  uint8_t *mask, rn_stream_t stream);

/* image = clamp(image + (1 - weights_sum) * bg, 0, 1); depth = max(depth - near, 0) / (far - near);
 * optional uint8 quantisation of the frame (image_u8 nullable): floor(image * 255). */
int rn_blend_frame(float *image, const float *weights_sum, const float *bg, float *depth, const float *nears,
                   const float *fars, uint32_t N, uint8_t *image_u8, rn_stream_t stream);

/* ---- occupancy-grid maintenance -------------------------------------------------------------------
 * NeRFRenderer.update_extra_state (nerf/renderer.py:383-499) and mark_untrained_grid (:318-379) without the Python block
 * loops: the cells are enumerated in MORTON order (element i of cascade c = the cell with morton code i), so the sample
 * buffer of the density query and the density grid share their indexing and nothing is ever scattered.
 *
 *   rn_occupancy_points      xyzs[C*H^3, 3]: cell centre (2 c / (H-1) - 1) * (bound_c - bound_c / H) + (u * 2 - 1) * bound_c / H
 *                            with u from `noise` ([C*H^3, 3] uniform [0,1), torch.rand_like in the reference) or, noise ==
 *                            NULL, from a counter-based hash of (seed, element index) -- rn_hash_u01_bits() * 2^-24.
 *   (density query)          rn_nerf_fused_forward(xyzs, NULL, NULL, C*H^3, ..., sigmas, rgbs = NULL, ...): sigma only.
 *   rn_occupancy_update      tmp = sigmas * density_scale; 6-neighbour max in morton space (raymarching.cu:304-341); where
 *                            grid >= 0 and tmp >= 0: grid = max(grid * decay, tmp); stats[0] = mean(max(grid, 0)) (summed in
 *                            double), stats[1] = min(stats[0], density_thresh); bitfield = packbits(grid, stats[1]).
 *                            workspace: rn_occupancy_workspace(C, H) bytes (per-workgroup partial sums; a one-workgroup
 *                            launch adds them up).  Nothing is read back by the host.
 *   rn_mark_untrained_grid   grid[c, cell] = -1 for cells no camera sees: poses [n, 4, 4] (pose_stride = 16 floats) or
 *                            [n, 3, 4] (12) cam2world, intrinsics as the Python floats fx, fy, cx, cy.
 *   rn_torso_grid_points     xys[H*H, 2] for element i = (column i % H, row i / H) (the transposed index of renderer.py:472);
 *   rn_torso_grid_update     5 x 5 max pool (-inf padding) of the H*H alphas, grid = max(grid * decay, pooled),
 *                            stats[0] = mean(grid).  The alphas come from rn_torso_fused(xys, ..., thresh = -1, ...). */
size_t rn_occupancy_workspace(uint32_t C, uint32_t H);
int rn_occupancy_points(uint32_t C, uint32_t H, float bound, const float *noise, uint32_t seed, float *xyzs, rn_stream_t stream);
int rn_occupancy_update(const float *sigmas, float density_scale, float *density_grid, uint32_t C, uint32_t H, float decay,
                        float density_thresh, uint8_t *bitfield, float *stats, void *workspace, rn_stream_t stream);
int rn_mark_untrained_grid(const float *poses, uint32_t n_poses, uint32_t pose_stride, double fx, double fy, double cx, double cy,
                           uint32_t C, uint32_t H, float bound, float *density_grid, rn_stream_t stream);
int rn_torso_grid_points(uint32_t H, const float *noise, uint32_t seed, float *xys, rn_stream_t stream);
int rn_torso_grid_update(const float *alphas, float *density_grid_torso, uint32_t H, float decay, float *stats, rn_stream_t stream);
uint32_t rn_hash_u01_bits(uint32_t seed, uint32_t idx);   /* the 24 random bits behind the built-in jitter (host-callable) */

/* ---- audio code ------------------------------------------------------------------------------
 * NeRFNetwork.encode_audio (nerf/network.py:170-185) = AudioNet (nerf/network.py:41-67: 4 x Conv1d(k3, s2, p1) +
 * LeakyReLU(0.02) over the 16-sample window, Linear-LeakyReLU-Linear) on each of the 8 frames of the attention window,
 * then AudioAttNet (nerf/network.py:10-37: 5 x Conv1d(k3, s1, p1) + LeakyReLU(0.02) over the 8 codes, Linear(8,8),
 * softmax, weighted sum) -- ~45 tiny PyTorch launches per frame in the reference; here one workgroup per (window, frame)
 * for AudioNet and one per window for the attention: two launches for any number of windows.
 * All weights are the modules' own tensors (Conv1d [cout, cin, 3] + bias, Linear [out, in] + bias), fp32. */
typedef struct {
    const float *conv_w[4], *conv_b[4];         /* audio_net.encoder_conv.{0,2,4,6}: dim_in->32->32->64->64 */
    const float *fc_w[2], *fc_b[2];             /* audio_net.encoder_fc1.{0,2}: 64->64->dim_aud */
    const float *att_conv_w[5], *att_conv_b[5]; /* audio_att_net.attentionConvNet.{0,2,4,6,8}: dim_aud->16->8->4->2->1 */
    const float *att_fc_w, *att_fc_b;           /* audio_att_net.attentionNet.0: [8,8], [8] */
    uint32_t dim_in, dim_aud, has_att;          /* has_att = 0: one frame per window, no attention (opt.att == 0) */
} rn_audio_weights_t;
#define RN_AUDIO_SEQ 8   /* frames per attention window */
#define RN_AUDIO_WIN 16  /* feature samples per frame */
/* enc[i, :] = encode_audio(auds[i]) for n windows; auds: [n, (has_att ? 8 : 1), dim_in, 16] as NeRFRenderer.run_cuda
 * receives them (nerf/renderer.py:186).  workspace: n * 8 * dim_aud floats of scratch for the per-frame codes (caller-
 * allocated like every buffer; may be NULL when has_att == 0). */
int rn_audio_encode_windows(const rn_audio_weights_t *w, const float *auds, uint32_t n, float *enc, float *workspace,
                            rn_stream_t stream);
/* Backward of rn_audio_encode_windows for the training step: given grad_enc [n, dim_aud] (and `codes`, the forward's
 * workspace with the per-frame codes), ADDS the gradients of every AudioNet / AudioAttNet parameter to the buffers of
 * `grads` (same shapes as the weights; zero them first for a plain gradient).  grad_codes: n * 8 * dim_aud floats of scratch
 * (has_att).  The input features receive no gradient (they are data). */
typedef struct {
    float *conv_w[4], *conv_b[4], *fc_w[2], *fc_b[2], *att_conv_w[5], *att_conv_b[5], *att_fc_w, *att_fc_b;
} rn_audio_grads_t;
int rn_audio_encode_windows_backward(const rn_audio_weights_t *w, const float *auds, uint32_t n, const float *codes,
                                     const float *grad_enc, const rn_audio_grads_t *grads, float *grad_codes,
                                     rn_stream_t stream);
/* The pair a training step uses: the forward keeps every AudioNet layer's output of every frame in `acts`
 * (rn_audio_train_acts_floats(n, has_att) floats), the backward starts from them instead of running the forward again
 * (five weight stagings and layer passes of a latency-bound kernel).  Same results as the pair above, bit for bit. */
size_t rn_audio_train_acts_floats(uint32_t n, int has_att);
int rn_audio_encode_windows_train(const rn_audio_weights_t *w, const float *auds, uint32_t n, float *enc, float *workspace,
                                  float *acts, rn_stream_t stream);
int rn_audio_encode_windows_backward_acts(const rn_audio_weights_t *w, const float *auds, uint32_t n, const float *codes,
                                          const float *grad_enc, const rn_audio_grads_t *grads, float *grad_codes,
                                          const float *acts, rn_stream_t stream);
/* The same for n consecutive frames (first + i) mod T of a feature stream feats [T, dim_in, 16]: the windows are cut on
 * the device exactly as get_audio_features(att_mode=2) does (nerf/utils.py:56-72: frames index-4 .. index+3, zero rows
 * outside the stream).  Needs T >= 8 and has_att. */
int rn_audio_encode_stream(const rn_audio_weights_t *w, const float *feats, uint32_t T, uint32_t first, uint32_t n,
                           float *enc, float *workspace, rn_stream_t stream);
/* Lip smoothing (nerf/renderer.py:190-194) folded over n codes in order:
 * state = state_valid ? lambda * state + (1 - lambda) * enc[i] : enc[i]; state [dim] is updated in place. */
int rn_audio_smooth(const float *enc, uint32_t n, uint32_t dim, float lambda, float *state, int state_valid,
                    rn_stream_t stream);

/* The smoothing recurrence with every intermediate state kept: out[i] = state after folding enc[i] (n x dim), `state` is
 * updated to out[n-1].  Lets a renderer that knows the next n frames' audio compute all their codes in one go. */
int rn_audio_smooth_seq(const float *enc, uint32_t n, uint32_t dim, float lambda, float *state, int state_valid, float *out,
                        rn_stream_t stream);

/* ---- ray generation -----------------------------------------------------------------------
 * get_rays for the full image (nerf/utils.py:249-333, N = -1 branch): pixel (row r, column c) -> ray r * W + c with
 * centre (c + 0.5, r + 0.5); dir = normalize(((c + 0.5) - cx) / fx, ((r + 0.5) - cy) / fy, 1); rays_d = dir @ R^T,
 * rays_o = pose[:3, 3].  ~12 PyTorch launches and two [N,3] round trips in the reference, one kernel here.
 * pose: device pointer to a row-major [4,4] (or [3,4]: row stride 4) cam2world matrix. */
int rn_get_rays(const float *pose, float fx, float fy, float cx, float cy, uint32_t H, uint32_t W, float *rays_o,
                float *rays_d, rn_stream_t stream);
/* get_bg_coords (nerf/utils.py:240-245): bg_coords [H*W, 2] in [-1, 1], component 0 along the image rows.
 * convert_poses (nerf/utils.py:231-237): n cam2world matrices [n, 4, 4] (row-major) -> poses6 [n, 6] = XYZ euler angles of the
 * rotation (matrix_to_euler_angles, :130-169) | translation -- the other two inputs of the torso pass, one launch each. */
int rn_get_bg_coords(uint32_t H, uint32_t W, float *bg_coords, rn_stream_t stream);
int rn_convert_poses(const float *poses, uint32_t n, float *poses6, rn_stream_t stream);

/* ---- the per-sample MLPs in training --------------------------------------------------------
 * nerf/network.py:69-88 (`MLP`: bias-free nn.Linear stack, ReLU between layers) with its autograd, for the shapes of the
 * path: hidden width 64, n_layers 2 or 3, in_dim <= 96, out_dim <= 4 (narrow) or 64 .. 68 (64 wide rows preceded by
 * out_dim - 64 narrow ones, as sigma_net's [sigma | geo_feat]).  x and grad_x are row-major [M, in_pad] with in_pad =
 * in_dim rounded up to a multiple of 4 (pad columns of x are read: keep them zero); out / grad_out are [M, out_dim].
 * Hidden activations h0 (h1: 3 layers) and pre-activation gradients dz0 (dz1) are opaque tile buffers of
 * rn_mlp64_tile_floats(M) floats each, written by forward / backward and read by backward / weight_grads.  `image` holds
 * rn_mlp64_image_floats() floats (16-byte aligned) and is rebuilt by rn_mlp64_pack whenever a weight changes (w0 [64,in_dim],
 * w1 [64,64] (3 layers, else NULL), w_last [out_dim,64]: the nn.Linear weights).  Weight gradients are written (not
 * accumulated) in the nn.Linear layout; `workspace` needs rn_mlp64_wgrad_workspace(n_layers) bytes.  fp32 MFMA throughout:
 * equal to torch up to summation order.
 * Constant input columns (the audio code, eye value, individual code: the same for every sample of a call) need not be
 * materialised: the MLP then reads only the first in_dim columns of a wider w0 (`ld0` = its row stride; gw0 is written with
 * the same stride, columns in_dim .. ld0-1 untouched) and the constants enter as bias0 [64] = w0[:, in_dim:] . constants, added
 * to the first layer's pre-activation; grad_bias0 [64] (nullable; needs in_dim <= 92) receives its gradient, from which the
 * caller derives the gradients of w0[:, in_dim:] (outer product with the constants) and of the constants. */
size_t rn_mlp64_image_floats(uint32_t in_dim, uint32_t out_dim, uint32_t n_layers);
size_t rn_mlp64_tile_floats(uint32_t M);
size_t rn_mlp64_wgrad_workspace(uint32_t n_layers);
int rn_mlp64_pack(const float *w0, uint32_t ld0, const float *w1, const float *w_last, uint32_t in_dim, uint32_t out_dim,
                  uint32_t n_layers, float *image, rn_stream_t stream);
int rn_mlp64_forward(const float *x, uint32_t M, const float *image, const float *bias0, uint32_t in_dim, uint32_t out_dim,
                     uint32_t n_layers, float *out, float *h0, float *h1, rn_stream_t stream);
int rn_mlp64_backward(const float *grad_out, uint32_t M, const float *image, uint32_t in_dim, uint32_t out_dim, uint32_t n_layers,
                      const float *h0, const float *h1, float *grad_x, float *dz0, float *dz1, rn_stream_t stream);
int rn_mlp64_weight_grads(const float *x, const float *grad_out, uint32_t M, uint32_t in_dim, uint32_t out_dim, uint32_t n_layers,
                          const float *h0, const float *h1, const float *dz0, const float *dz1, float *gw0, uint32_t ld0,
                          float *gw1, float *gw_last, float *grad_bias0, void *workspace, rn_stream_t stream);

/* ---- elementwise glue of the training step (one kernel per direction where PyTorch takes 3 - 15) -------------------------
 * rn_head_mid_*: between sigma_net and color_net (nerf/network.py:266-276): sigma = trunc_exp(h[:, 0]) (activation.py:5-17),
 * x_color [M, n_sh + 64] = cat[enc_d [M, n_sh], h[:, 1:65]]; backward assembles grad_h [M, 65] from grad_sigma and the geo_feat
 * columns of grad_x_color.  rn_abs_sum2_*: ambient.abs().sum(-1) for [M, 2] (nerf/renderer.py:216).
 * rn_train_loss: the head's training loss (nerf/utils.py:772-803: mean squared error + 1e-4 mean binary entropy of
 * clamp(weights_sum, 1e-5, 1 - 1e-5) + *w_amb * mean(ambient * (1 - face))) over N rays and its gradients with respect to
 * pred [N, 3], weights_sum [N], ambient [N] in one launch; face [N] is the 0 / 1 face mask as floats, w_amb a device scalar. */
int rn_head_mid_forward(const float *h, const float *enc_d, uint32_t M, uint32_t n_sh, float *sigma, float *x_color, rn_stream_t stream);
int rn_head_mid_backward(const float *h, const float *grad_sigma, const float *grad_x_color, uint32_t M, uint32_t n_sh, float *grad_h,
                         rn_stream_t stream);
int rn_abs_sum2_forward(const float *a, uint32_t M, float *out, rn_stream_t stream);
int rn_abs_sum2_backward(const float *a, const float *grad_out, uint32_t M, float *grad_a, rn_stream_t stream);
int rn_train_loss(const float *pred, const float *target, const float *weights_sum, const float *ambient, const float *face,
                  const float *w_amb, uint32_t N, float *loss, float *grad_pred, float *grad_weights_sum, float *grad_ambient,
                  rn_stream_t stream);

/* ---- optimizer update of the training step ------------------------------------------------------------------------
 * torch.optim.Adam as main.py:204 configures it (betas, eps; no weight decay, no amsgrad) for `count` tensors in ONE
 * launch (+ a one-thread launch that advances *step and computes the bias corrections in double, as Python does):
 *   m = m + (g - m)(1 - beta1);  v = beta2 v + (1 - beta2) g^2;  p -= lr / (1 - beta1^step) * m / (sqrt(v) / sqrt(1 - beta2^step) + eps)
 * `tensors` is a HOST array (the pointers travel as kernel arguments, 48 tensors per launch); *step (device int32) counts
 * the steps taken, corr is a 2-float device scratch.  Enqueue-only, capturable in a hipGraph. */
typedef struct {
    float *param;
    const float *grad;
    float *exp_avg, *exp_avg_sq;
    uint32_t numel;
    float lr;
} rn_adam_tensor_t;
int rn_adam_step(const rn_adam_tensor_t *tensors, uint32_t count, float beta1, float beta2, float eps, int32_t *step, float *corr,
                 rn_stream_t stream);
/* The same with the learning rates read from device memory at run time (lr_dev [count] floats, one per tensor; NULL = the
 * `lr` fields): a launch captured in a hipGraph then follows the reference's schedule (main.py:219: LambdaLR over the
 * optimizer's param_groups) -- the host refreshes lr_dev between replays instead of re-capturing. */
int rn_adam_step_lr(const rn_adam_tensor_t *tensors, uint32_t count, float beta1, float beta2, float eps, int32_t *step, float *corr,
                    const float *lr_dev, rn_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* RADNERF_FUSED_H */
