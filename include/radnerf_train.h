/*
 * radnerf_train.h -- C ABI of the fused TRAINING pass of the per-sample network (libradnerf_hip.so, gfx950).
 *
 * What is replaced: NeRFNetwork.forward (nerf/network.py:222-283) under autograd in the train branch of
 * NeRFRenderer.run_cuda (nerf/renderer.py:206-223), i.e. per training step of Trainer.train_step (nerf/utils.py:718-806):
 * two grid encodes with their backward (gridencoder/grid.py:24-89, gridencoder.cu:87-368), the SH encode, eight bias-free
 * nn.Linear layers with ReLU / tanh / trunc_exp / sigmoid (nerf/network.py:69-88, activation.py:5-17), three `repeat` + `cat`
 * of per-call constants, and the autograd of all of it -- ~130 launches in stock PyTorch.  Here:
 *
 *   rn_train_head_pack       weight images (forward + transposed) and the three first-layer bias vectors    1 launch
 *   rn_train_head_forward    xyz grid -> ambient net -> tanh -> ambient grid (+ d/dx) -> sigma net -> exp,
 *                            SH -> colour net -> sigmoid, per 32-sample tile on fp32 MFMA, saving every hidden
 *                            activation in the matrix-core register layout                                    1 launch
 *   rn_train_head_backward   the same tile walked back: pre-activation gradients of all eight layers, the
 *                            gradient of the ambient coordinates through the 2-D grid, and the feature
 *                            gradients of both grids in level-major [L, M, 2] layout                          1 launch
 *   rn_train_head_weight_grads   dW = dZ X^T for all eight layers in one launch (+ one reduction launch, + one
 *                            launch for the constant columns: audio code / eye / individual code and their
 *                            weight columns)                                                                   3 launches
 *   rn_grid_scatter_lbc      table gradient: scatter-add of the level-major feature gradients, merged per
 *                            64-byte line of the table in LDS before anything goes to memory                   1 launch per grid
 *
 * Conventions are those of radnerf_hip.h (device pointers, caller allocates, explicit stream, int status).  Supported
 * network shape = rn_nerf_fused_forward's (radnerf_fused.h): grids L = 16, C = 2 (xyz D = 3, ambient D = 2), hidden width 64,
 * geo_feat 64, SH degree 4, ambient_dim 2, fp32 tables.  Sample rows at index >= the live count (m_dev, nullable) receive
 * no output and contribute no gradient, exactly like the zero rows past the marcher's counter in the reference
 * (raymarching/raymarching.py:231-257).
 */
#ifndef RADNERF_TRAIN_H
#define RADNERF_TRAIN_H

#include "radnerf_fused.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Floats of the packed image rn_train_head_pack writes: forward image | transposed image | bias [192]. */
size_t rn_train_head_image_floats(void);
/* Floats of the activation / gradient workspace for a capacity of M sample rows (opaque; written by forward and
 * backward, read by backward and weight_grads). */
size_t rn_train_head_workspace_floats(uint32_t M);
/* Bytes of the weight-gradient workspace (partial sums of the reduction over the samples). */
size_t rn_train_head_wgrad_workspace(void);

/* Pack the weights (call once per step: the optimizer changed them) and fold the per-call constants into the first-layer
 * biases: W_amb0[:, 32:] enc_a | W_sig0[:, 64] eye | W_col0[:, 80:] ind_code (nerf/network.py:236, 262, 274). */
int rn_train_head_pack(const rn_nerf_weights_t *w, const float *enc_a, const float *eye, const float *ind_code, float *image,
                       rn_stream_t stream);
/* The same with the individual code picked on the device: ind_table = individual_codes [rows, ind_dim], *ind_index (int64 device
 * scalar) the row -- nerf/renderer.py:199's `self.individual_codes[index]` without an index_select launch. */
int rn_train_head_pack_row(const rn_nerf_weights_t *w, const float *enc_a, const float *eye, const float *ind_table,
                           const int64_t *ind_index, float *image, rn_stream_t stream);

/* Forward for M sample rows (m_dev: device int32 live count, clipped to M; NULL = M).  xyzs in [-bound, bound], dirs unit
 * vectors.  Outputs: sigmas [M], rgbs [M,3], ambient [M,2] (after tanh), ambient_abs [M] (|a0| + |a1|, nerf/renderer.py:216;
 * nullable).  xn [M,3] / wn [M,2]: the normalised grid inputs (x + bound) / (2 bound) and (ambient + 1) / 2 as the grid
 * kernels see them (gridencoder/grid.py:151), kept for the table scatter. */
int rn_train_head_forward(const float *xyzs, const float *dirs, uint32_t M, const int32_t *m_dev, const rn_grid_t *grid_xyz,
                          const rn_grid_t *grid_amb, const float *image, float bound, float *sigmas, float *rgbs,
                          float *ambient, float *ambient_abs, float *xn, float *wn, float *workspace, rn_stream_t stream);

/* Backward.  grad_sigmas [M], grad_rgbs [M,3], grad_ambient [M,2] (nullable), grad_ambient_abs [M] (nullable) are the
 * gradients of the forward's outputs; sigmas / rgbs / ambient its saved outputs.  Writes the feature gradients of the two
 * grids level-major: grad_enc_x [16, M, 2], grad_enc_w [16, M, 2] (rows >= live count are not written: the scatter takes
 * the same m_dev), and the pre-activation gradients into the workspace. */
int rn_train_head_backward(const float *grad_sigmas, const float *grad_rgbs, const float *grad_ambient,
                           const float *grad_ambient_abs, const float *rgbs, const float *ambient, uint32_t M,
                           const int32_t *m_dev, const float *image, float *workspace, float *grad_enc_x, float *grad_enc_w,
                           rn_stream_t stream);

/* Gradients of the eight weight matrices in the nn.Linear layout (written, not accumulated; the full [64, 32 + audio_dim]
 * etc. shapes including the constant columns) and of the constants: grad_enc_a [audio_dim], grad_eye [1] (nullable when
 * has_eye == 0), grad_ind_code [ind_dim] (nullable when ind_dim == 0). */
typedef struct {
    float *amb_w0, *amb_w1, *amb_w2, *sig_w0, *sig_w1, *sig_w2, *col_w0, *col_w1;
    float *enc_a, *eye, *ind_code;
} rn_train_head_grads_t;
int rn_train_head_weight_grads(const rn_nerf_weights_t *w, const float *enc_a, const float *eye, const float *ind_code,
                               uint32_t M, const int32_t *m_dev, const float *workspace, const rn_train_head_grads_t *grads,
                               void *wgrad_workspace, rn_stream_t stream);
/* Row form (see rn_train_head_pack_row): ind_table / g->ind_code are [ind_rows, ind_dim]; the launch that writes the constants'
 * gradients writes the picked row of g->ind_code and zeros everywhere else -- the whole gradient of individual_codes, which
 * index_select's backward builds with a memset and an index_add. */
int rn_train_head_weight_grads_row(const rn_nerf_weights_t *w, const float *enc_a, const float *eye, const float *ind_table,
                                   const int64_t *ind_index, uint32_t ind_rows, uint32_t M, const int32_t *m_dev, const float *workspace,
                                   const rn_train_head_grads_t *g, void *wgrad_workspace, rn_stream_t stream);

/* Table gradient of one grid from level-major feature gradients: grad_table[row(l, corner)] += w_corner * grad[l, b, :]
 * (kernel_grid_backward, gridencoder.cu:247-339) for b < live count; inputs [M, D] normalised coordinates (rows outside
 * [0, 1] contribute nothing, gridencoder.cu:275-280).  grad_table [rows, 2] fp32 must be zeroed by the caller.  D = 2 / 3,
 * C = 2, fp32, align_corners = false, linear interpolation.  One workgroup merges the rows of 64 (D = 3) / 128 (D = 2) samples of one level per
 * 64-byte line of the table in LDS and issues one atomic request per touched line. */
int rn_grid_scatter_lbc(const float *grad, const float *inputs, uint32_t M, const int32_t *m_dev, const rn_grid_t *grid,
                        float *grad_table, rn_stream_t stream);

/* The same gradient with the large levels summed by TABLE REGION instead of by workgroup (two launches).  A level with at least
 * 16 buckets of 4096 rows that is HASHED is "binned": the first kernel appends its (row, w g) entries to the bucket that
 * owns the row, the second has one workgroup per bucket add them in LDS and update the region with plain loads and stores -- no
 * global float atomic on those levels (the hashed levels of a T = 2^19 table are touched ~5 times per launch, but never twice by
 * one workgroup, so a per-workgroup merge leaves one memory-side request per corner pair there).  The other levels go through
 * the LDS line merge of rn_grid_scatter_lbc inside the same first launch.  offsets_host: host copy of grid->offsets [L + 1];
 * workspace: rn_grid_scatter_workspace() bytes, 256-byte aligned, ZEROED once by the caller before first use (bucket cursors
 * live at its start and are left zero by every call).  Same results up to summation order.  The rows of a BINNED level are
 * WRITTEN (each by the one workgroup that owns its region), not accumulated: the caller need not zero them, and must not
 * expect earlier contents to survive; rows of the other levels are accumulated into (zero them first). */
size_t rn_grid_scatter_workspace(uint32_t M, const rn_grid_t *grid, const int32_t *offsets_host);
/* Bit l set: level l of this grid is binned (its rows of grad_table are written, not accumulated into). */
uint32_t rn_grid_scatter_binned_levels(const rn_grid_t *grid, const int32_t *offsets_host);
/* One or two grids at once -- what the training step calls for its xyz and ambient grids: the line-merged levels of both grids
 * share ONE launch, job 0's binned levels (offsets_host + workspace given) take the two bucket launches.  3 launches in all. */
typedef struct {
    const float *grad, *inputs;      /* [L, M, 2] level-major feature gradients, [M, D] normalised coordinates */
    const rn_grid_t *grid;
    const int32_t *offsets_host;     /* host copy of grid->offsets (nullable).  With it the library knows which levels are hashed:
                                        large hashed levels skip the LDS merge (a workgroup never touches one of their lines twice)
                                        and send each x-pair's four floats from four adjacent lanes -- one request per pair; job 0's
                                        may additionally be binned when a workspace is given */
    float *grad_table;
} rn_scatter_job_t;
int rn_grid_scatter_jobs(const rn_scatter_job_t *jobs, uint32_t n_jobs, uint32_t M, const int32_t *m_dev, void *workspace,
                         size_t workspace_bytes, rn_stream_t stream);
int rn_grid_scatter_binned(const float *grad, const float *inputs, uint32_t M, const int32_t *m_dev, const rn_grid_t *grid,
                           const int32_t *offsets_host, float *grad_table, void *workspace, size_t workspace_bytes, rn_stream_t stream);

/* Head loss of the training step on the composited rays (nerf/renderer.py:306 + nerf/utils.py:772-803):
 *   pred = clamp(image + (1 - weights_sum) * bg, 0, 1);
 *   loss = mean_n mean_c (pred - target)^2 + 1e-4 mean_n H(clamp(ws, 1e-5, 1 - 1e-5)) + *w_amb mean_n (ambient_n (1 - face_n))
 * and its gradients with respect to image [N,3], weights_sum [N], ambient [N] in one launch.  pred (nullable) receives the
 * blended prediction.  Row strides (floats) of bg / target / face let them be columns of one packed batch table. */
int rn_train_head_loss(const float *image, const float *weights_sum, const float *ambient, const float *bg, uint32_t bg_stride,
                       const float *target, uint32_t target_stride, const float *face, uint32_t face_stride, const float *w_amb,
                       uint32_t N, float *loss, float *pred, float *grad_image, float *grad_weights_sum, float *grad_ambient,
                       rn_stream_t stream);

/* A training batch from a per-pixel table: out = [n, widths[0]] | [n, widths[1]] | ... (each section contiguous), section s
 * holding columns (widths[0] + .. + widths[s-1]) .. of the rows table[idx[i], :] (idx: int64 device array, row_floats = sum of
 * the <= 8 widths, given as a HOST array).  One launch where indexing + making each strided column block contiguous takes one
 * gather and one copy per section (the loader side of nerf/provider.py:588-690 for rays that are already on the device). */
int rn_train_batch_gather(const float *table, uint32_t row_floats, const int64_t *idx, uint32_t n, const uint32_t *widths,
                          uint32_t sections, float *out, rn_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* RADNERF_TRAIN_H */
