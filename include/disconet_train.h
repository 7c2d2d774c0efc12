/*
 * disconet_train.h -- C ABI of the training-step kernels (same library,
 * libdisconet_hip.so).  SURVEY.md §8(f) "next" row 1.
 *
 * Replaces what autograd + torch.optim run under
 *   upstream:coperception/utils/CoDetModule.py :: CoDetModule.step
 *     (model(...) in train() mode; loss_calculator; loss.backward(); optimizer.step())
 *   upstream:coperception/utils/loss.py :: SoftmaxFocalClassificationLoss,
 *     WeightedSmoothL1LocalizationLoss
 * (invoked by the `python train_codet.py ... --com disco` line, /root/reference/README.md:54-63;
 * the sources are in the un-vendored submodule, /root/reference/.gitmodules:1-3, so there
 * are no line numbers to cite).
 *
 * Same conventions as disconet_hip.h: device pointers, fp32 NHWC maps with an explicit
 * row stride ("ld", in floats) where a tensor may be a channel slice of a wider one,
 * caller-owned buffers and workspaces, everything enqueued on `stream`, int status +
 * dn_last_error().
 */
#ifndef DISCONET_TRAIN_H
#define DISCONET_TRAIN_H

#include "disconet_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- conv backward ------------------------------------------------------------------- */

/* dW of the layer `d` describes (the same descriptor its forward dn_conv2d used; d->ldo is
 * the row stride of dz).  dw[co][ci][ky][kx] with ci running over c0 + c1; `dw_cin_total`
 * (0 = c0 + c1) is the ci extent of the tensor dw points into, so a column block of a wider
 * weight (the attention MLP's [W_ego | W_nbr]) can be written in place.  Slices are summed
 * in a fixed order: deterministic.  accumulate != 0 adds to dw. */
size_t dn_conv_wgrad_workspace(const dn_conv_desc* d);
/* workspace of the per-channel reductions below (dn_bn_train_stats, dn_bn_train_backward, dn_channel_sum): the
 * folded sums AND every workgroup's partial.  The size depends on the library version (it grew with the
 * deterministic reductions of dn_version 112), so those entry points take the size of what was allocated
 * (`sums_bytes`) and refuse a workspace that is too small instead of writing past it. */
size_t dn_reduce_workspace_bytes(int n_groups, long rows_per_group, int c);
int dn_conv_wgrad(const dn_conv_desc* d, const float* src0, const float* src1, const float* dz,
                  void* workspace, float* dw, int dw_cin_total, int accumulate, void* stream);

/* dn_conv_wgrad on the f16 MFMA with split operands (csrc/wgrad_sp.inl): every dz and x value enters as an f16 hi + lo pair
 * (half(v) + half(v - half(v)), 22 significand bits) and a product is hi.hi + hi.lo + lo.hi on v_mfma_f32_32x32x16_f16 with
 * fp32 accumulation -- 5.3 x the fp32 MFMA's rate per product.  The maps stay what dn_conv_wgrad takes (float32 NHWC, same
 * descriptor, same gather: x2 upsample of source 0, concat); the kernel multiplies by the lifts, splits and transposes while
 * it stages (K = pixels: a lane's fragment is 8 consecutive pixels of one channel).
 *   dz_lift, x_lift: powers of two.  dz_lift brings max |dz| * dz_lift to ~2^8 (the lift of dn_bn_train_backward_finish_sp);
 *   x_lift (activations: 16) keeps the lo halves of small activations out of the f16 subnormal range.  A lifted value beyond
 *   +-65504 is clamped and sets bit 0 of dn_sp_range_flags: that step's gradients are invalid.  1 / (dz_lift * x_lift) is
 *   applied by the fixed-order slice sum (exact).
 * Layers: 3x3; stride 2 with ONE 16-byte loadable source of 32 k channels and c_out >= 32 (the encoder's down-sampling layers: the
 * patch is staged as two column-parity planes so that a fragment is still consecutive dwords); stride 1 with 16-byte aligned dz
 * (and sources, with the exception below), and either c_out >= 64 with c0 and c1 multiples of 64 (a workgroup owns a
 * 64 x 64 (co, ci) block) or c_out >= 32 (32 x 32 blocks: the 32-channel layers of the full-resolution maps) with either two
 * sources of 32 k channels each or ONE source of any width, which then need not be 16-byte loadable (the 13-channel voxel
 * grid of conv_pre_1: dword loads, the last block partial).  dn_conv_wgrad_sp_supported returns that block size, 0 = no such
 * kernel for the layer.  Deterministic like dn_conv_wgrad.  Workspace: dn_conv_wgrad_sp_workspace(d) bytes. */
int dn_conv_wgrad_sp_supported(const dn_conv_desc* d);
size_t dn_conv_wgrad_sp_workspace(const dn_conv_desc* d);
int dn_conv_wgrad_sp(const dn_conv_desc* d, const float* src0, const float* src1, const float* dz, void* workspace, float* dw,
                     int dw_cin_total, int accumulate, float dz_lift, float x_lift, void* stream);
/* dn_conv_wgrad_sp with dz taken from its SP copy (round 6): dz_sp = dz * dz_lift as the SP tensor [n][c_out / 16][4][h_out][w_out]
 * that dn_bn_train_backward_finish_sp / _finish_bias(_deferred) write for the data gradient (c_out % 16 == 0) -- the same hi / lo
 * halves the fp32 form derives while it stages dz, so dw is the same bits; the staging of the dz tile becomes byte permutes, and
 * a layer whose data gradient and bias gradient do not read the fp32 dz either can pass dz = NULL to those BatchNorm entry
 * points and save the fp32 copy's write (a quarter of that launch's bytes). */
int dn_conv_wgrad_sp_z(const dn_conv_desc* d, const float* src0, const float* src1, const void* dz_sp, void* workspace, float* dw,
                       int dw_cin_total, int accumulate, float dz_lift, float x_lift, void* stream);

/* Weights of the data-gradient conv: wt[ci][co][2-ky][2-kx] = w[co][ci][ky][kx] for the
 * c_in columns starting at `ci_first` of a [c_out][cin_total][k][k] tensor.  dx is then
 * dn_conv2d(dz; pack(wt)) with stride 1 -- for a stride-2 layer with src0 read zero-stuffed
 * (dn_conv_desc.up0 = 2: source pixel (y/2, x/2) where y and x are even, 0 elsewhere). */
int dn_conv_dgrad_weights(const float* w_oihw, int c_out, int cin_total, int ci_first, int c_in,
                          int ksize, float* wt_oihw, void* stream);

/* Parity-phase data gradient of a STRIDE-2 3x3 layer (even h_in, w_in): input pixel (2 m + py, 2 n + px) receives
 * gradient only through the taps whose offset has its parity, so dx splits into four stride-1 convs over dz, one per
 * class (py, px), of 1, 2, 2 and 4 taps -- a quarter of the MFMAs of the zero-stuffed form (dn_conv_desc.up0 = 2, kept for
 * odd map sizes).  dn_conv_dgrad_class_weights writes class (py, px)'s 3x3 kernel v[ci][co][3][3] (unused taps zero) and
 * the mask of the taps it uses; dn_conv2d_taps (disconet_hip.h) runs it over dz [n][h_out][w_out][c_out] with only those
 * taps multiplied, writing output pixel (m, n) to dx[2 m + py][2 n + px] through the output strides. */
int dn_conv_dgrad_class_weights(const float* w_oihw, int c_out, int cin_total, int ci_first, int c_in, int py, int px,
                                float* v_oihw, int* tap_mask, void* stream);

/* ---- batch norm in training mode (+ ReLU) ---------------------------------------------- */

/* Statistics over `rows_per_group` pixels for each of `n_groups` consecutive groups of rows
 * (n_groups = 1: nn.BatchNorm2d over the batch; n_groups = calls: the attention MLP's BN
 * layers, which see one (ego, neighbour) pair of 1 x C x 32 x 32 per call).
 * sums: workspace of dn_reduce_workspace_bytes(n_groups, rows_per_group, c) bytes (the folded sums and every
 * workgroup's partial: the per-channel sums are DETERMINISTIC -- fixed thread, workgroup and fold order, no
 * atomics).  var is the biased variance. */
int dn_bn_train_stats(const float* z, int n_groups, long rows_per_group, int c, int ldz,
                      double* sums, size_t sums_bytes, float* mean, float* var, void* stream);

/* Two-phase form for a BatchNorm batch that is spread over several ranks (agent-parallel training, disconet_amd/sharded.py):
 *   dn_bn_train_stats_partial   this rank's rows -> the folded sums [n_groups][2 c] doubles (sum z, sum z^2) at the START of `sums`
 *   (the caller all-reduces those n_groups * 2 * c doubles over the ranks)
 *   dn_bn_train_stats_finish    mean / biased var from the sums over `norm_rows` rows per group (the GLOBAL count).
 * dn_bn_train_stats is the two phases back to back with norm_rows = rows_per_group. */
int dn_bn_train_stats_partial(const float* z, int n_groups, long rows_per_group, int c, int ldz,
                              double* sums, size_t sums_bytes, void* stream);
int dn_bn_train_stats_finish(const double* sums, int n_groups, long norm_rows, int c, float* mean, float* var, void* stream);

/* dn_bn_train_stats for ONE group that also applies the running-statistics update of dn_bn_update_running (momentum, unbiased
 * variance with rows / (rows - 1)) from the launch that finishes the statistics (round 6: one launch instead of three behind the
 * reduction).  Bit for bit the separate calls. */
int dn_bn_train_stats_running(const float* z, long rows, int c, int ldz, double* sums, size_t sums_bytes, float* mean, float* var,
                              float* running_mean, float* running_var, float momentum, void* stream);

/* y = act((z - mean) * rsqrt(var + eps) * gamma + beta), act = ReLU if relu */
int dn_bn_train_apply(const float* z, const float* mean, const float* var, const float* gamma,
                      const float* beta, float eps, int relu, int n_groups, long rows_per_group,
                      int c, int ldz, float* y, void* stream);

/* dn_bn_train_apply with ReLU that also writes the ReLU gate of the backward as ONE BYTE PER FOUR CHANNELS:
 * relu_mask[(row * c + ch) / 4] bit (ch % 4) = (y[row][ch] > 0) -- rows * c / 4 bytes, 1/16 of y.  The backward entry points
 * below take it in place of y with relu = 2: their two passes then read 1/4 byte per element where y costs 4 (same gate, same
 * results bit for bit).  c % 4 == 0, 16-byte aligned tensors. */
int dn_bn_train_apply_mask(const float* z, const float* mean, const float* var, const float* gamma, const float* beta,
                           float eps, int n_groups, long rows_per_group, int c, int ldz, float* y,
                           unsigned char* relu_mask, void* stream);

/* dn_bn_train_apply_mask (one group) that ALSO writes y as an SP tensor (include/disconet_hip.h "SP tensor":
 * [image][c / 16][4 quarters][hw][8 halves]; y_sp of dn_sp_tensor_bytes(rows / hw, ., ., c) bytes) -- the operand of the NEXT
 * layer's forward conv on the split-f16 LDS-DMA engine (dn_spconv2d_nhwc), so that the training forward runs the inference
 * engine's kernels instead of splitting fp32 rows on the VALU while staging them (round 6).  The split is dn_sp_from_nhwc's
 * (clamp to +-65504, hi = half(y), lo = half(y - hi)); a clamped value sets the engine's sticky range flag.  y (fp32) is
 * still written: the weight gradient of the next layer and the skip consumers read it.  rows % hw == 0, c % 16 == 0, c / 4 a
 * power of two, 16-byte aligned tensors. */
int dn_bn_train_apply_mask_sp(const float* z, const float* mean, const float* var, const float* gamma, const float* beta,
                              float eps, long rows, int hw, int c, int ldz, float* y, unsigned char* relu_mask, void* y_sp,
                              void* stream);

/* running = (1 - momentum) * running + momentum * batch stat, group after group in the order
 * `order` lists them (null = 0..n_groups-1); running_var takes the unbiased variance. */
int dn_bn_update_running(const float* mean, const float* var, int n_groups, long rows_per_group,
                         int c, const int* order, float momentum, float* running_mean,
                         float* running_var, void* stream);

/* Backward of y = act(bn(z)).  The incoming gradient is dy_a (+ dy_b if not null); each has
 * its own row stride, and dy_a may live at twice the resolution (up_a = 1: the 2 x 2 block
 * sum, i.e. the backward of the decoder's nearest upsample; h, w are then y's dims).
 *   up_a = 2: dy_a is the SPACE-TO-DEPTH image of the gradient, [img][h / 2][w / 2][4 c] with pixel (y, x), channel ch at
 *   (y / 2, x / 2), channel ((y & 1) * 2 + (x & 1)) * c + ch -- what one split-f16 launch over the four parity classes of a
 *   stride-2 layer's data gradient writes (round 5; even h, w; ld_a >= 4 c).
 *   g = (dy_a + dy_b) * (y > 0)      dbeta = sum g      dgamma = sum g * zhat
 *   (relu = 2: `y` is not the map but dn_bn_train_apply_mask's byte mask of (y > 0), cast to const float*)
 *   dz = gamma * rstd * (g - mean(g) - zhat * mean(g * zhat))       (means per group)
 * dgamma / dbeta are summed over groups and ADDED when accumulate != 0.
 * sums: workspace of dn_reduce_workspace_bytes(n_groups, images_per_group * h * w, c) bytes. */
int dn_bn_train_backward(const float* dy_a, int ld_a, int up_a, const float* dy_b, int ld_b,
                         const float* y, const float* z, const float* mean, const float* var,
                         const float* gamma, float eps, int relu, int n_groups, int h, int w,
                         int images_per_group, int c, double* sums, size_t sums_bytes, float* dz,
                         float* dgamma, float* dbeta, int accumulate, void* stream);

/* Two-phase form of the backward (see dn_bn_train_stats_partial): `_partial` leaves this rank's folded sums of g and
 * g * zhat at the start of `sums` and writes dgamma / dbeta out of THESE rows (they are plain sums over rows: the ranks'
 * shares meet in the gradient all-reduce); the caller all-reduces the n_groups * 2 * c doubles; `_finish` writes dz of this
 * rank's rows with the means taken over `norm_rows` rows per group. */
int dn_bn_train_backward_partial(const float* dy_a, int ld_a, int up_a, const float* dy_b, int ld_b, const float* y,
                                 const float* z, const float* mean, const float* var, float eps, int relu, int n_groups,
                                 int h, int w, int images_per_group, int c, double* sums, size_t sums_bytes, float* dgamma,
                                 float* dbeta, int accumulate, void* stream);
int dn_bn_train_backward_finish(const float* dy_a, int ld_a, int up_a, const float* dy_b, int ld_b, const float* y,
                                const float* z, const float* mean, const float* var, const float* gamma, float eps, int relu,
                                int n_groups, int h, int w, int images_per_group, int c, const double* sums, long norm_rows,
                                float* dz, void* stream);

/* dn_bn_train_backward_finish that ALSO writes dz * sp_lift as the SP tensor [n][c / 16][4][h][w] x 16 bytes of the inference conv
 * engine (disconet_hip.h "SP tensor") -- the pre-split operand of the split-f16 data gradient (dn_spconv2d_nhwc): the split is
 * paid once, by the kernel that produces dz, not by the conv's staging.  One group, c % 16 == 0, 16-byte aligned tensors.
 * sp_lift: a power of two that brings max |dz| * sp_lift to ~2^8 -- a value is stored as half(x) + half(x - half(x)): 2^-22
 * relative while 2^-3 <= |x| <= 65504, so the top 19 binades of the tensor keep full precision, smaller elements an absolute floor
 * of 2^-25 / sp_lift, and 256 x of head room remains before the clamp (which sets bit 0 of dn_sp_range_flags: the gradients of
 * that step are invalid).  The caller folds 1 / sp_lift into the data gradient's scale vector (exact). */
int dn_bn_train_backward_finish_sp(const float* dy_a, int ld_a, int up_a, const float* dy_b, int ld_b, const float* y,
                                   const float* z, const float* mean, const float* var, const float* gamma, float eps, int relu,
                                   int n_groups, int h, int w, int images_per_group, int c, const double* sums, long norm_rows,
                                   float* dz, void* dz_sp, float sp_lift, void* stream);

/* dn_bn_train_backward_finish (dz_sp NULL) / _finish_sp with the BIAS GRADIENT of the conv in front of this BatchNorm fused in
 * (round 6): dbias[ch] = sum over this call's rows of dz[.][ch], written by the same launch that writes dz (a thread adds its
 * own values in fp32, a workgroup its threads in double, fold in a fixed order: deterministic) -- in place of a
 * dn_channel_sum pass that reads dz again.  One group; c / 4 a power of two; bias_ws of dn_bn_bias_workspace_bytes(rows, c).
 * dz may be NULL when dz_sp is given (and in dn_bn_train_backward_finish_sp where the one-group fast form runs): only the SP
 * copy is written (dn_conv_wgrad_sp_z reads that). */
size_t dn_bn_bias_workspace_bytes(long rows, int c);
int dn_bn_train_backward_finish_bias(const float* dy_a, int ld_a, int up_a, const float* dy_b, int ld_b, const float* y,
                                     const float* z, const float* mean, const float* var, const float* gamma, float eps,
                                     int relu, int h, int w, int images, int c, const double* sums, long norm_rows, float* dz,
                                     void* dz_sp /* may be NULL */, float sp_lift, float* dbias, double* bias_ws,
                                     size_t bias_ws_bytes, void* stream);

/* out[c] (+)= sum over rows of x[row][c] (bias gradients); sums: dn_reduce_workspace_bytes(1, rows, c) bytes */
int dn_channel_sum(const float* x, long rows, int c, int ld, double* sums, size_t sums_bytes, float* out,
                   int accumulate, void* stream);

/* The same sums with their FOLDS DEFERRED (round 6).  A bias gradient is a leaf of the backward: nothing reads it before the
 * optimizer step, so the launch that leaves the per-workgroup partials need not be followed by a fold of its own (26 launches of
 * ~5 us per training step on a stream with nothing to run beside them).  dn_bn_train_backward_finish_bias_deferred /
 * dn_channel_sum_partial are dn_bn_train_backward_finish_bias / dn_channel_sum without the fold: the partials stay in the
 * workspace ([c] doubles for the folded sums, then [*n_blocks][c] partials), which must stay untouched until
 * dn_channel_sum_fold_multi has folded it: one launch for every job (32 per launch), each in dn_channel_sum's fixed order --
 * out[ch] = (accumulate ? out[ch] : 0) + (float) sum, the same bits as the undeferred calls. */
typedef struct dn_fold_job {
  const double* partials; /* workspace + c */
  double* sums;           /* workspace: receives the folded doubles */
  float* out;
  int32_t n_blocks, c, accumulate, reserved;
} dn_fold_job;
int dn_bn_train_backward_finish_bias_deferred(const float* dy_a, int ld_a, int up_a, const float* dy_b, int ld_b, const float* y,
                                              const float* z, const float* mean, const float* var, const float* gamma, float eps,
                                              int relu, int h, int w, int images, int c, const double* sums, long norm_rows,
                                              float* dz, void* dz_sp /* may be NULL */, float sp_lift, double* bias_ws,
                                              size_t bias_ws_bytes, int* n_blocks, void* stream);
int dn_channel_sum_partial(const float* x, long rows, int c, int ld, double* sums, size_t sums_bytes, int* n_blocks, void* stream);
int dn_channel_sum_fold_multi(const dn_fold_job* jobs, int n_jobs, void* stream);

/* Backward of the decoder's nearest x2 upsample as a pass of its own (only needed where no
 * BatchNorm backward follows directly: the fusion on layer 4): out [n, h, w, c] dense = 2 x 2
 * block sums of g [n, 2h, 2w, c'] read with row stride ld. */
int dn_upsample2_sum(const float* g, int ld, int n_images, int h, int w, int c, float* out,
                     void* stream);

/* a[row][0..c) += b[row][0..c) */
int dn_add_rows(float* a, int ld_a, const float* b, int ld_b, long rows, int c, void* stream);

/* ---- DiscoGraph fusion, training mode ------------------------------------------------------ */

/* z1[p] += e[ego_image[p]] for every pair p (rows_per_image pixels of c channels each) */
int dn_pair_add_ego(float* z1, const float* e, const int* ego_image, int n_pairs,
                    int rows_per_image, int c, void* stream);
/* de[img] = sum of dz1[p] over the pairs listed for img: pairs[first[img] .. first[img + 1]) */
int dn_pair_sum_ego(const float* dz1, const int* first, const int* pairs, int n_images,
                    int rows_per_image, int c, float* de, void* stream);

/* Softmax over a scene-agent's neighbour list and the weighted sum (forward), per pixel:
 *   s_k = relu(z4[pair_k]);  w_k = exp(s_k) / sum_j exp(s_j);  fused = sum_k w_k * maps[nbr_k]
 * lists: for ego e in [0, n_egos): entries first[e] .. first[e+1) of (pair_index, map_image);
 * ego_out[e] = image of `fused` to write.  maps holds own and warped maps in one buffer. */
int dn_fuse_combine(const float* z4, const float* maps, const int* first, const int* pair_index,
                    const int* map_image, const int* ego_out, int n_egos, int hw, int c,
                    float* weights, float* fused, void* stream);
/* backward: dmaps[map_image_k] = w_k * dfused (assignment: every map is in one list),
 * dz4[pair_k] = w_k * (<dfused, map_k> - sum_j w_j <dfused, map_j>) * (z4 > 0) */
int dn_fuse_combine_backward(const float* dfused, int ld_df, const float* z4, const float* weights,
                             const float* maps, const int* first, const int* pair_index,
                             const int* map_image, const int* ego_out, int n_egos, int hw, int c,
                             float* dmaps, float* dz4, void* stream);

/* Backward of dn_warp_neighbors for a list of warps: the gradient of warp w (source image
 * src_image[w] in [0, n_src_images), 4x4 pose at poses + 16 * w, row-major, neighbour -> ego) goes
 * back through both bilinear passes and is ADDED to d_src[src_image[w]].  scratch: n_warps maps.
 * rigid != 0 (every pose a rotation + translation, which V2X poses are) on square maps: gather
 * form, deterministic, no atomics.  Otherwise: scatter with float atomics. */
int dn_warp_backward(const float* d_warped, const float* poses, const int* src_image, int n_warps,
                     int n_src_images, int h, int w, int c, int rigid, float* scratch, float* d_src,
                     void* stream);
/* forward over the same list (training keeps every warp, in list order) */
int dn_warp_list(const float* src, const float* poses, const int* src_image, int n_warps, int h,
                 int w, int c, float* warped, void* stream);

/* ---- loss and optimiser --------------------------------------------------------------------- */

/* Focal softmax classification loss + masked smooth-L1 localisation loss, forward and the
 * gradient w.r.t. the logits / box codes in one pass:
 *   cls [n, 2] logits, labels [n, 2] one-hot;  loc, targets [n, code] ; mask [n] (0/1 floats)
 *   loss_cls = sum_n alpha_t (1 - p_t)^gamma (-log p_t) / norm
 *   loss_loc = sum_n mask * sum_code smoothL1_sigma(loc - target) / norm
 * losses: 2 doubles (cls, loc), zeroed by the call. */
int dn_det_loss(const float* cls, const float* labels, const float* loc, const float* targets,
                const float* mask, long n, int code, float alpha, float gamma, float sigma,
                float norm, double* losses, float* dcls, float* dloc, void* stream);

/* Knowledge distillation term of CoDetModule.step (kd_flag = 1): for NHWC maps viewed as
 * [rows, c], nn.KLDivLoss(size_average=True)(log_softmax(student, 1), softmax(teacher, 1)) -- the
 * mean over ALL rows * c elements -- times kd_weight: pass scale = kd_weight / (rows * c).
 * *loss (double) is zeroed first if zero_loss, then the term is added; dstudent = d(term)/d(student). */
int dn_kd_kl_loss(const float* student, const float* teacher, long rows, int c, float scale,
                  double* loss, float* dstudent, int zero_loss, void* stream);

/* torch.optim.Adam (no amsgrad) on a flat parameter buffer; step counts from 1 */
int dn_adam_step(float* p, const float* g, float* m, float* v, long n, float lr, float beta1,
                 float beta2, float eps, float weight_decay, int step, void* stream);

#ifdef __cplusplus
}
#endif
#endif
