/*
 * disconet_seg.h -- C ABI of the segmentation variant of `--com disco`
 * (BASELINE.json configs[3]: "DiscoNet seg head, 5-agent, 256x256 BEV").
 *
 * The reference's seg path is upstream:coperception/models/seg/{SegModelBase,DiscoNet}.py +
 * upstream:coperception/utils/SegModule.py, reached by upstream:tools/seg/{train,test}_seg.py;
 * none of it is in the mount (/root/reference/coperception is an empty submodule directory,
 * /root/reference/.gitmodules:1-3) -- the only mounted mention of the task is
 * /root/reference/README.md:15 -- so every entry point cites the upstream call it replaces by path.
 *
 * The UNet's convolutions run on the SP conv engine of disconet_hip.h (dn_spconv2d), the
 * DiscoGraph fusion at the 512-channel bottleneck on dn_warp_neighbors + dn_conv2d (layer 1 of the
 * attention MLP) + dn_disco_fuse_tail.  This header adds the three ops that are new:
 * conventions (device pointers, caller-owned buffers, stream, int status) as disconet_hip.h.
 */
#ifndef DISCONET_SEG_H
#define DISCONET_SEG_H

#include "disconet_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* nn.MaxPool2d(2) of the Down blocks (upstream SegModelBase.py :: Down.maxpool_conv[0]) on a
 * split-planar tensor [n][ceil(c/16)][4][h][w] x 16 B -> [n][..][4][h/2][w/2]; h, w even.  The
 * (hi, lo) pair of the largest of the four values is copied: exact. */
int dn_sp_maxpool2(const void* src_sp, int n_images, int h, int w, int channels, void* dst_sp,
                   void* stream);

/* nn.Upsample(scale_factor=2, mode="bilinear", align_corners=True) of the Up blocks
 * (upstream SegModelBase.py :: Up.up), SP in -> SP out [n][..][4][2h][2w]; ATen's arithmetic:
 * src = dst * (in - 1) / (out - 1), fp32 lerp weights. */
int dn_sp_upsample2_bilinear(const void* src_sp, int n_images, int h, int w, int channels,
                             void* dst_sp, void* stream);

/* Label census for the mean reduction of nn.CrossEntropyLoss (ignore_index = -100):
 *   counts[0] = labels in [0, classes)  (the divisor of the mean),
 *   counts[1] = labels that are neither valid nor -100 (torch raises on these; they are treated as
 *               ignored here and reported so that the host can refuse them).
 * counts: 2 device int32, zeroed by the call. */
int dn_seg_label_count(const int32_t* labels, long pixels, int classes, int32_t* counts, void* stream);

/* Per-pixel cross entropy over `classes` logits (upstream SegModule.py :: step,
 * nn.CrossEntropyLoss): logits [pixels][ld] float32, labels [pixels] int32 (outside [0, classes) = ignored).
 *   loss_sum (device double, zeroed by the call) += sum over the live pixels of (logsumexp(z_p) - z_p[y_p])
 *   dlogits [pixels][ld] (may be NULL) = (softmax(z_p) - onehot(y_p)) * g, zero rows for ignored pixels;
 *   g = grad_scale, or grad_scale / counts[0] when `counts` (from dn_seg_label_count, may be NULL) is given:
 *   the reference's mean is over the NON-IGNORED pixels, loss = loss_sum / counts[0]. */
int dn_seg_ce_loss(const float* logits, const int32_t* labels, long pixels, int classes, int ld,
                   float grad_scale, const int32_t* counts, double* loss_sum, float* dlogits, void* stream);

/* ---- training forms on float32 NHWC maps (SegModule.step: the UNet's reverse pass; conventions of
 * disconet_train.h -- a gradient argument may be a channel slice of a wider map: pointer at its first
 * channel + the pixel stride ld in floats).  c % 4 == 0, 16-byte aligned buffers. ---- */
/* nn.MaxPool2d(2): x [n][h][w][c] -> y [n][h/2][w/2][c] */
int dn_maxpool2_nhwc(const float* x, int n_images, int h, int w, int c, float* y, void* stream);
/* its backward: dx [n][h][w][c] = dy routed to each window's FIRST maximum in scan order (ATen's recorded
 * index), zero elsewhere; every element written once (no zero fill, deterministic). */
int dn_maxpool2_nhwc_backward(const float* x, const float* dy, int ld_dy, int n_images, int h, int w, int c,
                              float* dx, void* stream);
/* nn.Upsample(x2, bilinear, align_corners=True): x [n][h][w][c] -> y [n][2h][2w][c] */
int dn_upsample2_bilinear_nhwc(const float* x, int n_images, int h, int w, int c, float* y, void* stream);
/* its backward in gather form: dy [n][2h][2w] (stride ld_dy) -> dx [n][h][w][c]; fixed summation order. */
int dn_upsample2_bilinear_nhwc_backward(const float* dy, int ld_dy, int n_images, int h, int w, int c,
                                        float* dx, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DISCONET_SEG_H */
