/*
 * gsr_loss.h -- C ABI of the fused training loss (part of libgsr_hip.so).  "Next" row 8f-1 of SURVEY.md.
 *
 * Replaces, for the train step of the reference (train.py:91-92),
 *     Ll1  = l1_loss(image, gt_image)                                   utils/loss_utils.py:17-18
 *     loss = (1 - lambda_dssim) * Ll1 + lambda_dssim * (1 - ssim(image, gt_image))   utils/loss_utils.py:33-63
 * (five grouped 11x11 convolutions + elementwise ops + their autograd, ~11.6 ms at 3x1080x1920 on MI355X)
 * with one fused kernel per direction.  Same conventions as gsr.h: device pointers, float32, caller-owned
 * buffers, work enqueued on `stream`, 0 = ok.
 */
#ifndef GSR_LOSS_H
#define GSR_LOSS_H
#include <stddef.h>
#include <stdint.h>
#include "gsr.h"
#ifdef __cplusplus
extern "C" {
#endif

/* bytes of the workspace written by the forward and read by the backward */
int32_t gsr_l1_ssim_workspace(int32_t C, int32_t H, int32_t W, size_t *bytes);

/* img, gt: [C,H,W].  out3 (device, 3 floats): loss, mean |img-gt|, mean SSIM. */
int32_t gsr_l1_ssim_forward(gsr_stream_t stream, int32_t C, int32_t H, int32_t W, const float *img, const float *gt,
                            float lambda_dssim, float *out3, void *ws, size_t ws_bytes);

/* grad_img [C,H,W] = grad_loss[0] * d loss / d img  (grad_loss: device scalar, NULL means 1). */
int32_t gsr_l1_ssim_backward(gsr_stream_t stream, int32_t C, int32_t H, int32_t W, const float *img, const float *gt,
                             float lambda_dssim, const float *grad_loss, const void *ws, size_t ws_bytes, float *grad_img);

#ifdef __cplusplus
}
#endif
#endif
