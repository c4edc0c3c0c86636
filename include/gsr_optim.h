/*
 * gsr_optim.h -- C ABI of the Adam step over the Gaussian parameter groups (part of libgsr_hip.so).
 * "Next" row 8f-1 of SURVEY.md (the fused step around the rasterizer).
 *
 * Replaces, for the training loop, the `self.optimizer.step()` of train.py:126 on the optimiser the reference builds at
 * scene/gaussian_model.py:155-164: torch.optim.Adam over six one-tensor groups, lr per group, betas (0.9, 0.999),
 * eps 1e-15, no weight decay, no amsgrad.  Same update, element for element:
 *     m = m + (g - m) * (1 - beta1);  v = v * beta2 + g * g * (1 - beta2)
 *     p = p - (lr / (1 - beta1^step)) * m / (sqrt(v) / sqrt(1 - beta2^step) + eps)
 * All groups go through ONE launch (28 bytes of traffic per element, nothing else): torch's fused path takes one
 * launch sequence per group and ~2x the time at 59 M floats.
 * Same conventions as gsr.h: device pointers, float32, caller-owned buffers, enqueued on `stream`, 0 = ok.
 */
#ifndef GSR_OPTIM_H
#define GSR_OPTIM_H
#include <stddef.h>
#include <stdint.h>
#include "gsr.h"
#ifdef __cplusplus
extern "C" {
#endif
#define GSR_ADAM_MAX_GROUPS 16
typedef struct {
    float *param;          /* [n] updated in place */
    const float *grad;     /* [n] */
    float *exp_avg;        /* [n] first moment, updated in place */
    float *exp_avg_sq;     /* [n] second moment, updated in place */
    int64_t n;
    float lr;
    int32_t step;          /* 1-based step count of this group AFTER this update (torch keeps it per tensor) */
} gsr_adam_group_t;
/* betas and eps are doubles: torch derives 1 - beta and the bias corrections in double (1 - 0.999f is 1.3e-5 off 0.001). */
int32_t gsr_adam_step(gsr_stream_t stream, int32_t n_groups, const gsr_adam_group_t *groups /* host array */, double beta1,
                      double beta2, double eps);
#ifdef __cplusplus
}
#endif
#endif
