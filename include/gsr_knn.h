/*
 * gsr_knn.h -- C ABI of the 3-nearest-neighbour distance used to initialise Gaussian scales
 * (part of libgsr_hip.so).  "Next" row 8f-2 of SURVEY.md.
 *
 * Replaces the reference's second native dependency, `simple_knn._C.distCUDA2`
 * (.gitmodules:1-3, an empty directory in the snapshot), called once at scene/gaussian_model.py:134:
 *     dist2 = torch.clamp_min(distCUDA2(points), 0.0000001)
 * mean_dist2[i] = mean of the three smallest squared distances from point i to the other points
 * (exact; coincident points count with distance 0; fewer than 4 points give +inf).
 * Same conventions as gsr.h: device pointers, float32, caller-owned buffers, enqueued on `stream`, 0 = ok.
 */
#ifndef GSR_KNN_H
#define GSR_KNN_H
#include <stddef.h>
#include <stdint.h>
#include "gsr.h"
#ifdef __cplusplus
extern "C" {
#endif
int32_t gsr_knn_workspace(int32_t N, size_t *bytes);
int32_t gsr_knn_mean_dist2(gsr_stream_t stream, int32_t N, const float *points /*[N,3]*/, float *mean_dist2 /*[N]*/,
                           void *ws, size_t ws_bytes);
#ifdef __cplusplus
}
#endif
#endif
