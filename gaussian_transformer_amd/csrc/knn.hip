// knn.hip -- mean squared distance to the 3 nearest neighbours of every point.
//
// "Next" row 8f-2 of SURVEY.md: the reference's second native dependency, `simple_knn._C.distCUDA2`
// (submodules/simple-knn, an EMPTY directory in the snapshot; used once, at scene/gaussian_model.py:134, to
// initialise the Gaussian scales from the SfM cloud: scales = log(sqrt(clamp_min(dist2, 1e-7)))).
// Semantics restated from that call site: for point i, the mean of the three smallest squared Euclidean
// distances to the OTHER points (index != i; coincident points count with distance 0).  Exact, not approximate.
//
// gfx950 shape: points are ordered along a 30-bit Morton curve (rocPRIM radix sort), cut into boxes of 1024
// consecutive points with their bounding boxes; one lane per point first looks at its 2x3 curve neighbours,
// then scans only the boxes whose distance to the point is below the current 3rd-best distance.  Init-time
// only (tens of thousands to ~1 M points): ~1e9 box tests + a few 1024-point scans per point.
#include <cstring>
#include <rocprim/rocprim.hpp>

#include <float.h>
#include <stdint.h>

#include "gsr_internal.h"

namespace gsr {

#define KNN_BOX 1024

struct KnnView {
    float *bbox;            // [6] min xyz, max xyz (as ordered uints during the reduction)
    uint32_t *codes, *codes_sorted, *idx_sorted;
    float *box_min, *box_max;   // [nbox][3]
    void *sort_temp;
    size_t sort_temp_bytes, total_bytes;
};

static hipError_t knn_sort_temp(int N, size_t *bytes) {
    size_t tb = 0;
    hipError_t e = rocprim::radix_sort_pairs(nullptr, tb, (const uint32_t *)nullptr, (uint32_t *)nullptr,
                                             rocprim::counting_iterator<uint32_t>(0), (uint32_t *)nullptr,
                                             (size_t)(N > 0 ? N : 1), 0u, 30u, (hipStream_t)0, false);
    *bytes = tb;
    return e;
}

static KnnView carve_knn(void *base, int N, size_t sort_tb) {
    KnnView v;
    const size_t n = (size_t)(N > 0 ? N : 1), nbox = (n + KNN_BOX - 1) / KNN_BOX;
    char *p = (char *)base;
    size_t off = 0;
    auto take = [&](size_t bytes) { char *r = p ? p + off : nullptr; off += align_up(bytes); return r; };
    v.bbox = (float *)take(6 * sizeof(float));
    v.codes = (uint32_t *)take(n * 4); v.codes_sorted = (uint32_t *)take(n * 4); v.idx_sorted = (uint32_t *)take(n * 4);
    v.box_min = (float *)take(nbox * 3 * 4); v.box_max = (float *)take(nbox * 3 * 4);
    v.sort_temp = take(sort_tb); v.sort_temp_bytes = sort_tb;
    v.total_bytes = off;
    return v;
}

// order-preserving float <-> uint mapping for atomicMin/Max on floats of either sign
__device__ __forceinline__ uint32_t f2ord(float f) { const uint32_t u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
__device__ __forceinline__ float ord2f(uint32_t u) { return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u); }

__global__ void knn_bbox_init_kernel(uint32_t *bbox) {
    if (threadIdx.x < 3) bbox[threadIdx.x] = 0xffffffffu;
    else if (threadIdx.x < 6) bbox[threadIdx.x] = 0u;
}

__global__ __launch_bounds__(256) void knn_bbox_kernel(int N, const float *__restrict__ pts, uint32_t *__restrict__ bbox) {
    float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    for (int i = blockIdx.x * 256 + threadIdx.x; i < N; i += gridDim.x * 256)
#pragma unroll
        for (int k = 0; k < 3; k++) { const float v = pts[3 * (size_t)i + k]; mn[k] = fminf(mn[k], v); mx[k] = fmaxf(mx[k], v); }
#pragma unroll
    for (int k = 0; k < 3; k++) {
#pragma unroll
        for (int m = 32; m > 0; m >>= 1) { mn[k] = fminf(mn[k], __shfl_xor(mn[k], m)); mx[k] = fmaxf(mx[k], __shfl_xor(mx[k], m)); }
        if ((threadIdx.x & 63) == 0) { atomicMin(&bbox[k], f2ord(mn[k])); atomicMax(&bbox[3 + k], f2ord(mx[k])); }
    }
}

__device__ __forceinline__ uint32_t spread10(uint32_t x) {   // 10 bits -> every third bit
    x &= 0x3ffu;
    x = (x | (x << 16)) & 0x030000ffu;
    x = (x | (x << 8)) & 0x0300f00fu;
    x = (x | (x << 4)) & 0x030c30c3u;
    x = (x | (x << 2)) & 0x09249249u;
    return x;
}

__global__ __launch_bounds__(256) void knn_morton_kernel(int N, const float *__restrict__ pts, const uint32_t *__restrict__ bbox,
                                                         uint32_t *__restrict__ codes) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    uint32_t c = 0;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const float lo = ord2f(bbox[k]), hi = ord2f(bbox[3 + k]);
        const float ext = hi - lo;
        float t = ext > 0.f ? (pts[3 * (size_t)i + k] - lo) / ext : 0.f;
        t = fminf(fmaxf(t, 0.f), 1.f);
        c |= spread10((uint32_t)(t * 1023.f)) << k;
    }
    codes[i] = c;
}

__global__ __launch_bounds__(256) void knn_boxes_kernel(int N, const float *__restrict__ pts, const uint32_t *__restrict__ idx_sorted,
                                                        float *__restrict__ box_min, float *__restrict__ box_max) {
    __shared__ float smn[4][3], smx[4][3];
    const int b = blockIdx.x;
    float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    for (int j = threadIdx.x; j < KNN_BOX; j += 256) {
        const int s = b * KNN_BOX + j;
        if (s < N) {
            const size_t i = idx_sorted[s];
#pragma unroll
            for (int k = 0; k < 3; k++) { const float v = pts[3 * i + k]; mn[k] = fminf(mn[k], v); mx[k] = fmaxf(mx[k], v); }
        }
    }
#pragma unroll
    for (int k = 0; k < 3; k++) {
#pragma unroll
        for (int m = 32; m > 0; m >>= 1) { mn[k] = fminf(mn[k], __shfl_xor(mn[k], m)); mx[k] = fmaxf(mx[k], __shfl_xor(mx[k], m)); }
        if ((threadIdx.x & 63) == 0) { smn[threadIdx.x >> 6][k] = mn[k]; smx[threadIdx.x >> 6][k] = mx[k]; }
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        const int k = threadIdx.x;
        box_min[3 * b + k] = fminf(fminf(smn[0][k], smn[1][k]), fminf(smn[2][k], smn[3][k]));
        box_max[3 * b + k] = fmaxf(fmaxf(smx[0][k], smx[1][k]), fmaxf(smx[2][k], smx[3][k]));
    }
}

__device__ __forceinline__ void keep3(float d, float best[3]) {   // best[0] <= best[1] <= best[2]
    if (d < best[2]) {
        if (d < best[1]) {
            best[2] = best[1];
            if (d < best[0]) { best[1] = best[0]; best[0] = d; } else best[1] = d;
        } else best[2] = d;
    }
}

__global__ __launch_bounds__(256) void knn_query_kernel(int N, int nbox, const float *__restrict__ pts,
                                                        const uint32_t *__restrict__ idx_sorted, const float *__restrict__ box_min,
                                                        const float *__restrict__ box_max, float *__restrict__ out) {
    const int s = blockIdx.x * 256 + threadIdx.x;     // position along the Morton curve
    if (s >= N) return;
    const size_t i = idx_sorted[s];
    const float px = pts[3 * i], py = pts[3 * i + 1], pz = pts[3 * i + 2];
    float best[3] = {FLT_MAX, FLT_MAX, FLT_MAX};
    for (int t = max(0, s - 3); t <= min(N - 1, s + 3); t++) {      // curve neighbours give a first bound
        if (t == s) continue;
        const size_t j = idx_sorted[t];
        const float dx = pts[3 * j] - px, dy = pts[3 * j + 1] - py, dz = pts[3 * j + 2] - pz;
        keep3(dx * dx + dy * dy + dz * dz, best);
    }
    // the curve neighbours only provide a rejection radius; the boxes are then scanned from scratch
    // (each neighbour lies in some box and must not be counted twice)
    const float reject = best[2];
    best[0] = best[1] = best[2] = FLT_MAX;
    for (int b = 0; b < nbox; b++) {
        const float ex = fmaxf(fmaxf(box_min[3 * b] - px, px - box_max[3 * b]), 0.f);
        const float ey = fmaxf(fmaxf(box_min[3 * b + 1] - py, py - box_max[3 * b + 1]), 0.f);
        const float ez = fmaxf(fmaxf(box_min[3 * b + 2] - pz, pz - box_max[3 * b + 2]), 0.f);
        const float bd = ex * ex + ey * ey + ez * ez;
        if (bd > reject || bd > best[2]) continue;                  // nothing in this box can enter the best three
        const int t0 = b * KNN_BOX, t1 = min(N, t0 + KNN_BOX);
        for (int t = t0; t < t1; t++) {
            if (t == s) continue;
            const size_t j = idx_sorted[t];
            const float dx = pts[3 * j] - px, dy = pts[3 * j + 1] - py, dz = pts[3 * j + 2] - pz;
            keep3(dx * dx + dy * dy + dz * dz, best);
        }
    }
    out[i] = (best[0] + best[1] + best[2]) / 3.f;
}

hipError_t knn_workspace_bytes(int N, size_t *bytes) {
    size_t tb = 0;
    hipError_t e = knn_sort_temp(N, &tb);
    *bytes = carve_knn(nullptr, N, tb).total_bytes;
    return e;
}

hipError_t launch_knn(int N, const float *pts, float *out, void *ws, hipStream_t s) {
    if (N <= 0) return hipSuccess;
    size_t tb = 0;
    hipError_t e = knn_sort_temp(N, &tb);
    if (e != hipSuccess) return e;
    KnnView v = carve_knn(ws, N, tb);
    const int nbox = (N + KNN_BOX - 1) / KNN_BOX;
    hipLaunchKernelGGL(knn_bbox_init_kernel, dim3(1), dim3(64), 0, s, (uint32_t *)v.bbox);
    hipLaunchKernelGGL(knn_bbox_kernel, dim3(min((N + 255) / 256, 1024)), dim3(256), 0, s, N, pts, (uint32_t *)v.bbox);
    hipLaunchKernelGGL(knn_morton_kernel, dim3((N + 255) / 256), dim3(256), 0, s, N, pts, (const uint32_t *)v.bbox, v.codes);
    size_t stb = v.sort_temp_bytes;
    e = rocprim::radix_sort_pairs(v.sort_temp, stb, (const uint32_t *)v.codes, v.codes_sorted, rocprim::counting_iterator<uint32_t>(0),
                                  v.idx_sorted, (size_t)N, 0u, 30u, s, false);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(knn_boxes_kernel, dim3(nbox), dim3(256), 0, s, N, pts, v.idx_sorted, v.box_min, v.box_max);
    hipLaunchKernelGGL(knn_query_kernel, dim3((N + 255) / 256), dim3(256), 0, s, N, nbox, pts, v.idx_sorted, v.box_min, v.box_max, out);
    return hipGetLastError();
}

}  // namespace gsr
