// ssim_loss.hip -- fused training loss  (1-lambda) * L1 + lambda * (1 - SSIM)  and its gradient.
//
// "Next" row 8f-1 of SURVEY.md: the loss that produces dL/dimage for the rasterizer backward.  Follows the
// reference's utils/loss_utils.py:17-63 (l1_loss; ssim: 11x11 Gaussian window sigma 1.5 built as the
// outer product of a normalised 1-D kernel, zero padding 5, C1 = 0.01^2, C2 = 0.03^2, mean over all
// C*H*W map entries) combined as in train.py:91-92.  The reference runs it as five grouped 11x11
// convolutions + elementwise ops (and their autograd); here ONE kernel per direction:
//   forward : 16x16 tile + 5-pixel halo of both images staged in LDS, separable 11-tap blur of the five
//             moments (x, y, x^2, y^2, xy), SSIM map value, the three partial-derivative maps
//             (df/dmu1, df/dm11, df/dm12) written for the backward, deterministic two-stage sum.
//   backward: separable blur of the three derivative maps (zero outside the image) and
//             dL/dx = (1-lambda)/N sign(x-y) - lambda/N (G*A + 2x G*B + y G*C), scaled by the upstream scalar.
// HBM-bound: forward reads 8 and writes 12 bytes per map entry, backward reads 20 and writes 4.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gsr_internal.h"

namespace gsr {

#define SSIM_R 5
#define SSIM_K 11
#define SSIM_T 16
#define SSIM_HALO (SSIM_T + 2 * SSIM_R)   // 26
#define SSIM_C1 0.0001f                   // 0.01^2
#define SSIM_C2 0.0009f                   // 0.03^2

struct SsimWeights { float g[SSIM_K]; };

// float32 1-D kernel exactly as utils/loss_utils.py:23-25 builds it: exp() in double, stored as float32,
// divided by the float32 sum
static SsimWeights make_weights() {
    SsimWeights w;
    float s = 0.f;
    for (int i = 0; i < SSIM_K; i++) {
        const double d = (double)(i - SSIM_K / 2);
        w.g[i] = (float)exp(-(d * d) / (2.0 * 1.5 * 1.5));
        s += w.g[i];
    }
    for (int i = 0; i < SSIM_K; i++) w.g[i] = w.g[i] / s;
    return w;
}

__global__ __launch_bounds__(256) void l1_ssim_fwd_kernel(int C, int H, int W, const float *__restrict__ img,
                                                          const float *__restrict__ gt, SsimWeights wts,
                                                          float *__restrict__ dmaps /*[3][C][H][W]*/,
                                                          float *__restrict__ partial /*[blocks][2]*/) {
    __shared__ float sx[SSIM_HALO][SSIM_HALO + 1], sy[SSIM_HALO][SSIM_HALO + 1];
    __shared__ float hz[5][SSIM_HALO][SSIM_T + 1];
    __shared__ float red[2][4];
    const int tid = threadIdx.x, lx = tid & 15, ly = tid >> 4;
    const int c = blockIdx.z, x0 = blockIdx.x * SSIM_T, y0 = blockIdx.y * SSIM_T;
    const size_t plane = (size_t)H * W, base = (size_t)c * plane;
    for (int i = tid; i < SSIM_HALO * SSIM_HALO; i += 256) {
        const int r = i / SSIM_HALO, q = i - r * SSIM_HALO;
        const int y = y0 + r - SSIM_R, x = x0 + q - SSIM_R;
        const bool in = x >= 0 && x < W && y >= 0 && y < H;
        sx[r][q] = in ? img[base + (size_t)y * W + x] : 0.f;        // zero padding (conv2d padding = 5)
        sy[r][q] = in ? gt[base + (size_t)y * W + x] : 0.f;
    }
    __syncthreads();
    for (int i = tid; i < SSIM_HALO * SSIM_T; i += 256) {            // horizontal pass on 26 rows x 16 columns
        const int r = i >> 4, q = i & 15;
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f, a4 = 0.f;
#pragma unroll
        for (int k = 0; k < SSIM_K; k++) {
            const float xv = sx[r][q + k], yv = sy[r][q + k], w = wts.g[k];
            a0 += w * xv; a1 += w * yv; a2 += w * xv * xv; a3 += w * yv * yv; a4 += w * xv * yv;
        }
        hz[0][r][q] = a0; hz[1][r][q] = a1; hz[2][r][q] = a2; hz[3][r][q] = a3; hz[4][r][q] = a4;
    }
    __syncthreads();
    float m1 = 0.f, m2 = 0.f, s11 = 0.f, s22 = 0.f, s12 = 0.f;
#pragma unroll
    for (int k = 0; k < SSIM_K; k++) {
        const float w = wts.g[k];
        m1 += w * hz[0][ly + k][lx]; m2 += w * hz[1][ly + k][lx]; s11 += w * hz[2][ly + k][lx];
        s22 += w * hz[3][ly + k][lx]; s12 += w * hz[4][ly + k][lx];
    }
    const int x = x0 + lx, y = y0 + ly;
    const bool in = x < W && y < H;
    float l1 = 0.f, ss = 0.f;
    if (in) {
        const float a1 = 2.f * m1 * m2 + SSIM_C1, sig12 = s12 - m1 * m2, a2 = 2.f * sig12 + SSIM_C2;
        const float b1 = m1 * m1 + m2 * m2 + SSIM_C1, b2 = (s11 - m1 * m1) + (s22 - m2 * m2) + SSIM_C2;
        const float invD = 1.f / (b1 * b2);
        const float f = a1 * a2 * invD;
        const size_t p = base + (size_t)y * W + x, CHW = (size_t)C * plane;
        dmaps[p] = (2.f * m2 * (a2 - a1) - f * 2.f * m1 * (b2 - b1)) * invD;   // d f / d mu1 (total)
        dmaps[CHW + p] = -f / b2;                                               // d f / d E[x^2]
        dmaps[2 * CHW + p] = 2.f * a1 * invD;                                   // d f / d E[xy]
        ss = f;
        l1 = fabsf(sx[ly + SSIM_R][lx + SSIM_R] - sy[ly + SSIM_R][lx + SSIM_R]);
    }
    // block sums (fixed order: deterministic)
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) { l1 += __shfl_xor(l1, m); ss += __shfl_xor(ss, m); }
    if ((tid & 63) == 0) { red[0][tid >> 6] = l1; red[1][tid >> 6] = ss; }
    __syncthreads();
    if (tid == 0) {
        const size_t b = ((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
        partial[2 * b] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
        partial[2 * b + 1] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
    }
}

// out[0] = loss, out[1] = L1 mean, out[2] = SSIM mean
__global__ __launch_bounds__(1024) void l1_ssim_finish_kernel(int nblocks, double inv_n, float lambda,
                                                              const float *__restrict__ partial, float *__restrict__ out) {
    __shared__ double r0[16], r1[16];
    double a = 0.0, b = 0.0;
    for (int i = threadIdx.x; i < nblocks; i += 1024) { a += (double)partial[2 * i]; b += (double)partial[2 * i + 1]; }
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) { a += __shfl_xor(a, m); b += __shfl_xor(b, m); }
    if ((threadIdx.x & 63) == 0) { r0[threadIdx.x >> 6] = a; r1[threadIdx.x >> 6] = b; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double sa = 0.0, sb = 0.0;
        for (int i = 0; i < 16; i++) { sa += r0[i]; sb += r1[i]; }
        const float l1 = (float)(sa * inv_n), ssim = (float)(sb * inv_n);
        out[0] = (1.f - lambda) * l1 + lambda * (1.f - ssim);
        out[1] = l1;
        out[2] = ssim;
    }
}

__global__ __launch_bounds__(256) void l1_ssim_bwd_kernel(int C, int H, int W, const float *__restrict__ img,
                                                          const float *__restrict__ gt, SsimWeights wts,
                                                          const float *__restrict__ dmaps, float lambda, float inv_n,
                                                          const float *__restrict__ grad_loss /*[1] or null*/,
                                                          float *__restrict__ grad_img) {
    __shared__ float sm[3][SSIM_HALO][SSIM_HALO + 1];
    __shared__ float hz[3][SSIM_HALO][SSIM_T + 1];
    const int tid = threadIdx.x, lx = tid & 15, ly = tid >> 4;
    const int c = blockIdx.z, x0 = blockIdx.x * SSIM_T, y0 = blockIdx.y * SSIM_T;
    const size_t plane = (size_t)H * W, base = (size_t)c * plane, CHW = (size_t)C * plane;
    for (int i = tid; i < SSIM_HALO * SSIM_HALO; i += 256) {
        const int r = i / SSIM_HALO, q = i - r * SSIM_HALO;
        const int y = y0 + r - SSIM_R, x = x0 + q - SSIM_R;
        const bool in = x >= 0 && x < W && y >= 0 && y < H;
        const size_t p = base + (size_t)(in ? y : 0) * W + (in ? x : 0);
        sm[0][r][q] = in ? dmaps[p] : 0.f;
        sm[1][r][q] = in ? dmaps[CHW + p] : 0.f;
        sm[2][r][q] = in ? dmaps[2 * CHW + p] : 0.f;
    }
    __syncthreads();
    for (int i = tid; i < SSIM_HALO * SSIM_T; i += 256) {
        const int r = i >> 4, q = i & 15;
        float a0 = 0.f, a1 = 0.f, a2 = 0.f;
#pragma unroll
        for (int k = 0; k < SSIM_K; k++) {
            const float w = wts.g[k];
            a0 += w * sm[0][r][q + k]; a1 += w * sm[1][r][q + k]; a2 += w * sm[2][r][q + k];
        }
        hz[0][r][q] = a0; hz[1][r][q] = a1; hz[2][r][q] = a2;
    }
    __syncthreads();
    float gA = 0.f, gB = 0.f, gC = 0.f;
#pragma unroll
    for (int k = 0; k < SSIM_K; k++) {
        const float w = wts.g[k];
        gA += w * hz[0][ly + k][lx]; gB += w * hz[1][ly + k][lx]; gC += w * hz[2][ly + k][lx];
    }
    const int x = x0 + lx, y = y0 + ly;
    if (x < W && y < H) {
        const size_t p = base + (size_t)y * W + x;
        const float xv = img[p], yv = gt[p];
        const float d = xv - yv;
        const float sgn = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);           // torch.abs backward: sign(0) = 0
        const float up = grad_loss ? grad_loss[0] : 1.f;
        grad_img[p] = up * inv_n * ((1.f - lambda) * sgn - lambda * (gA + 2.f * xv * gB + yv * gC));
    }
}

hipError_t launch_l1_ssim_forward(int C, int H, int W, const float *img, const float *gt, float lambda, float *dmaps,
                                  float *partial, float *out, hipStream_t s) {
    const SsimWeights w = make_weights();
    const dim3 grid((W + SSIM_T - 1) / SSIM_T, (H + SSIM_T - 1) / SSIM_T, C);
    hipLaunchKernelGGL(l1_ssim_fwd_kernel, grid, dim3(256), 0, s, C, H, W, img, gt, w, dmaps, partial);
    const int nblocks = (int)(grid.x * grid.y * grid.z);
    hipLaunchKernelGGL(l1_ssim_finish_kernel, dim3(1), dim3(1024), 0, s, nblocks, 1.0 / ((double)C * H * W), lambda, partial, out);
    return hipGetLastError();
}

hipError_t launch_l1_ssim_backward(int C, int H, int W, const float *img, const float *gt, float lambda, const float *dmaps,
                                   const float *grad_loss, float *grad_img, hipStream_t s) {
    const SsimWeights w = make_weights();
    const dim3 grid((W + SSIM_T - 1) / SSIM_T, (H + SSIM_T - 1) / SSIM_T, C);
    hipLaunchKernelGGL(l1_ssim_bwd_kernel, grid, dim3(256), 0, s, C, H, W, img, gt, w, dmaps, lambda,
                       (float)(1.0 / ((double)C * H * W)), grad_loss, grad_img);
    return hipGetLastError();
}

}  // namespace gsr
