// adam.hip -- one launch for the Adam step of all parameter groups (include/gsr_optim.h).  Pure streaming:
// 16 bytes read + 12 written per element, float4 wide where the group's length and pointers allow.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "../../include/gsr_optim.h"
#include "gsr_internal.h"

namespace gsr {

struct AdamGroupDev {
    float *p;
    const float *g;
    float *m, *v;
    long long n;
    float step_size, inv_sqrt_bc2;
    unsigned first_block;        // the group's blocks are [first_block, next group's first_block)
    int vec4;
};
struct AdamArgs {
    AdamGroupDev grp[GSR_ADAM_MAX_GROUPS];
    int n_groups;
    float beta2, omb1, omb2, eps;      // 1 - beta computed in double on the host (1 - 0.999f is 4.7e-5 off 0.001)
};

#define ADAM_PER_BLOCK (256 * 4 * 4)      // 256 threads x 4 float4

__device__ __forceinline__ void adam_one(float &p, float g, float &m, float &v, float omb1, float b2, float omb2, float eps, float step_size,
                                         float isb2) {
    m = m + (g - m) * omb1;
    v = v * b2 + g * g * omb2;
    const float denom = sqrtf(v) * isb2 + eps;
    p = p - step_size * (m / denom);
}

__global__ __launch_bounds__(256) void adam_kernel(AdamArgs a) {
    int gi = 0;
#pragma unroll 1
    for (int k = 1; k < a.n_groups; k++)
        if (blockIdx.x >= a.grp[k].first_block) gi = k;          // block-uniform
    const AdamGroupDev &G = a.grp[gi];
    const long long base = (long long)(blockIdx.x - G.first_block) * ADAM_PER_BLOCK;
    if (G.vec4) {
#pragma unroll
        for (int it = 0; it < 4; it++) {
            const long long i = base + ((long long)it * 256 + threadIdx.x) * 4;
            if (i + 3 < G.n) {
                float4 p = *reinterpret_cast<float4 *>(G.p + i), m = *reinterpret_cast<float4 *>(G.m + i), v = *reinterpret_cast<float4 *>(G.v + i);
                const float4 g = *reinterpret_cast<const float4 *>(G.g + i);
                adam_one(p.x, g.x, m.x, v.x, a.omb1, a.beta2, a.omb2, a.eps, G.step_size, G.inv_sqrt_bc2);
                adam_one(p.y, g.y, m.y, v.y, a.omb1, a.beta2, a.omb2, a.eps, G.step_size, G.inv_sqrt_bc2);
                adam_one(p.z, g.z, m.z, v.z, a.omb1, a.beta2, a.omb2, a.eps, G.step_size, G.inv_sqrt_bc2);
                adam_one(p.w, g.w, m.w, v.w, a.omb1, a.beta2, a.omb2, a.eps, G.step_size, G.inv_sqrt_bc2);
                *reinterpret_cast<float4 *>(G.p + i) = p; *reinterpret_cast<float4 *>(G.m + i) = m; *reinterpret_cast<float4 *>(G.v + i) = v;
            } else {
                for (long long j = i; j < G.n && j < i + 4; j++) {
                    float p = G.p[j], m = G.m[j], v = G.v[j];
                    adam_one(p, G.g[j], m, v, a.omb1, a.beta2, a.omb2, a.eps, G.step_size, G.inv_sqrt_bc2);
                    G.p[j] = p; G.m[j] = m; G.v[j] = v;
                }
            }
        }
    } else {
        for (int it = 0; it < 16; it++) {
            const long long j = base + (long long)it * 256 + threadIdx.x;
            if (j < G.n) {
                float p = G.p[j], m = G.m[j], v = G.v[j];
                adam_one(p, G.g[j], m, v, a.omb1, a.beta2, a.omb2, a.eps, G.step_size, G.inv_sqrt_bc2);
                G.p[j] = p; G.m[j] = m; G.v[j] = v;
            }
        }
    }
}

hipError_t launch_adam(int n_groups, const gsr_adam_group_t *groups, double beta1, double beta2, double eps, hipStream_t s) {
    AdamArgs a;
    a.n_groups = 0; a.beta2 = (float)beta2; a.omb1 = (float)(1.0 - beta1); a.omb2 = (float)(1.0 - beta2); a.eps = (float)eps;
    unsigned blocks = 0;
    for (int k = 0; k < n_groups; k++) {
        const gsr_adam_group_t &h = groups[k];
        if (h.n <= 0) continue;
        AdamGroupDev &d = a.grp[a.n_groups++];
        d.p = h.param; d.g = h.grad; d.m = h.exp_avg; d.v = h.exp_avg_sq; d.n = h.n;
        const double bc1 = 1.0 - pow(beta1, (double)h.step), bc2 = 1.0 - pow(beta2, (double)h.step);
        d.step_size = (float)((double)h.lr / bc1);
        d.inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
        d.first_block = blocks;
        d.vec4 = ((((uintptr_t)h.param | (uintptr_t)h.grad | (uintptr_t)h.exp_avg | (uintptr_t)h.exp_avg_sq) & 15) == 0) ? 1 : 0;
        blocks += (unsigned)((h.n + ADAM_PER_BLOCK - 1) / ADAM_PER_BLOCK);
    }
    if (blocks == 0) return hipSuccess;
    hipLaunchKernelGGL(adam_kernel, dim3(blocks), dim3(256), 0, s, a);
    return hipGetLastError();
}

}  // namespace gsr
