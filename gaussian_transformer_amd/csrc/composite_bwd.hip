// composite_bwd.hip -- reverse compositing (S10): per pixel back-to-front over the tile's
// depth-sorted splat list, producing dL/d{rgb, mean2D, conic, opacity} per Gaussian.
//
// CDNA4 shape (v1): same decomposition as the forward -- four independent wave64s per 16x16
// tile, one per 8x8 quadrant, wave-private LDS staging, no workgroup barrier.  All 64 lanes of a
// wave visit the same splat at the same step, so the nine partial gradients are reduced across
// the wave in registers (DPP quad_perm / row_half_mirror / row_mirror inside each 16-lane row,
// then two cross-row exchanges) and leave as ONE 9-lane global_atomic_add_f32 into the splat's
// 64-byte accumulator row: one memory-side atomic request per (quadrant, splat) instead of
// 9 x 64.  Splats that no pixel of the quadrant blends are skipped before the reduction.
#include "gsr_device.h"
#include "gsr_internal.h"

namespace gsr {

#define LOG2E 1.4426950408889634f

__device__ __forceinline__ int xcd_band_tile_b(int b, int nblocks_padded) {
    const int chunk = nblocks_padded >> 3;
    return (b & 7) * chunk + (b >> 3);
}

template <int CTRL>
__device__ __forceinline__ float dpp_add(float v) {
    const int moved = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true);
    return v + __int_as_float(moved);
}
// sum over the 16 lanes of each DPP row, result in every lane of the row
__device__ __forceinline__ float row_allreduce(float v) {
    v = dpp_add<0xB1>(v);   // quad_perm [1,0,3,2]
    v = dpp_add<0x4E>(v);   // quad_perm [2,3,0,1]
    v = dpp_add<0x141>(v);  // row_half_mirror
    v = dpp_add<0x140>(v);  // row_mirror
    return v;
}

__global__ __launch_bounds__(256) void composite_bwd_kernel(CompositeBwdArgs a, int nblocks_padded) {
    __shared__ float4 stage[4][64 * 3];
    __shared__ uint32_t stage_id[4][64];
    const int T = a.gridx * a.gridy;
    const int tile = xcd_band_tile_b(blockIdx.x, nblocks_padded);
    if (tile >= T) return;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int tx = tile % a.gridx, ty = tile / a.gridx;
    const int x = tx * GSR_TILE + (wave & 1) * 8 + (lane & 7);
    const int y = ty * GSR_TILE + (wave >> 1) * 8 + (lane >> 3);
    const bool inside = x < a.W && y < a.H;
    const float fx = (float)x, fy = (float)y;
    const uint2 range = a.ranges[tile];
    float4 *my = stage[wave];
    uint32_t *my_id = stage_id[wave];
    const float4 *rec4 = reinterpret_cast<const float4 *>(a.rec);
    const size_t pix = (size_t)(inside ? y : 0) * a.W + (inside ? x : 0), HW = (size_t)a.W * a.H;

    const float Tfinal = inside ? a.final_T[pix] : 1.f;
    const int last = inside ? (int)a.n_contrib[pix] : 0;
    const float d0 = inside ? a.dL_dpix[pix] : 0.f, d1 = inside ? a.dL_dpix[HW + pix] : 0.f,
                d2 = inside ? a.dL_dpix[2 * HW + pix] : 0.f;
    const float bg_dot = a.bg[0] * d0 + a.bg[1] * d1 + a.bg[2] * d2;
    const float halfW = 0.5f * (float)a.W, halfH = 0.5f * (float)a.H;

    int max_last = last;
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) max_last = max(max_last, __shfl_xor(max_last, m));
    if (max_last == 0) return;   // wave-uniform

    float Tr = Tfinal, acc0 = 0.f, acc1 = 0.f, acc2 = 0.f, last_alpha = 0.f, lc0 = 0.f, lc1 = 0.f, lc2 = 0.f;
    const int k16 = lane & 15;

    for (int base = ((max_last - 1) >> 6) << 6; base >= 0; base -= 64) {
        const int cnt = min(64, max_last - base);
        __builtin_amdgcn_wave_barrier();
        if (lane < cnt) {
            const uint32_t g = a.point_list[range.x + base + lane];
            my[lane * 3 + 0] = rec4[3 * (size_t)g];
            my[lane * 3 + 1] = rec4[3 * (size_t)g + 1];
            my[lane * 3 + 2] = make_float4(a.rec[GSR_REC_FLOATS * (size_t)g + 8], 0.f, 0.f, 0.f);
            my_id[lane] = g;
        }
        __builtin_amdgcn_wave_barrier();
        for (int j = cnt - 1; j >= 0; j--) {
            const float4 r0 = my[j * 3 + 0], r1 = my[j * 3 + 1];
            const float cb = reinterpret_cast<const float *>(my)[j * 12 + 8];
            const float dx = r0.x - fx, dy = r0.y - fy;
            const float power = -0.5f * (r0.z * dx * dx + r1.x * dy * dy) - r0.w * dx * dy;
            const float G = __builtin_amdgcn_exp2f(power * LOG2E);
            const float alpha = fminf(GSR_ALPHA_MAX, r1.y * G);
            const bool ok = (base + j < last) && !(power > 0.f) && !(alpha < GSR_ALPHA_MIN);
            if (!__any(ok)) continue;                       // wave-uniform skip
            const float one_m = 1.f - alpha;
            const float inv = __builtin_amdgcn_rcpf(one_m);
            const float Tk = ok ? Tr * inv : Tr;            // transmittance in front of this splat
            // colour accumulated behind this splat
            const float n0 = last_alpha * lc0 + (1.f - last_alpha) * acc0;
            const float n1 = last_alpha * lc1 + (1.f - last_alpha) * acc1;
            const float n2 = last_alpha * lc2 + (1.f - last_alpha) * acc2;
            float dL_dalpha = (r1.z - n0) * d0 + (r1.w - n1) * d1 + (cb - n2) * d2;
            dL_dalpha = dL_dalpha * Tk - Tfinal * inv * bg_dot;
            dL_dalpha = ok ? dL_dalpha : 0.f;
            const float w = ok ? alpha * Tk : 0.f;
            if (ok) { acc0 = n0; acc1 = n1; acc2 = n2; lc0 = r1.z; lc1 = r1.w; lc2 = cb; last_alpha = alpha; }
            Tr = Tk;
            const float dL_dG = r1.y * dL_dalpha;
            const float Gs = ok ? G : 0.f;                  // exp2 of a skipped lane may be inf
            const float gdx = Gs * dx, gdy = Gs * dy;
            const float dG_ddx = -gdx * r0.z - gdy * r0.w, dG_ddy = -gdy * r1.x - gdx * r0.w;
            // nine partial gradients of this pixel
            float v0 = w * d0, v1 = w * d1, v2 = w * d2;
            float v3 = dL_dG * dG_ddx * halfW, v4 = dL_dG * dG_ddy * halfH;
            float v5 = -0.5f * gdx * dx * dL_dG, v6 = -0.5f * gdx * dy * dL_dG, v7 = -0.5f * gdy * dy * dL_dG;
            float v8 = Gs * dL_dalpha;
            v0 = row_allreduce(v0); v1 = row_allreduce(v1); v2 = row_allreduce(v2);
            v3 = row_allreduce(v3); v4 = row_allreduce(v4); v5 = row_allreduce(v5);
            v6 = row_allreduce(v6); v7 = row_allreduce(v7); v8 = row_allreduce(v8);
            // lane (16 r + k) keeps row r's sum of value k, then rows are summed lane-wise
            float sel = v0;
            sel = k16 == 1 ? v1 : sel; sel = k16 == 2 ? v2 : sel; sel = k16 == 3 ? v3 : sel;
            sel = k16 == 4 ? v4 : sel; sel = k16 == 5 ? v5 : sel; sel = k16 == 6 ? v6 : sel;
            sel = k16 == 7 ? v7 : sel; sel = k16 == 8 ? v8 : sel;
            sel += __shfl_xor(sel, 16);
            sel += __shfl_xor(sel, 32);
            if (lane < 9) {
                const uint32_t g = my_id[j];
                atomicAdd(a.acc + GSR_ACC_FLOATS * (size_t)g + lane, sel);
            }
        }
    }
}

hipError_t launch_composite_bwd(const CompositeBwdArgs &a, hipStream_t s) {
    const int T = a.gridx * a.gridy;
    if (T <= 0) return hipSuccess;
    const int padded = (T + 7) / 8 * 8;
    hipLaunchKernelGGL(composite_bwd_kernel, dim3(padded), dim3(256), 0, s, a, padded);
    return hipGetLastError();
}

}  // namespace gsr
