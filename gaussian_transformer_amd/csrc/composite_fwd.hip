// composite_fwd.hip -- front-to-back alpha compositing (S9).
//
// CDNA4 shape: a 256-thread workgroup owns one 16x16 tile, but its four wave64s are fully
// independent: wave w composites the 8x8 pixel quadrant w of the tile, stages the tile's
// depth-sorted splat list 64 records at a time into a wave-private 3 KiB LDS slice (one record
// gathered per lane, read back as wave-uniform broadcast ds_read_b128), and stops as soon as its
// own 64 pixels are saturated (64-bit ballot).  There is no workgroup barrier anywhere, so a
// finished quadrant never waits for the slowest pixel of the tile.
// Conic terms are pre-scaled by -0.5*log2(e) / -log2(e) at staging time so the per-pixel
// exponent feeds v_exp_f32 directly.
// blockIdx is remapped so that each XCD (blocks b, b+8, ... share one) walks a contiguous band
// of tiles: neighbouring tiles share splats, which keeps the record gathers in that XCD's L2.
#include "gsr_device.h"
#include "gsr_internal.h"

namespace gsr {

#define LOG2E 1.4426950408889634f

__device__ __forceinline__ int xcd_band_tile(int b, int nblocks_padded) {
    const int chunk = nblocks_padded >> 3;
    return (b & 7) * chunk + (b >> 3);
}

__global__ __launch_bounds__(256) void composite_fwd_kernel(CompositeArgs a, int nblocks_padded) {
    __shared__ float4 stage[4][64 * 3];
    const int T = a.gridx * a.gridy;
    const int tile = xcd_band_tile(blockIdx.x, nblocks_padded);
    if (tile >= T) return;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int tx = tile % a.gridx, ty = tile / a.gridx;
    const int x = tx * GSR_TILE + (wave & 1) * 8 + (lane & 7);
    const int y = ty * GSR_TILE + (wave >> 1) * 8 + (lane >> 3);
    const bool inside = x < a.W && y < a.H;
    const float fx = (float)x, fy = (float)y;
    const uint2 range = a.ranges[tile];
    const int n = (int)(range.y - range.x);
    float4 *my = stage[wave];
    const float4 *rec4 = reinterpret_cast<const float4 *>(a.rec);

    float Tr = 1.f, C0 = 0.f, C1 = 0.f, C2 = 0.f;
    uint32_t last = 0;
    bool done = !inside;

    for (int base = 0; base < n; base += 64) {
        if (__all(done)) break;
        const int cnt = min(64, n - base);
        __builtin_amdgcn_wave_barrier();
        if (lane < cnt) {
            const uint32_t g = a.point_list[range.x + base + lane];
            float4 r0 = rec4[3 * (size_t)g], r1 = rec4[3 * (size_t)g + 1];
            const float b = a.rec[GSR_REC_FLOATS * (size_t)g + 8];
            // (px, py, -0.5*log2e*A, -log2e*B) (-0.5*log2e*C, opacity, r, g) (b)
            r0.z *= -0.5f * LOG2E; r0.w *= -LOG2E; r1.x *= -0.5f * LOG2E;
            my[lane * 3 + 0] = r0; my[lane * 3 + 1] = r1; my[lane * 3 + 2] = make_float4(b, 0.f, 0.f, 0.f);
        }
        __builtin_amdgcn_wave_barrier();
        for (int j = 0; j < cnt; j++) {
            const float4 r0 = my[j * 3 + 0], r1 = my[j * 3 + 1];
            const float cb = reinterpret_cast<const float *>(my)[j * 12 + 8];
            const float dx = r0.x - fx, dy = r0.y - fy;
            const float power = (r0.z * dx + r0.w * dy) * dx + (r1.x * dy) * dy;   // log2 units
            const float alpha = fminf(GSR_ALPHA_MAX, r1.y * __builtin_amdgcn_exp2f(power));
            const bool ok = !done && !(power > 0.f) && !(alpha < GSR_ALPHA_MIN);
            const float Tn = Tr * (1.f - alpha);
            const bool stop = ok && (Tn < GSR_T_MIN);
            done = done || stop;
            const bool blend = ok && !stop;
            const float w = blend ? alpha * Tr : 0.f;
            C0 += r1.z * w; C1 += r1.w * w; C2 += cb * w;
            Tr = blend ? Tn : Tr;
            last = blend ? (uint32_t)(base + j + 1) : last;
            if (__any(stop)) {                      // wave-uniform
                if (__all(done)) break;
            }
        }
    }
    if (inside) {
        const size_t pix = (size_t)y * a.W + x, HW = (size_t)a.W * a.H;
        a.final_T[pix] = Tr;
        a.n_contrib[pix] = last;
        a.out_color[pix] = C0 + Tr * a.bg[0];
        a.out_color[HW + pix] = C1 + Tr * a.bg[1];
        a.out_color[2 * HW + pix] = C2 + Tr * a.bg[2];
    }
}

hipError_t launch_composite_fwd(const CompositeArgs &a, hipStream_t s) {
    const int T = a.gridx * a.gridy;
    if (T <= 0) return hipSuccess;
    const int padded = (T + 7) / 8 * 8;
    hipLaunchKernelGGL(composite_fwd_kernel, dim3(padded), dim3(256), 0, s, a, padded);
    return hipGetLastError();
}

}  // namespace gsr
