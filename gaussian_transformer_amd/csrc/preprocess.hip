// preprocess.hip -- per-Gaussian forward stage (S1-S6): cull, 3D covariance, EWA projection,
// conic + radius, pixel centre + tile rectangle, SH colour.  One lane per Gaussian, wave64,
// 256-thread workgroups.  HBM-streaming: reads 44 + 12*K bytes, writes 48 (record) + 4 (depth)
// + 8 (rect) + 4 (tiles) + 4 (radii) + 1 (clamp mask) per Gaussian.
// Compiled with -ffp-contract=off: discrete outputs are bit-identical to oracle/gsr_ref.c.
#include "gsr_device.h"
#include "gsr_internal.h"

namespace gsr {

// RAW / SPLIT = the fused-step extension (raw parameters / split SH tensors), separate instantiations so
// that the reference path keeps its register budget.  The SH row is requested before the geometry is computed
// (102 VGPRs, 4 waves/SIMD at D = 3): hiding that latency is worth more than the fifth wave (88 -> 76 us at 1 M).
template <int D, bool RAW, bool SPLIT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4))) void preprocess_fwd_kernel(PreprocessArgs a) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    for (int j = i; j < GSR_DO_ZERO_WORDS; j += gridDim.x * 256) a.g.dord.hdr[j] = 0u;    // counters of depth_order.hip
    if (i == 0) *a.g.touch_mark = a.touch_mark;           // this frame's "staged" mark, for pergauss_bwd.hip (GeomView::touched)
    for (int j = i; j < GSR_SEG_HDR_WORDS && a.seg_hdr; j += gridDim.x * 256) a.seg_hdr[j] = 0u;    // checkpoint pool / work-unit counters of the compositing kernels (SegView)
    // no early exit: the workgroup reduces the depth extrema of its emitting Gaussians at the end.  Lanes past
    // the end recompute Gaussian P-1 and store nothing.
    const bool live = i < a.P;
    const size_t si = (size_t)(live ? i : a.P - 1);

    // defaults for a culled Gaussian
    int radius_out = 0;
    uint32_t tiles_out = 0;
    float4 r0 = make_float4(0.f, 0.f, 0.f, 0.f), r1 = r0, r2 = r0;
    uint4 rect_out = make_uint4(0u, 0u, ~0u, ~0u);
    uint8_t clamp_out = 0;
    float depth_out = 0.f;
    uint32_t ss_y = 0u;                        // supertile_sort.hip's record (GeomView::ss_rec): kind 0, nothing emitted
    uint64_t ss_w = 0ull;

    const float p[3] = {a.means3D[3 * si], a.means3D[3 * si + 1], a.means3D[3 * si + 2]};
    // Scale, rotation, opacity and the camera are requested here, with the mean, whatever becomes of the Gaussian (32 bytes more for a
    // culled one): where they are used they sit behind a branch each (near plane, empty rectangle), and a load behind a branch is a
    // memory round trip of its own.  The barrier keeps the compiler from sinking them back (behind it, it would also fetch the camera
    // matrices with vector loads where they are used: hence the copies).  Same box, same run: 73.6 -> 71.0 us at config 3.
    float s_in[3] = {0.f, 0.f, 0.f};
    float4 q4_in = make_float4(1.f, 0.f, 0.f, 0.f);
    if (!a.cov3D_precomp) {                                                         // uniform
        s_in[0] = a.scales[3 * si]; s_in[1] = a.scales[3 * si + 1]; s_in[2] = a.scales[3 * si + 2];
        q4_in = reinterpret_cast<const float4 *>(a.rotations)[si];
    }
    const float opacity_in = a.opacities[si];
    float campos_in[3] = {0.f, 0.f, 0.f};
    if (a.campos) { campos_in[0] = a.campos[0]; campos_in[1] = a.campos[1]; campos_in[2] = a.campos[2]; }
    float vm[16], pm[16];
#pragma unroll
    for (int k = 0; k < 16; k++) { vm[k] = a.viewmatrix[k]; pm[k] = a.projmatrix[k]; }
    asm volatile("" ::: "memory");
    float pv[3];
    xform4x3(vm, p, pv);
    // the SH row is requested as soon as the near-plane test passes: its latency then hides behind the geometry
    constexpr int KK = (D + 1) * (D + 1);
    float c[3 * KK + 3];
    const bool early_sh = a.shs != nullptr && pv[2] > GSR_NEAR_Z;
    if (early_sh) {
        if (SPLIT) load_sh_row_split<KK>(a.shs, a.shs_rest, si, a.M, c);
        else load_sh_row<KK>(a.shs, si, a.M, c);
    }
    if (pv[2] > GSR_NEAR_Z) {                                                       // S1
        float ph[4];
        xform4x4(pm, p, ph);
        const float pw = 1.f / (ph[3] + GSR_W_EPS);
        const float ndcx = ph[0] * pw, ndcy = ph[1] * pw;
        float c6[6];
        if (a.cov3D_precomp) {
#pragma unroll
            for (int k = 0; k < 6; k++) c6[k] = a.cov3D_precomp[6 * si + k];
        } else {                                                                    // S2
            float s[3] = {s_in[0], s_in[1], s_in[2]};
            const float4 q4 = q4_in;
            float q[4] = {q4.x, q4.y, q4.z, q4.w};
            if (RAW) {              // fused activations: exp / normalize
                s[0] = expf(s[0]); s[1] = expf(s[1]); s[2] = expf(s[2]);
                float inv_norm;
                act_normalize4(q, inv_norm);
            }
            cov3d_from_scale_rot(s, a.scale_modifier, q, c6);
        }
        Ewa e;
        ewa_project(pv, c6, vm, a.tanfovx, a.tanfovy, a.W, a.H, e);       // S3
        const float det = e.a * e.c - e.b * e.b;                                    // S4
        if (det != 0.f) {
            const float det_inv = 1.f / det;
            const float conA = e.c * det_inv, conB = -e.b * det_inv, conC = e.a * det_inv;
            const float mid = 0.5f * (e.a + e.c);
            float disc = mid * mid - det;
            if (disc < GSR_LAMBDA_FLOOR) disc = GSR_LAMBDA_FLOOR;
            const float l1 = mid + sqrtf(disc), l2 = mid - sqrtf(disc);
            const float lmax = l1 > l2 ? l1 : l2;
            const int radius = (int)ceilf(GSR_SIGMA_EXTENT * sqrtf(lmax));
            const float px = ((ndcx + 1.f) * (float)a.W - 1.f) * 0.5f;             // S5
            const float py = ((ndcy + 1.f) * (float)a.H - 1.f) * 0.5f;
            int x0, y0, x1, y1;
            tile_rect(px, py, radius, a.gridx, a.gridy, x0, y0, x1, y1);
            const int area = (x1 - x0) * (y1 - y0);
            if (area != 0) {
                const float opacity = (RAW) ? act_sigmoid(opacity_in) : opacity_in;
                float tau = 0.f;
                int pairs = area;
                // the row spans go to tile_lists.hip in one word when the rectangle is small enough
                const bool small = (y1 - y0) <= 8 && (x1 - x0) <= 15 && (((x1 - 1) >> 3) - (x0 >> 3)) <= 1;
                uint64_t sp = 0ull;
                // supertile_sort.hip bins by super-tiles of 4 x 4 tiles: a rectangle inside a 2 x 2 block of them (8 x 8 tiles
                // from (sbx, sby)) gets its four 16-bit masks here, bit (ty & 3) * 4 + (tx & 3), mask (dy * 2 + dx)
                const int sbx = x0 & ~(GSR_SS_TILES - 1), sby = y0 & ~(GSR_SS_TILES - 1);
                const bool small4 = x1 - sbx <= 2 * GSR_SS_TILES && y1 - sby <= 2 * GSR_SS_TILES;
                uint32_t m4lo = 0u, m4hi = 0u;             // super-tile row 0 / row 1 of the block: mask dx = 0 | mask dx = 1 << 16
                auto row_masks = [&](int ty, int c0, int c1) {
                    if (small4 && c1 > c0) {
                        const int r = ty - sby;
                        const uint32_t rowbits = ((1u << (c1 - c0)) - 1u) << (c0 - sbx);                       // columns sbx .. sbx + 7
                        const uint32_t word = ((rowbits & 15u) | ((rowbits >> 4) << 16)) << ((r & 3) * 4);
                        if (r & 4) m4hi |= word; else m4lo |= word;
                    }
                };
                if (a.exact_cull) {                      // count only the tiles the ellipse can reach
                    tau = cull_tau(opacity);
                    const CullParams cp = make_cull(conA, conB, conC, tau);
                    pairs = 0;
                    for (int ty = y0; ty < y1; ty++) {
                        int c0, c1;
                        tile_row_span(cp, px, py, conA, conB, ty, a.W, a.H, x0, x1, c0, c1);
                        pairs += c1 - c0;
                        sp |= (uint64_t)(uint32_t)((c0 - x0) | ((c1 - x0) << 4)) << (8 * ((ty - y0) & 7));
                        row_masks(ty, c0, c1);
                    }
                } else {
                    for (int k = 0; k < y1 - y0 && k < 8; k++) {
                        sp |= (uint64_t)(uint32_t)((x1 - x0) << 4) << (8 * k);
                        row_masks(y0 + k, x0, x1);
                    }
                }
                {   // kind 1: the four masks; kind 3: <= 8 rows x <= 15 columns, the row spans travel in the record together with the
                    // rectangle's offset inside its first super-tile and its size; kind 2: larger (supertile_sort.hip reads rect / rec)
                    const bool mid = y1 - y0 <= 8 && x1 - x0 <= 15;
                    const uint32_t kind = pairs > 0 ? (small4 ? 1u : (mid ? 3u : 2u)) : 0u;
                    const uint32_t bin0 = (uint32_t)((y0 / GSR_SS_TILES) * ((a.gridx + GSR_SS_TILES - 1) / GSR_SS_TILES) + x0 / GSR_SS_TILES);
                    ss_y = (bin0 & 0x3ffffu) | ((uint32_t)(x0 & 3) << 18) | ((uint32_t)(y0 & 3) << 20) | ((uint32_t)((y1 - y0 - 1) & 7) << 22) |
                           ((uint32_t)((x1 - x0 - 1) & 15) << 25) | (kind << 29);
                    ss_w = small4 ? ((uint64_t)m4lo | ((uint64_t)m4hi << 32)) : sp;
                }
                if (!small) sp = ~0ull;
                float rgb[3];
                if (a.colors_precomp) {
                    rgb[0] = a.colors_precomp[3 * si]; rgb[1] = a.colors_precomp[3 * si + 1]; rgb[2] = a.colors_precomp[3 * si + 2];
                } else {                                                            // S6
                    float dir[3] = {p[0] - campos_in[0], p[1] - campos_in[1], p[2] - campos_in[2]};
                    const float il = 1.f / sqrtf(dir[0] * dir[0] + dir[1] * dir[1] + dir[2] * dir[2]);
                    dir[0] *= il; dir[1] *= il; dir[2] *= il;
                    float b[16];
                    sh_basis<D>(dir, b);
                    constexpr int K = KK;
#pragma unroll
                    for (int ch = 0; ch < 3; ch++) {
                        float v = 0.f;
#pragma unroll
                        for (int k = 0; k < K; k++) v += b[k] * c[k * 3 + ch];
                        v += 0.5f;
                        if (v < 0.f) clamp_out |= (uint8_t)(1u << ch);
                        rgb[ch] = v < 0.f ? 0.f : v;
                    }
                }
                radius_out = radius;
                tiles_out = (uint32_t)pairs;
                depth_out = pv[2];
                rect_out = make_uint4((uint32_t)x0 | ((uint32_t)x1 << 16), (uint32_t)y0 | ((uint32_t)y1 << 16), (uint32_t)sp, (uint32_t)(sp >> 32));
                r0 = make_float4(px, py, conA, conB);
                r1 = make_float4(conC, opacity, rgb[0], rgb[1]);
                r2 = make_float4(rgb[2], pv[2], tau, 0.f);
            }
        }
    }
    if (live) {
        float4 *rec = reinterpret_cast<float4 *>(a.g.rec) + 3 * si;
        rec[0] = r0; rec[1] = r1; rec[2] = r2;
        a.g.depth[si] = depth_out;
        a.g.opac[si] = r1.y;
        a.g.rect[si] = rect_out;
        a.g.ss_rec[si] = make_uint4(__float_as_uint(depth_out), ss_y, (uint32_t)ss_w, (uint32_t)(ss_w >> 32));
        a.g.tiles[si] = tiles_out;
        a.g.clamped[si] = clamp_out;
        a.radii[si] = radius_out;
    }
    // depth-bit extrema of this workgroup's emitting Gaussians (depth_order.hip derives its bucket map from them)
    __shared__ uint32_t s_mn[4], s_mx[4], s_en[4];
    const bool emits = live && tiles_out > 0u;
    // (Gaussian, super-tile) entries of tile_lists.hip: super-tiles of 8 x 8 tiles under the tile rectangle
    uint32_t ent = 0u;
    if (emits) {
        const uint32_t x0 = rect_out.x & 0xffffu, x1 = rect_out.x >> 16, y0 = rect_out.y & 0xffffu, y1 = rect_out.y >> 16;
        ent = (((x1 + 7u) >> 3) - (x0 >> 3)) * (((y1 + 7u) >> 3) - (y0 >> 3));
    }
    uint32_t mn = emits ? __float_as_uint(depth_out) : 0xffffffffu, mx = emits ? __float_as_uint(depth_out) : 0u;
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) {
        mn = min(mn, (uint32_t)__shfl_xor((int)mn, m)); mx = max(mx, (uint32_t)__shfl_xor((int)mx, m));
        ent += (uint32_t)__shfl_xor((int)ent, m);
    }
    if ((threadIdx.x & 63) == 0) { s_mn[threadIdx.x >> 6] = mn; s_mx[threadIdx.x >> 6] = mx; s_en[threadIdx.x >> 6] = ent; }
    __syncthreads();
    if (threadIdx.x == 0) {
        a.g.dord.blkent[blockIdx.x] = s_en[0] + s_en[1] + s_en[2] + s_en[3];
        a.g.dord.blkmin[blockIdx.x] = min(min(s_mn[0], s_mn[1]), min(s_mn[2], s_mn[3]));
        a.g.dord.blkmax[blockIdx.x] = max(max(s_mx[0], s_mx[1]), max(s_mx[2], s_mx[3]));
    }
}

hipError_t launch_preprocess_fwd(const PreprocessArgs &a, hipStream_t s) {
    if (a.P <= 0) return hipSuccess;
    const dim3 grid((a.P + 255) / 256), block(256);
    const int d = a.shs ? a.D : 0;
    const bool raw = a.raw_params != 0, split = a.shs_rest != nullptr;
#define GSR_LAUNCH(DD)                                                                                   \
    do {                                                                                                 \
        if (!raw && !split) hipLaunchKernelGGL((preprocess_fwd_kernel<DD, false, false>), grid, block, 0, s, a);           \
        else if (raw && !split) hipLaunchKernelGGL((preprocess_fwd_kernel<DD, true, false>), grid, block, 0, s, a);        \
        else if (!raw && split) hipLaunchKernelGGL((preprocess_fwd_kernel<DD, false, true>), grid, block, 0, s, a);        \
        else hipLaunchKernelGGL((preprocess_fwd_kernel<DD, true, true>), grid, block, 0, s, a);                            \
    } while (0)
    switch (d) {
        case 0: GSR_LAUNCH(0); break;
        case 1: GSR_LAUNCH(1); break;
        case 2: GSR_LAUNCH(2); break;
        default: GSR_LAUNCH(3); break;
    }
#undef GSR_LAUNCH
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void mark_visible_kernel(int P, const float *__restrict__ means3D,
                                                           const float *__restrict__ V, uint8_t *__restrict__ present) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    const float p[3] = {means3D[3 * (size_t)i], means3D[3 * (size_t)i + 1], means3D[3 * (size_t)i + 2]};
    float pv[3];
    xform4x3(V, p, pv);
    present[i] = pv[2] > GSR_NEAR_Z ? 1 : 0;
}

__global__ __launch_bounds__(256) void composited_mask_kernel(int P, const uint8_t *__restrict__ touched, const uint32_t *__restrict__ mark,
                                                             const int *__restrict__ radii_unused, uint8_t *__restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < P) out[i] = touched[i] == (uint8_t)*mark ? 1 : 0;
}

hipError_t launch_composited_mask(int P, const uint8_t *touched, const uint32_t *mark, uint8_t *out, hipStream_t s) {
    if (P <= 0) return hipSuccess;
    hipLaunchKernelGGL(composited_mask_kernel, dim3((P + 255) / 256), dim3(256), 0, s, P, touched, mark, nullptr, out);
    return hipGetLastError();
}

hipError_t launch_mark_visible(int P, const float *means3D, const float *viewmatrix, uint8_t *present, hipStream_t s) {
    if (P <= 0) return hipSuccess;
    hipLaunchKernelGGL(mark_visible_kernel, dim3((P + 255) / 256), dim3(256), 0, s, P, means3D, viewmatrix, present);
    return hipGetLastError();
}

}  // namespace gsr
