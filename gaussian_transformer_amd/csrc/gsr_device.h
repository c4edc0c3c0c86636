// gsr_device.h -- per-Gaussian math shared by the forward and backward HIP kernels (gfx950).
//
// Stages S1-S6 / S11-S13 of SURVEY.md 8a.  The operation order matches oracle/gsr_ref.c and
// these translation units are compiled with -ffp-contract=off, so that the discrete outputs of
// the per-Gaussian stage (radius, tile rectangle, depth key) are bit-identical to the float32
// CPU restatement: sqrt and division are IEEE-correct under hipcc's default
// -fhip-fp32-correctly-rounded-divide-sqrt.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define GSR_TILE 16
#define GSR_NEAR_Z 0.2f
#define GSR_W_EPS 0.0000001f
#define GSR_FOV_CLAMP 1.3f
#define GSR_DILATION 0.3f
#define GSR_LAMBDA_FLOOR 0.1f
#define GSR_SIGMA_EXTENT 3.0f
#define GSR_ALPHA_MAX 0.99f
#define GSR_ALPHA_MIN (1.0f / 255.0f)
#define GSR_T_MIN 0.0001f
#define GSR_DENOM_EPS 0.0000001f

// SH basis constants (utils/sh_utils.py:26-43 of the reference)
#define SH_C0 0.28209479177387814f
#define SH_C1 0.4886025119029199f
#define SH_C2_0 1.0925484305920792f
#define SH_C2_1 -1.0925484305920792f
#define SH_C2_2 0.31539156525252005f
#define SH_C2_3 -1.0925484305920792f
#define SH_C2_4 0.5462742152960396f
#define SH_C3_0 -0.5900435899266435f
#define SH_C3_1 2.890611442640554f
#define SH_C3_2 -0.4570457994644658f
#define SH_C3_3 0.3731763325901154f
#define SH_C3_4 -0.4570457994644658f
#define SH_C3_5 1.445305721320277f
#define SH_C3_6 -0.5900435899266435f

namespace gsr {

// x' = m[0]x + m[4]y + m[8]z + m[12]  (scene/cameras.py:54-57 memory layout)
__device__ __forceinline__ void xform4x3(const float *__restrict__ m, const float p[3], float o[3]) {
    o[0] = m[0] * p[0] + m[4] * p[1] + m[8] * p[2] + m[12];
    o[1] = m[1] * p[0] + m[5] * p[1] + m[9] * p[2] + m[13];
    o[2] = m[2] * p[0] + m[6] * p[1] + m[10] * p[2] + m[14];
}
__device__ __forceinline__ void xform4x4(const float *__restrict__ m, const float p[3], float o[4]) {
    xform4x3(m, p, o);
    o[3] = m[3] * p[0] + m[7] * p[1] + m[11] * p[2] + m[15];
}

// rotation of an (r,x,y,z) quaternion used as-is (utils/general_utils.py:85-98)
__device__ __forceinline__ void quat_to_rot(const float q[4], float Rm[3][3]) {
    const float r = q[0], x = q[1], y = q[2], z = q[3];
    Rm[0][0] = 1.f - 2.f * (y * y + z * z);
    Rm[0][1] = 2.f * (x * y - r * z);
    Rm[0][2] = 2.f * (x * z + r * y);
    Rm[1][0] = 2.f * (x * y + r * z);
    Rm[1][1] = 1.f - 2.f * (x * x + z * z);
    Rm[1][2] = 2.f * (y * z - r * x);
    Rm[2][0] = 2.f * (x * z - r * y);
    Rm[2][1] = 2.f * (y * z + r * x);
    Rm[2][2] = 1.f - 2.f * (x * x + y * y);
}

// S2: Sigma = (R S)(R S)^T packed xx,xy,xz,yy,yz,zz
__device__ __forceinline__ void cov3d_from_scale_rot(const float s[3], float mod, const float q[4], float c6[6]) {
    float Rm[3][3], Mm[3][3];
    quat_to_rot(q, Rm);
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) Mm[i][j] = Rm[i][j] * (mod * s[j]);
    int k = 0;
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = i; j < 3; j++) c6[k++] = Mm[i][0] * Mm[j][0] + Mm[i][1] * Mm[j][1] + Mm[i][2] * Mm[j][2];
}

struct Ewa {
    float t[3];
    bool clampx, clampy;
    float T[2][3];
    float a, b, c;
    float fx, fy;
    float Wm[3][3];
    float TS[2][3];
};

// S3: EWA projection of the 3D covariance
__device__ __forceinline__ void ewa_project(const float pview[3], const float c6[6], const float *__restrict__ V,
                                            float tanfovx, float tanfovy, int W, int H, Ewa &e) {
    const float fx = (float)W / (2.f * tanfovx), fy = (float)H / (2.f * tanfovy);
    const float limx = GSR_FOV_CLAMP * tanfovx, limy = GSR_FOV_CLAMP * tanfovy;
    const float tz = pview[2];
    const float txtz = pview[0] / tz, tytz = pview[1] / tz;
    e.clampx = (txtz < -limx) || (txtz > limx);
    e.clampy = (tytz < -limy) || (tytz > limy);
    const float cx = txtz < -limx ? -limx : (txtz > limx ? limx : txtz);
    const float cy = tytz < -limy ? -limy : (tytz > limy ? limy : tytz);
    const float tx = cx * tz, ty = cy * tz;
    e.t[0] = tx; e.t[1] = ty; e.t[2] = tz;
    e.fx = fx; e.fy = fy;
    const float J00 = fx / tz, J02 = -(fx * tx) / (tz * tz);
    const float J11 = fy / tz, J12 = -(fy * ty) / (tz * tz);
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
        for (int c = 0; c < 3; c++) e.Wm[r][c] = V[c * 4 + r];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        e.T[0][k] = J00 * e.Wm[0][k] + J02 * e.Wm[2][k];
        e.T[1][k] = J11 * e.Wm[1][k] + J12 * e.Wm[2][k];
    }
    const float S[3][3] = {{c6[0], c6[1], c6[2]}, {c6[1], c6[3], c6[4]}, {c6[2], c6[4], c6[5]}};
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int k = 0; k < 3; k++) e.TS[i][k] = e.T[i][0] * S[0][k] + e.T[i][1] * S[1][k] + e.T[i][2] * S[2][k];
    e.a = e.TS[0][0] * e.T[0][0] + e.TS[0][1] * e.T[0][1] + e.TS[0][2] * e.T[0][2] + GSR_DILATION;
    e.b = e.TS[0][0] * e.T[1][0] + e.TS[0][1] * e.T[1][1] + e.TS[0][2] * e.T[1][2];
    e.c = e.TS[1][0] * e.T[1][0] + e.TS[1][1] * e.T[1][1] + e.TS[1][2] * e.T[1][2] + GSR_DILATION;
}

// SH basis b[0..(D+1)^2) at unit direction d (sign pattern of utils/sh_utils.py:74-100)
template <int D>
__device__ __forceinline__ void sh_basis(const float d[3], float b[16]) {
    const float x = d[0], y = d[1], z = d[2];
    b[0] = SH_C0;
    if (D > 0) {
        b[1] = -SH_C1 * y; b[2] = SH_C1 * z; b[3] = -SH_C1 * x;
        if (D > 1) {
            const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
            b[4] = SH_C2_0 * xy; b[5] = SH_C2_1 * yz; b[6] = SH_C2_2 * (2.f * zz - xx - yy);
            b[7] = SH_C2_3 * xz; b[8] = SH_C2_4 * (xx - yy);
            if (D > 2) {
                b[9] = SH_C3_0 * y * (3.f * xx - yy);
                b[10] = SH_C3_1 * xy * z;
                b[11] = SH_C3_2 * y * (4.f * zz - xx - yy);
                b[12] = SH_C3_3 * z * (2.f * zz - 3.f * xx - 3.f * yy);
                b[13] = SH_C3_4 * x * (4.f * zz - xx - yy);
                b[14] = SH_C3_5 * z * (xx - yy);
                b[15] = SH_C3_6 * x * (xx - 3.f * yy);
            }
        }
    }
}

// d(basis_k)/d(x,y,z), x,y,z independent; entries not listed are zero
template <int D>
__device__ __forceinline__ void sh_basis_grad(const float d[3], float g[16][3]) {
    const float x = d[0], y = d[1], z = d[2];
#pragma unroll
    for (int k = 0; k < 16; k++) { g[k][0] = 0.f; g[k][1] = 0.f; g[k][2] = 0.f; }
    if (D > 0) {
        g[1][1] = -SH_C1; g[2][2] = SH_C1; g[3][0] = -SH_C1;
        if (D > 1) {
            const float xx = x * x, yy = y * y, zz = z * z;
            g[4][0] = SH_C2_0 * y; g[4][1] = SH_C2_0 * x;
            g[5][1] = SH_C2_1 * z; g[5][2] = SH_C2_1 * y;
            g[6][0] = SH_C2_2 * -2.f * x; g[6][1] = SH_C2_2 * -2.f * y; g[6][2] = SH_C2_2 * 4.f * z;
            g[7][0] = SH_C2_3 * z; g[7][2] = SH_C2_3 * x;
            g[8][0] = SH_C2_4 * 2.f * x; g[8][1] = SH_C2_4 * -2.f * y;
            if (D > 2) {
                g[9][0] = SH_C3_0 * 6.f * x * y; g[9][1] = SH_C3_0 * (3.f * xx - 3.f * yy);
                g[10][0] = SH_C3_1 * y * z; g[10][1] = SH_C3_1 * x * z; g[10][2] = SH_C3_1 * x * y;
                g[11][0] = SH_C3_2 * -2.f * x * y; g[11][1] = SH_C3_2 * (4.f * zz - xx - 3.f * yy);
                g[11][2] = SH_C3_2 * 8.f * y * z;
                g[12][0] = SH_C3_3 * -6.f * x * z; g[12][1] = SH_C3_3 * -6.f * y * z;
                g[12][2] = SH_C3_3 * (6.f * zz - 3.f * xx - 3.f * yy);
                g[13][0] = SH_C3_4 * (4.f * zz - 3.f * xx - yy); g[13][1] = SH_C3_4 * -2.f * x * y;
                g[13][2] = SH_C3_4 * 8.f * x * z;
                g[14][0] = SH_C3_5 * 2.f * x * z; g[14][1] = SH_C3_5 * -2.f * y * z; g[14][2] = SH_C3_5 * (xx - yy);
                g[15][0] = SH_C3_6 * (3.f * xx - 3.f * yy); g[15][1] = SH_C3_6 * -6.f * x * y;
            }
        }
    }
}

// clamp(int(v), 0, hi) with C truncation; NaN and negatives -> 0
__device__ __forceinline__ int clampi_from_float(float v, int hi) {
    if (!(v > 0.f)) return 0;
    if (v >= (float)hi) return hi;
    return (int)v;
}

// S5 tile rectangle
__device__ __forceinline__ void tile_rect(float px, float py, int radius, int gridx, int gridy, int &x0, int &y0,
                                          int &x1, int &y1) {
    const float r = (float)radius;
    x0 = clampi_from_float((px - r) / (float)GSR_TILE, gridx);
    y0 = clampi_from_float((py - r) / (float)GSR_TILE, gridy);
    x1 = clampi_from_float((px + r + (float)(GSR_TILE - 1)) / (float)GSR_TILE, gridx);
    y1 = clampi_from_float((py + r + (float)(GSR_TILE - 1)) / (float)GSR_TILE, gridy);
}

// ---- exact (conservative) splat-vs-tile-row intersection ------------------------------------
// A (Gaussian, tile) pair can only ever be blended if some pixel of the tile has
// alpha = o*exp(-q/2) >= 1/255, i.e. q(d) = A dx^2 + 2B dx dy + C dy^2 <= tau = 2 ln(255 o).
// Upstream enumerates every tile of the 3-sigma bounding SQUARE; pairs outside the ellipse
// {q <= tau} are dead weight for the sort and the compositing passes (45 % of them on the headline
// config) and dropping them changes no output: every pixel would skip them at the alpha test.
// The tile set stays a subset of upstream's rectangle.  tau carries a safety margin so that
// float rounding in the compositing kernels can never resurrect a dropped pair.
struct CullParams {          // per Gaussian
    float tau;               // inflated threshold on q; <= 0: the splat can never reach alpha_min
    float xmax;              // half extent of {q<=tau} in x
    float ymax;              // half extent in y
    float dy_at_xmax;        // dy of the ellipse point with dx = +xmax
    float det;               // A*C - B*B of the conic
};
#define GSR_CULL_EPS_PX 0.01f
__device__ __forceinline__ float cull_tau(float opacity) {
    // 2 ln(255 o) with margin; o <= 0 or NaN -> -1 (never visible)
    if (!(opacity > 0.f)) return -1.f;
    const float t = (2.f * 0.6931471805599453f) * __builtin_amdgcn_logf(255.f * opacity);   // v_log_f32 (log2), ~1 ulp: far inside the margin
    return t > 0.f ? t * 1.0001f + 0.01f : -1.f;
}
__device__ __forceinline__ CullParams make_cull(float A, float B, float C, float tau) {
    CullParams c;
    c.tau = tau;
    // The culling math is the library's own (the oracle has no counterpart to match bit for bit) and only has to be
    // conservative and identical wherever it is evaluated: hardware rcp / sqrt (1 ulp) instead of the IEEE expansions,
    // errors ~1e-4 px against the 0.01 px span margin.
    c.det = A * C - B * B;
    const float inv = __builtin_amdgcn_rcpf(c.det);
    c.xmax = __builtin_amdgcn_sqrtf(fmaxf(tau * C * inv, 0.f));
    c.ymax = __builtin_amdgcn_sqrtf(fmaxf(tau * A * inv, 0.f));
    c.dy_at_xmax = -(B * __builtin_amdgcn_rcpf(C)) * c.xmax;
    return c;
}
// columns [c0, c1) of tile row ty (inside the rectangle columns [rx0, rx1)) that the ellipse can reach
__device__ __forceinline__ void tile_row_span(const CullParams &c, float px, float py, float A, float B, int ty,
                                              int W, int H, int rx0, int rx1, int &c0, int &c1) {
    c0 = rx0; c1 = rx0;                                     // empty
    if (!(c.tau > 0.f)) return;
    if (!(c.det > 0.f) || !(A > 0.f)) { c1 = rx1; return; }  // not a positive-definite conic (only reachable with a
                                                            // caller-supplied cov3D): no culling, upstream's rule
    const float ya = (float)(ty * GSR_TILE);
    const float yb = (float)min(ty * GSR_TILE + GSR_TILE - 1, H - 1);
    const float e0 = fmaxf(py - yb, -c.ymax), e1 = fminf(py - ya, c.ymax);   // dy = py - y over the band
    if (!(e0 <= e1)) return;
    const float s0 = __builtin_amdgcn_sqrtf(fmaxf(c.tau * A - c.det * e0 * e0, 0.f));
    const float s1 = __builtin_amdgcn_sqrtf(fmaxf(c.tau * A - c.det * e1 * e1, 0.f));
    const float invA = __builtin_amdgcn_rcpf(A);
    float dxhi, dxlo;
    if (c.dy_at_xmax >= e0 && c.dy_at_xmax <= e1) dxhi = c.xmax;
    else dxhi = fmaxf((-B * e0 + s0) * invA, (-B * e1 + s1) * invA);
    if (-c.dy_at_xmax >= e0 && -c.dy_at_xmax <= e1) dxlo = -c.xmax;
    else dxlo = fminf((-B * e0 - s0) * invA, (-B * e1 - s1) * invA);
    const float xl = ceilf(px - dxhi - GSR_CULL_EPS_PX), xr = floorf(px - dxlo + GSR_CULL_EPS_PX);   // dx = px - x
    const float xlc = fmaxf(xl, 0.f), xrc = fminf(xr, (float)(W - 1));
    if (!(xlc <= xrc)) return;
    const int t0 = max(rx0, ((int)xlc) >> 4), t1 = min(rx1, (((int)xrc) >> 4) + 1);
    if (t1 > t0) { c0 = t0; c1 = t1; }
}

// exact minimum of q(d) = A dx^2 + 2 B dx dy + C dy^2 over the pixel rectangle [xa,xb] x [ya,yb]
// (d = splat centre - pixel); compared with the splat's culling threshold tau
__device__ __forceinline__ bool block_reachable(float px, float py, float A, float B, float C, float invA, float invC,
                                                float tau, float xa, float xb, float ya, float yb) {
    if (!(A > 0.f) || !(C > 0.f) || !(A * C - B * B > 0.f)) return true;      // not positive definite: no culling
    const float dxa = px - xa, dxb = px - xb, dya = py - ya, dyb = py - yb;   // dxb <= dx <= dxa, dyb <= dy <= dya
    if (dxb <= 0.f && dxa >= 0.f && dyb <= 0.f && dya >= 0.f) return tau > 0.f;   // centre inside the block
    float q;
    {
        float dy = fminf(fmaxf(-B * dxa * invC, dyb), dya);
        q = (A * dxa + 2.f * B * dy) * dxa + C * dy * dy;
        dy = fminf(fmaxf(-B * dxb * invC, dyb), dya);
        q = fminf(q, (A * dxb + 2.f * B * dy) * dxb + C * dy * dy);
        float dx = fminf(fmaxf(-B * dya * invA, dxb), dxa);
        q = fminf(q, (A * dx + 2.f * B * dya) * dx + C * dya * dya);
        dx = fminf(fmaxf(-B * dyb * invA, dxb), dxa);
        q = fminf(q, (A * dx + 2.f * B * dyb) * dx + C * dyb * dyb);
    }
    return q <= tau;
}

// ---- the per-(pixel, splat) evaluation shared by the forward and the reverse compositing kernels ----------------
// Both passes must take the SAME discrete decisions for a pair (power > 0 and alpha < 1/255 skips): if the reverse pass
// skipped a pair the forward pass blended, its transmittance chain T/(1 - alpha) would be off for the rest of the
// pixel (upstream uses one expression in both kernels).  The expression is therefore spelled out with explicit
// fma / mul (no contraction freedom left to the compiler) on the conic pre-scaled at staging time by
// stage_conic(): the exponent comes out in log2 units and feeds v_exp_f32 directly.
#define GSR_LOG2E 1.4426950408889634f
struct StagedConic { float a, b, c; };            // (-0.5 log2e A, -log2e B, -0.5 log2e C)
__device__ __forceinline__ StagedConic stage_conic(float A, float B, float C) {
    return {A * (-0.5f * GSR_LOG2E), B * (-GSR_LOG2E), C * (-0.5f * GSR_LOG2E)};
}
__device__ __forceinline__ float splat_power_log2(const StagedConic &k, float dx, float dy) {
    const float t = __builtin_fmaf(k.b, dy, k.a * dx);          // a dx + b dy
    return __builtin_fmaf(k.c * dy, dy, t * dx);                // (a dx + b dy) dx + (c dy) dy
}
// The two blocks a wave of the default decomposition owns lie side by side: they share dy, so the terms of the exponent that do not
// depend on dx are formed once per splat visit (u = b dy, w = (c dy) dy) and each block adds two fma: (a dx + u) dx + w.
// The same polynomial as splat_power_log2 with another rounding; BOTH compositing passes must use the same form for a given
// decomposition (they do: 2 blocks per wave -> this one, 1 or 4 -> splat_power_log2).
struct RowTerms { float u, w; };
__device__ __forceinline__ RowTerms splat_row_terms(const StagedConic &k, float dy) { return {k.b * dy, (k.c * dy) * dy}; }
__device__ __forceinline__ float splat_power_log2_row(const StagedConic &k, const RowTerms &r, float dx) {
    return __builtin_fmaf(__builtin_fmaf(k.a, dx, r.u), dx, r.w);
}
// araw = opacity * G (alpha before the 0.99 cap) and the skip decision of S9: ok = !(power > 0) && !(alpha < 1/255).
// alpha = min(0.99, araw) < 1/255 exactly when araw < 1/255, so the decision is taken on araw and the cap is left to
// the caller (the reverse pass caps after masking).
// Returned as a 64-bit lane mask (two ballots combined by scalar logic): the kernels keep all their per-pair decisions
// in SGPR masks and turn them back into a predicate with inverse_ballot only where lanes must be switched off.
__device__ __forceinline__ unsigned long long splat_alpha(float power_log2, float opacity, float &araw) {
    araw = opacity * __builtin_amdgcn_exp2f(power_log2);
    return __builtin_amdgcn_ballot_w64(!(power_log2 > 0.f)) & __builtin_amdgcn_ballot_w64(!(araw < GSR_ALPHA_MIN));
}

// Load the first 3*K floats of one Gaussian's SH row [M,3] into c[].  16-byte vector loads when
// the row stride keeps every row 16-byte aligned (M = 4, 8, 12, 16 ...), scalar loads otherwise.
template <int K>
__device__ __forceinline__ void load_sh_row(const float *__restrict__ shs, size_t i, int M, float c[3 * K + 3]) {
    const float *row = shs + i * (size_t)M * 3;
    if (((M * 3) & 3) == 0 && ((reinterpret_cast<uintptr_t>(shs) & 15) == 0)) {
        const float4 *r4 = reinterpret_cast<const float4 *>(row);
        constexpr int NV = (3 * K + 3) / 4;
#pragma unroll
        for (int v = 0; v < NV; v++) {
            const float4 t = r4[v];
            c[4 * v] = t.x;
            if (4 * v + 1 < 3 * K + 3) c[4 * v + 1] = t.y;
            if (4 * v + 2 < 3 * K + 3) c[4 * v + 2] = t.z;
            if (4 * v + 3 < 3 * K + 3) c[4 * v + 3] = t.w;
        }
    } else {
#pragma unroll
        for (int v = 0; v < 3 * K; v++) c[v] = row[v];
    }
}

// SH row given as two tensors, as the reference stores them (scene/gaussian_model.py:108-111):
// dc [P,1,3] holds coefficient 0, rest [P,M-1,3] holds coefficients 1..M-1.  Avoids the torch.cat.
template <int K>
__device__ __forceinline__ void load_sh_row_split(const float *__restrict__ dc, const float *__restrict__ rest, size_t i,
                                                  int M, float c[3 * K + 3]) {
    c[0] = dc[3 * i]; c[1] = dc[3 * i + 1]; c[2] = dc[3 * i + 2];
    const float *row = rest + i * (size_t)(M - 1) * 3;
#pragma unroll
    for (int v = 3; v < 3 * K; v++) c[v] = row[v - 3];
}

// activations of scene/gaussian_model.py:33-41 for the fused (raw parameter) path
__device__ __forceinline__ float act_sigmoid(float x) { return 1.f / (1.f + expf(-x)); }
__device__ __forceinline__ void act_normalize4(float q[4], float &inv_norm) {     // F.normalize: x / max(|x|, 1e-12)
    const float n = sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    inv_norm = 1.f / fmaxf(n, 1e-12f);
    q[0] *= inv_norm; q[1] *= inv_norm; q[2] *= inv_norm; q[3] *= inv_norm;
}

}  // namespace gsr
