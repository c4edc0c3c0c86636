/*
 * oracle/gsr_ref.c -- CPU restatement of the differentiable Gaussian rasterizer.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (gaussian_transformer_amd/,
 * diff_gaussian_rasterization/) may import, link or call this file; only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg do, and only as the checker.
 *
 * PARITY STATUS: "parity unpinned" at the rasterizer boundary.  The algorithm lives in the
 * third-party module `diff_gaussian_rasterization` (github.com/graphdeco-inria/
 * diff-gaussian-rasterization, pinned version unknown: /root/reference/.gitmodules:4-6 names the
 * URL, the submodule directory is empty and the snapshot has no .git), which is absent from
 * /root/reference, and the reference holds no tests or golden vectors for it.
 * This file restates the published 3DGS algorithm (stages S1-S13 of SURVEY.md 8a) and is
 * anchored on what the reference DOES hold:
 *   - call site / argument meaning:  gaussian_renderer/__init__.py:36-49,85-93
 *   - SH basis, constants and signs: utils/sh_utils.py:26-43,74-100      (golden fixture sh_eval)
 *   - quaternion -> rotation, cov3D packing: utils/general_utils.py:64-98 (golden fixture cov3d)
 *   - matrix memory layout (transposed, row-major): scene/cameras.py:54-57,
 *     utils/graphics_utils.py:38-71; perspective epsilon 1e-7: utils/graphics_utils.py:28
 *     (golden fixtures camera, geom_transform)
 *   - colour clamp max(sh+0.5, 0): gaussian_renderer/__init__.py:78
 * and is cross-checked against a float64 dense PyTorch restatement with autograd and finite
 * differences (oracle/dense_ref.py).
 *
 * Build:  make -C oracle     ->  oracle/_build/libgsr_ref_f32.so, libgsr_ref_f64.so
 * REAL selects the arithmetic type (float by default = what the HIP path computes in).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#ifdef GSR_REF_DOUBLE
typedef double REAL;
#define R_SQRT sqrt
#define R_EXP exp
#define R_CEIL ceil
#define R_FLOOR floor
#else
typedef float REAL;
#define R_SQRT sqrtf
#define R_EXP expf
#define R_CEIL ceilf
#define R_FLOOR floorf
#endif
#define R(x) ((REAL)(x))

/* ---- named constants of the algorithm (SURVEY.md 8a "load-bearing constants") ---- */
#define GSR_TILE 16              /* tile edge in pixels; membership granularity          */
#define GSR_NEAR_Z R(0.2)        /* S1: culled iff view-space z <= 0.2                     */
#define GSR_W_EPS R(0.0000001)   /* S1: perspective divide epsilon                        */
#define GSR_FOV_CLAMP R(1.3)     /* S3: clamp of t.x/t.z, t.y/t.z to 1.3*tanfov           */
#define GSR_DILATION R(0.3)      /* S3: low-pass added to the 2D covariance diagonal      */
#define GSR_LAMBDA_FLOOR R(0.1)  /* S4: floor under the eigenvalue discriminant           */
#define GSR_SIGMA_EXTENT R(3.0)  /* S4: radius = ceil(3 sigma_max)                         */
#define GSR_ALPHA_MAX R(0.99)    /* S9: alpha cap                                         */
#define GSR_ALPHA_MIN (R(1.0) / R(255.0)) /* S9: skip threshold                           */
#define GSR_T_MIN R(0.0001)      /* S9: stop when transmittance would fall below          */
#define GSR_DENOM_EPS R(0.0000001) /* S11: 1/(det^2 + eps)                                */

/* SH basis constants: utils/sh_utils.py:26-43 */
static const REAL SH_C0 = R(0.28209479177387814);
static const REAL SH_C1 = R(0.4886025119029199);
static const REAL SH_C2[5] = {R(1.0925484305920792), R(-1.0925484305920792), R(0.31539156525252005),
                              R(-1.0925484305920792), R(0.5462742152960396)};
static const REAL SH_C3[7] = {R(-0.5900435899266435), R(2.890611442640554), R(-0.4570457994644658),
                              R(0.3731763325901154), R(-0.4570457994644658), R(1.445305721320277),
                              R(-0.5900435899266435)};

typedef struct gsr_ref_scene {
    int32_t P, D, M, W, H;
    int32_t prefiltered;
    REAL scale_modifier, tanfovx, tanfovy;
    const REAL *bg;             /* [3] */
    const REAL *means3D;        /* [P,3] */
    const REAL *shs;            /* [P,M,3] or NULL */
    const REAL *colors_precomp; /* [P,3] or NULL */
    const REAL *opacities;      /* [P] */
    const REAL *scales;         /* [P,3] or NULL */
    const REAL *rotations;      /* [P,4] (r,x,y,z) or NULL */
    const REAL *cov3D_precomp;  /* [P,6] or NULL */
    const REAL *viewmatrix;     /* [16] memory of world_view_transform (scene/cameras.py:54) */
    const REAL *projmatrix;     /* [16] memory of full_proj_transform (scene/cameras.py:56)  */
    const REAL *campos;         /* [3] */
} gsr_ref_scene;

typedef struct gsr_ref_state {
    int32_t P, W, H, gridx, gridy;
    int64_t N;
    /* per-Gaussian */
    REAL *depth, *xy, *cov3D, *conic_o, *rgb;
    int32_t *radii;
    uint32_t *tiles_touched, *offsets;
    uint8_t *clamped;
    /* binning */
    uint64_t *keys;
    uint32_t *vals;
    uint32_t *ranges; /* [T,2] */
    /* image */
    REAL *final_T;
    uint32_t *n_contrib;
    REAL *margin; /* [H*W] smallest margin of any discrete decision of S9 the pixel took (see below) */
    /* wall-clock seconds of the last forward / backward: preprocess, scan+emit+sort+ranges, composite | composite, per-Gaussian */
    double t_fwd[3], t_bwd[2];
} gsr_ref_state;

/* x' = m[0]x + m[4]y + m[8]z + m[12]: the tensors are transposes stored row-major
 * (scene/cameras.py:54-57), i.e. column-major memory of the column-vector matrix. */
static inline void xform4x3(const REAL *m, const REAL *p, REAL *o) {
    o[0] = m[0] * p[0] + m[4] * p[1] + m[8] * p[2] + m[12];
    o[1] = m[1] * p[0] + m[5] * p[1] + m[9] * p[2] + m[13];
    o[2] = m[2] * p[0] + m[6] * p[1] + m[10] * p[2] + m[14];
}
static inline void xform4x4(const REAL *m, const REAL *p, REAL *o) {
    xform4x3(m, p, o);
    o[3] = m[3] * p[0] + m[7] * p[1] + m[11] * p[2] + m[15];
}

/* rotation matrix of an (r,x,y,z) quaternion used as-is: utils/general_utils.py:85-98 */
static inline void quat_to_rot(const REAL *q, REAL Rm[3][3]) {
    REAL r = q[0], x = q[1], y = q[2], z = q[3];
    Rm[0][0] = R(1) - R(2) * (y * y + z * z);
    Rm[0][1] = R(2) * (x * y - r * z);
    Rm[0][2] = R(2) * (x * z + r * y);
    Rm[1][0] = R(2) * (x * y + r * z);
    Rm[1][1] = R(1) - R(2) * (x * x + z * z);
    Rm[1][2] = R(2) * (y * z - r * x);
    Rm[2][0] = R(2) * (x * z - r * y);
    Rm[2][1] = R(2) * (y * z + r * x);
    Rm[2][2] = R(1) - R(2) * (x * x + y * y);
}

/* S2: Sigma = (R S)(R S)^T packed xx,xy,xz,yy,yz,zz (utils/general_utils.py:67-72,
 * scene/gaussian_model.py:27-31) */
static void cov3d_from_scale_rot(const REAL *s, REAL mod, const REAL *q, REAL *c6) {
    REAL Rm[3][3], Mm[3][3];
    quat_to_rot(q, Rm);
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) Mm[i][j] = Rm[i][j] * (mod * s[j]);
    int k = 0;
    for (int i = 0; i < 3; i++)
        for (int j = i; j < 3; j++) c6[k++] = Mm[i][0] * Mm[j][0] + Mm[i][1] * Mm[j][1] + Mm[i][2] * Mm[j][2];
}

typedef struct {
    REAL t[3];      /* clamped view-space point */
    int clampx, clampy;
    REAL T[2][3];   /* J * W */
    REAL a, b, c;   /* dilated 2D covariance */
    REAL fx, fy;
    REAL Wm[3][3];  /* rotation part of world->view, Wm[r][c] */
    REAL TS[2][3];  /* T * Sigma */
} ewa_t;

/* S3: EWA projection of the 3D covariance */
static void ewa_project(const REAL *pview, const REAL *c6, const REAL *V, REAL tanfovx, REAL tanfovy,
                        int W, int H, ewa_t *e) {
    REAL fx = (REAL)W / (R(2) * tanfovx), fy = (REAL)H / (R(2) * tanfovy);
    REAL limx = GSR_FOV_CLAMP * tanfovx, limy = GSR_FOV_CLAMP * tanfovy;
    REAL tz = pview[2];
    REAL txtz = pview[0] / tz, tytz = pview[1] / tz;
    e->clampx = (txtz < -limx) || (txtz > limx);
    e->clampy = (tytz < -limy) || (tytz > limy);
    REAL cx = txtz < -limx ? -limx : (txtz > limx ? limx : txtz);
    REAL cy = tytz < -limy ? -limy : (tytz > limy ? limy : tytz);
    REAL tx = cx * tz, ty = cy * tz;
    e->t[0] = tx; e->t[1] = ty; e->t[2] = tz;
    e->fx = fx; e->fy = fy;
    REAL J00 = fx / tz, J02 = -(fx * tx) / (tz * tz);
    REAL J11 = fy / tz, J12 = -(fy * ty) / (tz * tz);
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) e->Wm[r][c] = V[c * 4 + r];
    for (int k = 0; k < 3; k++) {
        e->T[0][k] = J00 * e->Wm[0][k] + J02 * e->Wm[2][k];
        e->T[1][k] = J11 * e->Wm[1][k] + J12 * e->Wm[2][k];
    }
    REAL S[3][3] = {{c6[0], c6[1], c6[2]}, {c6[1], c6[3], c6[4]}, {c6[2], c6[4], c6[5]}};
    for (int i = 0; i < 2; i++)
        for (int k = 0; k < 3; k++) e->TS[i][k] = e->T[i][0] * S[0][k] + e->T[i][1] * S[1][k] + e->T[i][2] * S[2][k];
    e->a = e->TS[0][0] * e->T[0][0] + e->TS[0][1] * e->T[0][1] + e->TS[0][2] * e->T[0][2] + GSR_DILATION;
    e->b = e->TS[0][0] * e->T[1][0] + e->TS[0][1] * e->T[1][1] + e->TS[0][2] * e->T[1][2];
    e->c = e->TS[1][0] * e->T[1][0] + e->TS[1][1] * e->T[1][1] + e->TS[1][2] * e->T[1][2] + GSR_DILATION;
}

/* SH basis values b[0..K) at unit direction d; sign pattern of utils/sh_utils.py:74-100 */
static void sh_basis(int deg, const REAL *d, REAL *b) {
    REAL x = d[0], y = d[1], z = d[2];
    b[0] = SH_C0;
    if (deg > 0) {
        b[1] = -SH_C1 * y; b[2] = SH_C1 * z; b[3] = -SH_C1 * x;
        if (deg > 1) {
            REAL xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
            b[4] = SH_C2[0] * xy; b[5] = SH_C2[1] * yz; b[6] = SH_C2[2] * (R(2) * zz - xx - yy);
            b[7] = SH_C2[3] * xz; b[8] = SH_C2[4] * (xx - yy);
            if (deg > 2) {
                b[9] = SH_C3[0] * y * (R(3) * xx - yy);
                b[10] = SH_C3[1] * xy * z;
                b[11] = SH_C3[2] * y * (R(4) * zz - xx - yy);
                b[12] = SH_C3[3] * z * (R(2) * zz - R(3) * xx - R(3) * yy);
                b[13] = SH_C3[4] * x * (R(4) * zz - xx - yy);
                b[14] = SH_C3[5] * z * (xx - yy);
                b[15] = SH_C3[6] * x * (xx - R(3) * yy);
            }
        }
    }
}
/* d(basis_k)/d(x,y,z) treating x,y,z as independent */
static void sh_basis_grad(int deg, const REAL *d, REAL g[16][3]) {
    REAL x = d[0], y = d[1], z = d[2];
    memset(g, 0, sizeof(REAL) * 16 * 3);
    if (deg > 0) {
        g[1][1] = -SH_C1; g[2][2] = SH_C1; g[3][0] = -SH_C1;
        if (deg > 1) {
            REAL xx = x * x, yy = y * y, zz = z * z;
            g[4][0] = SH_C2[0] * y; g[4][1] = SH_C2[0] * x;
            g[5][1] = SH_C2[1] * z; g[5][2] = SH_C2[1] * y;
            g[6][0] = SH_C2[2] * R(-2) * x; g[6][1] = SH_C2[2] * R(-2) * y; g[6][2] = SH_C2[2] * R(4) * z;
            g[7][0] = SH_C2[3] * z; g[7][2] = SH_C2[3] * x;
            g[8][0] = SH_C2[4] * R(2) * x; g[8][1] = SH_C2[4] * R(-2) * y;
            if (deg > 2) {
                g[9][0] = SH_C3[0] * R(6) * x * y; g[9][1] = SH_C3[0] * (R(3) * xx - R(3) * yy);
                g[10][0] = SH_C3[1] * y * z; g[10][1] = SH_C3[1] * x * z; g[10][2] = SH_C3[1] * x * y;
                g[11][0] = SH_C3[2] * R(-2) * x * y; g[11][1] = SH_C3[2] * (R(4) * zz - xx - R(3) * yy);
                g[11][2] = SH_C3[2] * R(8) * y * z;
                g[12][0] = SH_C3[3] * R(-6) * x * z; g[12][1] = SH_C3[3] * R(-6) * y * z;
                g[12][2] = SH_C3[3] * (R(6) * zz - R(3) * xx - R(3) * yy);
                g[13][0] = SH_C3[4] * (R(4) * zz - R(3) * xx - yy); g[13][1] = SH_C3[4] * R(-2) * x * y;
                g[13][2] = SH_C3[4] * R(8) * x * z;
                g[14][0] = SH_C3[5] * R(2) * x * z; g[14][1] = SH_C3[5] * R(-2) * y * z; g[14][2] = SH_C3[5] * (xx - yy);
                g[15][0] = SH_C3[6] * (R(3) * xx - R(3) * yy); g[15][1] = SH_C3[6] * R(-6) * x * y;
            }
        }
    }
}

static inline int clampi_from_real(REAL v, int hi) { /* clamp(int(v), 0, hi), C truncation */
    if (!(v > R(0))) return 0;       /* negatives, -0.x (truncates to 0) and NaN */
    if (v >= (REAL)hi) return hi;
    return (int)v;
}

/* tile rectangle of a splat: S5 */
static void tile_rect(REAL px, REAL py, int radius, int gridx, int gridy, int *x0, int *y0, int *x1, int *y1) {
    REAL r = (REAL)radius;
    *x0 = clampi_from_real((px - r) / R(GSR_TILE), gridx);
    *y0 = clampi_from_real((py - r) / R(GSR_TILE), gridy);
    *x1 = clampi_from_real((px + r + R(GSR_TILE - 1)) / R(GSR_TILE), gridx);
    *y1 = clampi_from_real((py + r + R(GSR_TILE - 1)) / R(GSR_TILE), gridy);
}

static void free_state(gsr_ref_state *s) {
    if (!s) return;
    free(s->depth); free(s->xy); free(s->cov3D); free(s->conic_o); free(s->rgb); free(s->radii);
    free(s->tiles_touched); free(s->offsets); free(s->clamped); free(s->keys); free(s->vals);
    free(s->ranges); free(s->final_T); free(s->n_contrib); free(s->margin); free(s);
}
void gsr_ref_free(gsr_ref_state *s) { free_state(s); }

/* stable LSD radix sort of (key,val) pairs on bits [0,nbits) */
static void radix_sort_pairs(uint64_t *k, uint32_t *v, int64_t n, int nbits) {
    if (n <= 1) return;
    uint64_t *k2 = (uint64_t *)malloc(sizeof(uint64_t) * n);
    uint32_t *v2 = (uint32_t *)malloc(sizeof(uint32_t) * n);
    uint64_t *ka = k, *kb = k2; uint32_t *va = v, *vb = v2;
    for (int shift = 0; shift < nbits; shift += 11) {
        int64_t cnt[2049]; memset(cnt, 0, sizeof(cnt));
        for (int64_t i = 0; i < n; i++) cnt[((ka[i] >> shift) & 2047) + 1]++;
        for (int i = 0; i < 2048; i++) cnt[i + 1] += cnt[i];
        for (int64_t i = 0; i < n; i++) { int64_t d = cnt[(ka[i] >> shift) & 2047]++; kb[d] = ka[i]; vb[d] = va[i]; }
        uint64_t *tk = ka; ka = kb; kb = tk; uint32_t *tv = va; va = vb; vb = tv;
    }
    if (ka != k) { memcpy(k, ka, sizeof(uint64_t) * n); memcpy(v, va, sizeof(uint32_t) * n); }
    free(k2); free(v2);
}

static double now_s(void) {
#ifdef _OPENMP
    return omp_get_wtime();
#else
    return 0.0;
#endif
}

static int ceil_log2_u32(uint32_t x) { int b = 0; while ((1u << b) < x && b < 31) b++; return b; }

/* ------------------------------------------------------------------------------------------ */
static inline int band_skips(int ty);
/* forward: S1..S9.  out_color [3,H,W], radii [P] (both caller-owned).  Returns state or NULL. */
gsr_ref_state *gsr_ref_forward(const gsr_ref_scene *sc, REAL *out_color, int32_t *radii, int32_t nthreads) {
    const int P = sc->P, W = sc->W, H = sc->H;
    gsr_ref_state *st = (gsr_ref_state *)calloc(1, sizeof(gsr_ref_state));
    st->P = P; st->W = W; st->H = H;
    st->gridx = (W + GSR_TILE - 1) / GSR_TILE; st->gridy = (H + GSR_TILE - 1) / GSR_TILE;
    const int gridx = st->gridx, gridy = st->gridy, T = gridx * gridy;
    size_t Pn = P > 0 ? (size_t)P : 1;
    st->depth = (REAL *)calloc(Pn, sizeof(REAL)); st->xy = (REAL *)calloc(Pn * 2, sizeof(REAL));
    st->cov3D = (REAL *)calloc(Pn * 6, sizeof(REAL)); st->conic_o = (REAL *)calloc(Pn * 4, sizeof(REAL));
    st->rgb = (REAL *)calloc(Pn * 3, sizeof(REAL)); st->radii = (int32_t *)calloc(Pn, sizeof(int32_t));
    st->tiles_touched = (uint32_t *)calloc(Pn, sizeof(uint32_t)); st->offsets = (uint32_t *)calloc(Pn, sizeof(uint32_t));
    st->clamped = (uint8_t *)calloc(Pn * 3, 1);
    st->ranges = (uint32_t *)calloc((size_t)T * 2 + 2, sizeof(uint32_t));
    st->final_T = (REAL *)calloc((size_t)W * H + 1, sizeof(REAL));
    st->n_contrib = (uint32_t *)calloc((size_t)W * H + 1, sizeof(uint32_t));
    st->margin = (REAL *)calloc((size_t)W * H + 1, sizeof(REAL));
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#else
    (void)nthreads;
#endif

    /* ---- S1..S6 per Gaussian ---- */
    double t0 = now_s();
#pragma omp parallel for schedule(static)
    for (int i = 0; i < P; i++) {
        radii[i] = 0; st->tiles_touched[i] = 0;
        const REAL *p = sc->means3D + 3 * (size_t)i;
        REAL pv[3], ph[4];
        xform4x3(sc->viewmatrix, p, pv);
        if (pv[2] <= GSR_NEAR_Z) continue;                       /* S1 */
        xform4x4(sc->projmatrix, p, ph);
        REAL pw = R(1) / (ph[3] + GSR_W_EPS);
        REAL ndcx = ph[0] * pw, ndcy = ph[1] * pw;
        REAL c6[6];
        if (sc->cov3D_precomp) memcpy(c6, sc->cov3D_precomp + 6 * (size_t)i, sizeof(c6));
        else cov3d_from_scale_rot(sc->scales + 3 * (size_t)i, sc->scale_modifier, sc->rotations + 4 * (size_t)i, c6); /* S2 */
        memcpy(st->cov3D + 6 * (size_t)i, c6, sizeof(c6));
        ewa_t e;
        ewa_project(pv, c6, sc->viewmatrix, sc->tanfovx, sc->tanfovy, W, H, &e);                                      /* S3 */
        REAL det = e.a * e.c - e.b * e.b;                                                                              /* S4 */
        if (det == R(0)) continue;
        REAL det_inv = R(1) / det;
        REAL conic[3] = {e.c * det_inv, -e.b * det_inv, e.a * det_inv};
        REAL mid = R(0.5) * (e.a + e.c);
        REAL disc = mid * mid - det; if (disc < GSR_LAMBDA_FLOOR) disc = GSR_LAMBDA_FLOOR;
        REAL l1 = mid + R_SQRT(disc), l2 = mid - R_SQRT(disc);
        REAL lmax = l1 > l2 ? l1 : l2;
        int radius = (int)R_CEIL(GSR_SIGMA_EXTENT * R_SQRT(lmax));
        REAL px = ((ndcx + R(1)) * (REAL)W - R(1)) * R(0.5);                                                           /* S5 */
        REAL py = ((ndcy + R(1)) * (REAL)H - R(1)) * R(0.5);
        int x0, y0, x1, y1;
        tile_rect(px, py, radius, gridx, gridy, &x0, &y0, &x1, &y1);
        if ((x1 - x0) * (y1 - y0) == 0) continue;
        REAL rgb[3];
        if (sc->colors_precomp) { memcpy(rgb, sc->colors_precomp + 3 * (size_t)i, sizeof(rgb)); }
        else {                                                                                                         /* S6 */
            REAL dir[3] = {p[0] - sc->campos[0], p[1] - sc->campos[1], p[2] - sc->campos[2]};
            REAL il = R(1) / R_SQRT(dir[0] * dir[0] + dir[1] * dir[1] + dir[2] * dir[2]);
            dir[0] *= il; dir[1] *= il; dir[2] *= il;
            REAL b[16]; sh_basis(sc->D, dir, b);
            int K = (sc->D + 1) * (sc->D + 1);
            const REAL *sh = sc->shs + (size_t)i * sc->M * 3;
            for (int ch = 0; ch < 3; ch++) {
                REAL v = R(0);
                for (int k = 0; k < K; k++) v += b[k] * sh[k * 3 + ch];
                v += R(0.5);
                st->clamped[3 * (size_t)i + ch] = (v < R(0));
                rgb[ch] = v < R(0) ? R(0) : v;
            }
        }
        st->depth[i] = pv[2];
        st->radii[i] = radius; radii[i] = radius;
        st->xy[2 * (size_t)i] = px; st->xy[2 * (size_t)i + 1] = py;
        st->conic_o[4 * (size_t)i] = conic[0]; st->conic_o[4 * (size_t)i + 1] = conic[1];
        st->conic_o[4 * (size_t)i + 2] = conic[2]; st->conic_o[4 * (size_t)i + 3] = sc->opacities[i];
        st->rgb[3 * (size_t)i] = rgb[0]; st->rgb[3 * (size_t)i + 1] = rgb[1]; st->rgb[3 * (size_t)i + 2] = rgb[2];
        st->tiles_touched[i] = (uint32_t)((x1 - x0) * (y1 - y0));
    }

    /* ---- scan ---- */
    double t1 = now_s(); st->t_fwd[0] = t1 - t0;
    uint64_t run = 0;
    for (int i = 0; i < P; i++) { run += st->tiles_touched[i]; st->offsets[i] = (uint32_t)run; }
    const int64_t N = (int64_t)run; st->N = N;
    st->keys = (uint64_t *)malloc(sizeof(uint64_t) * (N > 0 ? N : 1));
    st->vals = (uint32_t *)malloc(sizeof(uint32_t) * (N > 0 ? N : 1));

    /* ---- S7 key emission: Gaussian order, then tile row-major ---- */
#pragma omp parallel for schedule(static)
    for (int i = 0; i < P; i++) {
        if (st->radii[i] <= 0 || st->tiles_touched[i] == 0) continue;
        int64_t off = i == 0 ? 0 : st->offsets[i - 1];
        int x0, y0, x1, y1;
        tile_rect(st->xy[2 * (size_t)i], st->xy[2 * (size_t)i + 1], st->radii[i], gridx, gridy, &x0, &y0, &x1, &y1);
        float df = (float)st->depth[i]; uint32_t dbits; memcpy(&dbits, &df, 4);
        for (int y = y0; y < y1; y++)
            for (int x = x0; x < x1; x++) {
                st->keys[off] = ((uint64_t)(uint32_t)(y * gridx + x) << 32) | dbits;
                st->vals[off] = (uint32_t)i; off++;
            }
    }
    /* ---- sort (stable) on 32 + ceil_log2(T) bits ---- */
    radix_sort_pairs(st->keys, st->vals, N, 32 + ceil_log2_u32((uint32_t)T));
    /* ---- S8 ranges ---- */
    for (int64_t j = 0; j < N; j++) {
        uint32_t t = (uint32_t)(st->keys[j] >> 32);
        if (j == 0 || (uint32_t)(st->keys[j - 1] >> 32) != t) st->ranges[2 * (size_t)t] = (uint32_t)j;
        if (j == N - 1 || (uint32_t)(st->keys[j + 1] >> 32) != t) st->ranges[2 * (size_t)t + 1] = (uint32_t)(j + 1);
    }

    /* ---- S9 compositing, per pixel front to back.  P == 0: image stays zero (not bg). ---- */
    double t2 = now_s(); st->t_fwd[1] = t2 - t1;
    if (P == 0) { memset(out_color, 0, sizeof(REAL) * 3 * (size_t)W * H); return st; }
#pragma omp parallel for schedule(dynamic, 1)
    for (int t = 0; t < T; t++) {
        int tx = t % gridx, ty = t / gridx;
        if (band_skips(ty)) continue;
        uint32_t r0 = st->ranges[2 * (size_t)t], r1 = st->ranges[2 * (size_t)t + 1];
        for (int ly = 0; ly < GSR_TILE; ly++)
            for (int lx = 0; lx < GSR_TILE; lx++) {
                int x = tx * GSR_TILE + lx, y = ty * GSR_TILE + ly;
                if (x >= W || y >= H) continue;
                REAL Tr = R(1), C[3] = {0, 0, 0};
                uint32_t last = 0, pos = 0;
                /* Decision margin (test instrumentation): S9 takes three discrete decisions per pair -- power > 0,
                 * alpha < 1/255, T(1 - alpha) < 1e-4 -- and an implementation whose exp() differs in the last bit can
                 * take a borderline one the other way, which moves the pixel by up to ~alpha_min * |c|.  margin = the
                 * smallest distance of any decision this pixel took from its threshold: relative for the alpha and
                 * transmittance tests, absolute for the sign of power (whose scale is 1). */
                REAL mg = R(1);
                for (uint32_t j = r0; j < r1; j++) {
                    pos++;
                    uint32_t g = st->vals[j];
                    REAL dx = st->xy[2 * (size_t)g] - (REAL)x, dy = st->xy[2 * (size_t)g + 1] - (REAL)y;
                    const REAL *co = st->conic_o + 4 * (size_t)g;
                    REAL power = R(-0.5) * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
                    REAL araw = co[3] * R_EXP(power);
                    if (araw >= GSR_ALPHA_MIN * R(0.5)) { REAL m = power < 0 ? -power : power; if (m < mg) mg = m; }
                    if (power > R(0)) continue;
                    REAL alpha = araw; if (alpha > GSR_ALPHA_MAX) alpha = GSR_ALPHA_MAX;
                    { REAL m = (alpha - GSR_ALPHA_MIN) / GSR_ALPHA_MIN; if (m < 0) m = -m; if (m < mg) mg = m; }
                    if (alpha < GSR_ALPHA_MIN) continue;
                    REAL Tn = Tr * (R(1) - alpha);
                    { REAL m = (Tn - GSR_T_MIN) / GSR_T_MIN; if (m < 0) m = -m; if (m < mg) mg = m; }
                    if (Tn < GSR_T_MIN) break;
                    const REAL *c = st->rgb + 3 * (size_t)g;
                    for (int ch = 0; ch < 3; ch++) C[ch] += c[ch] * alpha * Tr;
                    Tr = Tn; last = pos;
                }
                size_t pix = (size_t)y * W + x;
                st->final_T[pix] = Tr; st->n_contrib[pix] = last; st->margin[pix] = mg;
                for (int ch = 0; ch < 3; ch++) out_color[(size_t)ch * W * H + pix] = C[ch] + Tr * sc->bg[ch];
            }
    }
    st->t_fwd[2] = now_s() - t2;
    return st;
}

/* Timing aid for bench.py's single-thread baseline leg: restrict the two compositing stages (S9, S10) to the tile rows
 * [ty0, ty1) so that a bounded sample of a large render can be timed; ty1 <= ty0 (the default) = the whole image.  Pixels
 * outside the band keep colour 0 / n_contrib 0 and contribute no gradient: NOT a valid render, timing only. */
static int g_band_ty0 = 0, g_band_ty1 = 0;
void gsr_ref_set_tile_row_band(int32_t ty0, int32_t ty1) { g_band_ty0 = ty0; g_band_ty1 = ty1; }
static inline int band_skips(int ty) { return g_band_ty1 > g_band_ty0 && (ty < g_band_ty0 || ty >= g_band_ty1); }

static inline void atomic_add_real(REAL *p, REAL v) {
#pragma omp atomic
    *p += v;
}

/* ------------------------------------------------------------------------------------------ */
/* backward: S10..S13.  All gradient buffers caller-owned and are OVERWRITTEN (zero-filled     */
/* first).  dL_dconic is [P,4] scratch-visible (xx, xy(half), unused, yy) for stage tests.     */
int gsr_ref_backward(const gsr_ref_scene *sc, gsr_ref_state *st, const REAL *dL_dpix,
                     REAL *dL_dmeans2D /*[P,3]*/, REAL *dL_dconic /*[P,4]*/, REAL *dL_dopacity /*[P]*/,
                     REAL *dL_dcolors /*[P,3]*/, REAL *dL_dmeans3D /*[P,3]*/, REAL *dL_dcov3D /*[P,6]*/,
                     REAL *dL_dsh /*[P,M,3] or NULL*/, REAL *dL_dscales /*[P,3] or NULL*/, REAL *dL_drots /*[P,4] or NULL*/,
                     int32_t nthreads) {
    const int P = sc->P, W = sc->W, H = sc->H, gridx = st->gridx, gridy = st->gridy, T = gridx * gridy;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#else
    (void)nthreads;
#endif
    memset(dL_dmeans2D, 0, sizeof(REAL) * 3 * (size_t)P); memset(dL_dconic, 0, sizeof(REAL) * 4 * (size_t)P);
    memset(dL_dopacity, 0, sizeof(REAL) * (size_t)P); memset(dL_dcolors, 0, sizeof(REAL) * 3 * (size_t)P);
    memset(dL_dmeans3D, 0, sizeof(REAL) * 3 * (size_t)P); memset(dL_dcov3D, 0, sizeof(REAL) * 6 * (size_t)P);
    if (dL_dsh) memset(dL_dsh, 0, sizeof(REAL) * 3 * (size_t)P * sc->M);
    if (dL_dscales) memset(dL_dscales, 0, sizeof(REAL) * 3 * (size_t)P);
    if (dL_drots) memset(dL_drots, 0, sizeof(REAL) * 4 * (size_t)P);
    if (P == 0) return 0;

    /* ---- S10: per pixel, back to front from the last contributor ---- */
    double tb0 = now_s();
#pragma omp parallel for schedule(dynamic, 1)
    for (int t = 0; t < T; t++) {
        int tx = t % gridx, ty = t / gridx;
        if (band_skips(ty)) continue;
        uint32_t r0 = st->ranges[2 * (size_t)t];
        for (int ly = 0; ly < GSR_TILE; ly++)
            for (int lx = 0; lx < GSR_TILE; lx++) {
                int x = tx * GSR_TILE + lx, y = ty * GSR_TILE + ly;
                if (x >= W || y >= H) continue;
                size_t pix = (size_t)y * W + x;
                const REAL Tfinal = st->final_T[pix];
                REAL Tr = Tfinal;
                REAL dLdC[3] = {dL_dpix[pix], dL_dpix[(size_t)W * H + pix], dL_dpix[2 * (size_t)W * H + pix]};
                REAL accum[3] = {0, 0, 0}, last_alpha = 0, last_c[3] = {0, 0, 0};
                REAL bg_dot = sc->bg[0] * dLdC[0] + sc->bg[1] * dLdC[1] + sc->bg[2] * dLdC[2];
                for (int64_t k = (int64_t)st->n_contrib[pix] - 1; k >= 0; k--) {
                    uint32_t g = st->vals[r0 + k];
                    REAL dx = st->xy[2 * (size_t)g] - (REAL)x, dy = st->xy[2 * (size_t)g + 1] - (REAL)y;
                    const REAL *co = st->conic_o + 4 * (size_t)g;
                    REAL power = R(-0.5) * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
                    if (power > R(0)) continue;
                    REAL G = R_EXP(power);
                    REAL alpha = co[3] * G; if (alpha > GSR_ALPHA_MAX) alpha = GSR_ALPHA_MAX;
                    if (alpha < GSR_ALPHA_MIN) continue;
                    Tr = Tr / (R(1) - alpha);
                    const REAL *c = st->rgb + 3 * (size_t)g;
                    REAL dL_dalpha = 0, w = alpha * Tr;
                    for (int ch = 0; ch < 3; ch++) {
                        accum[ch] = last_alpha * last_c[ch] + (R(1) - last_alpha) * accum[ch];
                        last_c[ch] = c[ch];
                        dL_dalpha += (c[ch] - accum[ch]) * dLdC[ch];
                        atomic_add_real(&dL_dcolors[3 * (size_t)g + ch], w * dLdC[ch]);
                    }
                    dL_dalpha *= Tr;
                    last_alpha = alpha;
                    dL_dalpha += (-Tfinal / (R(1) - alpha)) * bg_dot;
                    REAL dL_dG = co[3] * dL_dalpha;
                    REAL gdx = G * dx, gdy = G * dy;
                    REAL dG_ddx = -gdx * co[0] - gdy * co[1], dG_ddy = -gdy * co[2] - gdx * co[1];
                    /* gradient w.r.t. NDC coordinates (pixel = ((ndc+1)*W-1)/2  =>  d pix/d ndc = W/2) */
                    atomic_add_real(&dL_dmeans2D[3 * (size_t)g], dL_dG * dG_ddx * R(0.5) * (REAL)W);
                    atomic_add_real(&dL_dmeans2D[3 * (size_t)g + 1], dL_dG * dG_ddy * R(0.5) * (REAL)H);
                    atomic_add_real(&dL_dconic[4 * (size_t)g], R(-0.5) * gdx * dx * dL_dG);
                    atomic_add_real(&dL_dconic[4 * (size_t)g + 1], R(-0.5) * gdx * dy * dL_dG);
                    atomic_add_real(&dL_dconic[4 * (size_t)g + 3], R(-0.5) * gdy * dy * dL_dG);
                    atomic_add_real(&dL_dopacity[g], G * dL_dalpha);
                }
            }
    }

    /* ---- S11..S13 per Gaussian ---- */
    double tb1 = now_s(); st->t_bwd[0] = tb1 - tb0;
    const int K = (sc->D + 1) * (sc->D + 1);
#pragma omp parallel for schedule(static)
    for (int i = 0; i < P; i++) {
        if (st->radii[i] <= 0) continue;
        const REAL *p = sc->means3D + 3 * (size_t)i;
        REAL dmean[3] = {0, 0, 0};
        /* S11: conic -> cov2D -> cov3D, mean (through the Jacobian) */
        REAL pv[3]; xform4x3(sc->viewmatrix, p, pv);
        const REAL *c6 = st->cov3D + 6 * (size_t)i;
        ewa_t e; ewa_project(pv, c6, sc->viewmatrix, sc->tanfovx, sc->tanfovy, W, H, &e);
        REAL a = e.a, b = e.b, c = e.c;
        REAL det = a * c - b * b;
        REAL d2inv = R(1) / (det * det + GSR_DENOM_EPS);
        REAL gA = dL_dconic[4 * (size_t)i], gB = dL_dconic[4 * (size_t)i + 1], gC = dL_dconic[4 * (size_t)i + 3];
        REAL dcov[6] = {0, 0, 0, 0, 0, 0};
        if (d2inv != R(0)) {
            REAL dL_da = d2inv * (-c * c * gA + R(2) * b * c * gB + (det - a * c) * gC);
            REAL dL_dc = d2inv * (-a * a * gC + R(2) * a * b * gB + (det - a * c) * gA);
            REAL dL_db = d2inv * R(2) * (b * c * gA - (det + R(2) * b * b) * gB + a * b * gC);
            const REAL(*Tm)[3] = e.T;
            dcov[0] = Tm[0][0] * Tm[0][0] * dL_da + Tm[0][0] * Tm[1][0] * dL_db + Tm[1][0] * Tm[1][0] * dL_dc;
            dcov[3] = Tm[0][1] * Tm[0][1] * dL_da + Tm[0][1] * Tm[1][1] * dL_db + Tm[1][1] * Tm[1][1] * dL_dc;
            dcov[5] = Tm[0][2] * Tm[0][2] * dL_da + Tm[0][2] * Tm[1][2] * dL_db + Tm[1][2] * Tm[1][2] * dL_dc;
            dcov[1] = R(2) * Tm[0][0] * Tm[0][1] * dL_da + (Tm[0][0] * Tm[1][1] + Tm[0][1] * Tm[1][0]) * dL_db + R(2) * Tm[1][0] * Tm[1][1] * dL_dc;
            dcov[2] = R(2) * Tm[0][0] * Tm[0][2] * dL_da + (Tm[0][0] * Tm[1][2] + Tm[0][2] * Tm[1][0]) * dL_db + R(2) * Tm[1][0] * Tm[1][2] * dL_dc;
            dcov[4] = R(2) * Tm[0][2] * Tm[0][1] * dL_da + (Tm[0][1] * Tm[1][2] + Tm[0][2] * Tm[1][1]) * dL_db + R(2) * Tm[1][1] * Tm[1][2] * dL_dc;
            REAL dT[2][3];
            for (int k = 0; k < 3; k++) {
                dT[0][k] = R(2) * dL_da * e.TS[0][k] + dL_db * e.TS[1][k];
                dT[1][k] = dL_db * e.TS[0][k] + R(2) * dL_dc * e.TS[1][k];
            }
            REAL dJ00 = 0, dJ02 = 0, dJ11 = 0, dJ12 = 0;
            for (int k = 0; k < 3; k++) {
                dJ00 += dT[0][k] * e.Wm[0][k]; dJ02 += dT[0][k] * e.Wm[2][k];
                dJ11 += dT[1][k] * e.Wm[1][k]; dJ12 += dT[1][k] * e.Wm[2][k];
            }
            REAL tz = R(1) / e.t[2], tz2 = tz * tz, tz3 = tz2 * tz;
            REAL dtx = (e.clampx ? R(0) : R(1)) * (-e.fx * tz2 * dJ02);
            REAL dty = (e.clampy ? R(0) : R(1)) * (-e.fy * tz2 * dJ12);
            REAL dtz = -e.fx * tz2 * dJ00 - e.fy * tz2 * dJ11 + R(2) * e.fx * e.t[0] * tz3 * dJ02 + R(2) * e.fy * e.t[1] * tz3 * dJ12;
            for (int k = 0; k < 3; k++) dmean[k] += e.Wm[0][k] * dtx + e.Wm[1][k] * dty + e.Wm[2][k] * dtz;
        }
        for (int k = 0; k < 6; k++) dL_dcov3D[6 * (size_t)i + k] = dcov[k];

        /* S12a: NDC mean gradient -> mean3D through the perspective divide */
        {
            REAL ph[4]; xform4x4(sc->projmatrix, p, ph);
            REAL mw = R(1) / (ph[3] + GSR_W_EPS);
            REAL mul1 = ph[0] * mw * mw, mul2 = ph[1] * mw * mw;
            const REAL *Pm = sc->projmatrix;
            REAL gx = dL_dmeans2D[3 * (size_t)i], gy = dL_dmeans2D[3 * (size_t)i + 1];
            for (int k = 0; k < 3; k++)
                dmean[k] += (Pm[4 * k] * mw - Pm[4 * k + 3] * mul1) * gx + (Pm[4 * k + 1] * mw - Pm[4 * k + 3] * mul2) * gy;
        }
        /* S12b: colour -> SH coefficients and view direction */
        if (sc->shs && dL_dsh) {
            REAL v[3] = {p[0] - sc->campos[0], p[1] - sc->campos[1], p[2] - sc->campos[2]};
            REAL len2 = v[0] * v[0] + v[1] * v[1] + v[2] * v[2], il = R(1) / R_SQRT(len2);
            REAL dir[3] = {v[0] * il, v[1] * il, v[2] * il};
            REAL bas[16], bg3[16][3];
            sh_basis(sc->D, dir, bas); sh_basis_grad(sc->D, dir, bg3);
            const REAL *sh = sc->shs + (size_t)i * sc->M * 3;
            REAL ddir[3] = {0, 0, 0};
            for (int ch = 0; ch < 3; ch++) {
                REAL gch = st->clamped[3 * (size_t)i + ch] ? R(0) : dL_dcolors[3 * (size_t)i + ch];
                for (int k = 0; k < K; k++) {
                    dL_dsh[((size_t)i * sc->M + k) * 3 + ch] = bas[k] * gch;
                    for (int ax = 0; ax < 3; ax++) ddir[ax] += bg3[k][ax] * sh[k * 3 + ch] * gch;
                }
            }
            /* through dir = v/|v| */
            REAL dot = dir[0] * ddir[0] + dir[1] * ddir[1] + dir[2] * ddir[2];
            for (int ax = 0; ax < 3; ax++) dmean[ax] += (ddir[ax] - dir[ax] * dot) * il;
        }
        for (int k = 0; k < 3; k++) dL_dmeans3D[3 * (size_t)i + k] = dmean[k];

        /* S13: cov3D -> scale, quaternion (no normalisation Jacobian) */
        if (sc->scales && dL_dscales && dL_drots) {
            const REAL *s = sc->scales + 3 * (size_t)i, *q = sc->rotations + 4 * (size_t)i;
            REAL Rm[3][3]; quat_to_rot(q, Rm);
            REAL sv[3] = {sc->scale_modifier * s[0], sc->scale_modifier * s[1], sc->scale_modifier * s[2]};
            REAL Ds[3][3] = {{dcov[0], R(0.5) * dcov[1], R(0.5) * dcov[2]},
                             {R(0.5) * dcov[1], dcov[3], R(0.5) * dcov[4]},
                             {R(0.5) * dcov[2], R(0.5) * dcov[4], dcov[5]}};
            REAL dM[3][3], dR[3][3];
            for (int r = 0; r < 3; r++)
                for (int j = 0; j < 3; j++) {
                    REAL acc = 0;
                    for (int l = 0; l < 3; l++) acc += Ds[r][l] * Rm[l][j] * sv[j];
                    dM[r][j] = R(2) * acc;
                }
            for (int j = 0; j < 3; j++) {
                REAL acc = 0;
                for (int r = 0; r < 3; r++) { acc += Rm[r][j] * dM[r][j]; dR[r][j] = dM[r][j] * sv[j]; }
                dL_dscales[3 * (size_t)i + j] = sc->scale_modifier * acc;
            }
            REAL r = q[0], x = q[1], y = q[2], z = q[3];
            dL_drots[4 * (size_t)i + 0] = R(2) * (-z * dR[0][1] + y * dR[0][2] + z * dR[1][0] - x * dR[1][2] - y * dR[2][0] + x * dR[2][1]);
            dL_drots[4 * (size_t)i + 1] = R(2) * (y * dR[0][1] + z * dR[0][2] + y * dR[1][0] - R(2) * x * dR[1][1] - r * dR[1][2] + z * dR[2][0] + r * dR[2][1] - R(2) * x * dR[2][2]);
            dL_drots[4 * (size_t)i + 2] = R(2) * (-R(2) * y * dR[0][0] + x * dR[0][1] + r * dR[0][2] + x * dR[1][0] + z * dR[1][2] - r * dR[2][0] + z * dR[2][1] - R(2) * y * dR[2][2]);
            dL_drots[4 * (size_t)i + 3] = R(2) * (-R(2) * z * dR[0][0] - r * dR[0][1] + x * dR[0][2] + r * dR[1][0] - R(2) * z * dR[1][1] + y * dR[1][2] + x * dR[2][0] + y * dR[2][1]);
        }
    }
    st->t_bwd[1] = now_s() - tb1;
    return 0;
}

/* markVisible: near-plane test only (unused by the reference: SURVEY.md 2b) */
void gsr_ref_mark_visible(int32_t P, const REAL *means3D, const REAL *viewmatrix, uint8_t *present) {
    for (int i = 0; i < P; i++) {
        REAL pv[3]; xform4x3(viewmatrix, means3D + 3 * (size_t)i, pv);
        present[i] = pv[2] > GSR_NEAR_Z;
    }
}

/* ---- accessors (stage-level tests) ---- */
int64_t gsr_ref_num_rendered(const gsr_ref_state *s) { return s->N; }
void gsr_ref_get_geom(const gsr_ref_state *s, REAL *depth, REAL *xy, REAL *conic_o, REAL *rgb, REAL *cov3D,
                      uint32_t *tiles_touched, uint8_t *clamped) {
    size_t P = (size_t)s->P;
    if (depth) memcpy(depth, s->depth, sizeof(REAL) * P);
    if (xy) memcpy(xy, s->xy, sizeof(REAL) * 2 * P);
    if (conic_o) memcpy(conic_o, s->conic_o, sizeof(REAL) * 4 * P);
    if (rgb) memcpy(rgb, s->rgb, sizeof(REAL) * 3 * P);
    if (cov3D) memcpy(cov3D, s->cov3D, sizeof(REAL) * 6 * P);
    if (tiles_touched) memcpy(tiles_touched, s->tiles_touched, sizeof(uint32_t) * P);
    if (clamped) memcpy(clamped, s->clamped, 3 * P);
}
void gsr_ref_get_binning(const gsr_ref_state *s, uint64_t *keys, uint32_t *vals, uint32_t *ranges) {
    if (keys) memcpy(keys, s->keys, sizeof(uint64_t) * (size_t)s->N);
    if (vals) memcpy(vals, s->vals, sizeof(uint32_t) * (size_t)s->N);
    if (ranges) memcpy(ranges, s->ranges, sizeof(uint32_t) * 2 * (size_t)s->gridx * s->gridy);
}
void gsr_ref_get_image_state(const gsr_ref_state *s, REAL *final_T, uint32_t *n_contrib) {
    size_t n = (size_t)s->W * s->H;
    if (final_T) memcpy(final_T, s->final_T, sizeof(REAL) * n);
    if (n_contrib) memcpy(n_contrib, s->n_contrib, sizeof(uint32_t) * n);
}
void gsr_ref_get_margin(const gsr_ref_state *s, REAL *margin) {
    memcpy(margin, s->margin, sizeof(REAL) * (size_t)s->W * s->H);
}
void gsr_ref_get_timings(const gsr_ref_state *s, double *fwd3, double *bwd2) {
    for (int i = 0; i < 3; i++) fwd3[i] = s->t_fwd[i];
    for (int i = 0; i < 2; i++) bwd2[i] = s->t_bwd[i];
}
int32_t gsr_ref_real_bytes(void) { return (int32_t)sizeof(REAL); }
int32_t gsr_ref_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
