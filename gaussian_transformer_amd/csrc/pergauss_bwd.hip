// pergauss_bwd.hip -- per-Gaussian backward chain (S11-S13), one fused kernel, one lane per
// Gaussian: conic -> 2D covariance -> 3D covariance and mean (through the EWA Jacobian),
// NDC mean gradient -> mean3D (perspective divide), colour -> SH coefficients and view
// direction, 3D covariance -> scale and quaternion.  Every output element is written exactly
// once (zeros for culled Gaussians), so no gradient buffer needs a zero-fill.
// HBM-streaming: reads 64 (accumulator row) + 44 + 12*K bytes, writes 64 + 12*M bytes per
// Gaussian.  Compiled with -ffp-contract=off to track oracle/gsr_ref.c.
#include <atomic>
#include "gsr_device.h"
#include "gsr_internal.h"

namespace gsr {

#define SH_TILE_ROW 13      // float4 per Gaussian row in the LDS tile: 12 + 1 of padding (52 floats: 16-byte LDS stores of
                            // eight consecutive rows then start in banks 0, 20, 8, 28, 16, 4, 24, 12 -- conflict-free)

// RAW / SPLIT: fused-step extension (raw parameters / split SH tensors) as separate instantiations, so the
// reference path keeps its register budget.
// DENSE: `i` comes from the workgroup's list of Gaussians with a gradient (pergauss_bwd_dense_kernel below): known visible, its rows
// are not next to its lane neighbours' (no LDS tile), nothing is written for anybody else (the rows were zero-filled, fill_zero_kernel)
// rows / nrows (DENSE): the wave's 64 list entries in LDS and how many of them exist: lanes beyond take part in copying the
// wave's dL/dshs rows out (each row to its own Gaussian's place), nothing else
template <int D, bool RAW, bool SPLIT, bool DENSE>
__device__ __forceinline__ void pergauss_one(const PergaussBwdArgs &a, const int i_in, float4 *sh_tile_dyn, const uint32_t *rows = nullptr,
                                             const int nrows = 0, const float4 *pre = nullptr) {
    if (!DENSE && i_in >= a.P) return;
    const bool active = !DENSE || (int)(threadIdx.x & 63) < nrows;
    const int i = active ? i_in : 0;
    const size_t si = (size_t)i;
    constexpr int K = (D + 1) * (D + 1);
    // radii and the flag byte are requested together and awaited together (the asm ties them: left alone, the compiler sinks the
    // byte load below the branch on radii, and the branch on its bit 7 then costs a second memory round trip)
    int rad = DENSE ? 1 : a.radii[si];
    uint32_t cl = DENSE ? __float_as_uint(pre[1].w) : a.clamped[si];         // bits 0-2: SH clamp mask; bit 7: the splat has replica accumulator rows
    uint32_t tch = DENSE ? 0u : a.touched[si];
    if (!DENSE) asm volatile("" : "+v"(rad), "+v"(cl), "+v"(tch));
    // byte != this frame's mark: no wave of the forward pass staged this Gaussian with a reachable block, so the reverse pass never met
    // it and all its gradients are 0: only the zeros are written (91 % of the Gaussians at config 3, whose dense cloud is mostly occluded)
    const bool visible = DENSE ? active : (rad > 0 && tch == *a.touch_mark);
    // skip_unmarked: the rows of the Gaussians that are not `visible` have been zero-filled already (the persistent reverse compositing
    // kernel's fill units, composite_bwd.hip, or fill_zero_kernel below); nothing is written for them here
    const bool writes = visible || !a.skip_unmarked;
    const unsigned long long wave_writes = DENSE ? 0ull : __builtin_amdgcn_ballot_w64(writes);
    // dL/dshs rows of a whole wave (64 Gaussians x 192 B at M = 16) are contiguous in memory: the lanes put their rows into a
    // wave-private LDS tile and the wave copies the tile out with 16-byte stores at consecutive addresses (12 x 1 KiB), instead
    // of 48 dword stores per lane at a 192-byte stride that leave every 128-byte line half written 48 times over.
    // Wave-uniform: M = 16, not the split layout, 16-byte aligned, every lane of the wave alive.
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int wave_first = blockIdx.x * 256 + wave * 64;
    const bool tile_path = !(SPLIT) && a.sh_tile && (DENSE || wave_first + 64 <= a.P);
    float4 *tile = sh_tile_dyn + wave * (64 * SH_TILE_ROW);

    float dmean[3] = {0.f, 0.f, 0.f}, dcov[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float dcol[3] = {0.f, 0.f, 0.f}, dm2[2] = {0.f, 0.f}, dop = 0.f;
    float dscale[3] = {0.f, 0.f, 0.f}, drot[4] = {0.f, 0.f, 0.f, 0.f};
    float q_inv_norm = 1.f;

    if (visible) {
        const float4 *acc4 = reinterpret_cast<const float4 *>(a.acc) + 4 * si;
        float4 A0 = acc4[0], A1 = acc4[1];
        float A8 = a.acc[GSR_ACC_FLOATS * si + 8];

        // DENSE: means, scales, rotation, opacity and the SH row come in `pre`, the Gaussian's record in the compact buffer that
        // gather_visible_kernel filled while the compositing kernel ran (or that the caller gathered itself): (mean, opacity)
        // (scale, flags) (rotation) (12 x SH)
        float p[3], c6[6], s[3] = {0.f, 0.f, 0.f}, q[4] = {1.f, 0.f, 0.f, 0.f}, opac, c[3 * K + 3];
        if (DENSE) {
            p[0] = pre[0].x; p[1] = pre[0].y; p[2] = pre[0].z; opac = pre[0].w;
            s[0] = pre[1].x; s[1] = pre[1].y; s[2] = pre[1].z;
            q[0] = pre[2].x; q[1] = pre[2].y; q[2] = pre[2].z; q[3] = pre[2].w;
#pragma unroll
            for (int v = 0; v < (3 * K + 3) / 4; v++) {
                c[4 * v] = pre[3 + v].x;
                if (4 * v + 1 < 3 * K + 3) c[4 * v + 1] = pre[3 + v].y;
                if (4 * v + 2 < 3 * K + 3) c[4 * v + 2] = pre[3 + v].z;
                if (4 * v + 3 < 3 * K + 3) c[4 * v + 3] = pre[3 + v].w;
            }
        } else {
            p[0] = a.means3D[3 * si]; p[1] = a.means3D[3 * si + 1]; p[2] = a.means3D[3 * si + 2];
            if (a.cov3D_precomp) {
#pragma unroll
                for (int k = 0; k < 6; k++) c6[k] = a.cov3D_precomp[6 * si + k];
            } else {
                s[0] = a.scales[3 * si]; s[1] = a.scales[3 * si + 1]; s[2] = a.scales[3 * si + 2];
                const float4 q4 = reinterpret_cast<const float4 *>(a.rotations)[si];
                q[0] = q4.x; q[1] = q4.y; q[2] = q4.z; q[3] = q4.w;
            }
            opac = a.opac[si];
            if (a.shs) {
                if (SPLIT) load_sh_row_split<K>(a.shs, a.shs_rest, si, a.M, c);
                else load_sh_row<K>(a.shs, si, a.M, c);
            }
        }
        // DENSE: few waves per SIMD, nobody covers a wave's memory latency: the accumulator row is requested here, with the record
        // (the compiler may not move a load across the barrier)
        if (DENSE) asm volatile("" ::: "memory");
        if (cl & 0x80u) {                                     // a splat over hundreds of tiles: its waves added into replica rows
            const uint32_t hot = a.hot[si];
            const size_t first = (size_t)a.P + (hot >> 4);
            for (uint32_t k = 0; k < (1u << (hot & 15u)); k++) {
                const float4 *r4 = reinterpret_cast<const float4 *>(a.acc) + 4 * (first + k);
                const float4 B0 = r4[0], B1 = r4[1];
                A0.x += B0.x; A0.y += B0.y; A0.z += B0.z; A0.w += B0.w;
                A1.x += B1.x; A1.y += B1.y; A1.z += B1.z; A1.w += B1.w;
                A8 += a.acc[GSR_ACC_FLOATS * (first + k) + 8];
            }
        }
        dcol[0] = A0.x; dcol[1] = A0.y; dcol[2] = A0.z;
        float pv[3];
        xform4x3(a.viewmatrix, p, pv);
        if (!a.cov3D_precomp) {
            if (RAW) {
                s[0] = expf(s[0]); s[1] = expf(s[1]); s[2] = expf(s[2]);
                act_normalize4(q, q_inv_norm);
            }
            cov3d_from_scale_rot(s, a.scale_modifier, q, c6);
        }
        // ---- S11 ----
        Ewa e;
        ewa_project(pv, c6, a.viewmatrix, a.tanfovx, a.tanfovy, a.W, a.H, e);
        const float ea = e.a, eb = e.b, ec = e.c;
        const float det = ea * ec - eb * eb;
        // The compositing kernel (composite_bwd.hip) accumulates moments of s = opacity * G * dL/dalpha over the pixels:
        // A0.w, A1.x = sum s*d;  A1.yzw = sum s * d d^T;  A8 = sum s.  The factors that are constant per Gaussian
        // are applied here:
        //   mean2D gradient w.r.t. NDC = -(W/2, H/2) * conic * sum s*d;   conic gradient = -1/2 * sum s * d d^T;
        //   dL/dopacity = sum G * dL/dalpha = (sum s) / opacity   (a Gaussian that blends anywhere has opacity >= 1/255)
        dop = opac > 0.f ? A8 / opac : 0.f;
        const float det_inv = 1.f / det;                        // the conic as the forward pass formed it (S4)
        const float cA = ec * det_inv, cB = -eb * det_inv, cC = ea * det_inv;
        dm2[0] = -0.5f * (float)a.W * (cA * A0.w + cB * A1.x); dm2[1] = -0.5f * (float)a.H * (cB * A0.w + cC * A1.x);
        const float gA = -0.5f * A1.y, gB = -0.5f * A1.z, gC = -0.5f * A1.w;
        const float d2inv = 1.f / (det * det + GSR_DENOM_EPS);
        if (d2inv != 0.f) {
            const float dL_da = d2inv * (-ec * ec * gA + 2.f * eb * ec * gB + (det - ea * ec) * gC);
            const float dL_dc = d2inv * (-ea * ea * gC + 2.f * ea * eb * gB + (det - ea * ec) * gA);
            const float dL_db = d2inv * 2.f * (eb * ec * gA - (det + 2.f * eb * eb) * gB + ea * eb * gC);
            const float(*Tm)[3] = e.T;
            dcov[0] = Tm[0][0] * Tm[0][0] * dL_da + Tm[0][0] * Tm[1][0] * dL_db + Tm[1][0] * Tm[1][0] * dL_dc;
            dcov[3] = Tm[0][1] * Tm[0][1] * dL_da + Tm[0][1] * Tm[1][1] * dL_db + Tm[1][1] * Tm[1][1] * dL_dc;
            dcov[5] = Tm[0][2] * Tm[0][2] * dL_da + Tm[0][2] * Tm[1][2] * dL_db + Tm[1][2] * Tm[1][2] * dL_dc;
            dcov[1] = 2.f * Tm[0][0] * Tm[0][1] * dL_da + (Tm[0][0] * Tm[1][1] + Tm[0][1] * Tm[1][0]) * dL_db + 2.f * Tm[1][0] * Tm[1][1] * dL_dc;
            dcov[2] = 2.f * Tm[0][0] * Tm[0][2] * dL_da + (Tm[0][0] * Tm[1][2] + Tm[0][2] * Tm[1][0]) * dL_db + 2.f * Tm[1][0] * Tm[1][2] * dL_dc;
            dcov[4] = 2.f * Tm[0][2] * Tm[0][1] * dL_da + (Tm[0][1] * Tm[1][2] + Tm[0][2] * Tm[1][1]) * dL_db + 2.f * Tm[1][1] * Tm[1][2] * dL_dc;
            float dT[2][3];
#pragma unroll
            for (int k = 0; k < 3; k++) {
                dT[0][k] = 2.f * dL_da * e.TS[0][k] + dL_db * e.TS[1][k];
                dT[1][k] = dL_db * e.TS[0][k] + 2.f * dL_dc * e.TS[1][k];
            }
            float dJ00 = 0.f, dJ02 = 0.f, dJ11 = 0.f, dJ12 = 0.f;
#pragma unroll
            for (int k = 0; k < 3; k++) {
                dJ00 += dT[0][k] * e.Wm[0][k]; dJ02 += dT[0][k] * e.Wm[2][k];
                dJ11 += dT[1][k] * e.Wm[1][k]; dJ12 += dT[1][k] * e.Wm[2][k];
            }
            const float tz = 1.f / e.t[2], tz2 = tz * tz, tz3 = tz2 * tz;
            const float dtx = (e.clampx ? 0.f : 1.f) * (-e.fx * tz2 * dJ02);
            const float dty = (e.clampy ? 0.f : 1.f) * (-e.fy * tz2 * dJ12);
            const float dtz = -e.fx * tz2 * dJ00 - e.fy * tz2 * dJ11 + 2.f * e.fx * e.t[0] * tz3 * dJ02 + 2.f * e.fy * e.t[1] * tz3 * dJ12;
#pragma unroll
            for (int k = 0; k < 3; k++) dmean[k] += e.Wm[0][k] * dtx + e.Wm[1][k] * dty + e.Wm[2][k] * dtz;
        }
        // ---- S12a ----
        {
            float ph[4];
            xform4x4(a.projmatrix, p, ph);
            const float mw = 1.f / (ph[3] + GSR_W_EPS);
            const float mul1 = ph[0] * mw * mw, mul2 = ph[1] * mw * mw;
            const float *Pm = a.projmatrix;
#pragma unroll
            for (int k = 0; k < 3; k++)
                dmean[k] += (Pm[4 * k] * mw - Pm[4 * k + 3] * mul1) * dm2[0] + (Pm[4 * k + 1] * mw - Pm[4 * k + 3] * mul2) * dm2[1];
        }
        // ---- S12b ----
        if (a.shs) {
            const float v[3] = {p[0] - a.campos[0], p[1] - a.campos[1], p[2] - a.campos[2]};
            const float len2 = v[0] * v[0] + v[1] * v[1] + v[2] * v[2], il = 1.f / sqrtf(len2);
            const float dir[3] = {v[0] * il, v[1] * il, v[2] * il};
            float bas[16], bg3[16][3];
            sh_basis<D>(dir, bas);
            sh_basis_grad<D>(dir, bg3);
            float ddir[3] = {0.f, 0.f, 0.f};
            float *out = (SPLIT) ? nullptr : a.dL_dsh + si * (size_t)a.M * 3;
            float gch[3];
#pragma unroll
            for (int ch = 0; ch < 3; ch++) gch[ch] = ((cl >> ch) & 1) ? 0.f : dcol[ch];
#pragma unroll
            for (int ch = 0; ch < 3; ch++)
#pragma unroll
                for (int k = 0; k < K; k++)
#pragma unroll
                    for (int ax = 0; ax < 3; ax++) ddir[ax] += bg3[k][ax] * c[k * 3 + ch] * gch[ch];
            if (!(SPLIT) && tile_path) {
                float row[48];
#pragma unroll
                for (int k = 0; k < 16; k++) {
                    row[k * 3 + 0] = k < K ? bas[k] * gch[0] : 0.f; row[k * 3 + 1] = k < K ? bas[k] * gch[1] : 0.f;
                    row[k * 3 + 2] = k < K ? bas[k] * gch[2] : 0.f;
                }
#pragma unroll
                for (int v = 0; v < 12; v++) tile[lane * SH_TILE_ROW + v] = make_float4(row[4 * v], row[4 * v + 1], row[4 * v + 2], row[4 * v + 3]);
            } else if (!(SPLIT)) {
#pragma unroll
                for (int k = 0; k < K; k++) {
                    out[k * 3 + 0] = bas[k] * gch[0]; out[k * 3 + 1] = bas[k] * gch[1]; out[k * 3 + 2] = bas[k] * gch[2];
                }
                for (int k = 3 * K; k < 3 * a.M; k++) out[k] = 0.f;
            } else {                                   // dc [P,1,3] and rest [P,M-1,3] written separately
                float *odc = a.dL_dsh + 3 * si, *orest = a.dL_dsh_rest + si * (size_t)(a.M - 1) * 3;
                odc[0] = bas[0] * gch[0]; odc[1] = bas[0] * gch[1]; odc[2] = bas[0] * gch[2];
#pragma unroll
                for (int k = 1; k < K; k++) {
                    orest[(k - 1) * 3 + 0] = bas[k] * gch[0]; orest[(k - 1) * 3 + 1] = bas[k] * gch[1]; orest[(k - 1) * 3 + 2] = bas[k] * gch[2];
                }
                for (int k = 3 * (K - 1); k < 3 * (a.M - 1); k++) orest[k] = 0.f;
            }
            const float dot = dir[0] * ddir[0] + dir[1] * ddir[1] + dir[2] * ddir[2];
#pragma unroll
            for (int ax = 0; ax < 3; ax++) dmean[ax] += (ddir[ax] - dir[ax] * dot) * il;
        }
        // ---- S13 ----
        if (!a.cov3D_precomp) {
            float Rm[3][3];
            quat_to_rot(q, Rm);
            const float sv[3] = {a.scale_modifier * s[0], a.scale_modifier * s[1], a.scale_modifier * s[2]};
            const float Ds[3][3] = {{dcov[0], 0.5f * dcov[1], 0.5f * dcov[2]},
                                    {0.5f * dcov[1], dcov[3], 0.5f * dcov[4]},
                                    {0.5f * dcov[2], 0.5f * dcov[4], dcov[5]}};
            float dM[3][3], dR[3][3];
#pragma unroll
            for (int r = 0; r < 3; r++)
#pragma unroll
                for (int j = 0; j < 3; j++) {
                    float accv = 0.f;
#pragma unroll
                    for (int l = 0; l < 3; l++) accv += Ds[r][l] * Rm[l][j] * sv[j];
                    dM[r][j] = 2.f * accv;
                }
#pragma unroll
            for (int j = 0; j < 3; j++) {
                float accv = 0.f;
#pragma unroll
                for (int r = 0; r < 3; r++) { accv += Rm[r][j] * dM[r][j]; dR[r][j] = dM[r][j] * sv[j]; }
                dscale[j] = a.scale_modifier * accv;
            }
            const float r = q[0], x = q[1], y = q[2], z = q[3];
            drot[0] = 2.f * (-z * dR[0][1] + y * dR[0][2] + z * dR[1][0] - x * dR[1][2] - y * dR[2][0] + x * dR[2][1]);
            drot[1] = 2.f * (y * dR[0][1] + z * dR[0][2] + y * dR[1][0] - 2.f * x * dR[1][1] - r * dR[1][2] + z * dR[2][0] + r * dR[2][1] - 2.f * x * dR[2][2]);
            drot[2] = 2.f * (-2.f * y * dR[0][0] + x * dR[0][1] + r * dR[0][2] + x * dR[1][0] + z * dR[1][2] - r * dR[2][0] + z * dR[2][1] - 2.f * y * dR[2][2]);
            drot[3] = 2.f * (-2.f * z * dR[0][0] - r * dR[0][1] + x * dR[0][2] + r * dR[1][0] - 2.f * z * dR[1][1] + y * dR[1][2] + x * dR[2][0] + y * dR[2][1]);
        }
        if (RAW) {                   // chain through sigmoid / exp / normalize (scene/gaussian_model.py:33-41)
            const float o = opac;
            dop *= o * (1.f - o);
            if (!a.cov3D_precomp) {
                dscale[0] *= s[0]; dscale[1] *= s[1]; dscale[2] *= s[2];
                const float dotq = q[0] * drot[0] + q[1] * drot[1] + q[2] * drot[2] + q[3] * drot[3];
#pragma unroll
                for (int k = 0; k < 4; k++) drot[k] = (drot[k] - q[k] * dotq) * q_inv_norm;
            }
        }
    } else if (a.shs && writes) {
        if (!(SPLIT) && tile_path) {
#pragma unroll
            for (int v = 0; v < 12; v++) tile[lane * SH_TILE_ROW + v] = make_float4(0.f, 0.f, 0.f, 0.f);
        } else if (!(SPLIT)) {
            float *out = a.dL_dsh + si * (size_t)a.M * 3;
            for (int k = 0; k < 3 * a.M; k++) out[k] = 0.f;
        } else {
            float *odc = a.dL_dsh + 3 * si, *orest = a.dL_dsh_rest + si * (size_t)(a.M - 1) * 3;
            odc[0] = odc[1] = odc[2] = 0.f;
            for (int k = 0; k < 3 * (a.M - 1); k++) orest[k] = 0.f;
        }
    }

    if (a.shs && tile_path) {            // the wave's 64 rows = 768 consecutive float4 in memory
        __builtin_amdgcn_wave_barrier();
        float4 *dst = reinterpret_cast<float4 *>(a.dL_dsh + (size_t)wave_first * 48);
#pragma unroll
        for (int t = 0; t < 12; t++) {
            const int q = t * 64 + lane;
            const int g = (q * 43691) >> 19;         // q / 12 for q < 768
            if (DENSE) {                              // 64 rows of 192 B, each at its own Gaussian's place: 5.3 whole rows per store instruction
                if (g < nrows) reinterpret_cast<float4 *>(a.dL_dsh + (size_t)rows[g] * 48)[q - 12 * g] = tile[g * SH_TILE_ROW + (q - 12 * g)];
            } else if ((wave_writes >> g) & 1ull) dst[q] = tile[g * SH_TILE_ROW + (q - 12 * g)];
        }
        if (DENSE) __builtin_amdgcn_wave_barrier();      // the tile is reused by the wave's next 64
    }
    if (!writes) return;
    a.dL_dmeans2D[3 * si] = dm2[0]; a.dL_dmeans2D[3 * si + 1] = dm2[1]; a.dL_dmeans2D[3 * si + 2] = 0.f;
    a.dL_dopacity[si] = dop;
    if (a.dL_dcolors) { a.dL_dcolors[3 * si] = dcol[0]; a.dL_dcolors[3 * si + 1] = dcol[1]; a.dL_dcolors[3 * si + 2] = dcol[2]; }
    a.dL_dmeans3D[3 * si] = dmean[0]; a.dL_dmeans3D[3 * si + 1] = dmean[1]; a.dL_dmeans3D[3 * si + 2] = dmean[2];
    if (a.dL_dcov3D) {
#pragma unroll
        for (int k = 0; k < 6; k++) a.dL_dcov3D[6 * si + k] = dcov[k];
    }
    if (a.dL_dscales) {
        a.dL_dscales[3 * si] = dscale[0]; a.dL_dscales[3 * si + 1] = dscale[1]; a.dL_dscales[3 * si + 2] = dscale[2];
        reinterpret_cast<float4 *>(a.dL_drots)[si] = make_float4(drot[0], drot[1], drot[2], drot[3]);
    }
}

template <int D, bool RAW, bool SPLIT>
__global__ __launch_bounds__(256) void pergauss_bwd_kernel(PergaussBwdArgs a) {
    extern __shared__ __align__(16) float4 sh_tile_dyn[];
    pergauss_one<D, RAW, SPLIT, false>(a, blockIdx.x * 256 + threadIdx.x, sh_tile_dyn);
}

// Dense variant (round 3), for the reference's training layout (SH tensor with M = 16, scales + rotations).  At config 3 nine
// Gaussians in a hundred have a gradient; a wave of the kernel above runs the whole chain (~1400 vector instructions) for its few
// visible lanes, fetches a whole line for every few bytes it needs of them (85 MB read for 24 MB used) and writes zeros for the rest.
// Here, while the compositing kernel runs (FP32-issue-bound, the memory system idle), gsr_backward's second stream
//   * writes the zeros (fill_zero_kernel), and
//   * lists the Gaussians with a gradient and copies their inputs -- mean, opacity, scale, flags, rotation, SH row: 15 x 16 bytes --
//     into a compact buffer, 64 records per 15 KiB chunk laid out slot-major so that a wave reads its chunk with fifteen coalesced
//     1-KiB loads (gather_visible_kernel: the scattered reads happen here, off the critical path);
// afterwards pergauss_bwd_dense_kernel runs the chain on full waves: one wave per chunk, the only scattered reads left are the
// accumulator rows.  List order is whatever the workgroups' atomics made it: every Gaussian is independent.
#define GSR_PG_REC_SLOTS 15
#define GSR_PG_GATHER_SPAN 1024
__device__ __forceinline__ void load_record_scattered(const PergaussBwdArgs &a, const size_t g, float4 rec[GSR_PG_REC_SLOTS]) {
    const float4 *sh4 = reinterpret_cast<const float4 *>(a.shs + g * 48);
#pragma unroll
    for (int v = 0; v < 12; v++) rec[3 + v] = sh4[v];
    rec[0] = make_float4(a.means3D[3 * g], a.means3D[3 * g + 1], a.means3D[3 * g + 2], a.opac[g]);
    rec[1] = make_float4(a.scales[3 * g], a.scales[3 * g + 1], a.scales[3 * g + 2], __uint_as_float((uint32_t)a.clamped[g]));
    rec[2] = reinterpret_cast<const float4 *>(a.rotations)[g];
}

__global__ __launch_bounds__(256) void gather_visible_kernel(PergaussBwdArgs a) {
    __shared__ uint32_t list[GSR_PG_GATHER_SPAN];
    __shared__ uint32_t count, base_s;
    if (threadIdx.x == 0) count = 0u;
    const uint32_t mark = *a.touch_mark;
    constexpr int PER = GSR_PG_GATHER_SPAN / 256;
    int rad[PER];
    uint32_t tch[PER];
    const int g0 = blockIdx.x * GSR_PG_GATHER_SPAN + threadIdx.x;
#pragma unroll
    for (int r = 0; r < PER; r++) {                  // unconditional loads: a branch per load would put a memory round trip between them
        const int g = min(g0 + 256 * r, a.P - 1);
        rad[r] = a.radii[g];
        tch[r] = (uint32_t)a.touched[g];
    }
    __syncthreads();
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int r = 0; r < PER; r++) {
        const bool vis = g0 + 256 * r < a.P && rad[r] > 0 && tch[r] == mark;
        const unsigned long long m = __builtin_amdgcn_ballot_w64(vis);
        if (m == 0ull) continue;
        uint32_t at = 0u;
        if (lane == 0) at = atomicAdd(&count, (uint32_t)__builtin_popcountll(m));
        at = __builtin_amdgcn_readfirstlane(at);
        if (vis) list[at + __builtin_popcountll(m & ((1ull << lane) - 1ull))] = (uint32_t)(g0 + 256 * r);
    }
    __syncthreads();
    const uint32_t n = count;
    if (n == 0u) return;
    if (threadIdx.x == 0) base_s = atomicAdd(a.vis_count, n);        // one device atomic per workgroup with anything to list
    __syncthreads();
    const uint32_t base = base_s;
    for (uint32_t i = threadIdx.x; i < n; i += 256u) {
        const uint32_t g = list[i], e = base + i;
        a.vis_list[e] = g;
        if (e < a.vis_cap) {
            float4 rec[GSR_PG_REC_SLOTS];
            load_record_scattered(a, (size_t)g, rec);
            float4 *dst = a.vis_rec + (size_t)(e >> 6) * (64 * GSR_PG_REC_SLOTS) + (e & 63u);
#pragma unroll
            for (int v = 0; v < GSR_PG_REC_SLOTS; v++) dst[v * 64] = rec[v];
        }
    }
}

template <int D, bool RAW>
__global__ __launch_bounds__(64) void pergauss_bwd_dense_kernel(PergaussBwdArgs a) {
    const uint32_t count = *a.vis_count;
    const uint32_t e0 = blockIdx.x * 64u;
    if (e0 >= count) return;                          // wave-uniform
    // the camera, read once into registers (scalar loads here, before anything was stored; behind the chain's load barrier the compiler
    // would fetch each matrix again with vector loads where it is used: three more round trips)
    float vm[16], pm[16], cp[3];
#pragma unroll
    for (int k = 0; k < 16; k++) { vm[k] = a.viewmatrix[k]; pm[k] = a.projmatrix[k]; }
    cp[0] = a.campos[0]; cp[1] = a.campos[1]; cp[2] = a.campos[2];
    PergaussBwdArgs al = a;
    al.viewmatrix = vm; al.projmatrix = pm; al.campos = cp;
    __shared__ uint32_t rows[64];
    const int lane = threadIdx.x;
    const int nrows = (int)min(64u, count - e0);
    const uint32_t g = a.vis_list[e0 + (uint32_t)min(lane, nrows - 1)];
    rows[lane] = g;
    float4 rec[GSR_PG_REC_SLOTS];
    if (e0 + 64u <= a.vis_cap) {                      // the whole chunk was copied
        const float4 *src = a.vis_rec + (size_t)blockIdx.x * (64 * GSR_PG_REC_SLOTS) + lane;
#pragma unroll
        for (int v = 0; v < GSR_PG_REC_SLOTS; v++) rec[v] = src[v * 64];
    } else load_record_scattered(a, (size_t)g, rec);  // more Gaussians with a gradient than the buffer holds: this wave gathers its own
    __builtin_amdgcn_wave_barrier();
    extern __shared__ __align__(16) float4 sh_tile_dyn[];
    pergauss_one<D, RAW, false, true>(al, (int)g, sh_tile_dyn, rows, nrows, rec);
}

// Zero-fill of all gradient outputs (the rows of the Gaussians with a gradient are overwritten afterwards): plain 16-byte stores over
// each tensor's 16-byte-aligned body, the few floats before and after it by the first lanes of the tensor's first workgroup.
#define GSR_FILL_SEGS 10
#define GSR_FILL_F4_PER_BLOCK 2048          // 32 KiB per workgroup of 256 lanes: 8 stores per lane
struct FillZeroArgs {
    float *p[GSR_FILL_SEGS];
    unsigned long long n[GSR_FILL_SEGS];     // floats
    unsigned first_block[GSR_FILL_SEGS + 1]; // prefix sums of the workgroups per tensor
    int nseg;
};
__global__ __launch_bounds__(256) void fill_zero_kernel(FillZeroArgs f) {
    int sgm = 0;
    while (sgm + 1 < f.nseg && blockIdx.x >= f.first_block[sgm + 1]) sgm++;
    const unsigned b = blockIdx.x - f.first_block[sgm];
    float *p = f.p[sgm];
    const unsigned long long n = f.n[sgm];
    const unsigned long long head = min(n, (unsigned long long)(((16u - ((uintptr_t)p & 15u)) & 15u) >> 2));   // floats before the aligned body
    const unsigned long long body4 = (n - head) >> 2;                                                            // float4s in it
    typedef float f4v __attribute__((ext_vector_type(4)));
    f4v *q = reinterpret_cast<f4v *>(p + head);
    const f4v z = {0.f, 0.f, 0.f, 0.f};
    const unsigned long long at = (unsigned long long)b * GSR_FILL_F4_PER_BLOCK + threadIdx.x;
#pragma unroll
    for (int r = 0; r < GSR_FILL_F4_PER_BLOCK / 256; r++) {
        const unsigned long long k = at + 256ull * r;
        if (k < body4) __builtin_nontemporal_store(z, &q[k]);      // streamed past the L2 working set of the compositing kernel running beside it
    }
    if (b == 0 && threadIdx.x < 8) {
        const unsigned long long tail0 = head + (body4 << 2);
        if (threadIdx.x < 4) { if (threadIdx.x < head) p[threadIdx.x] = 0.f; }
        else if (tail0 + (threadIdx.x - 4) < n) p[tail0 + (threadIdx.x - 4)] = 0.f;
    }
}

// can the dense variant serve this call?  (the layout the records are made for)
bool pergauss_dense_eligible(const PergaussBwdArgs &a) {
    return a.shs && !a.shs_rest && a.M == 16 && a.scales && a.rotations && !a.cov3D_precomp && a.campos &&
           (reinterpret_cast<uintptr_t>(a.dL_dsh) & 15) == 0 && (reinterpret_cast<uintptr_t>(a.shs) & 15) == 0 &&
           (reinterpret_cast<uintptr_t>(a.rotations) & 15) == 0;
}
hipError_t launch_gather_visible(const PergaussBwdArgs &a, hipStream_t s) {
    if (a.P <= 0) return hipSuccess;
    hipLaunchKernelGGL(gather_visible_kernel, dim3((a.P + GSR_PG_GATHER_SPAN - 1) / GSR_PG_GATHER_SPAN), dim3(256), 0, s, a);
    return hipGetLastError();
}

hipError_t launch_fill_zero(const PergaussBwdArgs &a, hipStream_t s) {
    if (a.P <= 0) return hipSuccess;
    FillZeroArgs f;
    f.nseg = 0;
    unsigned blocks = 0;
    const unsigned long long P = (unsigned long long)a.P;
    auto add = [&](float *p, unsigned long long n) {
        if (!p || n == 0 || f.nseg >= GSR_FILL_SEGS) return;
        f.p[f.nseg] = p; f.n[f.nseg] = n; f.first_block[f.nseg] = blocks;
        blocks += (unsigned)((n / 4 + GSR_FILL_F4_PER_BLOCK - 1) / GSR_FILL_F4_PER_BLOCK) + (n / 4 == 0 ? 1u : 0u);
        f.nseg++;
    };
    const bool split = a.shs_rest != nullptr;
    add(a.dL_dmeans2D, 3 * P); add(a.dL_dopacity, P); add(a.dL_dcolors, 3 * P); add(a.dL_dmeans3D, 3 * P); add(a.dL_dcov3D, 6 * P);
    if (a.shs) {
        if (split) { add(a.dL_dsh, 3 * P); add(a.dL_dsh_rest, 3 * P * (unsigned long long)(a.M - 1)); }
        else add(a.dL_dsh, 3 * P * (unsigned long long)a.M);
    }
    if (a.dL_dscales) { add(a.dL_dscales, 3 * P); add(a.dL_drots, 4 * P); }
    f.first_block[f.nseg] = blocks;
    if (blocks == 0) return hipSuccess;
    hipLaunchKernelGGL(fill_zero_kernel, dim3(blocks), dim3(256), 0, s, f);
    return hipGetLastError();
}

hipError_t launch_pergauss_bwd(const PergaussBwdArgs &a_in, hipStream_t s) {
    if (a_in.P <= 0) return hipSuccess;
    PergaussBwdArgs a = a_in;
    const dim3 grid((a.P + 255) / 256), block(256);
    const int d = a.shs ? a.D : 0;
    const bool raw = a.raw_params != 0, split = a.shs_rest != nullptr;
    a.sh_tile = (a.shs && !split && a.M == 16 && (reinterpret_cast<uintptr_t>(a.dL_dsh) & 15) == 0) ? 1 : 0;
    const size_t lds = a.sh_tile ? (size_t)4 * 64 * SH_TILE_ROW * sizeof(float4) : 0;      // 52 KiB per workgroup: 3 workgroups per CU
    if (lds > 48 * 1024) {
        static std::atomic<uint64_t> attr_set{0};            // one bit per device (per-device attribute, idempotent)
        int dev = 0;
        (void)hipGetDevice(&dev);
        if (!(attr_set.load() & (1ull << (dev & 63)))) {
            hipError_t e = hipSuccess;
            const void *fns[] = {(const void *)pergauss_bwd_kernel<0, false, false>, (const void *)pergauss_bwd_kernel<1, false, false>,
                                 (const void *)pergauss_bwd_kernel<2, false, false>, (const void *)pergauss_bwd_kernel<3, false, false>,
                                 (const void *)pergauss_bwd_kernel<0, true, false>, (const void *)pergauss_bwd_kernel<1, true, false>,
                                 (const void *)pergauss_bwd_kernel<2, true, false>, (const void *)pergauss_bwd_kernel<3, true, false>};
            for (const void *f : fns)
                if (e == hipSuccess) e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return e;
            attr_set.fetch_or(1ull << (dev & 63));
        }
    }
    if (a.dense) {                                    // gsr_backward checked: SH tensor with M = 16, scales + rotations, not split
        const dim3 dgrid((a.P + 63) / 64), dblock(64);
        const size_t dlds = lds / 4;                 // one wave's tile
        switch (d * 2 + (raw ? 1 : 0)) {
            case 0: hipLaunchKernelGGL((pergauss_bwd_dense_kernel<0, false>), dgrid, dblock, dlds, s, a); break;
            case 1: hipLaunchKernelGGL((pergauss_bwd_dense_kernel<0, true>), dgrid, dblock, dlds, s, a); break;
            case 2: hipLaunchKernelGGL((pergauss_bwd_dense_kernel<1, false>), dgrid, dblock, dlds, s, a); break;
            case 3: hipLaunchKernelGGL((pergauss_bwd_dense_kernel<1, true>), dgrid, dblock, dlds, s, a); break;
            case 4: hipLaunchKernelGGL((pergauss_bwd_dense_kernel<2, false>), dgrid, dblock, dlds, s, a); break;
            case 5: hipLaunchKernelGGL((pergauss_bwd_dense_kernel<2, true>), dgrid, dblock, dlds, s, a); break;
            case 6: hipLaunchKernelGGL((pergauss_bwd_dense_kernel<3, false>), dgrid, dblock, dlds, s, a); break;
            default: hipLaunchKernelGGL((pergauss_bwd_dense_kernel<3, true>), dgrid, dblock, dlds, s, a); break;
        }
        return hipGetLastError();
    }
#define GSR_LAUNCH(DD)                                                                                   \
    do {                                                                                                 \
        if (!raw && !split) hipLaunchKernelGGL((pergauss_bwd_kernel<DD, false, false>), grid, block, lds, s, a);           \
        else if (raw && !split) hipLaunchKernelGGL((pergauss_bwd_kernel<DD, true, false>), grid, block, lds, s, a);        \
        else if (!raw && split) hipLaunchKernelGGL((pergauss_bwd_kernel<DD, false, true>), grid, block, 0, s, a);        \
        else hipLaunchKernelGGL((pergauss_bwd_kernel<DD, true, true>), grid, block, 0, s, a);                            \
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

}  // namespace gsr
