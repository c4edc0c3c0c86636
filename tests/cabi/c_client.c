/* Plain-C client of libgsr_hip.so: proves the drop-in boundary is a C ABI (no torch, no C++ types).
 * Built and run by tests/test_gpu_cabi.py on the GPU box:
 *   hipcc -x c ... or gcc c_client.c -I<repo>/include -I/opt/rocm/include -L<pkg> -lgsr_hip -L/opt/rocm/lib -lamdhip64
 * Renders 3 splats into a 48x32 image, runs the backward, exercises the error paths. */
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "gsr.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %d at line %d\n", (int)e_, __LINE__); return 2; } } while (0)

static void *g_bin = NULL; static size_t g_bin_bytes = 0; static int g_fail_alloc = 0;
static void *alloc_cb(void *user, size_t bytes) {
    (void)user;
    if (g_fail_alloc) return NULL;
    if (hipMalloc(&g_bin, bytes ? bytes : 1) != hipSuccess) return NULL;
    g_bin_bytes = bytes;
    return g_bin;
}
static float *dev_copy(const float *h, size_t n) { float *d; if (hipMalloc((void **)&d, n * 4) != hipSuccess) return NULL; hipMemcpy(d, h, n * 4, hipMemcpyHostToDevice); return d; }

int main(void) {
    const int P = 3, W = 48, H = 32, D = 0, M = 1;
    const float tanx = 0.5f, tany = 0.5f * H / W;
    /* camera at the origin looking down +z: world_view_transform = I; projection as utils/graphics_utils.py:51-71, transposed */
    float view[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    const float zn = 0.01f, zf = 100.f;
    float proj[16]; memset(proj, 0, sizeof proj);
    proj[0] = 1.f / tanx; proj[5] = 1.f / tany; proj[10] = zf / (zf - zn); proj[11] = 1.f; proj[14] = -(zf * zn) / (zf - zn);
    float campos[3] = {0, 0, 0}, bg[3] = {0.1f, 0.2f, 0.3f};
    float means[9] = {0, 0, 2.f, 0.3f, 0.1f, 3.f, 0, 0, -1.f};          /* third splat behind the camera */
    float shs[9] = {1.f, 0, 0, 0, 1.f, 0, 0, 0, 1.f}, opac[3] = {0.8f, 0.6f, 0.9f};
    float scales[9] = {0.2f, 0.2f, 0.2f, 0.3f, 0.1f, 0.2f, 0.2f, 0.2f, 0.2f}, rots[12] = {1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0};
    float *d_view = dev_copy(view, 16), *d_proj = dev_copy(proj, 16), *d_cam = dev_copy(campos, 3), *d_bg = dev_copy(bg, 3);
    float *d_means = dev_copy(means, 9), *d_shs = dev_copy(shs, 9), *d_op = dev_copy(opac, 3), *d_sc = dev_copy(scales, 9), *d_rot = dev_copy(rots, 12);
    size_t gb, ib, bb;
    if (gsr_abi_version() != GSR_ABI_VERSION) { printf("ABI version mismatch\n"); return 1; }
    if (gsr_workspace_sizes(P, W, H, &gb, &ib, &bb) != GSR_OK) { printf("sizes: %s\n", gsr_last_error()); return 1; }
    void *geom, *img, *bwd; float *color; int32_t *radii;
    CK(hipMalloc(&geom, gb)); CK(hipMalloc(&img, ib)); CK(hipMalloc(&bwd, bb));
    CK(hipMalloc((void **)&color, 3 * W * H * 4)); CK(hipMalloc((void **)&radii, P * 4));
    int64_t n = -1;
    /* error paths first: they must return codes, not abort, and leave the device usable */
    int rc = gsr_forward(NULL, P, D, M, W, H, d_bg, d_means, d_shs, d_shs, d_op, d_sc, 1.f, d_rot, NULL, d_view, d_proj, d_cam, tanx, tany, 0, 0,
                         color, radii, geom, gb, alloc_cb, NULL, img, ib, &n, NULL, 0);
    if (rc != GSR_ERR_INVALID_ARGUMENT) { printf("expected INVALID_ARGUMENT for shs+colors, got %d\n", rc); return 1; }
    rc = gsr_forward(NULL, P, D, M, W, H, d_bg, d_means, d_shs, NULL, d_op, d_sc, 1.f, d_rot, NULL, d_view, d_proj, d_cam, tanx, tany, 0, 0,
                     color, radii, geom, gb / 2, alloc_cb, NULL, img, ib, &n, NULL, 0);
    if (rc != GSR_ERR_WORKSPACE) { printf("expected WORKSPACE, got %d\n", rc); return 1; }
    g_fail_alloc = 1;
    rc = gsr_forward(NULL, P, D, M, W, H, d_bg, d_means, d_shs, NULL, d_op, d_sc, 1.f, d_rot, NULL, d_view, d_proj, d_cam, tanx, tany, 0, 0,
                     color, radii, geom, gb, alloc_cb, NULL, img, ib, &n, NULL, 0);
    if (rc != GSR_ERR_ALLOC || strlen(gsr_last_error()) == 0) { printf("expected ALLOC with a message, got %d\n", rc); return 1; }
    g_fail_alloc = 0;
    /* the real call, debug = 1 (synchronises after every stage) */
    rc = gsr_forward(NULL, P, D, M, W, H, d_bg, d_means, d_shs, NULL, d_op, d_sc, 1.f, d_rot, NULL, d_view, d_proj, d_cam, tanx, tany, 0, 1,
                     color, radii, geom, gb, alloc_cb, NULL, img, ib, &n, NULL, 0);
    if (rc != GSR_OK) { printf("forward failed: %s\n", gsr_last_error()); return 1; }
    float *h_color = (float *)malloc(3 * W * H * 4); int32_t h_r[3];
    CK(hipMemcpy(h_color, color, 3 * W * H * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(h_r, radii, 12, hipMemcpyDeviceToHost));
    if (!(h_r[0] > 0 && h_r[1] > 0 && h_r[2] == 0) || n <= 0) { printf("radii %d %d %d n=%lld\n", h_r[0], h_r[1], h_r[2], (long long)n); return 1; }
    const float centre_r = h_color[0 * W * H + (H / 2) * W + W / 2], corner_b = h_color[2 * W * H + 0];
    if (!(centre_r > 0.5f) || fabsf(corner_b - 0.3f) > 1e-6f) { printf("centre red %f corner blue %f\n", centre_r, corner_b); return 1; }
    /* backward with dL/dpix = 1 */
    float *ones = (float *)malloc(3 * W * H * 4); for (int i = 0; i < 3 * W * H; i++) ones[i] = 1.f;
    float *d_dl = dev_copy(ones, 3 * W * H);
    float *g2, *go, *gc, *g3, *gcov, *gsh, *gs, *gr;
    CK(hipMalloc((void **)&g2, P * 12)); CK(hipMalloc((void **)&go, P * 4)); CK(hipMalloc((void **)&gc, P * 12)); CK(hipMalloc((void **)&g3, P * 12));
    CK(hipMalloc((void **)&gcov, P * 24)); CK(hipMalloc((void **)&gsh, P * M * 12)); CK(hipMalloc((void **)&gs, P * 12)); CK(hipMalloc((void **)&gr, P * 16));
    rc = gsr_backward(NULL, P, D, M, n, W, H, d_bg, d_means, radii, d_shs, NULL, d_sc, 1.f, d_rot, NULL, d_view, d_proj, d_cam, tanx, tany, d_dl,
                      geom, gb, g_bin, g_bin_bytes, img, ib, bwd, bb, g2, go, gc, g3, gcov, gsh, gs, gr, 1, NULL, 0, NULL);
    if (rc != GSR_OK) { printf("backward failed: %s\n", gsr_last_error()); return 1; }
    float h_g3[9], h_go[3];
    CK(hipMemcpy(h_g3, g3, 36, hipMemcpyDeviceToHost)); CK(hipMemcpy(h_go, go, 12, hipMemcpyDeviceToHost));
    for (int i = 0; i < 9; i++) if (!isfinite(h_g3[i])) { printf("non-finite gradient\n"); return 1; }
    if (h_g3[6] != 0.f || h_g3[7] != 0.f || h_g3[8] != 0.f || h_go[2] != 0.f) { printf("culled splat has a gradient\n"); return 1; }
    if (h_go[0] == 0.f) { printf("visible splat has no opacity gradient\n"); return 1; }
    printf("C client ok: num_rendered=%lld centre_r=%.4f dL/dopacity0=%.5f\n", (long long)n, centre_r, h_go[0]);
    return 0;
}
