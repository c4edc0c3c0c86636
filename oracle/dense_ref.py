"""Oracle A: float64 dense PyTorch restatement of the rasterizer forward (tiny sizes only).

TEST INFRASTRUCTURE ONLY.  Its job is to check oracle/gsr_ref.c (the analytic backward in
particular) through autograd and finite differences; it is O(P * H * W) per step in Python.
"Parity unpinned" applies here as in gsr_ref.c: it follows SURVEY.md 8a S1-S10, not an
executable reference.

Discrete decisions (cull, tile membership, sort order, alpha<1/255 skip, power>0 skip,
T<1e-4 stop) are constants for autograd.  Three straight-through conventions of the
published backward are reproduced with .detach():
  * alpha = min(0.99, o*G) passes the gradient of o*G even when capped (S10),
  * when t.x/t.z (t.y/t.z) hits the 1.3*tanfov clamp the clamped coordinate is a constant (S11),
  * 1/(det^2+1e-7) in S11 is NOT reproduced (exact inverse here; relative effect <= 1.3e-5
    because det >= 0.09 after the 0.3 dilation) -- tests allow for it.
"""
from __future__ import annotations

import math

import torch

TILE = 16
C0 = 0.28209479177387814
C1 = 0.4886025119029199
C2 = [1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396]
C3 = [-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154, -0.4570457994644658,
      1.445305721320277, -0.5900435899266435]


def _sh_color(deg, sh, d):
    """sh [P,M,3], d [P,3] unit -> [P,3] (utils/sh_utils.py:74-100 sign pattern)."""
    x, y, z = d[:, 0:1], d[:, 1:2], d[:, 2:3]
    res = C0 * sh[:, 0]
    if deg > 0:
        res = res - C1 * y * sh[:, 1] + C1 * z * sh[:, 2] - C1 * x * sh[:, 3]
    if deg > 1:
        xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
        res = (res + C2[0] * xy * sh[:, 4] + C2[1] * yz * sh[:, 5] + C2[2] * (2 * zz - xx - yy) * sh[:, 6]
               + C2[3] * xz * sh[:, 7] + C2[4] * (xx - yy) * sh[:, 8])
    if deg > 2:
        res = (res + C3[0] * y * (3 * xx - yy) * sh[:, 9] + C3[1] * xy * z * sh[:, 10]
               + C3[2] * y * (4 * zz - xx - yy) * sh[:, 11] + C3[3] * z * (2 * zz - 3 * xx - 3 * yy) * sh[:, 12]
               + C3[4] * x * (4 * zz - xx - yy) * sh[:, 13] + C3[5] * z * (xx - yy) * sh[:, 14]
               + C3[6] * x * (xx - 3 * yy) * sh[:, 15])
    return res


def _rot(q):
    r, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R = torch.stack([
        1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
        2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
        2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], dim=1)
    return R.view(-1, 3, 3)


def dense_render(W, H, tanfovx, tanfovy, viewmatrix, projmatrix, campos, bg, means3D, opacities,
                 sh_degree=0, shs=None, colors_precomp=None, scales=None, rotations=None, cov3D_precomp=None,
                 scale_modifier=1.0, means2D=None):
    """All tensor args float64 torch tensors.  Returns (color [3,H,W], radii [P] int, aux dict)."""
    P = means3D.shape[0]
    dt = means3D.dtype
    V = viewmatrix.reshape(4, 4)     # memory layout: p_row @ V
    Pm = projmatrix.reshape(4, 4)
    ones = torch.ones(P, 1, dtype=dt)
    ph = torch.cat([means3D, ones], 1) @ Pm
    pv = (torch.cat([means3D, ones], 1) @ V)[:, :3]
    pw = 1.0 / (ph[:, 3] + 1e-7)
    ndc = ph[:, :3] * pw[:, None]
    vis = pv[:, 2] > 0.2

    if cov3D_precomp is not None:
        c6 = cov3D_precomp
        Sig = torch.stack([c6[:, 0], c6[:, 1], c6[:, 2], c6[:, 1], c6[:, 3], c6[:, 4], c6[:, 2], c6[:, 4], c6[:, 5]], 1).view(P, 3, 3)
    else:
        Rm = _rot(rotations)
        Mm = Rm * (scale_modifier * scales)[:, None, :]
        Sig = Mm @ Mm.transpose(1, 2)

    fx, fy = W / (2.0 * tanfovx), H / (2.0 * tanfovy)
    limx, limy = 1.3 * tanfovx, 1.3 * tanfovy
    tz = pv[:, 2]
    tz_safe = torch.where(vis, tz, torch.ones_like(tz))
    txtz, tytz = pv[:, 0] / tz_safe, pv[:, 1] / tz_safe
    cx = txtz.clamp(-limx, limx); cy = tytz.clamp(-limy, limy)
    clx = (txtz < -limx) | (txtz > limx); cly = (tytz < -limy) | (tytz > limy)
    tx = torch.where(clx, (cx * tz_safe).detach(), pv[:, 0])
    ty = torch.where(cly, (cy * tz_safe).detach(), pv[:, 1])
    zero = torch.zeros_like(tz)
    J = torch.stack([fx / tz_safe, zero, -fx * tx / tz_safe ** 2, zero, fy / tz_safe, -fy * ty / tz_safe ** 2], 1).view(P, 2, 3)
    Wm = V[:3, :3].t()               # standard rotation: Wm[r][c] = V_mem[c*4+r]
    Tm = J @ Wm
    cov2 = Tm @ Sig @ Tm.transpose(1, 2)
    a = cov2[:, 0, 0] + 0.3; b = cov2[:, 0, 1]; c = cov2[:, 1, 1] + 0.3
    det = a * c - b * b
    ok = vis & (det != 0)
    det_s = torch.where(ok, det, torch.ones_like(det))
    conA, conB, conC = c / det_s, -b / det_s, a / det_s
    mid = 0.5 * (a + c)
    disc = (mid * mid - det).clamp_min(0.1)
    lam = torch.maximum(mid + disc.sqrt(), mid - disc.sqrt())
    radius = torch.ceil(3.0 * lam.sqrt()).detach()
    m2x = means2D[:, 0] if means2D is not None else 0.0   # dummy NDC offset whose grad is dL/dmean2D
    m2y = means2D[:, 1] if means2D is not None else 0.0
    px = ((ndc[:, 0] + m2x + 1.0) * W - 1.0) * 0.5
    py = ((ndc[:, 1] + m2y + 1.0) * H - 1.0) * 0.5
    gridx, gridy = (W + TILE - 1) // TILE, (H + TILE - 1) // TILE

    def cl(v, hi):
        return torch.clamp(torch.trunc(v), 0, hi)
    x0 = cl((px.detach() - radius) / TILE, gridx); x1 = cl((px.detach() + radius + TILE - 1) / TILE, gridx)
    y0 = cl((py.detach() - radius) / TILE, gridy); y1 = cl((py.detach() + radius + TILE - 1) / TILE, gridy)
    ok = ok & ((x1 - x0) * (y1 - y0) > 0)
    radii = torch.where(ok, radius, torch.zeros_like(radius)).to(torch.int32)

    if colors_precomp is not None:
        rgb = colors_precomp
    else:
        d = means3D - campos[None, :]
        d = d / d.norm(dim=1, keepdim=True)
        rgb = torch.clamp_min(_sh_color(sh_degree, shs, d) + 0.5, 0.0)

    # depth order, ties by index (stable)
    depth32 = pv[:, 2].detach().to(torch.float32)          # keys carry the float32 depth bits
    order = sorted([i for i in range(P) if bool(ok[i])], key=lambda i: (float(depth32[i]), i))

    ys, xs = torch.meshgrid(torch.arange(H, dtype=dt), torch.arange(W, dtype=dt), indexing="ij")
    tyx = torch.div(ys, TILE, rounding_mode="floor"); txx = torch.div(xs, TILE, rounding_mode="floor")
    Tr = torch.ones(H, W, dtype=dt)
    done = torch.zeros(H, W, dtype=torch.bool)
    C = torch.zeros(3, H, W, dtype=dt)
    o = opacities.reshape(-1)
    for g in order:
        member = (txx >= x0[g]) & (txx < x1[g]) & (tyx >= y0[g]) & (tyx < y1[g])
        dx = px[g] - xs; dy = py[g] - ys
        power = -0.5 * (conA[g] * dx * dx + conC[g] * dy * dy) - conB[g] * dx * dy
        G = torch.exp(torch.clamp(power, max=0.0))
        raw = o[g] * G
        alpha = raw + (torch.clamp(raw, max=0.99) - raw).detach()
        active = member & ~done & (power.detach() <= 0) & (alpha.detach() >= 1.0 / 255.0)
        Tn = Tr * (1 - alpha)
        stop = active & (Tn.detach() < 1e-4)
        done = done | stop
        blend = active & ~stop
        w = torch.where(blend, alpha * Tr, torch.zeros_like(Tr))
        C = C + rgb[g][:, None, None] * w[None]
        Tr = torch.where(blend, Tn, Tr)
    color = C + Tr[None] * bg[:, None, None]
    if P == 0:
        color = torch.zeros(3, H, W, dtype=dt)
    aux = dict(px=px, py=py, conic=torch.stack([conA, conB, conC], 1), rgb=rgb, depth=pv[:, 2], final_T=Tr, order=order)
    return color, radii, aux
