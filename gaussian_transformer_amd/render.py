"""render(): the reference's render wrapper, restated (gaussian_renderer/__init__.py:18-100).

Same selection logic (SH vs precomputed colour :59-82, scale/rotation vs Python covariance
:62-66), same dummy `screenspace_points` whose .grad carries dL/dmean2D (:26-30), same result
dict (:97-100).  The reference's own file cannot travel to the GPU box, so the harness, the
bench and the tests call this one; with the reference checkout on sys.path its own render()
works unchanged against `diff_gaussian_rasterization` from this repo.
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import torch

from .rasterizer import GaussianRasterizationSettings, GaussianRasterizer, rasterize_gaussians_fused


@dataclass
class PipelineParams:                 # arguments/__init__.py:64-69
    convert_SHs_python: bool = False
    compute_cov3D_python: bool = False
    debug: bool = False


def eval_sh_torch(deg: int, sh: torch.Tensor, dirs: torch.Tensor) -> torch.Tensor:
    """Optional Python SH path (utils/sh_utils.py:57-112); sh [..., C, coeffs], dirs [..., 3]."""
    C0, C1 = 0.28209479177387814, 0.4886025119029199
    C2 = [1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396]
    C3 = [-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154, -0.4570457994644658,
          1.445305721320277, -0.5900435899266435]
    result = C0 * sh[..., 0]
    if deg > 0:
        x, y, z = dirs[..., 0:1], dirs[..., 1:2], dirs[..., 2:3]
        result = result - C1 * y * sh[..., 1] + C1 * z * sh[..., 2] - C1 * x * sh[..., 3]
        if deg > 1:
            xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
            result = (result + C2[0] * xy * sh[..., 4] + C2[1] * yz * sh[..., 5] + C2[2] * (2.0 * zz - xx - yy) * sh[..., 6]
                      + C2[3] * xz * sh[..., 7] + C2[4] * (xx - yy) * sh[..., 8])
            if deg > 2:
                result = (result + C3[0] * y * (3 * xx - yy) * sh[..., 9] + C3[1] * xy * z * sh[..., 10]
                          + C3[2] * y * (4 * zz - xx - yy) * sh[..., 11] + C3[3] * z * (2 * zz - 3 * xx - 3 * yy) * sh[..., 12]
                          + C3[4] * x * (4 * zz - xx - yy) * sh[..., 13] + C3[5] * z * (xx - yy) * sh[..., 14]
                          + C3[6] * x * (xx - 3 * yy) * sh[..., 15])
    return result


def render(viewpoint_camera, pc, pipe, bg_color: torch.Tensor, scaling_modifier: float = 1.0, override_color=None):
    """viewpoint_camera: anything with image_height/width, FoVx/FoVy, world_view_transform,
    full_proj_transform, camera_center (torch tensors on the device).  pc: GaussianParams-like."""
    xyz = pc.get_xyz
    screenspace_points = torch.zeros_like(xyz, dtype=xyz.dtype, requires_grad=True, device=xyz.device) + 0
    try:
        screenspace_points.retain_grad()
    except Exception:
        pass
    tanfovx = math.tan(viewpoint_camera.FoVx * 0.5)
    tanfovy = math.tan(viewpoint_camera.FoVy * 0.5)
    raster_settings = GaussianRasterizationSettings(
        image_height=int(viewpoint_camera.image_height), image_width=int(viewpoint_camera.image_width),
        tanfovx=tanfovx, tanfovy=tanfovy, bg=bg_color, scale_modifier=scaling_modifier,
        viewmatrix=viewpoint_camera.world_view_transform, projmatrix=viewpoint_camera.full_proj_transform,
        sh_degree=pc.active_sh_degree, campos=viewpoint_camera.camera_center, prefiltered=False, debug=pipe.debug)
    rasterizer = GaussianRasterizer(raster_settings=raster_settings)
    means3D, means2D, opacity = xyz, screenspace_points, pc.get_opacity
    scales = rotations = cov3D_precomp = None
    if pipe.compute_cov3D_python:
        cov3D_precomp = pc.get_covariance(scaling_modifier)
    else:
        scales, rotations = pc.get_scaling, pc.get_rotation
    shs = colors_precomp = None
    if override_color is None:
        if pipe.convert_SHs_python:
            shs_view = pc.get_features.transpose(1, 2).view(-1, 3, (pc.max_sh_degree + 1) ** 2)
            dir_pp = xyz - viewpoint_camera.camera_center.repeat(pc.get_features.shape[0], 1)
            dir_pp_normalized = dir_pp / dir_pp.norm(dim=1, keepdim=True)
            colors_precomp = torch.clamp_min(eval_sh_torch(pc.active_sh_degree, shs_view, dir_pp_normalized) + 0.5, 0.0)
        else:
            shs = pc.get_features
    else:
        colors_precomp = override_color
    rendered_image, radii = rasterizer(means3D=means3D, means2D=means2D, shs=shs, colors_precomp=colors_precomp,
                                       opacities=opacity, scales=scales, rotations=rotations, cov3D_precomp=cov3D_precomp)
    return {"render": rendered_image, "viewspace_points": screenspace_points, "visibility_filter": radii > 0, "radii": radii}


def render_fused(viewpoint_camera, pc, pipe, bg_color: torch.Tensor, scaling_modifier: float = 1.0):
    """Same result dict as render(), through the fused-step entry point: the raw parameters
    of the model go straight to the kernels -- no exp / sigmoid / normalize / cat tensors, no autograd nodes
    for them.  Only the default configuration of render() (SH colours, scale/rotation covariance) is
    covered; use render() for override_color / convert_SHs_python / compute_cov3D_python."""
    xyz = pc._xyz
    screenspace_points = torch.zeros_like(xyz, requires_grad=True) + 0
    try:
        screenspace_points.retain_grad()
    except Exception:
        pass
    rs = GaussianRasterizationSettings(
        image_height=int(viewpoint_camera.image_height), image_width=int(viewpoint_camera.image_width),
        tanfovx=math.tan(viewpoint_camera.FoVx * 0.5), tanfovy=math.tan(viewpoint_camera.FoVy * 0.5), bg=bg_color,
        scale_modifier=scaling_modifier, viewmatrix=viewpoint_camera.world_view_transform,
        projmatrix=viewpoint_camera.full_proj_transform, sh_degree=pc.active_sh_degree,
        campos=viewpoint_camera.camera_center, prefiltered=False, debug=pipe.debug)
    image, radii = rasterize_gaussians_fused(xyz, screenspace_points, pc._features_dc, pc._features_rest, pc._opacity,
                                             pc._scaling, pc._rotation, rs)
    return {"render": image, "viewspace_points": screenspace_points, "visibility_filter": radii > 0, "radii": radii}


class TorchCamera:
    """CameraMatrices (numpy) moved to a device, with the attribute names render() reads."""

    def __init__(self, cam, device):
        self.image_width, self.image_height = cam.image_width, cam.image_height
        self.FoVx, self.FoVy = cam.FoVx, cam.FoVy
        t = lambda a: torch.tensor(a, dtype=torch.float32, device=device)
        self.world_view_transform = t(cam.world_view_transform)
        self.full_proj_transform = t(cam.full_proj_transform)
        self.camera_center = t(cam.camera_center)
