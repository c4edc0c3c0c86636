#!/usr/bin/env python3
"""Per-stage hipEvent times of config 3 with SH colours vs precomputed colours (what the SH rows cost preprocess / pergauss)."""
import ctypes as C, json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gaussian_transformer_amd import _lib, synth
from gaussian_transformer_amd.rasterizer import GaussianRasterizationSettings, GaussianRasterizer
from gaussian_transformer_amd.render import TorchCamera

dev = torch.device("cuda:0")
sc = synth.make_config(sys.argv[1] if len(sys.argv) > 1 else "cfg3_synth_1M_1080p", seed=0)
cam = TorchCamera(sc.camera, dev)
t = lambda a, g=False: torch.tensor(a, dtype=torch.float32, device=dev).requires_grad_(g)
means3D, opac, shs, scales, rots = t(sc.means3D, True), t(sc.opacities, True), t(sc.shs, True), t(sc.scales, True), t(sc.rotations, True)
colors = torch.rand((sc.P, 3), device=dev, requires_grad=True)
dL, bg = t(sc.dL_dimage), t(sc.bg)
rs = GaussianRasterizationSettings(image_height=cam.image_height, image_width=cam.image_width, tanfovx=sc.camera.tanfovx, tanfovy=sc.camera.tanfovy,
                                   bg=bg, scale_modifier=1.0, viewmatrix=cam.world_view_transform, projmatrix=cam.full_proj_transform,
                                   sh_degree=sc.sh_degree, campos=cam.camera_center, prefiltered=False, debug=False)
lib = _lib.load()
names = (C.c_char_p * _lib.GSR_NUM_STAGES)(); ms = (C.c_float * _lib.GSR_NUM_STAGES)()
for mode, exact in (("shs", 1), ("colors_precomp", 1), ("shs", 0), ("colors_precomp", 0)):
    _lib.set_option("exact_tile_cull", exact)
    def step():
        means2D = torch.zeros((sc.P, 3), device=dev, requires_grad=True)
        kw = dict(shs=shs) if mode == "shs" else dict(colors_precomp=colors)
        color, radii = GaussianRasterizer(raster_settings=rs)(means3D=means3D, means2D=means2D, opacities=opac, scales=scales, rotations=rots, **kw)
        params = [means3D, opac, scales, rots] + ([shs] if mode == "shs" else [colors])
        torch.autograd.grad(color, params, grad_outputs=dL)
        return radii
    for _ in range(3): radii = step()
    torch.cuda.synchronize()
    lib.gsr_set_profiling(1)
    acc = np.zeros(_lib.GSR_NUM_STAGES)
    for _ in range(10):
        step(); lib.gsr_get_stage_times(names, ms); acc += np.array(list(ms))
    lib.gsr_set_profiling(0)
    print(json.dumps({"mode": mode, "exact_tile_cull": exact, "visible_frac": float((radii > 0).float().mean()),
                      **{names[i].decode(): round(float(acc[i] / 10), 4) for i in range(_lib.GSR_NUM_STAGES)}}))
