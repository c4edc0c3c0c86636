#!/usr/bin/env python3
"""Soak: many forward+backward renders of random sizes back to back (no oracle), watching for errors, NaNs and
allocator growth.  python scripts/soak.py [--seconds 60]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gaussian_transformer_amd import GaussianRasterizationSettings, GaussianRasterizer, synth
from gaussian_transformer_amd.render import TorchCamera
ap = argparse.ArgumentParser(); ap.add_argument("--seconds", type=float, default=60.0); a = ap.parse_args()
dev = torch.device("cuda", 0)
rng = np.random.default_rng(0)
t0 = time.time(); n = 0; peak = 0
while time.time() - t0 < a.seconds:
    P = int(rng.choice([1, 50, 1000, 5000, 40000, 200000])); W = int(rng.integers(16, 1500)); H = int(rng.integers(16, 900))
    sc = synth.make_scene(P=P, width=W, height=H, sh_degree=int(rng.integers(0, 4)), s0=float(10 ** rng.uniform(-2.5, -0.5)),
                          seed=int(rng.integers(1 << 30)), zmin=float(rng.choice([0.05, 1.0, 3.0])), zmax=float(rng.choice([3.0, 10.0, 200.0])))
    cam = TorchCamera(sc.camera, dev)
    t = lambda x, g=False: torch.tensor(x, dtype=torch.float32, device=dev).requires_grad_(g)
    ps = [t(sc.means3D, True), t(sc.opacities, True), t(sc.shs, True), t(sc.scales, True), t(sc.rotations, True)]
    rs = GaussianRasterizationSettings(H, W, sc.camera.tanfovx, sc.camera.tanfovy, t(sc.bg), 1.0, cam.world_view_transform, cam.full_proj_transform,
                                       sc.sh_degree, cam.camera_center, False, False)
    for _ in range(3):
        m2 = torch.zeros((P, 3), device=dev, requires_grad=True)
        color, radii = GaussianRasterizer(raster_settings=rs)(means3D=ps[0], means2D=m2, shs=ps[2], opacities=ps[1], scales=ps[3], rotations=ps[4])
        gr = torch.autograd.grad(color, ps, grad_outputs=torch.ones_like(color) / color.numel())
        n += 1
    assert torch.isfinite(color).all() and all(torch.isfinite(g).all() for g in gr), (P, W, H)
    peak = max(peak, torch.cuda.max_memory_allocated(dev))
torch.cuda.synchronize()
print(f"soak ok: {n} renders in {time.time() - t0:.1f} s, peak allocated {peak / 2**20:.0f} MiB")
