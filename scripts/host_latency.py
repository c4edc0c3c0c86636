#!/usr/bin/env python
"""Where a small scene's step goes on the host (config 1: 10 k Gaussians, 256 x 256; the transformer scripts of the reference render
small predicted sets over and over, train_transformer.py:79,124,186,213): step time, host time inside the forward / backward calls
of the backend, and a cProfile listing.

    python scripts/host_latency.py [--config cfg1_plumbing_10k_256] [--profile]
"""
import argparse
import cProfile
import io
import json
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="cfg1_plumbing_10k_256")
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--profile", action="store_true")
    args = ap.parse_args()
    import torch
    from gaussian_transformer_amd import GaussianRasterizationSettings, GaussianRasterizer, synth
    from gaussian_transformer_amd.rasterizer import get_backend
    sc = synth.make_config(args.config)
    dev = "cuda"; cam = sc.camera
    t = lambda a, g=False: torch.tensor(a, dtype=torch.float32, device=dev).requires_grad_(g)
    inp = dict(means3D=t(sc.means3D, True), opacities=t(sc.opacities, True), shs=t(sc.shs, True), scales=t(sc.scales, True), rotations=t(sc.rotations, True))
    rs = GaussianRasterizationSettings(cam.image_height, cam.image_width, cam.tanfovx, cam.tanfovy, t(sc.bg), 1.0, t(cam.world_view_transform),
                                       t(cam.full_proj_transform), sc.sh_degree, t(cam.camera_center), False, False)
    dL = t(sc.dL_dimage); params = list(inp.values()); P = sc.P

    def step():
        m2 = torch.zeros((P, 3), device=dev, requires_grad=True)
        c, r = GaussianRasterizer(raster_settings=rs)(means2D=m2, **inp)
        return torch.autograd.grad(c, params, grad_outputs=dL)
    for _ in range(30):
        step()
    torch.cuda.synchronize()
    n = args.steps
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    out = {"config": args.config, "ms_per_step": (time.perf_counter() - t0) / n * 1e3}
    # host time inside the backend calls (no synchronisation: what the Python thread spends there)
    be = get_backend()
    acc = {"fwd": 0.0, "bwd": 0.0}
    of, ob = be.forward, be.backward

    def tf(*a, **k):
        s = time.perf_counter(); r = of(*a, **k); acc["fwd"] += time.perf_counter() - s; return r

    def tb(*a, **k):
        s = time.perf_counter(); r = ob(*a, **k); acc["bwd"] += time.perf_counter() - s; return r
    be.forward, be.backward = tf, tb
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    tot = time.perf_counter() - t0
    be.forward, be.backward = of, ob
    out.update(host_ms_in_backend_forward=acc["fwd"] / n * 1e3, host_ms_in_backend_backward=acc["bwd"] / n * 1e3,
               host_ms_elsewhere=(tot - acc["fwd"] - acc["bwd"]) / n * 1e3)
    with torch.no_grad():
        t0 = time.perf_counter()
        for _ in range(n):
            GaussianRasterizer(raster_settings=rs)(means2D=torch.zeros((P, 3), device=dev), **inp)
        torch.cuda.synchronize()
        out["forward_only_ms"] = (time.perf_counter() - t0) / n * 1e3
    print(json.dumps(out))
    if args.profile:
        pr = cProfile.Profile(); pr.enable()
        for _ in range(200):
            step()
        torch.cuda.synchronize(); pr.disable()
        s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(25); print(s.getvalue()[:4500])


if __name__ == "__main__":
    main()
