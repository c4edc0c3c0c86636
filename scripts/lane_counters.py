#!/usr/bin/env python
"""Lane-slot accounting of the two compositing kernels on a BASELINE config.

    python scripts/lane_counters.py [--config cfg3_synth_1M_1080p] [--npx 1 2 4]

Runs one forward + backward render with the instrumented kernels (gsr_set_option("count_lanes", 1)) per
blocks-per-wave setting and prints, per direction: list entries staged, splat visits, 8x8 block visits (64 lane
slots each), lanes that blended, why the other lanes were idle, reductions/atomics issued.
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def count_once(sc, dev, opts=(), mode=1):
    import torch
    from gaussian_transformer_amd import GaussianRasterizationSettings, GaussianRasterizer, _lib
    from gaussian_transformer_amd.render import TorchCamera
    cam = TorchCamera(sc.camera, dev)
    t = lambda a, g=False: torch.tensor(a, dtype=torch.float32, device=dev).requires_grad_(g)
    means3D, opac, shs = t(sc.means3D, True), t(sc.opacities, True), t(sc.shs, True)
    scales, rots = t(sc.scales, True), t(sc.rotations, True)
    rs = GaussianRasterizationSettings(
        image_height=cam.image_height, image_width=cam.image_width, tanfovx=sc.camera.tanfovx, tanfovy=sc.camera.tanfovy,
        bg=t(sc.bg), scale_modifier=1.0, viewmatrix=cam.world_view_transform, projmatrix=cam.full_proj_transform,
        sh_degree=sc.sh_degree, campos=cam.camera_center, prefiltered=False, debug=False)
    for k, v in opts:
        _lib.set_option(k, v)
    _lib.set_option("count_lanes", mode)
    try:
        _lib.read_lane_counters()                       # reset
        means2D = torch.zeros((sc.P, 3), dtype=torch.float32, device=dev, requires_grad=True)
        color, _ = GaussianRasterizer(raster_settings=rs)(means3D=means3D, means2D=means2D, shs=shs, opacities=opac,
                                                          scales=scales, rotations=rots)
        torch.autograd.grad(color, [means3D, opac, shs, scales, rots], grad_outputs=t(sc.dL_dimage))
        torch.cuda.synchronize()
        return _lib.read_lane_counters()
    finally:
        _lib.set_option("count_lanes", 0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="cfg3_synth_1M_1080p")
    ap.add_argument("--npx", type=int, nargs="*", default=[2])
    args = ap.parse_args()
    import torch
    from gaussian_transformer_amd import synth
    dev = torch.device("cuda", 0)
    sc = synth.make_config(args.config, seed=0)
    for npx in args.npx:
        c = count_once(sc, dev, (("fwd_blocks_per_wave", npx), ("bwd_blocks_per_wave", npx)))
        print(json.dumps({"config": args.config, "blocks_per_wave": npx, **c}))


if __name__ == "__main__":
    main()
