#!/usr/bin/env python
"""Renders a set of scenes (random ones over the list builders' regimes, BASELINE configs 2, 3 and 4, a 2560 x 1440 image, long unsaturated lists with the
segmented reverse pass) forward + backward and reads the capacity-assert words of the library after every one.

    GSR_LIB_PATH=gaussian_transformer_amd/libgsr_hip_dbg.so python scripts/debug_bounds_run.py [--scenes 40]

Meant for the debug build (python -m gaussian_transformer_amd.build --debug-bounds), where every index bounded by a host-side plan is
checked on the device (csrc/gsr_internal.h, GSR_IDX_OK).  Prints one JSON line; exit code 1 if any bound was violated."""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scenes", type=int, default=40)
    args = ap.parse_args()
    import torch
    from gaussian_transformer_amd import _lib, synth
    from gaussian_transformer_amd.rasterizer import GaussianRasterizationSettings, get_backend
    be = get_backend()
    lib = be.lib
    t = lambda a: torch.tensor(np.ascontiguousarray(a, dtype=np.float32), device="cuda")
    words = np.zeros(12, np.uint32)
    report = dict(scenes=0, violations=[], debug_build=None, selftest=None)

    def run(sc, tag):
        cam = sc.camera
        P = sc.P
        rs = GaussianRasterizationSettings(cam.image_height, cam.image_width, cam.tanfovx, cam.tanfovy, t(sc.bg), 1.0, t(cam.world_view_transform),
                                           t(cam.full_proj_transform), sc.sh_degree, t(cam.camera_center), False, False)
        e = torch.empty(0, device="cuda")
        a = (t(sc.means3D), t(sc.shs), e, t(sc.opacities.reshape(P, 1)), t(sc.scales), t(sc.rotations), e)
        n, color, radii, geom, binning, img = be.forward(rs, *a)
        be.backward(rs, n, torch.ones_like(color), a[0], radii, a[1], a[2], a[4], a[5], a[6], geom, binning, img)
        _lib.check(lib.gsr_debug_read_bound_errors(torch.cuda.current_stream().cuda_stream, P, geom.data_ptr(), cam.image_width, cam.image_height,
                                                   img.data_ptr(), words.ctypes.data), "read bound errors")
        report["scenes"] += 1
        report["debug_build"] = int(words[8]); report["selftest"] = [int(x) for x in words[9:12]]
        if words[3] or words[7]:
            report["violations"].append(dict(scene=tag, lists=[int(x) for x in words[0:4]], compositing=[int(x) for x in words[4:8]]))

    rng = np.random.default_rng(2024)
    for i in range(args.scenes):
        P = int(rng.choice([1, 7, 300, 4000, 9000, 30000, 90000]))
        W = int(rng.integers(16, 1600)); H = int(rng.integers(16, 1000))
        kw = dict(P=P, width=W, height=H, sh_degree=int(rng.integers(0, 4)), s0=float(10 ** rng.uniform(-2.3, -0.2)), seed=int(rng.integers(1 << 30)),
                  zmin=float(rng.choice([0.05, 1.0, 3.0])), zmax=float(rng.choice([3.0, 10.0, 200.0])))
        for opts in ((), (("tile_lists", 1),)) if i % 4 == 0 else ((),):          # round 1's list builder now and then
            for k, v in opts:
                _lib.set_option(k, v)
            try:
                run(synth.make_scene(**kw), f"random {kw} {opts}")
            finally:
                _lib.set_option("tile_lists", 2)
    # a crowded super-tile (bin beyond the small LDS buffer), many tiles per splat
    run(synth.make_scene(P=13000, width=64, height=64, sh_degree=0, s0=0.8, seed=3), "crowded 64x64")
    run(synth.make_scene(P=20000, width=32, height=32, sh_degree=0, s0=0.8, seed=4), "overfull bin 32x32 (falls back)")
    run(synth.make_config("cfg2_table_300k_800"), "cfg2")
    run(synth.make_config("cfg4_tiramisu_303k_1600x900"), "cfg4")
    # a large image with many Gaussians: the reverse pass in order of length, the dense per-Gaussian stage, both assembly walks
    run(synth.make_config("cfg3_synth_1M_1080p"), "cfg3")
    run(synth.make_scene(P=200000, width=2560, height=1440, sh_degree=3, s0=0.004, seed=12), "large image 2560x1440")
    # long unsaturated lists: checkpoints, pool exhaustion, overlong remainders, every unit class
    for seg in (64, 256):
        _lib.set_option("persistent_bwd", 1); _lib.set_option("segment_entries", seg)
        try:
            sc = synth.make_scene(P=30000, width=48, height=48, sh_degree=1, s0=0.03, seed=9)
            sc.opacities = (sc.opacities * 0.03 + 0.004).astype(np.float32)
            run(sc, f"long lists seg={seg}")
        finally:
            _lib.set_option("persistent_bwd", 2); _lib.set_option("segment_entries", 256)
    print(json.dumps(report))
    return 1 if report["violations"] else 0


if __name__ == "__main__":
    sys.exit(main())
