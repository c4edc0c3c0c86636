#!/usr/bin/env python
"""Per-tile list lengths of a BASELINE config as the HIP forward pass left them, and what they mean for load balance.

    python scripts/tile_stats.py [--config cfg2_table_300k_800 ...]

For every 16x16 tile: n = list length, last = the largest "last contributor" position of its pixels (what the reverse pass
walks; the forward pass stops once every pixel is saturated, at about the same place).  Prints percentiles, the share of
the work in the longest tiles and a lower bound on the makespan of "one wave pair per tile" against a perfectly balanced
machine (8192 resident waves).
"""
import argparse
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def tile_stats(config, dev):
    import torch
    from gaussian_transformer_amd import GaussianRasterizationSettings, _lib, synth
    from gaussian_transformer_amd.rasterizer import get_backend
    from gaussian_transformer_amd.render import TorchCamera
    sc = synth.make_config(config, seed=0)
    cam = TorchCamera(sc.camera, dev)
    t = lambda a: torch.tensor(a, dtype=torch.float32, device=dev)
    W, H = cam.image_width, cam.image_height
    rs = GaussianRasterizationSettings(
        image_height=H, image_width=W, tanfovx=sc.camera.tanfovx, tanfovy=sc.camera.tanfovy, bg=t(sc.bg), scale_modifier=1.0,
        viewmatrix=cam.world_view_transform, projmatrix=cam.full_proj_transform, sh_degree=sc.sh_degree,
        campos=cam.camera_center, prefiltered=False, debug=False)
    be = get_backend()
    empty = torch.empty(0, device=dev)
    N, color, radii, geom, binning, img = be.forward(rs, t(sc.means3D), t(sc.shs), empty, t(sc.opacities), t(sc.scales),
                                                      t(sc.rotations), empty)
    gx, gy = (W + 15) // 16, (H + 15) // 16
    T = gx * gy
    ranges = np.zeros((T, 2), dtype=np.uint32)
    lib = _lib.load()
    stream = torch.cuda.current_stream(dev).cuda_stream
    _lib.check(lib.gsr_debug_read_binning(stream, N, W, H, binning.data_ptr(), img.data_ptr(), None, None,
                                          ranges.ctypes.data_as(C.c_void_p)), "read binning")
    finalT = np.zeros((H, W), dtype=np.float32)
    ncon = np.zeros((H, W), dtype=np.uint32)
    _lib.check(lib.gsr_debug_read_image_state(stream, W, H, img.data_ptr(), finalT.ctypes.data_as(C.c_void_p),
                                              ncon.ctypes.data_as(C.c_void_p)), "read image state")
    n = (ranges[:, 1].astype(np.int64) - ranges[:, 0].astype(np.int64))
    pad = np.zeros((gy * 16, gx * 16), dtype=np.uint32)
    pad[:H, :W] = ncon
    last = pad.reshape(gy, 16, gx, 16).max(axis=(1, 3)).reshape(-1).astype(np.int64)
    # per 8-row half (what one wave of the default kernels owns)
    half = pad.reshape(gy, 2, 8, gx, 16).max(axis=(2, 4)).transpose(0, 2, 1).reshape(-1).astype(np.int64)
    q = lambda a: [int(x) for x in np.quantile(a, [0.5, 0.9, 0.99, 0.999, 1.0])]
    slots = 8192
    out = dict(config=config, P=sc.P, W=W, H=H, tiles=T, pairs=int(N), n_pct=q(n), last_pct=q(last), sum_n=int(n.sum()),
               sum_last=int(last.sum()), saturated_px_frac=float((finalT < 1e-3).mean()))
    # makespan model for the reverse pass: wave cost ~ its half-tile's last; all waves resident at once when 2T <= slots
    w = np.sort(half)[::-1]
    out["waves"] = int(w.size)
    out["longest_wave"] = int(w[0])
    out["balanced_per_slot"] = float(w.sum() / slots)
    out["imbalance_longest_over_balanced"] = float(w[0] / max(w.sum() / slots, 1))
    for thr in (256, 512, 1024, 2048):
        out[f"share_of_work_in_waves_over_{thr}"] = float(w[w > thr].sum() / max(w.sum(), 1))
        out[f"waves_over_{thr}"] = int((w > thr).sum())
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", nargs="*", default=["cfg2_table_300k_800", "cfg3_synth_1M_1080p", "cfg4_tiramisu_303k_1600x900"])
    args = ap.parse_args()
    import torch
    dev = torch.device("cuda", 0)
    for c in args.config:
        print(json.dumps(tile_stats(c, dev)), flush=True)


if __name__ == "__main__":
    main()
