#!/usr/bin/env python3
"""Full 3DGS optimisation-step rate around the HIP rasterizer (informational; bench.py stays the
headline metric).  One iteration = the reference's train.py:84-128 without densification:
render (activations + rasterizer) -> 0.8 L1 + 0.2 (1-SSIM) -> backward -> Adam step (6 groups, eps 1e-15,
scene/gaussian_model.py:149-167) -> zero_grad.  Synthetic scene of bench.py, random target image.

    python scripts/train_step_bench.py [--config cfg3_synth_1M_1080p] [--iters 30]
"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from gaussian_transformer_amd import synth
from gaussian_transformer_amd.loss import fused_l1_ssim_loss, training_loss
from gaussian_transformer_amd.model import GaussianParams
from gaussian_transformer_amd.render import PipelineParams, TorchCamera, render, render_fused

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="cfg3_synth_1M_1080p")
ap.add_argument("--iters", type=int, default=30)
ap.add_argument("--warmup", type=int, default=3)
ap.add_argument("--loss", choices=["fused", "torch"], default="fused", help="fused = HIP L1+SSIM kernel (include/gsr_loss.h); torch = grouped conv2d path")
ap.add_argument("--adam", choices=["hip", "fused", "default"], default="hip", help="hip = one-launch HIP Adam (include/gsr_optim.h); fused/default = torch.optim.Adam")
ap.add_argument("--render", choices=["fused", "reference"], default="fused", help="fused = raw parameters into the kernels (render_fused)")
a = ap.parse_args()
dev = torch.device("cuda", 0)
sc = synth.make_config(a.config)
pc = GaussianParams.from_synthetic(sc, dev)
cam = TorchCamera(sc.camera, dev)
bg = torch.tensor(sc.bg, device=dev)
gt = torch.rand((3, cam.image_height, cam.image_width), device=dev)
lrs = [0.00016, 0.0025, 0.0025 / 20.0, 0.05, 0.005, 0.001]          # arguments/__init__.py:74-82
groups = [{"params": [p], "lr": lr} for p, lr in zip(pc.parameters(), lrs)]
if a.adam == "hip":
    from gaussian_transformer_amd.optim import HipAdam
    opt = HipAdam(groups, lr=0.0, eps=1e-15)
else:
    opt = torch.optim.Adam(groups, lr=0.0, eps=1e-15, **({"fused": True} if a.adam == "fused" else {}))
training_loss = fused_l1_ssim_loss if a.loss == "fused" else training_loss
render = render_fused if a.render == "fused" else render
pipe = PipelineParams()
parts = {"render": 0.0, "loss": 0.0, "backward": 0.0, "adam": 0.0}
ev = lambda: torch.cuda.Event(enable_timing=True)
losses = []
for it in range(a.warmup + a.iters):
    e = [ev() for _ in range(5)]
    e[0].record()
    pkg = render(cam, pc, pipe, bg)
    e[1].record()
    loss = training_loss(pkg["render"], gt)
    e[2].record()
    loss.backward()
    e[3].record()
    opt.step(); opt.zero_grad(set_to_none=True)
    e[4].record()
    torch.cuda.synchronize()
    if it >= a.warmup:
        for k, i in zip(parts, range(4)):
            parts[k] += e[i].elapsed_time(e[i + 1])
        losses.append(float(loss))
torch.cuda.synchronize()
t0 = time.perf_counter()          # wall clock without per-iteration syncs
for it in range(a.iters):
    pkg = render(cam, pc, pipe, bg)
    loss = training_loss(pkg["render"], gt)
    loss.backward()
    opt.step(); opt.zero_grad(set_to_none=True)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(json.dumps({"metric": "full optimisation steps/s (render + L1/SSIM loss + backward + Adam)", "value": round(a.iters / dt, 2),
                  "ms_per_iter": round(dt / a.iters * 1e3, 3), "config": a.config, "loss_impl": a.loss, "adam_impl": a.adam, "render_impl": a.render,
                  "ms_breakdown": {k: round(v / a.iters, 3) for k, v in parts.items()},
                  "loss_first": round(losses[0], 5), "loss_last": round(losses[-1], 5)}))
