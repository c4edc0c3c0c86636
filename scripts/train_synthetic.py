#!/usr/bin/env python3
"""A whole 3DGS optimisation loop on the HIP path, shaped like the reference's train.py:66-128: camera of the iteration
-> render -> 0.8 L1 + 0.2 (1 - SSIM) -> backward -> densification statistics / clone / split / prune / opacity reset
-> Adam step.  Target images come from a hidden "ground truth" set of Gaussians seen from a ring of cameras; the trained
set starts from a perturbed, thinned copy.  Informational (bench.py is the headline metric).

    python scripts/train_synthetic.py [--iters 600] [--P 20000] [--size 256]
"""
import argparse, json, math, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from gaussian_transformer_amd import synth
from gaussian_transformer_amd.camera import look_at_camera
from gaussian_transformer_amd.densify import DensityController, OptimizationParams
from gaussian_transformer_amd.loss import fused_l1_ssim_loss, psnr
from gaussian_transformer_amd.model import GaussianParams
from gaussian_transformer_amd.render import PipelineParams, TorchCamera, render_fused


def run(iters=600, P=20000, size=256, ncam=8, seed=0, densify_from=100, densify_every=100, log=None):
    dev = torch.device("cuda", 0)
    sc = synth.make_scene(P=P, width=size, height=size, sh_degree=1, s0=0.03, seed=seed, zmin=3.0, zmax=6.0)
    truth = GaussianParams.from_synthetic(sc, dev, requires_grad=False)
    centre = np.array([0.0, 0.0, 4.5])
    cams = []
    for k in range(ncam):
        a = (k - (ncam - 1) / 2.0) * math.radians(8.0)
        eye = centre + 4.5 * np.array([math.sin(a), 0.0, -math.cos(a)])
        cams.append(TorchCamera(look_at_camera(eye, centre, (0.0, -1.0, 0.0), sc.camera.FoVx, size, size), dev))
    pipe, bg = PipelineParams(), torch.zeros(3, device=dev)
    with torch.no_grad():
        targets = [render_fused(c, truth, pipe, bg)["render"].clone() for c in cams]
    # the trained model: every third Gaussian of the truth, displaced, grey, fat and faint
    rng = np.random.default_rng(seed + 1)
    keep = np.arange(0, P, 3)
    sc0 = synth.SyntheticScene(sc.camera, (sc.means3D[keep] + rng.normal(0, 0.02, (len(keep), 3))).astype(np.float32),
                               (sc.scales[keep] * 1.5).astype(np.float32), sc.rotations[keep], np.full((len(keep), 1), 0.3, np.float32),
                               (sc.shs[keep] * 0.0).astype(np.float32), sc.sh_degree, sc.bg, sc.dL_dimage)
    model = GaussianParams.from_synthetic(sc0, dev)
    opt = OptimizationParams(densify_from_iter=densify_from, densification_interval=densify_every, opacity_reset_interval=10 ** 9,
                             densify_until_iter=iters)
    ctl = DensityController(model, opt, spatial_lr_scale=1.0, adam="hip")
    gen = torch.Generator(device=dev).manual_seed(seed)

    def evaluate():
        with torch.no_grad():
            imgs = [render_fused(c, model, pipe, bg)["render"] for c in cams]
            return (float(np.mean([float(fused_l1_ssim_loss(i, t)) for i, t in zip(imgs, targets)])),
                    float(np.mean([float(psnr(i[None], t[None]).mean()) for i, t in zip(imgs, targets)])))
    loss0, psnr0 = evaluate()
    hist = []
    t0 = time.perf_counter()
    for it in range(1, iters + 1):
        ctl.update_learning_rate(it)
        k = int(rng.integers(ncam))
        pkg = render_fused(cams[k], model, pipe, bg)
        loss = fused_l1_ssim_loss(pkg["render"], targets[k])
        loss.backward()
        with torch.no_grad():
            ev = ctl.after_backward(it, pkg["viewspace_points"], pkg["visibility_filter"], pkg["radii"], extent=3.0, generator=gen)
            ctl.optimizer.step()
            ctl.optimizer.zero_grad(set_to_none=True)
        if it % 50 == 0 or ev:
            rec = {"it": it, "loss": round(float(loss.detach()), 5), "P": int(model._xyz.shape[0]), "event": ev}
            hist.append(rec)
            if log:
                log(rec)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    loss1, psnr1 = evaluate()
    return {"iters": iters, "seconds": round(dt, 2), "it_per_s": round(iters / dt, 1), "P_start": len(keep), "P_end": int(model._xyz.shape[0]),
            "loss_first": round(loss0, 5), "loss_last": round(loss1, 5), "psnr_first": round(psnr0, 2), "psnr": round(psnr1, 2),
            "history": hist}


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=600)
    ap.add_argument("--P", type=int, default=20000)
    ap.add_argument("--size", type=int, default=256)
    a = ap.parse_args()
    out = run(a.iters, a.P, a.size, log=lambda r: print(json.dumps(r), flush=True))
    out.pop("history")
    print(json.dumps(out))
