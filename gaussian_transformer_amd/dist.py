"""One camera per GPU: data-parallel rendering with a single gradient all-reduce.

The reference is single-device (utils/general_utils.py:133 pins cuda:0; no torch.distributed
anywhere), so this is new functionality layered on its train step (train.py:84-93,113-128):
every rank holds a replica of the Gaussian parameters, renders its own camera of the batch,
back-propagates its own loss, and the parameter gradients of all cameras are summed with ONE
collective over a flat bucket:
    59 floats per Gaussian at M = 16 (xyz 3, f_dc 3, f_rest 3(M-1), opacity 1, scaling 3, rotation 4)
  +  2 floats per Gaussian of per-view densification statistics
       (|dL/dmean2D[:, :2]| * visible, visible)  -- scene/gaussian_model.py:405-407 is per view,
       so the norm is taken before the reduction, not after.
The screen-space radii (train.py:115) are max-reduced separately (4 bytes per Gaussian), which makes the density
control of densify.py rank-consistent: same statistics + an identically seeded split generator = same Gaussians.
One process per GPU (torch.distributed.run); backend "nccl" is RCCL over xGMI on ROCm, "gloo"
in the CPU tests.  No collective is issued inside the rasterizer itself.
"""
from __future__ import annotations

from typing import Callable, Optional, Sequence

import torch
import torch.distributed as dist


def direct_all_reduce(flat: torch.Tensor, group=None, scratch: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Sum `flat` over the ranks with two direct exchanges instead of a ring: every rank owns one 1/n slice,
    receives that slice from each peer at once (n-1 point-to-point transfers, one per xGMI link), sums, and sends the
    reduced slice back to every peer.  Per link and phase S/n bytes, against 2(n-1)/n S through the slowest link of a ring.
    Point-to-point only (batch_isend_irecv), so it also runs on gloo.  In place; returns `flat`.
    Not the default of bench.py: which of the two is faster on an 8-GPU node has not been measured (one GPU per box here)."""
    world = dist.get_world_size(group)
    if world == 1:
        return flat
    rank = dist.get_rank(group)
    n = flat.numel()
    per = (n + world - 1) // world
    lo = lambda r: min(r * per, n)
    hi = lambda r: min((r + 1) * per, n)
    mine = flat[lo(rank):hi(rank)]
    if scratch is None or scratch.numel() < (world - 1) * per or scratch.device != flat.device:
        scratch = torch.empty(((world - 1) * per,), dtype=flat.dtype, device=flat.device)
    peers = [r for r in range(world) if r != rank]
    to_global = (lambda r: dist.get_global_rank(group, r)) if group is not None else (lambda r: r)
    # phase 1: my slice of everyone's buffer comes to me
    ops, bufs = [], []
    for k, r in enumerate(peers):
        buf = scratch[k * per:k * per + mine.numel()]
        bufs.append(buf)
        if mine.numel():
            ops.append(dist.P2POp(dist.irecv, buf, to_global(r), group))
        if hi(r) > lo(r):
            ops.append(dist.P2POp(dist.isend, flat[lo(r):hi(r)], to_global(r), group))
    for w in (dist.batch_isend_irecv(ops) if ops else []):
        w.wait()
    for buf in bufs:
        mine += buf
    # phase 2: the reduced slices go back to everyone
    ops = []
    for r in peers:
        if hi(r) > lo(r):
            ops.append(dist.P2POp(dist.irecv, flat[lo(r):hi(r)], to_global(r), group))
        if mine.numel():
            ops.append(dist.P2POp(dist.isend, mine, to_global(r), group))
    for w in (dist.batch_isend_irecv(ops) if ops else []):
        w.wait()
    return flat


class GradientBucket:
    """Flat, reusable float32 buffer holding every parameter gradient + the densification stats."""

    def __init__(self, params: Sequence[torch.Tensor], num_points: int):
        self.shapes = [tuple(p.shape) for p in params]
        self.sizes = [int(p.numel()) for p in params]
        self.P = int(num_points)
        total = sum(self.sizes) + 2 * self.P
        dev = params[0].device
        self.flat = torch.zeros((total,), dtype=torch.float32, device=dev)

    def pack(self, params: Sequence[torch.Tensor], grad_norm_vis: torch.Tensor, vis: torch.Tensor) -> None:
        off = 0
        for p, n in zip(params, self.sizes):
            g = p.grad
            if g is None:
                self.flat[off:off + n].zero_()
            else:
                self.flat[off:off + n].copy_(g.reshape(-1))
            off += n
        self.flat[off:off + self.P].copy_(grad_norm_vis.reshape(-1)); off += self.P
        self.flat[off:off + self.P].copy_(vis.reshape(-1).to(torch.float32))

    def unpack(self, params: Sequence[torch.Tensor], scale: float = 1.0):
        off = 0
        for p, n, shp in zip(params, self.sizes, self.shapes):
            g = self.flat[off:off + n].view(shp)
            if p.grad is None:
                p.grad = (g * scale).clone()
            else:
                p.grad.copy_(g * scale)
            off += n
        gnorm = self.flat[off:off + self.P].clone(); off += self.P
        count = self.flat[off:off + self.P].clone()
        return gnorm, count

    @property
    def nbytes(self) -> int:
        return self.flat.numel() * 4


def data_parallel_step(pc, camera, pipe, bg: torch.Tensor, gt_image: Optional[torch.Tensor],
                       render_fn: Callable, loss_fn: Callable, bucket: Optional[GradientBucket] = None,
                       group=None, average: bool = True, dL_dimage: Optional[torch.Tensor] = None):
    """One data-parallel render step on this rank's camera.

    Mirrors train.py:86-93 (render -> loss -> backward) and :113-116 (densification statistics),
    then reduces.  Returns dict(loss, render, radii, visibility_filter, bucket).
    If dL_dimage is given it is used as the upstream gradient instead of a loss (bench mode).
    """
    params = pc.parameters()
    for p in params:
        p.grad = None
    pkg = render_fn(camera, pc, pipe, bg)
    image, vsp, vis, radii = pkg["render"], pkg["viewspace_points"], pkg["visibility_filter"], pkg["radii"]
    if dL_dimage is not None:
        loss = (image * dL_dimage).sum()
    else:
        loss = loss_fn(image, gt_image)
    loss.backward()
    P = pc.get_xyz.shape[0]
    if bucket is None:
        bucket = GradientBucket(params, P)
    g2 = vsp.grad if vsp.grad is not None else torch.zeros_like(vsp)
    gnorm = torch.norm(g2[:, :2], dim=-1) * vis.to(g2.dtype)
    bucket.pack(params, gnorm, vis)
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world > 1:
        dist.all_reduce(bucket.flat, op=dist.ReduceOp.SUM, group=group)
    gnorm_sum, vis_count = bucket.unpack(params, scale=(1.0 / world) if average else 1.0)
    # per-view statistics, summed over the views of the batch (add_densification_stats once per view)
    if getattr(pc, "xyz_gradient_accum", None) is not None:
        pc.xyz_gradient_accum += gnorm_sum[:, None]
        pc.denom += vis_count[:, None]
    # largest screen-space radius over the views of the batch (train.py:115), so that every rank prunes the same
    # Gaussians afterwards (densify.py: seed the split generator identically on all ranks)
    if getattr(pc, "max_radii2D", None) is not None:
        r = torch.where(vis, radii.to(pc.max_radii2D.dtype), torch.zeros_like(pc.max_radii2D))
        if world > 1:
            dist.all_reduce(r, op=dist.ReduceOp.MAX, group=group)
        pc.max_radii2D = torch.max(pc.max_radii2D, r)
    return {"loss": loss.detach(), "render": image.detach(), "radii": radii, "visibility_filter": vis, "bucket": bucket}
