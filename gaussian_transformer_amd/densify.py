"""Adaptive density control and the optimiser state that follows it.

Restates the behaviour of the reference's scene/gaussian_model.py: training_setup :149-167 (six Adam groups, eps 1e-15),
update_learning_rate :169-175 with utils/general_utils.py:29-61 (log-linear decay of the position rate),
reset_opacity :210-213, the optimiser-state surgery :258-322, densify_and_split :349-371, densify_and_clone :373-387,
densify_and_prune :389-403 -- and of train.py:113-123, which drives it.  Everything is torch index arithmetic on the
device that holds the Gaussians; no kernel of its own is needed.

One deliberate extension: `densify_and_split` draws its samples from a caller-supplied torch.Generator, so
data-parallel ranks that seed it identically split identically (the reference uses the global RNG).
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Callable, Dict, Optional

import torch

from .model import GaussianParams, inverse_sigmoid

GROUPS = ("xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation")          # order of training_setup :155-162
_ATTR = {"xyz": "_xyz", "f_dc": "_features_dc", "f_rest": "_features_rest", "opacity": "_opacity", "scaling": "_scaling",
         "rotation": "_rotation"}


@dataclass
class OptimizationParams:          # arguments/__init__.py:71-90 (this fork's defaults)
    iterations: int = 30_000
    position_lr_init: float = 0.00016
    position_lr_final: float = 0.0000016
    position_lr_delay_mult: float = 0.01
    position_lr_max_steps: int = 30_000
    feature_lr: float = 0.0025
    opacity_lr: float = 0.05
    scaling_lr: float = 0.005
    rotation_lr: float = 0.001
    percent_dense: float = 0.01
    lambda_dssim: float = 0.2
    densification_interval: int = 500
    opacity_reset_interval: int = 3000
    densify_from_iter: int = 100
    densify_until_iter: int = 10_000
    densify_grad_threshold: float = 0.0002
    random_background: bool = False


def expon_lr(lr_init: float, lr_final: float, lr_delay_steps: int = 0, lr_delay_mult: float = 1.0,
             max_steps: int = 1_000_000) -> Callable[[int], float]:
    """Log-linear interpolation lr_init -> lr_final over max_steps, optionally eased in (utils/general_utils.py:29-61)."""
    def at(step: int) -> float:
        if step < 0 or (lr_init == 0.0 and lr_final == 0.0):
            return 0.0
        delay = 1.0
        if lr_delay_steps > 0:
            delay = lr_delay_mult + (1.0 - lr_delay_mult) * math.sin(0.5 * math.pi * min(max(step / lr_delay_steps, 0.0), 1.0))
        t = min(max(step / max_steps, 0.0), 1.0)
        return delay * math.exp(math.log(lr_init) * (1.0 - t) + math.log(lr_final) * t)
    return at


def quaternion_to_rotation(q: torch.Tensor) -> torch.Tensor:
    """[P,4] (r,x,y,z), any norm -> [P,3,3] (utils/general_utils.py:78-99)."""
    q = q / torch.sqrt(q[:, 0] * q[:, 0] + q[:, 1] * q[:, 1] + q[:, 2] * q[:, 2] + q[:, 3] * q[:, 3])[:, None]
    r, x, y, z = q.unbind(dim=1)
    rows = [1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
            2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
            2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)]
    return torch.stack(rows, dim=1).view(-1, 3, 3)


class DensityController:
    """Owns the Adam optimiser of a GaussianParams and keeps it consistent while Gaussians are cloned, split and pruned."""

    def __init__(self, model: GaussianParams, opt: Optional[OptimizationParams] = None, spatial_lr_scale: float = 1.0,
                 fused_adam: bool = False, adam: str = "torch"):
        """adam: "torch" (torch.optim.Adam; fused_adam=True for its fused kernels) or "hip" (optim.HipAdam, one launch)."""
        self.model = model
        self.opt = opt or OptimizationParams()
        self.spatial_lr_scale = spatial_lr_scale
        o = self.opt
        P, dev = model._xyz.shape[0], model._xyz.device
        model.xyz_gradient_accum = torch.zeros((P, 1), device=dev)
        model.denom = torch.zeros((P, 1), device=dev)
        model.max_radii2D = torch.zeros((P,), device=dev)
        lrs = {"xyz": o.position_lr_init * spatial_lr_scale, "f_dc": o.feature_lr, "f_rest": o.feature_lr / 20.0,
               "opacity": o.opacity_lr, "scaling": o.scaling_lr, "rotation": o.rotation_lr}
        for n in GROUPS:           # the optimiser must own leaf tensors it can replace
            setattr(model, _ATTR[n], torch.nn.Parameter(getattr(model, _ATTR[n]).detach().clone().requires_grad_(True)))
        groups = [{"params": [getattr(model, _ATTR[n])], "lr": lrs[n], "name": n} for n in GROUPS]
        if adam == "hip":
            from .optim import HipAdam
            self.optimizer = HipAdam(groups, lr=0.0, eps=1e-15)
        else:
            self.optimizer = torch.optim.Adam(groups, lr=0.0, eps=1e-15, **({"fused": True} if fused_adam else {}))
        self._xyz_lr = expon_lr(o.position_lr_init * spatial_lr_scale, o.position_lr_final * spatial_lr_scale,
                                lr_delay_mult=o.position_lr_delay_mult, max_steps=o.position_lr_max_steps)

    # ---- learning rate ----
    def update_learning_rate(self, iteration: int) -> float:
        lr = self._xyz_lr(iteration)
        for g in self.optimizer.param_groups:
            if g["name"] == "xyz":
                g["lr"] = lr
        return lr

    # ---- optimiser-state surgery: every change of the Gaussian set is one row map applied to parameter and moments ----
    def _remap(self, param_fn: Callable[[str, torch.Tensor], torch.Tensor],
               moment_fn: Callable[[str, torch.Tensor], torch.Tensor], only: Optional[str] = None) -> None:
        for g in self.optimizer.param_groups:
            name = g["name"]
            if only is not None and name != only:
                continue
            old = g["params"][0]
            state = self.optimizer.state.pop(old, None)
            new = torch.nn.Parameter(param_fn(name, old.detach()).requires_grad_(True))
            if state is not None:
                state["exp_avg"] = moment_fn(name, state["exp_avg"])
                state["exp_avg_sq"] = moment_fn(name, state["exp_avg_sq"])
                self.optimizer.state[new] = state
            g["params"][0] = new
            setattr(self.model, _ATTR[name], new)

    def prune_points(self, mask: torch.Tensor) -> None:
        """Removes the Gaussians where `mask` is True (:287-301)."""
        keep = ~mask
        self._remap(lambda n, p: p[keep], lambda n, m: m[keep])
        m = self.model
        m.xyz_gradient_accum = m.xyz_gradient_accum[keep]
        m.denom = m.denom[keep]
        m.max_radii2D = m.max_radii2D[keep]

    def append_points(self, new: Dict[str, torch.Tensor]) -> None:
        """Appends Gaussians with zero Adam moments and resets the densification statistics (:303-347)."""
        self._remap(lambda n, p: torch.cat((p, new[n]), dim=0), lambda n, m: torch.cat((m, torch.zeros_like(new[n])), dim=0))
        m = self.model
        P, dev = m._xyz.shape[0], m._xyz.device
        m.xyz_gradient_accum = torch.zeros((P, 1), device=dev)
        m.denom = torch.zeros((P, 1), device=dev)
        m.max_radii2D = torch.zeros((P,), device=dev)

    def replace_tensor(self, name: str, tensor: torch.Tensor) -> None:
        """New values for one group, its Adam moments zeroed (:258-271)."""
        self._remap(lambda n, p: tensor, lambda n, m: torch.zeros_like(tensor), only=name)

    # ---- statistics (train.py:113-116) ----
    def record(self, viewspace_points: torch.Tensor, visibility: torch.Tensor, radii: torch.Tensor) -> None:
        m = self.model
        m.max_radii2D[visibility] = torch.max(m.max_radii2D[visibility], radii[visibility].to(m.max_radii2D.dtype))
        m.add_densification_stats(viewspace_points, visibility)

    # ---- densification ----
    def _selected(self, grads: torch.Tensor, threshold: float, extent: float, large: bool) -> torch.Tensor:
        size = self.model.get_scaling.max(dim=1).values
        cut = self.opt.percent_dense * extent
        return (grads >= threshold) & ((size > cut) if large else (size <= cut))

    def densify_and_clone(self, grads: torch.Tensor, threshold: float, extent: float) -> int:
        """Small Gaussians with a large view-space gradient are duplicated in place (:373-387)."""
        m = self.model
        sel = self._selected(grads.norm(dim=-1), threshold, extent, large=False)
        self.append_points({n: getattr(m, _ATTR[n]).detach()[sel] for n in GROUPS})
        return int(sel.sum())

    def densify_and_split(self, grads: torch.Tensor, threshold: float, extent: float, N: int = 2,
                          generator: Optional[torch.Generator] = None, normal_fn: Optional[Callable] = None) -> int:
        """Large Gaussians with a large gradient are replaced by N samples of themselves, 1.6x smaller (:349-371).
        normal_fn(stds) -> samples ~ N(0, stds^2) replaces torch.normal when given (e.g. drawn on the host, so that the
        result does not depend on which device's generator the model sits on)."""
        m = self.model
        P = m._xyz.shape[0]
        padded = torch.zeros((P,), device=m._xyz.device)
        padded[:grads.shape[0]] = grads.squeeze()                    # clones appended just before have no statistics yet
        sel = self._selected(padded, threshold, extent, large=True)
        scale = m.get_scaling.detach()[sel]
        stds = scale.repeat(N, 1)
        samples = normal_fn(stds) if normal_fn is not None else torch.normal(mean=torch.zeros_like(stds), std=stds, generator=generator)
        R = quaternion_to_rotation(m._rotation.detach()[sel]).repeat(N, 1, 1)
        new = {
            "xyz": torch.bmm(R, samples.unsqueeze(-1)).squeeze(-1) + m._xyz.detach()[sel].repeat(N, 1),
            "scaling": torch.log(scale.repeat(N, 1) / (0.8 * N)),
            "rotation": m._rotation.detach()[sel].repeat(N, 1),
            "f_dc": m._features_dc.detach()[sel].repeat(N, 1, 1),
            "f_rest": m._features_rest.detach()[sel].repeat(N, 1, 1),
            "opacity": m._opacity.detach()[sel].repeat(N, 1),
        }
        n_sel = int(sel.sum())
        self.append_points(new)
        self.prune_points(torch.cat((sel, torch.zeros(N * n_sel, dtype=torch.bool, device=sel.device))))
        return n_sel

    def densify_and_prune(self, max_grad: float, min_opacity: float, extent: float, max_screen_size: Optional[float],
                          generator: Optional[torch.Generator] = None, normal_fn: Optional[Callable] = None) -> Dict[str, int]:
        """:389-403.  Returns how many Gaussians were cloned / split / pruned."""
        m = self.model
        grads = m.xyz_gradient_accum / m.denom
        grads[grads.isnan()] = 0.0
        cloned = self.densify_and_clone(grads, max_grad, extent)
        split = self.densify_and_split(grads, max_grad, extent, generator=generator, normal_fn=normal_fn)
        prune = (m.get_opacity.detach() < min_opacity).squeeze(-1)
        if max_screen_size:
            prune = prune | (m.max_radii2D > max_screen_size) | (m.get_scaling.detach().max(dim=1).values > 0.1 * extent)
        self.prune_points(prune)
        return {"cloned": cloned, "split": split, "pruned": int(prune.sum())}

    def reset_opacity(self) -> None:
        """Opacity clamped to at most 0.01, its Adam moments zeroed (:210-213)."""
        o = self.model.get_opacity.detach()
        self.replace_tensor("opacity", inverse_sigmoid(torch.min(o, torch.full_like(o, 0.01))))

    # ---- the schedule of train.py:69-73, 113-123 ----
    def after_backward(self, iteration: int, viewspace_points: torch.Tensor, visibility: torch.Tensor, radii: torch.Tensor,
                       extent: float, white_background: bool = False, generator: Optional[torch.Generator] = None):
        """Call after loss.backward() and before optimizer.step(), with torch.no_grad()."""
        o = self.opt
        out = None
        if iteration < o.densify_until_iter:
            self.record(viewspace_points, visibility, radii)
            if iteration > o.densify_from_iter and iteration % o.densification_interval == 0:
                size_threshold = 20 if iteration > o.opacity_reset_interval else None
                out = self.densify_and_prune(o.densify_grad_threshold, 0.005, extent, size_threshold, generator=generator)
            if iteration % o.opacity_reset_interval == 0 or (white_background and iteration == o.densify_from_iter):
                self.reset_opacity()
        return out
