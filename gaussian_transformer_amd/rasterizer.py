"""GaussianRasterizationSettings / GaussianRasterizer: the reference's rasterizer API on HIP.

Drop-in for the names the reference imports at gaussian_renderer/__init__.py:14 and uses at
:36-49 (settings), :51 (constructor), :85-93 (keyword-only forward returning (color, radii)).
Same field order, argument meaning, validation messages and gradient order as the absent
upstream package; the native layer underneath is libgsr_hip.so through ctypes
(_lib.py) instead of pybind11/CUDA.
"""
from __future__ import annotations

import ctypes as C
from typing import NamedTuple, Optional

import torch
from torch import nn

from . import _lib


class GaussianRasterizationSettings(NamedTuple):
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool
    debug: bool


def _ptr(t: Optional[torch.Tensor]):
    if t is None or t.numel() == 0:
        return None
    if not t.is_contiguous():      # the C ABI takes dense row-major buffers; the public entry points make them so
        raise _lib.GsrError(f"non-contiguous tensor of shape {tuple(t.shape)} passed to the native rasterizer")
    return t.data_ptr()


def _f32c(t: torch.Tensor, name: str, device) -> torch.Tensor:
    if t.dtype != torch.float32:
        raise _lib.GsrError(f"{name} must be float32, got {t.dtype}")
    if t.device != device:
        t = t.to(device)
    return t.contiguous()


class HipBackend:
    """Calls the C ABI.  The only backend the product ever constructs."""

    name = "hip"

    def __init__(self):
        self.lib = _lib.load()
        self._ws_cache = {}
        self._announced = {}         # device index -> (geom pointer, P, M, gradient tensors) of a forward(announce_backward=True)
        self._bin_hint = {}          # (P, W, H) -> bytes of binning workspace the last forward pass of that shape asked for

    def _sizes(self, P, W, H):
        key = (P, W, H)
        v = self._ws_cache.get(key)
        if v is None:
            g, i, b = C.c_size_t(), C.c_size_t(), C.c_size_t()
            _lib.check(self.lib.gsr_workspace_sizes(P, W, H, C.byref(g), C.byref(i), C.byref(b)), "gsr_workspace_sizes")
            v = (g.value, i.value, b.value)
            if len(self._ws_cache) > 64:
                self._ws_cache.clear()
            self._ws_cache[key] = v
        return v

    def _gradient_outputs(self, dev, P, M, Mrest, has_sr, has_colors, has_cov, arena):
        """The tensors gsr_backward writes (include/gsr.h): carved from the gradient arena when one is set."""
        f32 = dict(dtype=torch.float32, device=dev)
        off = [0]

        def out(shape):
            n = 1
            for d in shape:
                n *= d
            if arena is None or n == 0:
                return torch.empty(shape, **f32)
            v = arena[off[0]:off[0] + n].view(shape)
            off[0] += n
            return v
        g_means3D = out((P, 3))
        g_sh = out((P, M, 3)) if M > 0 else torch.empty((0,), **f32)
        g_sh_rest = out((P, Mrest, 3)) if Mrest > 0 else None
        g_opacity = out((P, 1))
        g_scales = out((P, 3)) if has_sr else torch.empty((0,), **f32)
        g_rots = out((P, 4)) if has_sr else torch.empty((0,), **f32)
        g_means2D = torch.empty((P, 3), **f32)
        # gradients of the inputs that were not given are not written at all (36 B per Gaussian less to store)
        g_colors = torch.empty((P, 3), **f32) if has_colors else None
        g_cov3D = torch.empty((P, 6), **f32) if has_cov else None
        return g_means3D, g_means2D, g_sh, g_colors, g_opacity, g_scales, g_rots, g_cov3D, g_sh_rest

    def forward(self, rs: GaussianRasterizationSettings, means3D, shs, colors_precomp, opacities, scales, rotations,
                cov3D_precomp, shs_rest=None, raw_params=False, announce_backward=False):
        """announce_backward: a backward() call for this render will follow.  Its gradient tensors are created now and announced
        to the library (gsr_backward_prefill), which writes their zeros beside this forward pass where that pays; backward() picks
        them up.  With a gradient arena they are its slices if the arena is set around this call as well as around backward() (the
        same one: anything else makes backward() carve and fill as before).  Not in the fused form."""
        dev = means3D.device
        if dev.type != "cuda":
            raise _lib.GsrError(f"the HIP rasterizer needs tensors on a HIP device, got {dev} (no CPU fallback)")
        P, H, W = int(means3D.shape[0]), int(rs.image_height), int(rs.image_width)
        M = int(shs.shape[1]) if shs.numel() else 0
        if shs_rest is not None:
            M += int(shs_rest.shape[1])
        with torch.cuda.device(dev):
            stream = torch.cuda.current_stream(dev).cuda_stream
            gb, ib, _ = self._sizes(P, W, H)
            u8 = dict(dtype=torch.uint8, device=dev)
            color = torch.empty((3, H, W), dtype=torch.float32, device=dev)
            radii = torch.empty((P,), dtype=torch.int32, device=dev)
            geom = torch.empty((gb,), **u8)
            img = torch.empty((ib,), **u8)
            holder = []
            # The library asks for the binning workspace in the middle of the forward pass, once the device has counted the pairs,
            # with the GPU waiting for the kernels that follow: torch.empty() there is host time on the critical path.  A buffer
            # of the size this shape needed last time (+5 %) is set aside beforehand; the callback hands it over if it is enough.
            key = (P, W, H)
            hint = self._bin_hint.get(key, 0)
            spare = torch.empty((int(hint * 1.05) + 4096,), **u8) if hint else None
            asked = [0]

            def alloc(_user, nbytes):
                try:
                    asked[0] = int(nbytes)
                    if spare is not None and not holder and nbytes <= spare.numel():
                        holder.append(spare)
                        return spare.data_ptr()
                    t = torch.empty((max(int(nbytes), 1),), **u8)
                    holder.append(t)
                    return t.data_ptr()
                except Exception:        # allocation failure surfaces as GSR_ERR_ALLOC
                    return None
            cb = _lib.ALLOC_FN(alloc)
            n = C.c_int64(0)
            bg = _f32c(rs.bg, "bg", dev); vm = _f32c(rs.viewmatrix, "viewmatrix", dev)
            pm = _f32c(rs.projmatrix, "projmatrix", dev); cp = _f32c(rs.campos, "campos", dev)
            stale = self._announced.pop(dev.index, None)      # kept allocated until gsr_forward has ordered its fill (include/gsr.h)
            grads = None
            arena = _grad_arena            # set around the forward call too: the announced tensors are its slices (and get zeroed now)
            if arena is not None and (arena.device != dev or arena.dtype != torch.float32 or not arena.is_contiguous()
                                      or arena.numel() < arena_floats(P, M, scales.numel() > 0)):
                announce_backward = False  # backward() raises the error
            if announce_backward and shs_rest is None and P > 0:
                grads = self._gradient_outputs(dev, P, M, 0, scales.numel() > 0, colors_precomp.numel() > 0, cov3D_precomp.numel() > 0, arena)
                g_means3D, g_means2D, g_sh, g_colors, g_opacity, g_scales, g_rots, g_cov3D, _ = grads
                _lib.check(self.lib.gsr_backward_prefill(P, M, _ptr(g_means2D), _ptr(g_opacity), _ptr(g_colors), _ptr(g_means3D), _ptr(g_cov3D),
                                                         _ptr(g_sh), _ptr(g_scales), _ptr(g_rots), None), "gsr_backward_prefill")
            rc = self.lib.gsr_forward(
                stream, P, int(rs.sh_degree), M, W, H, _ptr(bg), _ptr(means3D), _ptr(shs), _ptr(colors_precomp),
                _ptr(opacities), _ptr(scales), float(rs.scale_modifier), _ptr(rotations), _ptr(cov3D_precomp),
                _ptr(vm), _ptr(pm), _ptr(cp), float(rs.tanfovx), float(rs.tanfovy), int(bool(rs.prefiltered)),
                int(bool(rs.debug)), color.data_ptr(), _ptr(radii), geom.data_ptr(), gb, cb, None, img.data_ptr(), ib,
                C.byref(n), _ptr(shs_rest), int(bool(raw_params)))
            if rc != 0 and grads is not None:
                self.lib.gsr_backward_prefill(0, 0, None, None, None, None, None, None, None, None, None)
            del stale
            _lib.check(rc, "gsr_forward")
            if grads is not None:
                # one render per device: the next forward() drops it
                self._announced[dev.index] = (geom.data_ptr(), P, M, grads, None if arena is None else (arena.data_ptr(), arena.numel()))
            if asked[0]:
                self._bin_hint[key] = asked[0]
                if len(self._bin_hint) > 64:
                    self._bin_hint.pop(next(iter(self._bin_hint)))
            binning = holder[0] if holder else torch.empty((0,), **u8)
        return int(n.value), color, radii, geom, binning, img

    def backward(self, rs, num_rendered, dL_dpix, means3D, radii, shs, colors_precomp, scales, rotations,
                 cov3D_precomp, geom, binning, img, shs_rest=None, raw_params=False):
        dev = means3D.device
        P, H, W = int(means3D.shape[0]), int(rs.image_height), int(rs.image_width)
        M = int(shs.shape[1]) if shs.numel() else 0
        Mrest = int(shs_rest.shape[1]) if shs_rest is not None else 0
        f32 = dict(dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            stream = torch.cuda.current_stream(dev).cuda_stream
            bbs = C.c_size_t()         # depends on the pair count while the deterministic mode is on
            _lib.check(self.lib.gsr_backward_workspace_bytes(P, int(num_rendered), C.byref(bbs)), "gsr_backward_workspace_bytes")
            bwd_ws = torch.empty((bbs.value,), dtype=torch.uint8, device=dev)
            has_sr = scales.numel() > 0
            arena = _grad_arena
            if arena is not None and (arena.device != dev or arena.dtype != torch.float32 or not arena.is_contiguous()
                                      or arena.numel() < arena_floats(P, M + Mrest, has_sr)):
                raise _lib.GsrError("gradient arena must be a contiguous float32 tensor on the render device with "
                                    f"at least {arena_floats(P, M + Mrest, has_sr)} elements")
            ann = self._announced.pop(dev.index, None)
            if ann is not None and ann[:3] == (geom.data_ptr(), P, M) and Mrest == 0 and \
                    ann[4] == (None if arena is None else (arena.data_ptr(), arena.numel())):
                grads = ann[3]                       # created (and, where it pays, zero-filled) by the forward pass
            else:
                grads = self._gradient_outputs(dev, P, M, Mrest, has_sr, colors_precomp.numel() > 0, cov3D_precomp.numel() > 0, arena)
            g_means3D, g_means2D, g_sh, g_colors, g_opacity, g_scales, g_rots, g_cov3D, g_sh_rest = grads
            bg = _f32c(rs.bg, "bg", dev); vm = _f32c(rs.viewmatrix, "viewmatrix", dev)
            pm = _f32c(rs.projmatrix, "projmatrix", dev); cp = _f32c(rs.campos, "campos", dev)
            dL = _f32c(dL_dpix, "grad of rendered image", dev)
            rc = self.lib.gsr_backward(
                stream, P, int(rs.sh_degree), M + Mrest, int(num_rendered), W, H, _ptr(bg), _ptr(means3D), _ptr(radii),
                _ptr(shs), _ptr(colors_precomp), _ptr(scales), float(rs.scale_modifier), _ptr(rotations),
                _ptr(cov3D_precomp), _ptr(vm), _ptr(pm), _ptr(cp), float(rs.tanfovx), float(rs.tanfovy), _ptr(dL),
                _ptr(geom), geom.numel(), _ptr(binning), binning.numel(), _ptr(img), img.numel(), _ptr(bwd_ws),
                bwd_ws.numel(), _ptr(g_means2D), _ptr(g_opacity), _ptr(g_colors), _ptr(g_means3D), _ptr(g_cov3D),
                _ptr(g_sh), _ptr(g_scales), _ptr(g_rots), int(bool(rs.debug)), _ptr(shs_rest), int(bool(raw_params)),
                _ptr(g_sh_rest))
            del ann
            _lib.check(rc, "gsr_backward")
        if shs_rest is not None:
            return g_means3D, g_means2D, g_sh, g_colors, g_opacity, g_scales, g_rots, g_cov3D, g_sh_rest
        return g_means3D, g_means2D, g_sh, g_colors, g_opacity, g_scales, g_rots, g_cov3D

    def composited_mask(self, geom, P):
        """bool [P]: the Gaussians the forward pass that filled `geom` composited at all (include/gsr.h gsr_composited_mask): a
        superset of those that can receive a gradient, usually far smaller than radii > 0."""
        dev = geom.device
        out = torch.empty((P,), dtype=torch.uint8, device=dev)
        with torch.cuda.device(dev):
            rc = self.lib.gsr_composited_mask(torch.cuda.current_stream(dev).cuda_stream, P, _ptr(geom), geom.numel(), _ptr(out))
            _lib.check(rc, "gsr_composited_mask")
        return out.bool()

    def mark_visible(self, positions, viewmatrix, projmatrix):
        dev = positions.device
        if dev.type != "cuda":
            raise _lib.GsrError(f"the HIP rasterizer needs tensors on a HIP device, got {dev} (no CPU fallback)")
        P = int(positions.shape[0])
        out = torch.empty((P,), dtype=torch.uint8, device=dev)
        with torch.cuda.device(dev):
            pos = _f32c(positions, "positions", dev); vm = _f32c(viewmatrix, "viewmatrix", dev)
            pm = _f32c(projmatrix, "projmatrix", dev)
            rc = self.lib.gsr_mark_visible(torch.cuda.current_stream(dev).cuda_stream, P, _ptr(pos), _ptr(vm), _ptr(pm), _ptr(out))
            _lib.check(rc, "gsr_mark_visible")
        return out.bool()


_backend = None
_last_forward = None    # (weak reference to the geometry workspace, P) of the most recent forward pass, for composited_mask()
_grad_arena = None      # optional flat float32 tensor the backward pass carves its parameter gradients from


def _remember_forward(geom, P):
    """composited_mask() without arguments refers to the most recent forward pass; a weak reference, so that the workspace (hundreds
    of MB at 5 M Gaussians) is released with its autograd graph instead of living on until the next render."""
    global _last_forward
    import weakref
    _last_forward = (weakref.ref(geom), P)


class gradient_arena:
    """Context manager: while active, gsr_backward writes dL/d{means3D, shs, opacities, scales, rotations}
    straight into consecutive slices of `flat` (59 floats per Gaussian at M = 16, in that order), so a
    data-parallel step can all-reduce `flat` without a packing copy.  The returned gradient tensors are
    views of `flat`.  Entered around the forward call as well, it lets the library zero those slices beside the forward pass already
    (gsr_backward_prefill): `flat` then belongs to that render from its forward call on."""

    def __init__(self, flat: torch.Tensor):
        self.flat = flat

    def __enter__(self):
        global _grad_arena
        self.prev, _grad_arena = _grad_arena, self.flat
        return self.flat

    def __exit__(self, *exc):
        global _grad_arena
        _grad_arena = self.prev
        return False


def arena_floats(P: int, M: int, has_scale_rot: bool = True) -> int:
    return P * (3 + 3 * M + 1 + (7 if has_scale_rot else 0))


def get_backend():
    global _backend
    if _backend is None:
        _backend = HipBackend()      # raises if libgsr_hip.so is missing: no fallback
    return _backend


def _set_backend_for_tests(backend):
    """Test hook: tests/ inject an oracle-backed stand-in to exercise the host logic (argument
    validation, autograd plumbing, the gloo data-parallel path) on machines without a GPU.
    Product code never calls this."""
    global _backend
    prev = _backend
    _backend = backend
    return prev


def _dump(path, *objs):
    try:
        torch.save(tuple(o.detach().cpu() if isinstance(o, torch.Tensor) else o for o in objs), path)
    except Exception:
        pass


def composited_mask(image: Optional[torch.Tensor] = None):
    """bool [P]: which Gaussians a forward pass composited at all.  Every Gaussian with a non-zero gradient is among them; a
    data-parallel trainer can restrict its gradient exchange to the union of the ranks' masks
    (dist.GradientExchange.launch(visible=...)).  `image`: the colour tensor a GaussianRasterizer call returned -- the mask of THAT
    render, whatever was rendered since (several cameras per step, several devices per process).  Without it: the most recent
    forward pass of this process, while its autograd graph is alive.  None if the backend does not track it or the render is gone."""
    be = get_backend()
    if not hasattr(be, "composited_mask"):
        return None
    if image is not None:
        fn = image.grad_fn
        if fn is None or not hasattr(fn, "saved_tensors") or not hasattr(fn, "gsr_geom_index"):
            return None
        saved = fn.saved_tensors
        return be.composited_mask(saved[fn.gsr_geom_index], int(saved[fn.gsr_means_index].shape[0]))
    if _last_forward is None:
        return None
    ref_, P = _last_forward
    geom = ref_()
    return None if geom is None else be.composited_mask(geom, P)


class _RasterizeGaussians(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, raster_settings):
        be = get_backend()
        dev = means3D.device
        if means3D.dim() != 2 or means3D.shape[1] != 3:
            raise _lib.GsrError("means3D must have dimensions (num_points, 3)")
        args = [_f32c(t, n, dev) for t, n in ((means3D, "means3D"), (sh, "shs"), (colors_precomp, "colors_precomp"),
                                              (opacities, "opacities"), (scales, "scales"), (rotations, "rotations"),
                                              (cov3Ds_precomp, "cov3D_precomp"))]
        means3D_c, sh_c, colors_c, opac_c, scales_c, rots_c, cov_c = args
        try:
            num_rendered, color, radii, geom, binning, img = be.forward(
                raster_settings, means3D_c, sh_c, colors_c, opac_c, scales_c, rots_c, cov_c,
                **({"announce_backward": True} if any(ctx.needs_input_grad) and hasattr(be, "_announced") else {}))
        except Exception:
            if raster_settings.debug:
                _dump("snapshot_fw.dump", *args, raster_settings._asdict())
                print("\nAn error occured in forward. Please forward snapshot_fw.dump for debugging.")
            raise
        ctx.raster_settings = raster_settings
        ctx.num_rendered = num_rendered
        _remember_forward(geom, int(means3D_c.shape[0]))
        ctx.save_for_backward(colors_c, means3D_c, scales_c, rots_c, cov_c, radii, sh_c, geom, binning, img)
        ctx.gsr_geom_index, ctx.gsr_means_index = 7, 1        # composited_mask(image) finds the workspace through image.grad_fn
        ctx.mark_non_differentiable(radii)
        ctx.set_materialize_grads(False)      # no zero tensor for the int32 radii output on every backward (a 4 P byte fill)
        return color, radii

    @staticmethod
    def backward(ctx, grad_out_color, _grad_radii):
        be = get_backend()
        rs = ctx.raster_settings
        if grad_out_color is None:            # materialisation is off: an unused colour output arrives as None
            grad_out_color = torch.zeros((3, int(rs.image_height), int(rs.image_width)), dtype=torch.float32, device=ctx.saved_tensors[1].device)
        colors_c, means3D_c, scales_c, rots_c, cov_c, radii, sh_c, geom, binning, img = ctx.saved_tensors
        try:
            g_means3D, g_means2D, g_sh, g_colors, g_opac, g_scales, g_rots, g_cov = be.backward(
                rs, ctx.num_rendered, grad_out_color, means3D_c, radii, sh_c, colors_c, scales_c, rots_c, cov_c,
                geom, binning, img)
        except Exception:
            if rs.debug:
                _dump("snapshot_bw.dump", grad_out_color, means3D_c, radii, sh_c, colors_c, scales_c, rots_c, cov_c,
                      ctx.num_rendered, rs._asdict())
                print("\nAn error occured in backward. Writing snapshot_bw.dump for debugging.\n")
            raise
        none_if_empty = lambda g, src: g if src.numel() > 0 else None
        return (g_means3D, g_means2D, none_if_empty(g_sh, sh_c), none_if_empty(g_colors, colors_c), g_opac,
                none_if_empty(g_scales, scales_c), none_if_empty(g_rots, rots_c), none_if_empty(g_cov, cov_c), None)


def rasterize_gaussians(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, raster_settings):
    return _RasterizeGaussians.apply(means3D, means2D, sh, colors_precomp, opacities, scales, rotations,
                                     cov3Ds_precomp, raster_settings)


class _RasterizeGaussiansFused(torch.autograd.Function):
    """Fused-step variant: takes the RAW parameters of scene/gaussian_model.py (xyz, features_dc,
    features_rest, opacity logits, log-scales, un-normalised quaternions); the activations (:33-41) and the
    torch.cat of get_features (:108-111) happen inside the per-Gaussian kernels, forward and backward."""

    @staticmethod
    def forward(ctx, means3D, means2D, features_dc, features_rest, opacity_raw, scaling_raw, rotation_raw, raster_settings):
        be = get_backend()
        dev = means3D.device
        if means3D.dim() != 2 or means3D.shape[1] != 3:
            raise _lib.GsrError("means3D must have dimensions (num_points, 3)")
        a = [_f32c(t, n, dev) for t, n in ((means3D, "means3D"), (features_dc, "features_dc"), (features_rest, "features_rest"),
                                            (opacity_raw, "opacity"), (scaling_raw, "scaling"), (rotation_raw, "rotation"))]
        m3, dc, rest, op, sc, rot = a
        if dc.dim() != 3 or dc.shape[1] != 1 or rest.dim() != 3 or rest.shape[1] < 1:
            raise _lib.GsrError("features_dc must be [P,1,3] and features_rest [P,M-1,3] with M >= 2")
        empty = m3.new_empty((0,))
        num_rendered, color, radii, geom, binning, img = be.forward(raster_settings, m3, dc, empty, op, sc, rot, empty,
                                                                    shs_rest=rest, raw_params=True)
        ctx.raster_settings, ctx.num_rendered = raster_settings, num_rendered
        _remember_forward(geom, int(m3.shape[0]))
        ctx.save_for_backward(m3, dc, rest, sc, rot, radii, geom, binning, img)
        ctx.gsr_geom_index, ctx.gsr_means_index = 6, 0
        ctx.mark_non_differentiable(radii)
        ctx.set_materialize_grads(False)      # no zero tensor for the int32 radii output on every backward (a 4 P byte fill)
        return color, radii

    @staticmethod
    def backward(ctx, grad_out_color, _grad_radii):
        be = get_backend()
        m3, dc, rest, sc, rot, radii, geom, binning, img = ctx.saved_tensors
        if grad_out_color is None:
            rs_ = ctx.raster_settings
            grad_out_color = torch.zeros((3, int(rs_.image_height), int(rs_.image_width)), dtype=torch.float32, device=m3.device)
        empty = m3.new_empty((0,))
        g_m3, g_m2, g_dc, _g_col, g_op, g_sc, g_rot, _g_cov, g_rest = be.backward(
            ctx.raster_settings, ctx.num_rendered, grad_out_color, m3, radii, dc, empty, sc, rot, empty, geom, binning, img,
            shs_rest=rest, raw_params=True)
        return g_m3, g_m2, g_dc, g_rest, g_op, g_sc, g_rot, None


def rasterize_gaussians_fused(means3D, means2D, features_dc, features_rest, opacity_raw, scaling_raw, rotation_raw, raster_settings):
    """(color, radii) from the raw Gaussian parameters; gradients flow to the raw parameters."""
    return _RasterizeGaussiansFused.apply(means3D, means2D, features_dc, features_rest, opacity_raw, scaling_raw,
                                          rotation_raw, raster_settings)


class GaussianRasterizer(nn.Module):
    """Constructed per render call by the reference (gaussian_renderer/__init__.py:51): O(us), no allocation."""

    def __init__(self, raster_settings: GaussianRasterizationSettings):
        super().__init__()
        self.raster_settings = raster_settings

    def markVisible(self, positions: torch.Tensor) -> torch.Tensor:
        with torch.no_grad():
            rs = self.raster_settings
            return get_backend().mark_visible(positions, rs.viewmatrix, rs.projmatrix)

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                cov3D_precomp=None):
        rs = self.raster_settings
        if (shs is None and colors_precomp is None) or (shs is not None and colors_precomp is not None):
            raise Exception("Please provide excatly one of either SHs or precomputed colors!")
        if ((scales is None or rotations is None) and cov3D_precomp is None) or \
                ((scales is not None or rotations is not None) and cov3D_precomp is not None):
            raise Exception("Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!")
        empty = means3D.new_empty((0,), dtype=torch.float32)
        shs = empty if shs is None else shs
        colors_precomp = empty if colors_precomp is None else colors_precomp
        scales = empty if scales is None else scales
        rotations = empty if rotations is None else rotations
        cov3D_precomp = empty if cov3D_precomp is None else cov3D_precomp
        return rasterize_gaussians(means3D, means2D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp, rs)
