"""`distCUDA2` on HIP: mean squared distance to the three nearest neighbours (include/gsr_knn.h).

The reference imports `from simple_knn._C import distCUDA2` (scene/gaussian_model.py:20) and calls it once,
at scene/gaussian_model.py:134, on the SfM points to initialise the Gaussian scales.  No CPU fallback."""
import ctypes as C

import torch

from gaussian_transformer_amd import _lib


def distCUDA2(points: torch.Tensor) -> torch.Tensor:
    lib = _lib.load()
    if points.device.type != "cuda":
        raise _lib.GsrError(f"distCUDA2 needs a tensor on a HIP device, got {points.device} (no CPU fallback)")
    if points.dim() != 2 or points.shape[1] != 3:
        raise _lib.GsrError("distCUDA2: points must have shape [N, 3]")
    pts = points.detach().to(torch.float32).contiguous()
    N = int(pts.shape[0])
    out = torch.empty((N,), dtype=torch.float32, device=pts.device)
    nb = C.c_size_t()
    _lib.check(lib.gsr_knn_workspace(N, C.byref(nb)), "gsr_knn_workspace")
    with torch.cuda.device(pts.device):
        ws = torch.empty((max(nb.value, 1),), dtype=torch.uint8, device=pts.device)
        _lib.check(lib.gsr_knn_mean_dist2(torch.cuda.current_stream(pts.device).cuda_stream, N, pts.data_ptr() if N else None,
                                          out.data_ptr() if N else None, ws.data_ptr(), ws.numel()), "gsr_knn_mean_dist2")
    return out
