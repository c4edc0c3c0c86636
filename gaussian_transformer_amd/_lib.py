"""ctypes binding of libgsr_hip.so (C ABI: include/gsr.h).

This is the Python-side stub a maintainer of the reference would add in place of the pybind11
module `diff_gaussian_rasterization._C`.  There is NO CPU fallback: if the
library is missing, cannot be loaded, or a tensor is not on a HIP device, calls raise.
"""
from __future__ import annotations

import ctypes as C
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GSR_LIB_PATH") or os.path.join(_HERE, "libgsr_hip.so")   # GSR_LIB_PATH: ablation builds (scripts/)

ALLOC_FN = C.CFUNCTYPE(C.c_void_p, C.c_void_p, C.c_size_t)
GSR_NUM_STAGES = 13

_p = C.c_void_p
_i32 = C.c_int32
_f = C.c_float
_sz = C.c_size_t

ADAM_MAX_GROUPS = 16


class AdamGroup(C.Structure):          # gsr_adam_group_t of include/gsr_optim.h
    _fields_ = [("param", C.c_void_p), ("grad", C.c_void_p), ("exp_avg", C.c_void_p), ("exp_avg_sq", C.c_void_p),
                ("n", C.c_int64), ("lr", C.c_float), ("step", C.c_int32)]


SIGNATURES = {
    "gsr_abi_version": (_i32, []),
    "gsr_last_error": (C.c_char_p, []),
    "gsr_workspace_sizes": (_i32, [_i32, _i32, _i32, C.POINTER(_sz), C.POINTER(_sz), C.POINTER(_sz)]),
    "gsr_binning_bytes": (_i32, [C.c_int64, _i32, _i32, C.POINTER(_sz)]),
    "gsr_backward_workspace_bytes": (_i32, [_i32, C.c_int64, C.POINTER(_sz)]),
    "gsr_forward": (_i32, [_p, _i32, _i32, _i32, _i32, _i32,          # stream P D M W H
                           _p, _p, _p, _p, _p,                        # bg means3D shs colors opacities
                           _p, _f, _p, _p,                            # scales mod rotations cov3D
                           _p, _p, _p, _f, _f,                        # view proj campos tanfovx tanfovy
                           _i32, _i32, _p, _p,                        # prefiltered debug out_color radii
                           _p, _sz, ALLOC_FN, _p, _p, _sz,            # geom, bytes, alloc, user, img, bytes
                           C.POINTER(C.c_int64),
                           _p, _i32]),                                # shs_rest raw_params (fused-step extension)
    "gsr_backward": (_i32, [_p, _i32, _i32, _i32, C.c_int64, _i32, _i32,   # stream P D M R W H
                            _p, _p, _p, _p, _p,                       # bg means3D radii shs colors
                            _p, _f, _p, _p,                           # scales mod rotations cov3D
                            _p, _p, _p, _f, _f,                       # view proj campos tanfovx tanfovy
                            _p, _p, _sz, _p, _sz, _p, _sz, _p, _sz,   # dL_dpix geom binning img bwd (+bytes)
                            _p, _p, _p, _p, _p, _p, _p, _p,           # 8 gradient outputs
                            _i32,                                     # debug
                            _p, _i32, _p]),                           # shs_rest raw_params dL_dsh_rest
    "gsr_mark_visible": (_i32, [_p, _i32, _p, _p, _p, _p]),
    "gsr_backward_prefill": (_i32, [_i32, _i32, _p, _p, _p, _p, _p, _p, _p, _p, _p]),   # P M + gsr_backward's 8 gradient outputs, dL_dsh_rest
    "gsr_composited_mask": (_i32, [_p, _i32, _p, C.c_size_t, _p]),
    "gsr_debug_read_geom": (_i32, [_p, _i32, _p, _p, _p, _p, _p, _p, _p]),
    "gsr_debug_read_binning": (_i32, [_p, C.c_int64, _i32, _i32, _p, _p, _p, _p, _p]),
    "gsr_debug_read_image_state": (_i32, [_p, _i32, _i32, _p, _p, _p]),
    "gsr_debug_read_lane_counters": (_i32, [C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "gsr_debug_read_wave_trace": (_i32, [_i32, _p, C.c_int64]),
    "gsr_debug_read_segments": (_i32, [_p, _i32, _i32, _p, _p]),
    "gsr_debug_read_bound_errors": (_i32, [_p, _i32, _p, _i32, _i32, _p, _p]),
    "gsr_l1_ssim_workspace": (_i32, [_i32, _i32, _i32, C.POINTER(_sz)]),
    "gsr_l1_ssim_forward": (_i32, [_p, _i32, _i32, _i32, _p, _p, _f, _p, _p, _sz]),
    "gsr_l1_ssim_backward": (_i32, [_p, _i32, _i32, _i32, _p, _p, _f, _p, _p, _sz, _p]),
    "gsr_knn_workspace": (_i32, [_i32, C.POINTER(_sz)]),
    "gsr_knn_mean_dist2": (_i32, [_p, _i32, _p, _p, _p, _sz]),
    "gsr_adam_step": (_i32, [_p, _i32, C.POINTER(AdamGroup), C.c_double, C.c_double, C.c_double]),
    "gsr_set_option": (_i32, [C.c_char_p, _i32]),
    "gsr_get_option": (_i32, [C.c_char_p, C.POINTER(_i32)]),
    "gsr_set_profiling": (_i32, [_i32]),
    "gsr_get_stage_times": (_i32, [C.POINTER(C.c_char_p), C.POINTER(_f)]),
}

_lock = threading.Lock()
_lib = None


class GsrError(RuntimeError):
    """A native failure of the HIP rasterizer (callers of the reference catch RuntimeError)."""


def load() -> C.CDLL:
    """Loads libgsr_hip.so once.  torch is imported first so that the library binds to the HIP
    runtime already living in the process (torch ships its own libamdhip64 with the same soname)."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        import torch  # noqa: F401  (loads libamdhip64 before ours resolves it)
        if not os.path.exists(LIB_PATH):
            raise GsrError(
                f"HIP extension not built: {LIB_PATH} is missing. Run `python -m gaussian_transformer_amd.build` "
                "(there is no CPU fallback).")
        try:
            lib = C.CDLL(LIB_PATH)
        except OSError as e:
            raise GsrError(f"cannot load {LIB_PATH}: {e}") from e
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)       # AttributeError if the symbol is missing: loud by design
            fn.restype = res
            fn.argtypes = args
        if lib.gsr_abi_version() != 2:
            raise GsrError(f"libgsr_hip.so ABI version {lib.gsr_abi_version()} != 2")
        _lib = lib
    return _lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = load().gsr_last_error()
        raise GsrError(f"{what} failed (code {rc}): {msg.decode(errors='replace') if msg else ''}")


def set_option(name: str, value: int) -> None:
    """Process-wide tuning knob of the native library (include/gsr.h: gsr_set_option)."""
    check(load().gsr_set_option(name.encode(), int(value)), f"gsr_set_option({name})")


def get_option(name: str) -> int:
    """Current value of a tuning knob (include/gsr.h: gsr_get_option)."""
    v = C.c_int32(0)
    check(load().gsr_get_option(name.encode(), C.byref(v)), f"gsr_get_option({name})")
    return int(v.value)


LANE_COUNTER_NAMES = ("staged", "visits", "block_visits", "lanes_ok", "lanes_past_last", "lanes_below_alpha",
                      "reductions", "dead_block_visits", "waves")


def read_lane_counters() -> dict:
    """Lane-slot accounting of the instrumented compositing kernels (set_option("count_lanes", 1) first); reads and
    resets the current device's counters.  Returns {"fwd": {...}, "bwd": {...}} with a derived "lane_efficiency"
    = lanes that blended / lane slots issued (64 per 8x8 block visit)."""
    f = (C.c_uint64 * 16)()
    b = (C.c_uint64 * 16)()
    check(load().gsr_debug_read_lane_counters(f, b), "gsr_debug_read_lane_counters")
    out = {}
    for tag, arr in (("fwd", f), ("bwd", b)):
        d = {n: int(arr[i]) for i, n in enumerate(LANE_COUNTER_NAMES)}
        slots = 64 * d["block_visits"]
        d["lane_slots"] = slots
        d["lane_efficiency"] = (d["lanes_ok"] / slots) if slots else 0.0
        out[tag] = d
    return out
